"""Learning A/B on the GPU box: GR1T1 rough-terrain curriculum, PPO with the reference's hyper-parameters,
seeds x {self-collision on / off} x {restitution on / off}; per arm the final-100-iteration mean, the slope over the last 300
iterations and the KL / learning-rate / noise-std traces.  Writes gpurun_out/learning_ab.json.
    python tools/train_ab.py [iterations] [seeds] [envs]"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, ".")
import numpy as np

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
os.makedirs("gpurun_out", exist_ok=True)
results = []
for sc in (1, 0):
    for rest in (1, 0):
        for seed in range(1, seeds + 1):
            os.environ["GRX_SELF_COLLISIONS"] = str(sc)
            if rest:
                os.environ.pop("GRX_NO_RESTITUTION", None)
            else:
                os.environ["GRX_NO_RESTITUTION"] = "1"
            import torch
            from wiki_grx_gym_amd.envs import GR1T1Cfg, GR1T1CfgPPO
            from wiki_grx_gym_amd.utils import get_args, task_registry
            args = get_args(["--task", "GR1T1", "--headless", "--num_envs", str(envs), "--seed", str(seed), "--max_iterations", str(iters)])
            cfg = GR1T1Cfg()
            cfg.terrain.mesh_type = "heightfield"
            cfg.seed = seed
            env, _ = task_registry.make_env("GR1T1", args=args, env_cfg=cfg)
            tcfg = GR1T1CfgPPO()
            tcfg.seed = seed
            tcfg.runner.save_interval = 10 ** 9
            out = f"gpurun_out/ab_sc{sc}_rest{rest}_s{seed}"
            runner, tcfg = task_registry.make_alg_runner(env, name="GR1T1", args=args, train_cfg=tcfg, log_root=out)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
            dt = time.time() - t0
            rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
            def series(tag):
                return np.array([r["value"] for r in rows if r["tag"] == tag], dtype=np.float64)
            rew, eplen = series("Train/mean_reward"), series("Train/mean_episode_length")
            tail = rew[-300:]
            slope = float(np.polyfit(np.arange(len(tail)), tail, 1)[0]) if len(tail) > 10 else 0.0
            tags = sorted({r["tag"] for r in rows})
            res = {"self_collisions": sc, "restitution": rest, "seed": seed, "iterations": iters, "wall_s": dt,
                   "final100_reward": float(rew[-100:].mean()), "final100_episode_length": float(eplen[-100:].mean()),
                   "best100_reward": float(max(rew[i:i + 100].mean() for i in range(0, max(1, len(rew) - 100), 50))),
                   "slope_last300_reward_per_iter": slope,
                   "reward_every_100": rew[::100].round(2).tolist(), "episode_length_every_100": eplen[::100].round(1).tolist(),
                   "noise_std_every_100": series("Policy/mean_noise_std")[::100].round(4).tolist(),
                   "learning_rate_every_100": (series("Loss/learning_rate")[::100].tolist() if "Loss/learning_rate" in tags else None),
                   "terrain_level_every_100": (series("Episode/terrain_level")[::100].round(2).tolist() if "Episode/terrain_level" in tags else None)}
            results.append(res)
            print(json.dumps({k: v for k, v in res.items() if not isinstance(v, list)}), flush=True)
            json.dump({"arms": results, "tags": tags}, open("gpurun_out/learning_ab.json", "w"), indent=1)
            env.close()
            del runner, env
            torch.cuda.empty_cache()
