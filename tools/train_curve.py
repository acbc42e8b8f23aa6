"""Learning-curve run on the GPU box: registered GR1T1 task (flat terrain), PPO with the reference's
hyper-parameters; writes the scalar log and a summary under gpurun_out/."""
import json, os, shutil, sys, time
sys.path.insert(0, ".")
import torch
from wiki_grx_gym_amd.envs import GR1T1Cfg, GR1T1CfgPPO
from wiki_grx_gym_amd.utils import get_args, task_registry

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
terrain = sys.argv[3] if len(sys.argv) > 3 else "plane"
out = "gpurun_out/train_" + terrain
os.makedirs(out, exist_ok=True)
args = get_args(["--task", "GR1T1", "--headless", "--num_envs", str(envs), "--seed", "1", "--max_iterations", str(iters)])
cfg = GR1T1Cfg()
cfg.terrain.mesh_type = terrain
env, _ = task_registry.make_env("GR1T1", args=args, env_cfg=cfg)
tcfg = GR1T1CfgPPO()
tcfg.runner.save_interval = 500
runner, tcfg = task_registry.make_alg_runner(env, name="GR1T1", args=args, train_cfg=tcfg, log_root=out)
import io, contextlib
t0 = time.time()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
dt = time.time() - t0
rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
def series(tag): return [(r["step"], r["value"]) for r in rows if r["tag"] == tag]
summary = {"iterations": iters, "num_envs": envs, "terrain": terrain, "wall_s": dt, "env_steps": iters * 64 * envs,
           "train_env_steps_per_s": iters * 64 * envs / dt,
           "mean_reward": series("Train/mean_reward")[::25], "mean_episode_length": series("Train/mean_episode_length")[::25],
           "fps": series("Perf/total_fps")[::100], "collection_time": series("Perf/collection time")[::100], "learning_time": series("Perf/learning_time")[::100],
           "noise_std": series("Policy/mean_noise_std")[::100],
           "rew_cmd_diff_lin_vel_x": series("Episode/rew_cmd_diff_lin_vel_x")[::50], "rew_feet_air_time": series("Episode/rew_feet_air_time")[::50]}
json.dump(summary, open(os.path.join(out, "learning_curve.json"), "w"))
print(json.dumps({k: (v if not isinstance(v, list) else v[-3:]) for k, v in summary.items()}))
