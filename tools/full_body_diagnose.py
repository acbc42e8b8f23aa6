"""What ends the episodes of the 32-DOF task UNDER A TRAINED POLICY?  Trains the registered "GR1T1_full_body" task for ITERS iterations, then
rolls the policy's mean action out (act_inference) and counts, per reset: time-out / terminating contact (which URDF links carry |F| > 1 N) / tilt
(|g_z| < 0.33), the step of the episode at which it happened, and the per-term reward of the last steps before a reset.
    python tools/full_body_diagnose.py [iters=300] [terrain=trimesh]"""
import contextlib, io, json, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
terrain = sys.argv[2] if len(sys.argv) > 2 else "trimesh"
os.environ["GRX_PUBLISH_DEBUG"] = "1"
from wiki_grx_gym_amd.envs import GR1T1FullBodyCfg, GR1T1FullBodyCfgPPO
from wiki_grx_gym_amd.utils import get_args, task_registry
from wiki_grx_gym_amd import _capi
args = get_args(["--task", "GR1T1_full_body", "--headless", "--num_envs", "4096", "--seed", "1", "--max_iterations", str(iters)])
cfg = GR1T1FullBodyCfg(); cfg.terrain.mesh_type = terrain; cfg.seed = 1
cfg.env.publish_reward_terms = True
for kv in os.environ.get("GRX_DIAG_SET", "").split(";"):      # e.g. "control.stiffness['ankle_roll']=28.6"
    if kv.strip():
        exec("cfg." + kv.strip(), {"cfg": cfg})
env, _ = task_registry.make_env("GR1T1_full_body", args=args, env_cfg=cfg)
tcfg = GR1T1FullBodyCfgPPO(); tcfg.seed = 1; tcfg.runner.save_interval = 10 ** 9
runner, _ = task_registry.make_alg_runner(env, name="GR1T1_full_body", args=args, train_cfg=tcfg, log_root=None)
with contextlib.redirect_stdout(io.StringIO()):
    runner.learn(num_learning_iterations=iters, init_at_random_ep_len=True)
policy = runner.get_inference_policy()
sim = env._sim
obs = env.get_observations()
N = env.num_envs
names = env.body_names
causes = {"time_out": 0, "contact": 0, "tilt": 0, "other": 0}
link_hits = np.zeros(len(names)); ep_len_at = []; rew_last = []; rew_all = []
term_links = set(env.termination_contact_indices.tolist())
with torch.inference_mode():
    for s in range(400):
        obs, pri, rew, done, ex = env.step(policy(obs))
        r = done.clone(); to = ex["time_outs"].bool() if "time_outs" in ex else torch.zeros_like(r)
        tc = sim.tensor("TERM_CONTACT").bool(); gz = sim.tensor("PROJECTED_GRAVITY")[:, 2]
        tilt = gz.abs() < 0.33
        causes["time_out"] += int((r & to).sum()); causes["contact"] += int((r & ~to & tc).sum())
        causes["tilt"] += int((r & ~to & ~tc & tilt).sum()); causes["other"] += int((r & ~to & ~tc & ~tilt).sum())
        if (r & tc).any():
            cf = env.contact_forces[r & tc]
            hit = (cf.norm(dim=-1) > 1.0).float().sum(0).cpu().numpy()
            link_hits[:len(hit)] += hit
        rew_all.append(float(rew.mean()))
# a synchronised episode: everybody reset, then the mean action -- how does the robot go down?
env.reset()
obs = env.get_observations()
series = []
alive = torch.ones(N, dtype=torch.bool, device=obs.device)
with torch.inference_mode():
    for s in range(90):
        obs, pri, rew, done, ex = env.step(policy(obs))
        g = sim.tensor("PROJECTED_GRAVITY"); root = sim.tensor("ROOT_STATES"); fc = sim.tensor("FEET_CONTACT").float()
        blv = sim.tensor("BASE_LIN_VEL")
        if s % 6 == 5:
            a = alive & ~done
            series.append({"step": s + 1, "alive": int(a.sum()), "g_x": round(float(g[a, 0].mean()), 3), "g_y_abs": round(float(g[a, 1].abs().mean()), 3), "g_y": round(float(g[a, 1].mean()), 3),
                           "g_z": round(float(g[a, 2].mean()), 3), "height_above_origin": round(float((root[a, 2] - env.env_origins[a, 2]).mean()), 3),
                           "v_x": round(float(blv[a, 0].mean()), 3), "feet_in_contact": round(float(fc[a].sum(1).mean()), 2),
                           "cmd_x": round(float(env.commands[a, 0].mean()), 3)})
        alive &= ~done
print(json.dumps(series))
print(json.dumps({"iters": iters, "terrain": terrain, "resets_by_cause_over_400_steps_x_4096_envs": causes,
                  "mean_reward_per_step_under_the_mean_action": float(np.mean(rew_all)),
                  "terminating_links": sorted(names[i] for i in term_links),
                  "links_loaded_when_a_contact_ended_an_episode": {names[i]: int(v) for i, v in enumerate(link_hits) if v > 0}}, indent=1))
