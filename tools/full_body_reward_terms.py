"""Which reward terms carry the 32-DOF task's per-step reward?  Zero actions and the fresh policy's noise (0.2 rad), rough terrain: mean of
every active term per step (dt-scaled, as summed into the reward), lower-limb task beside it."""
import os, sys; sys.path.insert(0, ".")
import numpy as np, torch
os.environ["GRX_PUBLISH_DEBUG"] = "1"
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = 1024
STDS = [float(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0.0, 0.2]
for robot, nd in (("GR1T1", 10), ("GR1T1Full", 32)):
    for std in STDS:
        cfg = make_cfg(robot, noise=True, dr=True, push=False, terrain="heightfield")
        ter = make_terrain(cfg, N, 1)
        c, keep, info = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
        s = HipSim(c, "cuda:0", keep); s.reset_all()
        gen = torch.Generator(device="cuda").manual_seed(0)
        acc = None; rew = 0.0; k = 0
        for i in range(30):
            a = torch.randn(N, nd, device="cuda", generator=gen) * std
            s.step(a, 0.0, i + 1)
            if i >= 5:
                t = s.tensor("REWARD_TERMS").float()
                t = t if t.shape[0] == N else t.t()
                acc = t.mean(0).cpu().numpy() if acc is None else acc + t.mean(0).cpu().numpy()
                rew += float(s.tensor("REW").mean()); k += 1
        from wiki_grx_gym_amd import _capi
        names = _capi.REWARD_TERMS
        acc = acc / k
        order = np.argsort(acc)
        print(robot, "action std", std, "| mean reward per step %.4f" % (rew / k), "| terms (mean per step):")
        print("    " + ", ".join(f"{names[j]}: {acc[j]:+.4f}" for j in order if abs(acc[j]) > 1e-4))
        s.close()
