"""Per-step wall times (synchronised) of HipSim.step: finds one-off stalls.  usage: python tools/gpu_step_times.py N [steps]"""
import sys, os, time; sys.path.insert(0, ".")
os.environ.setdefault("GRX_PUBLISH_DEBUG", "0")
import torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = make_cfg(noise=True, dr=True, push=True, terrain="heightfield")
ter = make_terrain(cfg, N, 1)
c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
s = HipSim(c, "cuda:0", keep); s.reset_all()
gen = torch.Generator().manual_seed(0)
acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
ts = []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.step(acts[i % 8], 5.0, i + 1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
import numpy as np
ts = np.array(ts)
print("median ms", np.median(ts), "max", ts.max(), "at", int(ts.argmax()), "outliers>1ms:", [(int(i), round(float(t), 2)) for i, t in enumerate(ts) if t > 1.0][:20])
