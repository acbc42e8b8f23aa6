"""Bench-like free-running loop with per-call host timestamps: locates sporadic stalls.  usage: N mode(0 none,1 every,8 stride)"""
import sys, os, time; sys.path.insert(0, ".")
os.environ.setdefault("GRX_PUBLISH_DEBUG", "0")
import numpy as np, torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = int(sys.argv[1]); mode = int(sys.argv[2])
cfg = make_cfg(noise=True, dr=True, push=True, terrain="plane")
c, keep, _ = build_config.build(cfg, cfg.sim.dt, N)
s = HipSim(c, "cuda:0", keep); s.reset_all()
gen = torch.Generator().manual_seed(0)
acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(16)]
for i in range(50): s.step(acts[i % 16], 5.0, i + 1)
if mode: s.kernel_time_ms(mode)
torch.cuda.synchronize()
ts = np.zeros(501)
ts[0] = time.perf_counter()
for i in range(500):
    s.step(acts[i % 16], 5.0, 100 + i)
    ts[i + 1] = time.perf_counter()
s.wait_idle()
t_idle = time.perf_counter()
torch.cuda.synchronize()
t_end = time.perf_counter()
d = np.diff(ts) * 1e3
big = [(int(i), round(float(x), 2)) for i, x in enumerate(d) if x > 0.5]
print(f"N={N} mode={mode}: total {1e3*(t_end-ts[0])/500:.4f} ms/step (to idle {1e3*(t_idle-ts[0])/500:.4f}); issue median {np.median(d)*1e3:.1f} us, max {d.max():.2f} ms; calls>0.5ms: {big[:12]} (n={len(big)}); tail wait {1e3*(t_end-ts[-1]):.1f} ms; kernel {s.kernel_time_ms(False) if mode else ''}")
import glob
for f in glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/stats_*/evicted_ms") + glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/stats_*/cu_occupancy"):
    try: print("   ", f.split("/")[-2:], open(f).read().strip())
    except Exception as e: print("   ", f, e)
