"""Host issue rate of HipSim.step (time to enqueue K steps, before synchronising) vs GPU time."""
import sys, os, time; sys.path.insert(0, ".")
os.environ.setdefault("GRX_PUBLISH_DEBUG", "0")
import torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
print("cpus", os.cpu_count(), "loadavg", os.getloadavg(), "affinity", len(os.sched_getaffinity(0)))
cfg = make_cfg(noise=True, dr=True, push=True, terrain="plane")
c, keep, _ = build_config.build(cfg, cfg.sim.dt, N)
s = HipSim(c, "cuda:0", keep); s.reset_all()
gen = torch.Generator().manual_seed(0)
acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
for timing in (False, True):
    for i in range(50): s.step(acts[i % 8], 5.0, i + 1)
    torch.cuda.synchronize()
    if timing: s.kernel_time_ms(True)
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(500): s.step(acts[i % 8], 5.0, 100 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"events={timing} issue {1e3*(t1-t0)/500:.4f} ms/step, total {1e3*(t2-t0)/500:.4f} ms/step")
    if timing: print(s.kernel_time_ms(False))
