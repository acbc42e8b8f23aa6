"""Throughput of the generic-tree kernel: full-body GR1T1 (32 DOF) and the lower-limb model forced through it."""
import sys, os, time; sys.path.insert(0, ".")
os.environ.setdefault("GRX_PUBLISH_DEBUG", "0")
import torch
from tests.helpers import make_cfg, make_sims, make_terrain, random_actions
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
for task, force, terrain, N in (("GR1T1Full", "0", "plane", 4096), ("GR1T1Full", "0", "heightfield", 4096), ("GR1T1Full", "0", "heightfield", 16384),
                                ("GR1T1", "1", "heightfield", 4096), ("GR1T1", "0", "heightfield", 4096)):
    if force == "1": os.environ["GRX_FORCE_GENERIC"] = "1"
    else: os.environ.pop("GRX_FORCE_GENERIC", None)
    cfg = make_cfg(task, noise=True, dr=True, push=True, terrain=terrain)
    ter = make_terrain(cfg, N, 1)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
    for i in range(10): s.step(acts[i % 8], 5.0, i + 1)
    s.wait_idle(); s.kernel_time_ms(1)
    t0 = time.perf_counter()
    K = 60
    for i in range(K): s.step(acts[i % 8], 5.0, 11 + i)
    s.wait_idle(); dt = time.perf_counter() - t0
    ms, n = s.kernel_time_ms(0)
    print(f"{task:10s} generic={'forced' if force=='1' else ('yes' if task=='GR1T1Full' else 'no '):6s} {terrain:12s} N={N:6d}: {N*K/dt/1e6:7.2f} M env-steps/s, kernel {ms*1e3:8.1f} us", flush=True)
    s.close()
