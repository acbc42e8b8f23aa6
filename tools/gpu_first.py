import sys, time; sys.path.insert(0,'.')
import torch
from tests.helpers import *
cfg = make_cfg()
hip, ora = make_sims(cfg, 64)
print("created", flush=True)
rep = compare_step(hip, ora, steps=3, cfg=cfg)
for k,v in rep["worst"].items(): print(k, v)
print("OK" if rep["ok"] else "MISMATCH")
# timing
cfg = make_cfg(noise=True, dr=True, push=True)
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
for N in (4096, 32768):
    c,keep,_ = build_config.build(cfg, cfg.sim.dt, N)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
    for i in range(20): s.step(acts[i%8], 5.0, i+1)
    torch.cuda.synchronize(); t=time.time()
    K=200
    for i in range(K): s.step(acts[i%8], 5.0, 21+i)
    torch.cuda.synchronize(); dt=time.time()-t
    print(f"N={N}: {dt/K*1e6:.1f} us/step  {N*K/dt/1e6:.2f} M env-steps/s  resets/step {s.tensor('RESET').float().mean().item():.4f}")
    s.close()
