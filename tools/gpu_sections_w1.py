"""Section-level cycle profile of the ONE-WAVE layout (grx_step_kernel<HF, 1>, > 16384 envs per GPU) with every SIMD busy:
32768 envs, -DGRX_PROFILE_SECTIONS build (tools/gpu_sections.py --build makes it).  Cycles of a block's wave, medians over blocks."""
import sys, ctypes as C, os; sys.path.insert(0, '.')
import numpy as np, torch
PROF = os.path.abspath(os.environ.get("GRX_PROF_LIB", "wiki-grx-gym_amd/csrc/variants/libgrx_prof.so"))
os.environ["GRX_HIP_LIB"] = PROF
os.environ["GRX_WAVES_PER_BLOCK"] = "1"
os.environ["GRX_LANES_PER_ENV"] = "2"
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = int(os.environ.get("N", 32768))
names = ["load", "substeps", "footkin", "update+heights", "timers", "reward", "reset", "obs", "store", "rows->HBM"]
sub = ["walk + bias + foot contacts", "rare contacts + self-collision", "inertia / bias recursion", "base 6x6", "acceleration pass", "integration"]
for terrain in ("plane", "heightfield"):
    cfg = make_cfg(noise=True, dr=True, push=True, terrain=terrain)
    pass   # (bench.py = the product default: on-demand tensors on refresh)
    ter = make_terrain(cfg, N, 1)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(4)]
    for i in range(int(os.environ.get("STEPS", 200))): s.step(acts[i % 4], 5.0, i + 1)
    torch.cuda.synchronize()
    lib = C.CDLL(PROF); nbmax = N // 32; buf = (C.c_longlong * (nbmax * 96))()
    lib.grx_debug_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    nb = lib.grx_debug_profile(s._h, buf, nbmax)
    full = np.array(buf[:], dtype=np.int64).reshape(nbmax, 96)[:nb]
    a = full[:, :11]
    d = np.diff(a, axis=1)
    tot = a[:, 10] - a[:, 0]
    print(terrain, "blocks", nb, "total cycles median", np.median(tot), "| 5 / 95 / 99 % and max of the blocks:", [int(x) for x in np.percentile(tot, [5, 95, 99, 100])],
          "| first start to last end over the launch:", int(a[:, 10].max() - a[:, 0].min()), "| spread of the starts:", int(a[:, 0].max() - a[:, 0].min()))
    for n, v in zip(names, np.median(d, axis=0)): print(f"   {n:16s} {v:9.0f} ticks")
    print("   sub-step sections, sum over the 10 sub-steps:")
    for n, v in zip(sub, np.median(full[:, 16:22], axis=0)): print(f"      {n:34s} {v:9.0f} ticks")
    r = np.median(full[:, 32:40], axis=0).astype(int).tolist()
    print("   rare contacts, sum over 10 sub-steps: cheap+fine test base lump", r[0], "thigh/shank tests", r[1], "compaction", r[2], "evaluation", r[3], "pick-up + netting", r[4], "| mean candidates per step", full[:, 38].mean(), "calls with any", full[:, 39].mean())
    q = np.mean(full[:, 40:48], axis=0).astype(int).tolist()
    print("   self-collision, sum over 10 sub-steps: cycles", q[0], "candidate envs", q[1], "lanes with a hit", q[2], "candidate groups", q[3], "| centres+extents", q[4], "ballot+staging", q[5], "pair tests", q[6], "forces", q[7])
    print("   obs sub-sections (cycles after tick 7): heights, noise load, side-0 puts:", np.median(full[:, 11:14] - full[:, 7:8], axis=0).astype(int).tolist())
    s.close()
