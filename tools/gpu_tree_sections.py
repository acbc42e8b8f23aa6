"""In-kernel cycle profile of grx_step_tree (the -DGRX_PROFILE_SECTIONS build of tools/gpu_sections.py --build): cycles per policy
step of wave 0 in every section of the ten sub-steps (grx_tree.h TT(i)), full body, rough terrain and plane."""
import sys, ctypes as C, os; sys.path.insert(0, '.')
import numpy as np, torch
PROF = os.path.abspath(os.environ.get("GRX_PROF_LIB", "wiki-grx-gym_amd/csrc/variants/libgrx_prof.so"))
os.environ["GRX_HIP_LIB"] = PROF
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
os.environ["GRX_PUBLISH_DEBUG"] = "0"
names = ["outward", "contacts", "base lump", "self-collision", "inward", "base solve", "accel", "integrate+avg"]
for terrain in ("plane", "heightfield"):
    cfg = make_cfg("GR1T1Full", noise=True, dr=True, push=True, terrain=terrain); N = int(os.environ.get("N", "4096"))
    pass   # (the product default: on-demand tensors on refresh)
    ter = make_terrain(cfg, N, 1)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, N, gen, 0.3).cuda() for _ in range(4)]
    for i in range(40): s.step(acts[i % 4], 5.0, i + 1)
    torch.cuda.synchronize()
    lib = C.CDLL(PROF); buf = (C.c_longlong * (64 * 96))()
    lib.grx_debug_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    nb = lib.grx_debug_profile(s._h, buf, 64)
    full = np.array(buf[:], dtype=np.int64).reshape(64, 96)[:nb]
    med = np.median(full[:, :10], axis=0)
    print(terrain, s.layout(), "cycles per policy step (wave 0, median over blocks): physics", int(med[8]), "whole kernel", int(med[9]))
    for n, v in zip(names, med[:8]): print(f"   {n:16s} {v:9.0f}  ({v / 10:7.0f} per sub-step)")
    m2 = np.median(full[:, 8:14], axis=0)
    print("   behind the sub-steps (cycles since the kernel's start): physics done", int(m2[0]), "final frames / link frames / feet done", int(m2[2]), "state update + height scan done", int(m2[3]), "rewards done", int(m2[4]), "reset done", int(m2[5]), "end", int(m2[1]))
    m4 = np.median(full[:, 20:24], axis=0)
    print("   inside 'outward': joint-local phase", int(m4[0] / 10), "walk", int(m4[1] / 10), "contact probe", int(m4[2] / 10), "(the rest: bias forces of all bodies);  inside 'inward': rigid inertias of all bodies", int(m4[3] / 10), " [cycles per sub-step]")
    lv = np.median(np.diff(full[:, 30:41], axis=1), axis=0); lo = np.median(np.diff(full[:, 44:55], axis=1), axis=0)
    print("   last sub-step, cycles per depth level: inward (leaf level first)", [int(x) for x in lv], " outward walk (level 0 first)", [int(x) for x in lo])
    m3 = np.median(full[:, 14:18], axis=0)
    print("   inside the rewards: per-joint sums done", int(m3[0]), "group sums done", int(m3[1]), "terms done", int(m3[2]), "scaled and summed", int(m3[3]))
    s.close()
