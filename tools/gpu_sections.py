"""Section-level cycle profile of grx_step_kernel (build with -DGRX_PROFILE_SECTIONS)."""
import sys, ctypes as C, subprocess, os; sys.path.insert(0,'.')
import numpy as np, torch
csrc="wiki-grx-gym_amd/csrc"
flags="-I../../include --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-hip-fp32-correctly-rounded-divide-sqrt -ffinite-math-only -fno-signed-zeros -fno-trapping-math -fassociative-math -fno-slp-vectorize -DGRX_REG_CONSTS=2 -mllvm -amdgpu-sched-strategy=iterative-ilp -DGRX_PROFILE_SECTIONS"
os.makedirs(csrc + "/variants", exist_ok=True)
PROF = os.path.abspath(csrc + "/variants/libgrx_prof.so")
if "--build" in sys.argv or not os.path.exists(PROF):   # hipcc cross-compiles in the build container; the .so travels with gpurun
    subprocess.run(f"cd {csrc} && hipcc {flags} -shared -o variants/libgrx_prof.so grx_kernels.hip grx_quad.hip grx_tree16.hip grx_capi.cpp 2>/dev/null", shell=True, check=True)
    if "--build" in sys.argv:
        sys.exit(0)
PROF = os.path.abspath(os.environ.get("GRX_PROF_LIB", PROF))
os.environ["GRX_HIP_LIB"] = PROF
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim, load_hip_library
from wiki_grx_gym_amd.envs import build_config
os.environ["GRX_PUBLISH_DEBUG"]="0"
names=["load","substeps","footkin","update+heights","timers","reward","reset","obs","store","rows->HBM"]
for terrain in ("plane","heightfield"):
    cfg = make_cfg(noise=True, dr=True, push=True, terrain=terrain, task=os.environ.get("TASK", "GR1T1")); N=int(os.environ.get("N", 4096))
    pass   # (bench.py = the product default: on-demand tensors on refresh)
    ter = make_terrain(cfg, N, 1)
    c,keep,_ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts=[random_actions(cfg,N,gen,1.0).cuda() for _ in range(4)]
    for i in range(40): s.step(acts[i%4],5.0,i+1)
    torch.cuda.synchronize()
    lib=C.CDLL(PROF); buf=(C.c_longlong*(1024*96))()
    lib.grx_debug_profile.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
    nb=lib.grx_debug_profile(s._h, buf, 1024)
    full=np.array(buf[:],dtype=np.int64).reshape(1024,96)[:nb]
    a=full[:,:11]
    print('   wave 0, sum over 10 sub-steps:', dict(zip(['wait bias forces','barrier after the sub-steps','wait foot / rare contacts','wait self-collision','wait rigid inertias','whole sub-steps'], np.median(full[:,16:22],axis=0).astype(int).tolist())))
    print('   helper waves (idle waiting for state, total) cycles:', {f"wave{w}": np.median(full[:,22+2*w:24+2*w],axis=0).astype(int).tolist() for w in (1,2,3)})
    print('   wave 3 rare contacts (sum over 10 sub-steps): cheap test, fine test, compaction, evaluation, pick-up + netting, -, candidates, calls with any:', np.median(full[:,32:40],axis=0).astype(int).tolist(), 'mean candidates', full[:,38].mean())
    print('   wave 1 self-collision (sum over 10 sub-steps): cycles, candidate envs, lanes with a hit, candidate groups | cycles: centres+extents, ballot+staging, pair tests, forces:', np.mean(full[:,40:48],axis=0).astype(int).tolist())
    ev = full[:, 48:63] - full[:, 48:49]
    names_ev = ['w0 start', 'w0 walk done', 'w0 inertia half + base factorised', 'w0 recursion done', 'w0 got foot', 'w0 got rare', 'w0 sub-step end', 'w0 bias half done', 'w2 frames out', 'w2 bias[4] out', 'w2 bias[0] out', 'w2 foot out', 'w3 rare out', 'w1 self out', 'w0 base assembled']
    print('   timeline of sub-step 5 (cycles after wave 0 starts it):', {n: int(v) for n, v in zip(names_ev, np.median(ev, axis=0)) if n != '-'})
    print('   w0 base factorised at', int(np.median(full[:,79]-full[:,48])))
    print('   arrival at the barrier that ends the sub-steps, waves 0..7, relative to wave 0:', np.median(full[:,80:88]-full[:,80:81],axis=0).astype(int).tolist())
    print('   wave 3, last sub-step, relative to wave 0 at the barrier: rare out, inputs of the link rows in, link rows written:', np.median(full[:,88:91]-full[:,80:81],axis=0).astype(int).tolist())
    if os.environ.get("GRX_QUAD_WAVES") == "8":
        print('   eight waves: w4 bias out, w6 rigid inertias out, w6 bias out, w5 factorisation out, w5 got X Y:', np.median(full[:,74:79]-full[:,48:49],axis=0).astype(int).tolist())
    if os.environ.get("GRX_LANES_PER_ENV") == "2" and os.environ.get("GRX_WAVES_PER_BLOCK") == "8":
        print('   lane pairs, eight waves: w0 reaches the wait for wave 5\'s rigid inertias, w0 has them:', np.median(full[:,91:93]-full[:,48:49],axis=0).astype(int).tolist())
    print('   wave 0: duration of each of the 10 sub-steps:', np.median(full[:, 64:74], axis=0).astype(int).tolist())
    print('   obs sub-sections (cycles after tick 7): heights, noise load, side-0 puts:', np.median(full[:,11:14]-full[:,7:8],axis=0).astype(int).tolist())
    print('   relative to tick 6 (FL_REW published): wave1 got FL_REW, wave1 rewards done, wave2 got FL_HZ, wave2 heights done, wave0 tick 9:', np.median(full[:,[14,15,30,31,9]]-full[:,6:7],axis=0).astype(int).tolist())
    print('   wave 0, store section (cycles after tick 8): leg columns stored, env columns stored, tick 9 (height-block sums in):', np.median(full[:,[93,94,9]]-full[:,8:9],axis=0).astype(int).tolist())
    d=np.diff(a,axis=1)
    print(terrain, "total cycles median", np.median(a[:,10]-a[:,0]))
    for n,v in zip(names, np.median(d,axis=0)): print(f"   {n:16s} {v:9.0f} ticks")
    s.close()
