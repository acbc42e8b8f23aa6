"""Compile libgrx_hip.so with different flag sets on the GPU box and time the step kernel."""
import subprocess, sys, os
sys.path.insert(0, ".")
base = "-I../../include --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-hip-fp32-correctly-rounded-divide-sqrt"
variants = {
    "default(noslp)": "-fno-slp-vectorize",
    "no-licm-barrier": "-fno-slp-vectorize -DGRX_NO_LICM_BARRIER",
}
if len(sys.argv) > 1:   # name=flags ... on the command line
    variants = {a.split("=", 1)[0]: "-fno-slp-vectorize " + a.split("=", 1)[1] for a in sys.argv[1:]}
_unused = {
    "default(noslp)": "-fno-slp-vectorize",
    "slp": "",
    "noslp max-ilp": "-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp",
    "noslp iterative-ilp": "-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp",
    "noslp max-memory-clause": "-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause",
    "slp max-ilp": "-mllvm -amdgpu-sched-strategy=max-ilp",
}
code = r'''
import sys, os; sys.path.insert(0, ".")
os.environ["GRX_PUBLISH_DEBUG"]="0"
import torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
for waves in os.environ.get("WAVES", "4").split(","):
  os.environ["GRX_WAVES_PER_BLOCK"] = waves
  for terrain in ("plane", "heightfield"):
      cfg = make_cfg(noise=True, dr=True, push=True, terrain=terrain); N=int(os.environ.get("NENVS", "4096"))
      ter = make_terrain(cfg, N, 1)
      c,keep,_ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
      s = HipSim(c, "cuda:0", keep); s.reset_all()
      gen = torch.Generator().manual_seed(0)
      acts=[random_actions(cfg,N,gen,1.0).cuda() for _ in range(8)]
      for i in range(40): s.step(acts[i%8],5.0,i+1)
      s.kernel_time_ms(True)
      for i in range(300): s.step(acts[i%8],5.0,41+i)
      torch.cuda.synchronize()
      ms,n=s.kernel_time_ms(False)
      print(f"   W={waves} {terrain:12s} {ms*1e3:7.1f} us", flush=True)
      s.close()
'''
for name, fl in variants.items():
    r = subprocess.run(f"cd wiki-grx-gym_amd/csrc && hipcc {base} {fl} -shared -o libgrx_hip.so grx_kernels.hip grx_capi.cpp 2>&1 | grep -E ' error' ", shell=True)
    print("==", name, flush=True)
    subprocess.run([sys.executable, "-c", code])
