import sys, time; sys.path.insert(0,'.')
import torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
def run(label, N=4096, terrain="plane", dec=10, z=None, grav=True, noise=True, mh=True, steps=200):
    cfg = make_cfg(noise=noise, dr=True, push=True, terrain=terrain)
    cfg.control.decimation = dec
    cfg.terrain.measure_heights = mh
    if not mh: cfg.env.num_pri_obs = 47
    if z is not None: cfg.init_state.pos = [0,0,z]
    if not grav: cfg.sim.gravity=[0,0,0]
    ter = make_terrain(cfg, N, 1)
    c,keep,_ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    s = HipSim(c, "cuda:0", keep); s.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
    for i in range(30): s.step(acts[i%8], 5.0, i+1)
    s.kernel_time_ms(True)
    for i in range(steps): s.step(acts[i%8], 5.0, 31+i)
    torch.cuda.synchronize()
    ms,n = s.kernel_time_ms(False)
    print(f"{label:40s} kernel {ms*1e3:7.1f} us  contacts {s.tensor('FEET_CONTACT').float().mean().item():.2f}")
    s.close()
import os
os.environ["GRX_PUBLISH_DEBUG"]="0"
run("flat dec10")
run("flat dec1", dec=1)
run("flat dec10 no-contact (z=50,g=0)", z=50., grav=False)
run("flat dec1 no-contact", dec=1, z=50., grav=False)
run("flat dec10 nonoise", noise=False)
run("rough dec10", terrain="heightfield")
run("rough dec1", terrain="heightfield", dec=1)
run("rough dec10 no-contact", terrain="heightfield", z=50., grav=False)
run("flat dec10 N=32768", N=32768)
run("rough dec10 N=32768", N=32768, terrain="heightfield")
