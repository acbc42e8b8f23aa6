"""SHA-256 over every published tensor after K policy steps of a seeded rollout (random actions, DR + noise + pushes): two builds of the library that
claim bit-identical results must print the same digests.  usage: [GRX_HIP_LIB=...] python tools/state_digest.py [envs=4096] [steps=150] [terrain=heightfield] [task=GR1T1]"""
import hashlib, os, sys
sys.path.insert(0, ".")
import torch
from tests.helpers import make_cfg, make_terrain, random_actions
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 150
terrain = sys.argv[3] if len(sys.argv) > 3 else "heightfield"
task = sys.argv[4] if len(sys.argv) > 4 else "GR1T1"
cfg = make_cfg(noise=True, dr=True, push=True, terrain=terrain, task=task)
ter = make_terrain(cfg, N, 1)
c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
s = HipSim(c, "cuda:0", keep); s.reset_all()
gen = torch.Generator().manual_seed(0)
acts = [random_actions(cfg, N, gen, 1.0).cuda() for _ in range(8)]
h = hashlib.sha256()
for i in range(K):
    s.step(acts[i % 8], 5.0, i + 1)
    if i % 10 == 9 or i == K - 1:
        for name in ("DOF_POS", "DOF_VEL", "ROOT_STATES", "TORQUES", "OBS", "PRI_OBS", "REW", "RESET", "FEET_CONTACT_FORCE", "CONTACT_FORCES"):
            h.update(s.tensor(name).contiguous().cpu().numpy().tobytes())
torch.cuda.synchronize()
print(s.layout()["kernel"], N, K, terrain, task, h.hexdigest()[:32], "resets", int(s.tensor("RESET").sum().item()))
s.close()
