"""Does the 32-DOF robot of BASELINE.json's fifth configuration stand?  Zero actions (the PD targets are the default joint angles) and
small random actions, plane and rough terrain: episode length, what ends the episodes, which links carry contact force when they end."""
import os, sys; sys.path.insert(0, ".")
import numpy as np, torch
from tests.helpers import *
from wiki_grx_gym_amd.sim import HipSim
from wiki_grx_gym_amd.envs import build_config
N = int(os.environ.get("N", 256)); STEPS = int(os.environ.get("STEPS", 400))
for terrain in ("plane", "heightfield"):
    for scale in (0.0, 0.1, 0.3):
        cfg = make_cfg("GR1T1Full", noise=False, dr=False, push=False, terrain=terrain)
        cfg.env.publish_rigid_body_states = False
        ter = make_terrain(cfg, N, 1)
        c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
        s = HipSim(c, "cuda:0", keep); s.reset_all()
        gen = torch.Generator().manual_seed(0)
        resets = 0; term = 0; tout = 0; first = None; linkhits = None
        for i in range(STEPS):
            a = random_actions(cfg, N, gen, scale).cuda() if scale > 0 else torch.zeros(N, 32, device="cuda")
            s.step(a, 0.0, i + 1)
            r = s.tensor("RESET").bool()
            tc = s.tensor("TERM_CONTACT").bool()
            resets += int(r.sum()); term += int(tc.sum()); tout += int(s.tensor("TIME_OUT").bool().sum())
            if tc.any():
                cf = s.tensor("CONTACT_FORCES")   # (N, links, 3)
                hit = (cf[tc].norm(dim=-1) > 1.0).float().sum(0).cpu().numpy()
                linkhits = hit if linkhits is None else linkhits + hit
                if first is None: first = i
        print(terrain, "action scale", scale, "| env-steps", N * STEPS, "resets", resets, "by contact", term, "by time-out", tout, "first at step", first,
              "| mean steps between resets", round(N * STEPS / max(resets, 1), 1))
        if linkhits is not None: print("    links with |F| > 1 N when an episode ended by contact (count per link index):", {i: int(v) for i, v in enumerate(linkhits) if v > 0})
        s.close()
