#!/usr/bin/env python3
"""SURVEY 8d baseline (i) / BASELINE.md section 3 "R-py": the REFERENCE's own Python obs / reward / termination stage
(LeggedRobot.post_physics_step: legged_robot.py:269-481, legged_robot_fftai.py:90-167, gr1t1.py:281-589) timed on
torch-CPU in the BUILD CONTAINER (the reference never travels to the GPU box).  Same stub import as tools/gen_golden.py:
Isaac Gym / PhysX is absent, so the physics half of the step is not in this number -- it covers rows B4-B15 only.

    python tools/time_ref_pipeline.py            # writes profiles/r02_ref_python_pipeline.json
"""
import json
import os
import platform
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_stage(N, reps, terrain_obj=None):
    env, g, body_names = gg.make_ref_env(N, 5, terrain_obj=terrain_obj)
    env.cfg.domain_rand.push_robots = False
    gg.randomize_state(env, g, body_names, N, 0)
    env.reset_idx = lambda ids: None     # reset_idx needs gym.set_*_tensor_indexed: mocked out, rows stay un-reset
    for _ in range(3):
        env.post_physics_step()
    t0 = time.perf_counter()
    for _ in range(reps):
        env.post_physics_step()
    dt = (time.perf_counter() - t0) / reps
    return dt


def main():
    gg.install_stub()
    from legged_gym.utils.terrain import Terrain
    from legged_gym.envs.base.legged_robot_config import LeggedRobotCfg
    import numpy as np
    threads = torch.get_num_threads()
    out = {"what": "reference Python obs/reward/termination stage (post_physics_step), torch-CPU, stubbed sim (no physics)",
           "where": "build container", "cpu_model": cpu_model(), "cores": os.cpu_count(), "torch_threads": threads,
           "torch": torch.__version__, "rows": []}
    tcfg = LeggedRobotCfg.terrain()
    tcfg.mesh_type = "heightfield"
    np.random.seed(1)
    ter = Terrain(tcfg, 4096)
    for N, reps in ((64, 200), (4096, 40)):
        for name, tobj in (("plane", None), ("rough heightfield (121-point scan)", ter)):
            dt = time_stage(N, reps, tobj)
            row = {"num_envs": N, "terrain": name, "ms_per_call": dt * 1e3, "env_steps_per_s_stage_only": N / dt, "reps": reps}
            print(row, flush=True)
            out["rows"].append(row)
    path = os.path.join(ROOT, "profiles", "r02_ref_python_pipeline.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
