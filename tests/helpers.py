"""Shared helpers of the test-suite: config variants, HIP-vs-oracle comparison."""
import numpy as np
import torch

from wiki_grx_gym_amd.envs import build_config, config

CMP_TENSORS = ("DOF_POS", "DOF_VEL", "ROOT_STATES", "TORQUES", "ACTIONS", "COMMANDS", "BASE_LIN_VEL", "BASE_ANG_VEL",
               "PROJECTED_GRAVITY", "FEET_CONTACT_FORCE", "FEET_POS", "FEET_HEIGHT", "FEET_AIR_TIME", "FEET_LAND_TIME",
               "AVG_FEET_FORCE", "AVG_FEET_SPEED", "BASE_HEIGHTS_OFFSET", "REW", "OBS", "PRI_OBS", "LAST_ACTIONS",
               "LAST_DOF_VEL", "EPISODE_SUMS", "REWARD_TERMS", "MEASURED_HEIGHTS")
CMP_EXACT = ("RESET", "TIME_OUT", "EPISODE_LENGTH", "FEET_CONTACT", "TERRAIN_LEVELS", "TERRAIN_TYPES")


def make_cfg(task="GR1T1", noise=False, dr=False, push=None, terrain="plane", curriculum=True):
    cfg = {"GR1T1": config.GR1T1Cfg, "GR1T2": config.GR1T2Cfg}[task]()
    d = cfg.domain_rand
    if not dr:
        d.randomize_friction = d.randomize_restitution = d.randomize_base_mass = d.randomize_base_com = False
        d.randomize_motor_strength = False
        d.randomize_init_dof_pos = d.randomize_init_base_velocity = False
    d.push_robots = d.push_robots if push is None else push
    if not dr and push is None:
        d.push_robots = False
    cfg.noise.add_noise = bool(noise)
    cfg.terrain.mesh_type = terrain
    cfg.terrain.curriculum = curriculum
    return cfg


def make_terrain(cfg, num_envs, seed=1):
    if cfg.terrain.mesh_type == "plane":
        return None
    from wiki_grx_gym_amd.utils.terrain import Terrain
    return Terrain(cfg.terrain, num_envs, seed=seed)


def make_sims(cfg, num_envs, device="cuda:0", precision="f32", seed=1, env_offset=0, total_envs=None, hip=True, terrain=None):
    """(HipSim or None, OracleSim) built from the same config."""
    from oracle.binding import OracleSim
    if terrain is None:
        terrain = make_terrain(cfg, total_envs or num_envs, seed)
    c1, keep1, _ = build_config.build(cfg, cfg.sim.dt, num_envs, env_offset, total_envs, seed, terrain)
    ora = OracleSim(c1, precision, keep1)
    h = None
    if hip:
        from wiki_grx_gym_amd.sim import HipSim
        c2, keep2, _ = build_config.build(cfg, cfg.sim.dt, num_envs, env_offset, total_envs, seed, terrain)
        h = HipSim(c2, device, keep2)
    return h, ora


def random_actions(cfg, num_envs, gen, scale=1.0):
    lo = torch.tensor(np.asarray(cfg.normalization.clip_actions_min, dtype=np.float32))
    hi = torch.tensor(np.asarray(cfg.normalization.clip_actions_max, dtype=np.float32))
    u = torch.rand(num_envs, lo.numel(), generator=gen)
    mid, half = (lo + hi) / 2, (hi - lo) / 2
    return (mid + (2 * u - 1) * half * scale).contiguous()


def tensor_diff(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs()
    tol = 1e-4 + 1e-4 * b.abs()
    return float(err.max()) if err.numel() else 0.0, float((err > tol).double().mean()) if err.numel() else 0.0


def compare_step(hip, ora, steps=1, seed=0, cfg=None, action_scale=0.3, delay=5.0, atol_frac=0.0, start_counter=1,
                 noise=False):
    """Reset both, step both with identical actions, compare every exposed tensor.
    Returns a dict with per-tensor max abs error and the fraction of elements outside 1e-4 (abs+rel)."""
    cfg = cfg or make_cfg()
    gen = torch.Generator().manual_seed(seed)
    N = ora.num_envs
    hip.reset_all()
    ora.reset_all()
    report = {"ok": True, "worst": {}}
    for s in range(steps):
        a = random_actions(cfg, N, gen, action_scale)
        nz = torch.rand(N, 39, generator=gen).contiguous() if noise else None
        ora.step(a, delay, start_counter + s, nz)
        hip.step(a.to(hip.device), delay, start_counter + s, nz.to(hip.device) if nz is not None else None)
        torch.cuda.synchronize()
        for name in CMP_TENSORS:
            mx, frac = tensor_diff(hip.tensor(name), ora.tensor(name))
            prev = report["worst"].get(name, (0.0, 0.0))
            report["worst"][name] = (max(prev[0], mx), max(prev[1], frac))
            if frac > atol_frac:
                report["ok"] = False
        for name in CMP_EXACT:
            same = torch.equal(hip.tensor(name).cpu().to(torch.int64), ora.tensor(name).to(torch.int64))
            if not same:
                report["ok"] = False
                report["worst"][name] = ("mismatch", float((hip.tensor(name).cpu().to(torch.int64) != ora.tensor(name).to(torch.int64)).double().mean()))
    return report
