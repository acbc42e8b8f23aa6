"""Shared helpers of the test-suite: config variants, HIP-vs-oracle comparison."""
import numpy as np
import torch

from wiki_grx_gym_amd.envs import build_config, config

CMP_TENSORS = ("DOF_POS", "DOF_VEL", "ROOT_STATES", "TORQUES", "ACTIONS", "COMMANDS", "BASE_LIN_VEL", "BASE_ANG_VEL",
               "PROJECTED_GRAVITY", "FEET_CONTACT_FORCE", "CONTACT_FORCES", "FEET_POS", "FEET_HEIGHT", "FEET_AIR_TIME", "FEET_LAND_TIME",
               "AVG_FEET_FORCE", "AVG_FEET_SPEED", "BASE_HEIGHTS_OFFSET", "REW", "OBS", "PRI_OBS", "LAST_ACTIONS",
               "LAST_DOF_VEL", "EPISODE_SUMS", "REWARD_TERMS", "MEASURED_HEIGHTS")
CMP_EXACT = ("RESET", "TIME_OUT", "EPISODE_LENGTH", "FEET_CONTACT", "TERRAIN_LEVELS", "TERRAIN_TYPES")


def make_cfg(task="GR1T1", noise=False, dr=False, push=None, terrain="plane", curriculum=True):
    cfg = {"GR1T1": config.GR1T1Cfg, "GR1T2": config.GR1T2Cfg, "GR1T1Full": config.GR1T1FullBodyCfg}[task]()
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


ATOL = {"TORQUES": 2e-3, "FEET_CONTACT_FORCE": 1e-2, "CONTACT_FORCES": 1e-2, "AVG_FEET_FORCE": 1e-2}   # N m / N scales: kp * 1e-6 rad etc.


def tensor_diff(a, b, name=None):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs()
    tol = ATOL.get(name, 1e-4) + 1e-4 * b.abs()
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


STATE_TENSORS = ("DOF_POS", "DOF_VEL", "ROOT_STATES", "ANCHORS", "LAST_ACTIONS", "LAST_DOF_VEL", "COMMANDS", "FEET_AIR_TIME",
                 "FEET_LAND_TIME", "FEET_CONTACT", "BASE_HEIGHTS_OFFSET", "EPISODE_LENGTH", "EPISODE_SUMS", "ENV_ORIGINS",
                 "TERRAIN_LEVELS")


def sync_state(hip, ora):
    """Overwrite the complete HIP simulation state with the oracle's (the library buffers are
    writable zero-copy views: the set_*_tensor role of the reference)."""
    for name in STATE_TENSORS:
        src = ora.tensor(name)
        hip.tensor(name).copy_(src.to(hip.device))


def lockstep(hip, ora, cfg, steps, seed=0, scale=0.3, delay=5.0, noise=False, resync=True, start=1, check=None):
    """Step both; with resync the HIP state is reset to the oracle's before every step so that
    each comparison is a ONE-STEP comparison from identical state (chaotic divergence of the
    contact dynamics is tested separately)."""
    gen = torch.Generator().manual_seed(seed)
    N = ora.num_envs
    worst = {}
    for s in range(steps):
        if resync and s > 0:
            sync_state(hip, ora)
        a = random_actions(cfg, N, gen, scale)
        nz = torch.rand(N, 39, generator=gen).contiguous() if noise else None
        ora.step(a, delay, start + s, nz)
        hip.step(a.to(hip.device), delay, start + s, nz.to(hip.device) if nz is not None else None)
        torch.cuda.synchronize()
        for name in CMP_TENSORS:
            mx, frac = tensor_diff(hip.tensor(name), ora.tensor(name), name)
            w = worst.get(name, (0.0, 0.0))
            worst[name] = (max(w[0], mx), max(w[1], frac))
        for name in CMP_EXACT:
            a_, b_ = hip.tensor(name).cpu().to(torch.int64), ora.tensor(name).to(torch.int64)
            w = worst.get(name, (0.0, 0.0))
            worst[name] = (0.0, max(w[1], float((a_ != b_).double().mean())))
        if check:
            check(s, hip, ora)
    return worst


def quat_to_R_np(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


PRE_KEYS = ("LAST_ACTIONS", "LAST_DOF_VEL", "COMMANDS", "FEET_AIR_TIME", "FEET_LAND_TIME", "FEET_CONTACT",
            "BASE_HEIGHTS_OFFSET", "EPISODE_LENGTH", "EPISODE_SUMS")
POST_KEYS = ("DOF_POS", "DOF_VEL", "ROOT_STATES", "ACTIONS", "TORQUES", "FEET_CONTACT_FORCE", "FEET_POS", "AVG_FEET_FORCE",
             "AVG_FEET_SPEED", "TERM_CONTACT", "MEASURED_HEIGHTS")


def oracle_pipeline_on_hip_state(ora, pre, post, common_step_counter, noise=None, plane=True):
    """Feed the HIP step's OWN post-physics state (q, qd, root, contact forces, ...) and the pre-step
    history into the oracle's post_physics_step restatement: the 'obs/reward on identical (q, qd,
    actions)' comparison of the north star, independent of the physics."""
    from oracle.binding import PipelineState
    N, nd = ora.num_envs, ora.num_dofs
    f = {k: v.detach().cpu().numpy() for k, v in {**pre, **post}.items()}
    ora.tensor("EPISODE_SUMS")  # make sure views exist
    for i in range(N):
        ps = PipelineState()
        for j in range(nd):
            ps.q[j] = f["DOF_POS"][i, j]; ps.qd[j] = f["DOF_VEL"][i, j]; ps.actions[j] = f["ACTIONS"][i, j]
            ps.last_actions[j] = f["LAST_ACTIONS"][i, j]; ps.last_last_actions[j] = f["LAST_ACTIONS"][i, j]
            ps.last_dof_vel[j] = f["LAST_DOF_VEL"][i, j]; ps.torques[j] = f["TORQUES"][i, j]
        for k in range(13):
            ps.root[k] = f["ROOT_STATES"][i, k]
        for k in range(3):
            ps.commands[k] = f["COMMANDS"][i, k]
        for ft in range(2):
            ps.air_time[ft] = f["FEET_AIR_TIME"][i, ft]; ps.land_time[ft] = f["FEET_LAND_TIME"][i, ft]
            ps.contact_last[ft] = int(f["FEET_CONTACT"][i, ft]); ps.avg_force[ft] = f["AVG_FEET_FORCE"][i, ft]
            for k in range(3):
                ps.feet_force[ft][k] = f["FEET_CONTACT_FORCE"][i, ft, k]; ps.feet_pos[ft][k] = f["FEET_POS"][i, ft, k]
                ps.avg_speed[ft][k] = f["AVG_FEET_SPEED"][i, ft, k]
        R = quat_to_R_np(f["ROOT_STATES"][i, 3:7]).reshape(-1)
        for k in range(9):
            ps.torso_R[k] = R[k]
        if plane:
            for k in range(f["MEASURED_HEIGHTS"].shape[1]):
                ps.heights[k] = 0.0
        ps.base_heights_offset = float(f["BASE_HEIGHTS_OFFSET"][i])
        ps.episode_length = int(f["EPISODE_LENGTH"][i])
        ps.term_contact = int(f["TERM_CONTACT"][i])
        ora.post_physics(i, ps, apply_reset=False, common_step_counter=common_step_counter, noise_uniform=noise)
