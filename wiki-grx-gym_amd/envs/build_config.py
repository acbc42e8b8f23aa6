"""Environment config (class tree) + robot model -> the C-ABI ``grx_config`` struct.

This is the host-side counterpart of what the reference spreads over ``_parse_cfg``
(legged_robot.py:91-104), ``_init_buffers`` (legged_robot.py:106-203: PD gains by name substring),
``_process_dof_props`` (582-616), ``_init_buffers_joint_indices`` (gr1t1.py:127-279),
``_init_height_points`` (legged_robot.py:1219-1233) and ``_prepare_reward_function`` (840-866).
"""
import ctypes as C
import math

import numpy as np

from .. import _capi
from ..model import RobotModel, asset_key_from_file, fill_model, _mask


def _set(arr, values):
    for i, v in enumerate(values):
        arr[i] = v


def resolve_gains(cfg, dof_names):
    """PD gains / default angles per DOF by name-substring match (legged_robot.py:176-192)."""
    kp, kd, q0 = [], [], []
    for name in dof_names:
        q0.append(float(cfg.init_state.default_joint_angles[name]))
        p = d = 0.0
        for key in cfg.control.stiffness.keys():
            if key in name:
                p = float(cfg.control.stiffness[key])
                d = float(cfg.control.damping[key])
        kp.append(p)
        kd.append(d)
    return kp, kd, q0


def publish_mode(v):
    import os
    if v is False or v is None or v == 0 or v == "never":
        return _capi.PUBLISH_NEVER
    # (True == 1 == PUBLISH_EVERY_STEP in Python: the bool is tested by identity first -- True means "publish", i.e. on refresh)
    every = v == "every_step" or (v is not True and v == _capi.PUBLISH_EVERY_STEP)
    on_refresh = v is True or v == "on_refresh" or v == _capi.PUBLISH_ON_REFRESH
    if every or (on_refresh and os.environ.get("GRX_PUBLISH_EVERY_STEP") == "1"):   # (the environment knob turns a published tensor into a step-written one)
        return _capi.PUBLISH_EVERY_STEP
    if on_refresh:
        return _capi.PUBLISH_ON_REFRESH
    raise ValueError(f"publish mode {v!r}: True / 'on_refresh', 'every_step' or False")


def dof_armature(value, dof_names):
    """cfg.asset.armature -> one value per DOF.  A number is the reference's asset_options.armature (legged_robot.py:958,
    legged_robot_config.py:125): every DOF gets it.  A dict {substring of the joint name: kg m^2} -- the form cfg.control.stiffness /
    damping take (legged_robot.py:1060-1072) -- names the joints that get one; the others keep the reference's default, 0."""
    if isinstance(value, dict):
        out = []
        for n in dof_names:
            hit = [v for k, v in value.items() if k in n]
            if len(hit) > 1 and len(set(hit)) > 1:
                raise ValueError(f"cfg.asset.armature: joint '{n}' matches several keys with different values")
            out.append(float(hit[0]) if hit else 0.0)
        return out
    return [float(value)] * len(dof_names)


def build(cfg, sim_dt, num_envs, env_offset=0, total_envs=None, seed=1, terrain=None):
    """Returns (cfg_struct, keepalive, meta).  ``terrain``: utils.terrain.Terrain or None."""
    total_envs = num_envs if total_envs is None else total_envs
    rm = RobotModel(getattr(cfg.asset, "model", None) or asset_key_from_file(cfg.asset.file))
    nd = rm.num_dofs
    if cfg.env.num_actions != nd:
        raise ValueError(f"cfg.env.num_actions={cfg.env.num_actions} but asset has {nd} DOFs")
    c = _capi.Config()
    c.abi_version = _capi.GRX_ABI_VERSION
    c.struct_size = C.sizeof(_capi.Config)
    c.num_envs, c.env_offset, c.total_envs = int(num_envs), int(env_offset), int(total_envs)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    g = cfg.sim.grx
    meta = fill_model(c.model, rm, cfg.asset.foot_name, cfg.asset.torso_name,
                      getattr(cfg.asset, "forehead_name", ""), cfg.asset.terminate_after_contacts_on,
                      cfg.asset.penalize_contacts_on, damp_alpha=g.damp_alpha, sim_dt=sim_dt,
                      armature=dof_armature(getattr(cfg.asset, "armature", 0.0), rm.dof_names))
    meta["model"] = rm
    c.contact.kn, c.contact.dn, c.contact.kt, c.contact.ct, c.contact.cv = g.kn, g.dn, g.kt, g.ct, g.cv
    c.contact.k_limit, c.contact.c_limit, c.contact.damp_alpha = g.k_limit, g.c_limit, g.damp_alpha
    c.contact.terrain_friction = cfg.terrain.static_friction
    c.sim_dt = float(sim_dt)
    c.decimation = int(cfg.control.decimation)
    _set(c.gravity, cfg.sim.gravity)
    if cfg.control.control_type not in _capi.CONTROL_TYPES:
        raise NameError(f"Unknown controller type: {cfg.control.control_type}")   # (the reference's own error, legged_robot.py:707)
    c.control_type = _capi.CONTROL_TYPES[cfg.control.control_type]
    kp, kd, q0 = resolve_gains(cfg, rm.dof_names)
    _set(c.kp, kp); _set(c.kd, kd); _set(c.default_dof_pos, q0)
    c.action_scale = float(cfg.control.action_scale)
    _set(c.clip_actions_min, np.asarray(cfg.normalization.clip_actions_min, dtype=np.float32))
    _set(c.clip_actions_max, np.asarray(cfg.normalization.clip_actions_max, dtype=np.float32))
    dt = cfg.control.decimation * sim_dt
    c.max_episode_length_s = float(cfg.env.episode_length_s)
    c.max_episode_length = float(np.ceil(cfg.env.episode_length_s / dt))
    c.resample_command_interval = int(cfg.commands.resampling_command_interval_s / dt)
    r = cfg.commands.ranges
    _set(c.cmd_lin_vel_x, r.lin_vel_x); _set(c.cmd_lin_vel_y, r.lin_vel_y); _set(c.cmd_ang_vel_yaw, r.ang_vel_yaw)
    c.heading_command = int(bool(cfg.commands.heading_command))   # legged_robot.py:320-326 (GRx tasks: False, gr1t1_config.py:151)
    _set(c.init_pos, cfg.init_state.pos); _set(c.init_rot, cfg.init_state.rot)
    _set(c.init_lin_vel, cfg.init_state.lin_vel); _set(c.init_ang_vel, cfg.init_state.ang_vel)
    dr = cfg.domain_rand
    c.randomize_friction = int(dr.randomize_friction); _set(c.friction_range, dr.friction_range)
    c.randomize_restitution = int(dr.randomize_restitution); _set(c.restitution_range, dr.restitution_range)
    c.terrain_restitution = float(cfg.terrain.restitution)
    c.bounce_threshold_velocity = float(getattr(cfg.sim.physx, "bounce_threshold_velocity", 0.5))
    c.self_collisions = int(cfg.asset.self_collisions == 0)   # the reference's flag is a bitwise FILTER: 0 = collide (legged_robot_config.py:121)
    c.randomize_base_mass = int(dr.randomize_base_mass); _set(c.base_mass_range, dr.multiply_base_mass_range)
    c.randomize_base_com = int(dr.randomize_base_com)
    for k, rng in enumerate((dr.add_base_com_range_x, dr.add_base_com_range_y, dr.add_base_com_range_z)):
        c.base_com_range[k][0], c.base_com_range[k][1] = rng
    c.randomize_motor_strength = int(dr.randomize_motor_strength); _set(c.motor_strength_range, dr.multiply_motor_strength)
    c.push_robots = int(dr.push_robots)
    c.push_interval = int(np.ceil(dr.push_interval_s / dt))
    c.max_push_vel_xy = float(dr.max_push_vel_xy)
    c.randomize_init_dof_pos = int(dr.randomize_init_dof_pos)
    c.randomize_init_base_velocity = int(dr.randomize_init_base_velocity)
    # rewards: scale / sigma per term, by name
    rw = cfg.rewards
    scales = {k: v for k, v in vars_of(rw.scales).items()}
    unknown = [k for k, v in scales.items() if v != 0 and k not in _capi.REWARD_TERMS]
    if unknown:
        raise ValueError(f"reward terms without an implementation: {unknown}")
    for t, name in enumerate(_capi.REWARD_TERMS):
        c.reward_scale[t] = float(scales.get(name, 0.0))
        c.reward_sigma[t] = float(getattr(rw, "sigma_" + name, 0.0))
    c.only_positive_rewards = int(rw.only_positive_rewards)
    c.base_height_target = rw.base_height_target
    c.swing_feet_height_target = getattr(rw, "swing_feet_height_target", 0.1)
    c.feet_stumble_ratio = getattr(rw, "feet_stumble_ratio", 5.0)
    c.feet_air_time_target = getattr(rw, "feet_air_time_target", 0.5)
    c.feet_land_time_max = getattr(rw, "feet_land_time_max", 1.0)
    c.soft_dof_pos_limit, c.soft_dof_vel_limit, c.soft_torque_limit = rw.soft_dof_pos_limit, rw.soft_dof_vel_limit, rw.soft_torque_limit
    a = cfg.asset
    c.knee_mask = _mask(rm.dofs_containing(getattr(a, "knee_name", "knee")))
    c.hip_roll_mask = _mask(rm.dofs_containing(getattr(a, "hip_roll_name", "hip_roll")))
    c.hip_yaw_mask = _mask(rm.dofs_containing(getattr(a, "hip_yaw_name", "hip_yaw")))
    ankle = rm.dofs_containing(getattr(a, "ankle_name", "ankle"))
    c.ankle_left_mask = _mask(ankle[:len(ankle) // 2])   # gr1t1.py:409,413: first / second half
    c.ankle_right_mask = _mask(ankle[len(ankle) // 2:])
    meta["ankle_indices"] = ankle
    # observations
    c.num_obs = int(cfg.env.num_obs)
    c.num_pri_obs = int(cfg.env.num_pri_obs) if cfg.env.num_pri_obs is not None else 0
    s = cfg.normalization.obs_scales
    c.obs_scale_action, c.obs_scale_lin_vel, c.obs_scale_ang_vel = s.action, s.lin_vel, s.ang_vel
    c.obs_scale_gravity, c.obs_scale_dof_pos, c.obs_scale_dof_vel = s.gravity, s.dof_pos, s.dof_vel
    c.obs_scale_height = s.height_measurements
    n = cfg.noise
    c.add_noise = int(n.add_noise); c.noise_level = n.noise_level
    ns = n.noise_scales
    c.noise_action, c.noise_lin_vel, c.noise_ang_vel, c.noise_gravity = ns.action, ns.lin_vel, ns.ang_vel, ns.gravity
    c.noise_dof_pos, c.noise_dof_vel, c.noise_height = ns.dof_pos, ns.dof_vel, ns.height_measurements
    c.clip_observations = cfg.normalization.clip_observations
    c.termination_force = 1.0       # legged_robot.py:341
    c.termination_gravity_z = 0.33  # legged_robot.py:347
    # terrain
    t = cfg.terrain
    keep = []
    c.measure_heights = int(t.measure_heights)
    pts = [(x, y) for x in t.measured_points_x for y in t.measured_points_y]  # meshgrid(x, y) 'ij' flatten
    if len(pts) > _capi.MAX_HEIGHT_POINTS:
        raise ValueError("too many height measurement points")
    c.num_height_points = len(pts)
    for k, (x, y) in enumerate(pts):
        c.height_points[k][0], c.height_points[k][1] = x, y
    c.env_spacing = float(cfg.env.env_spacing)
    c.publish_reward_terms = int(getattr(cfg.env, "publish_reward_terms", True))   # make_env turns it off
    # tensors nobody needs on every step (include/grx.h grx_publish_mode): True / "on_refresh" -- the default -- = materialised when somebody
    # reads them (sim.tensor / env.rigid_body_states / env.measured_heights call grx_refresh: the gym.refresh_*_tensor model), "every_step" =
    # written by the step kernel (1.9 KB + 0.5 KB per env-step), False = no such tensor (rigid_body_states only).  GRX_PUBLISH_EVERY_STEP=1
    # forces the step-written mode (A/B runs, tests)
    c.publish_rigid_body_states = publish_mode(getattr(cfg.env, "publish_rigid_body_states", True))
    c.publish_measured_heights = publish_mode(getattr(cfg.env, "publish_measured_heights", True)) or _capi.PUBLISH_EVERY_STEP
    c.horizontal_scale, c.vertical_scale, c.border_size = t.horizontal_scale, t.vertical_scale, t.border_size
    c.terrain_length = t.terrain_length
    if t.mesh_type == "plane":
        c.terrain_type = _capi.TERRAIN_PLANE
        c.curriculum = 0
    elif t.mesh_type in ("heightfield", "trimesh"):
        if terrain is None:
            raise ValueError("heightfield terrain requested but no Terrain object given")
        c.terrain_type = _capi.TERRAIN_HEIGHTFIELD
        hs = np.ascontiguousarray(terrain.heightsamples, dtype=np.int16)
        org = np.ascontiguousarray(terrain.env_origins, dtype=np.float32)
        keep += [hs, org]
        c.height_samples = hs.ctypes.data
        c.hf_rows, c.hf_cols = hs.shape
        c.terrain_origins = org.ctypes.data
        c.curriculum = int(t.curriculum)
        c.num_terrain_rows, c.num_terrain_cols = int(t.num_rows), int(t.num_cols)
        c.max_init_terrain_level = int(t.max_init_terrain_level)
        c.vertical_faces = int(t.mesh_type == "trimesh")     # legged_robot.py:903-921: the slope-corrected mesh
        c.slope_threshold = float(getattr(t, "slope_treshold", 0.75) or 0.75)
    else:
        raise ValueError(f"Terrain mesh type '{t.mesh_type}' not supported (plane, heightfield, trimesh)")
    meta["dt"] = dt
    meta["active_terms"] = [nm for nm in _capi.REWARD_TERMS if scales.get(nm, 0.0) != 0]
    return c, keep, meta


def vars_of(section):
    """public attributes of a config section instance/class, in dir() order"""
    return {k: getattr(section, k) for k in dir(section) if not k.startswith("_")}


def soft_dof_pos_limits(rm, soft):
    """legged_robot.py:606-610"""
    mid = (rm.dof_lower + rm.dof_upper) / 2
    rng = rm.dof_upper - rm.dof_lower
    return np.stack([mid - 0.5 * rng * soft, mid + 0.5 * rng * soft], axis=1)
