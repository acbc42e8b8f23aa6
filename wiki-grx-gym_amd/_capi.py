"""ctypes mirror of include/grx.h (the C ABI of libgrx_hip.so).

Field order and sizes must match the header exactly; ``grx_create`` rejects a mismatching
``struct_size`` (GRX_ERR_ABI_MISMATCH) so drift fails loudly instead of corrupting memory.
"""
import ctypes as C

GRX_ABI_VERSION = 6
PUBLISH_NEVER, PUBLISH_EVERY_STEP, PUBLISH_ON_REFRESH = 0, 1, 2   # grx_publish_mode
CONTROL_TYPES = {"P": 0, "V": 1, "T": 2}   # grx_control_type (legged_robot.py:693-707)
MAX_BODIES = 36
MAX_DOFS = 32
MAX_SPHERES = 48
MAX_PAIRS = 192
NUM_FEET = 2
MAX_HEIGHT_POINTS = 128
MAX_LINKS = 40
STATS_HISTORY = 128

REWARD_TERMS = (
    "action_diff", "action_diff_diff", "action_diff_knee", "cmd_diff_ang_vel_pitch",
    "cmd_diff_ang_vel_roll", "cmd_diff_ang_vel_yaw", "cmd_diff_base_height", "cmd_diff_base_orient",
    "cmd_diff_forehead_orient", "cmd_diff_lin_vel_x", "cmd_diff_lin_vel_y", "cmd_diff_lin_vel_z",
    "cmd_diff_torso_orient", "collision", "dof_acc_new", "dof_tor_ankle_feet_lift_up", "dof_tor_new",
    "dof_tor_new_hip_roll", "dof_vel_new", "dof_vel_new_knee", "feet_air_force", "feet_air_height",
    "feet_air_time", "feet_land_time", "feet_speed_xy_close_to_ground",
    "feet_speed_z_close_to_height_target", "feet_stumble", "limits_actions", "limits_dof_pos",
    "limits_dof_tor", "limits_dof_vel", "on_the_air", "pose_offset", "pose_offset_hip_yaw",
    "stand_still", "termination",
)
NUM_REWARD_TERMS = len(REWARD_TERMS)
assert list(REWARD_TERMS) == sorted(REWARD_TERMS)  # alphabetical == the reference's dir() order

SPH_FOOT_LEFT, SPH_FOOT_RIGHT, SPH_TERMINATE, SPH_PENALISE = 1, 2, 4, 8
TERRAIN_PLANE, TERRAIN_HEIGHTFIELD = 0, 1

TENSOR_IDS = (
    "OBS", "PRI_OBS", "REW", "RESET", "TIME_OUT", "EPISODE_LENGTH", "DOF_POS", "DOF_VEL", "TORQUES",
    "ACTIONS", "LAST_ACTIONS", "LAST_DOF_VEL", "COMMANDS", "ROOT_STATES", "BASE_LIN_VEL",
    "BASE_ANG_VEL", "PROJECTED_GRAVITY", "FEET_CONTACT_FORCE", "FEET_POS", "FEET_HEIGHT",
    "FEET_AIR_TIME", "FEET_LAND_TIME", "FEET_CONTACT", "AVG_FEET_FORCE", "AVG_FEET_SPEED",
    "MEASURED_HEIGHTS", "BASE_HEIGHTS_OFFSET", "EPISODE_SUMS", "REWARD_TERMS", "TERRAIN_LEVELS",
    "TERRAIN_TYPES", "ENV_ORIGINS", "MOTOR_STRENGTH", "FRICTION", "BASE_MASS_COM", "TERM_CONTACT",
    "EPISODE_STATS", "ANCHORS", "CONTACT_FORCES", "EPISODE_STATS_HISTORY", "RIGID_BODY_STATES", "AVG_FEET_SPEED_RPY",
)
T = {name: i for i, name in enumerate(TENSOR_IDS)}
DTYPE_F32, DTYPE_U8, DTYPE_I32, DTYPE_I64 = 0, 1, 2, 3

f32, i32, u32, u64, i64 = C.c_float, C.c_int32, C.c_uint32, C.c_uint64, C.c_int64


class Model(C.Structure):
    _fields_ = [
        ("num_bodies", i32),
        ("parent", i32 * MAX_BODIES),
        ("joint_axis", (f32 * 3) * MAX_BODIES),
        ("joint_rot0", (f32 * 9) * MAX_BODIES),
        ("joint_pos", (f32 * 3) * MAX_BODIES),
        ("mass", f32 * MAX_BODIES),
        ("com", (f32 * 3) * MAX_BODIES),
        ("inertia", (f32 * 6) * MAX_BODIES),
        ("base_link_mass", f32), ("base_link_com", f32 * 3), ("base_link_inertia", f32 * 6),
        ("base_rest_mass", f32), ("base_rest_com", f32 * 3), ("base_rest_inertia", f32 * 6),
        ("dof_lower", f32 * MAX_DOFS), ("dof_upper", f32 * MAX_DOFS),
        ("dof_vel_limit", f32 * MAX_DOFS), ("dof_effort", f32 * MAX_DOFS),
        ("dof_armature", f32 * MAX_DOFS),
        ("num_spheres", i32),
        ("sph_body", i32 * MAX_SPHERES),
        ("sph_pos", (f32 * 3) * MAX_SPHERES),
        ("sph_radius", f32 * MAX_SPHERES),
        ("sph_flags", u32 * MAX_SPHERES),
        ("sph_link", i32 * MAX_SPHERES),
        ("sph_damp_max", f32 * MAX_SPHERES),
        ("num_pairs", i32),
        ("pair_a", C.c_int16 * MAX_PAIRS), ("pair_b", C.c_int16 * MAX_PAIRS),
        ("foot_body", i32 * NUM_FEET),
        ("foot_pos", (f32 * 3) * NUM_FEET),
        ("torso_body", i32),
        ("torso_rot", f32 * 9),
        ("forehead_body", i32),
        ("forehead_rot", f32 * 9),
        ("num_links", i32),
        ("link_body", i32 * MAX_LINKS),
        ("link_pos", (f32 * 3) * MAX_LINKS),
        ("link_rot", (f32 * 9) * MAX_LINKS),
    ]


class ContactParams(C.Structure):
    _fields_ = [("kn", f32), ("dn", f32), ("kt", f32), ("ct", f32), ("cv", f32),
                ("k_limit", f32), ("c_limit", f32), ("damp_alpha", f32), ("terrain_friction", f32)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", i32), ("struct_size", i32),
        ("num_envs", i32), ("env_offset", i32), ("total_envs", i32),
        ("seed", u64),
        ("model", Model),
        ("contact", ContactParams),
        ("sim_dt", f32), ("decimation", i32), ("gravity", f32 * 3),
        ("kp", f32 * MAX_DOFS), ("kd", f32 * MAX_DOFS), ("default_dof_pos", f32 * MAX_DOFS),
        ("action_scale", f32),
        ("clip_actions_min", f32 * MAX_DOFS), ("clip_actions_max", f32 * MAX_DOFS),
        ("max_episode_length", f32), ("max_episode_length_s", f32),
        ("resample_command_interval", i32),
        ("cmd_lin_vel_x", f32 * 2), ("cmd_lin_vel_y", f32 * 2), ("cmd_ang_vel_yaw", f32 * 2),
        ("init_pos", f32 * 3), ("init_rot", f32 * 4), ("init_lin_vel", f32 * 3), ("init_ang_vel", f32 * 3),
        ("randomize_friction", i32), ("friction_range", f32 * 2),
        ("randomize_restitution", i32), ("restitution_range", f32 * 2),
        ("terrain_restitution", f32), ("bounce_threshold_velocity", f32), ("self_collisions", i32),
        ("randomize_base_mass", i32), ("base_mass_range", f32 * 2),
        ("randomize_base_com", i32), ("base_com_range", (f32 * 2) * 3),
        ("randomize_motor_strength", i32), ("motor_strength_range", f32 * 2),
        ("push_robots", i32), ("push_interval", i32), ("max_push_vel_xy", f32),
        ("randomize_init_dof_pos", i32), ("randomize_init_base_velocity", i32),
        ("reward_scale", f32 * NUM_REWARD_TERMS), ("reward_sigma", f32 * NUM_REWARD_TERMS),
        ("only_positive_rewards", i32),
        ("base_height_target", f32), ("swing_feet_height_target", f32), ("feet_stumble_ratio", f32),
        ("feet_air_time_target", f32), ("feet_land_time_max", f32),
        ("soft_dof_pos_limit", f32), ("soft_dof_vel_limit", f32), ("soft_torque_limit", f32),
        ("knee_mask", u32), ("hip_roll_mask", u32), ("hip_yaw_mask", u32),
        ("ankle_left_mask", u32), ("ankle_right_mask", u32),
        ("num_obs", i32), ("num_pri_obs", i32),
        ("obs_scale_action", f32), ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32),
        ("obs_scale_gravity", f32), ("obs_scale_dof_pos", f32), ("obs_scale_dof_vel", f32),
        ("obs_scale_height", f32),
        ("add_noise", i32), ("noise_level", f32),
        ("noise_action", f32), ("noise_lin_vel", f32), ("noise_ang_vel", f32), ("noise_gravity", f32),
        ("noise_dof_pos", f32), ("noise_dof_vel", f32), ("noise_height", f32),
        ("clip_observations", f32),
        ("termination_force", f32), ("termination_gravity_z", f32),
        ("terrain_type", i32), ("measure_heights", i32), ("num_height_points", i32),
        ("height_points", (f32 * 2) * MAX_HEIGHT_POINTS),
        ("height_samples", C.c_void_p),
        ("hf_rows", i32), ("hf_cols", i32),
        ("horizontal_scale", f32), ("vertical_scale", f32), ("border_size", f32),
        ("vertical_faces", i32), ("slope_threshold", f32),
        ("curriculum", i32), ("num_terrain_rows", i32), ("num_terrain_cols", i32),
        ("max_init_terrain_level", i32),
        ("terrain_origins", C.c_void_p),
        ("terrain_length", f32), ("env_spacing", f32),
        ("publish_reward_terms", i32),
        ("publish_rigid_body_states", i32),
        ("publish_measured_heights", i32),
        ("control_type", i32), ("heading_command", i32),
    ]


class TensorDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", i32), ("ndim", i32),
                ("shape", i64 * 4), ("stride", i64 * 4)]


class StepArgs(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("delay_substeps", f32),
                ("common_step_counter", i64), ("noise_uniform", C.c_void_p),
                ("obs_out", C.c_void_p), ("pri_obs_out", C.c_void_p), ("stats_slot", i64), ("stats_seq", i64)]


class PipelineState(C.Structure):
    """grx_pipeline_state (TEST-ONLY entry grx_debug_post_physics; the oracle's gro_debug_post_physics takes the same record)."""
    _fields_ = [
        ("q", f32 * MAX_DOFS), ("qd", f32 * MAX_DOFS), ("root", f32 * 13),
        ("actions", f32 * MAX_DOFS), ("last_actions", f32 * MAX_DOFS),
        ("last_last_actions", f32 * MAX_DOFS), ("last_dof_vel", f32 * MAX_DOFS),
        ("torques", f32 * MAX_DOFS), ("commands", f32 * 3),
        ("air_time", f32 * 2), ("land_time", f32 * 2), ("contact_last", i32 * 2),
        ("feet_force", (f32 * 3) * 2), ("feet_pos", (f32 * 3) * 2),
        ("avg_force", f32 * 2), ("avg_speed", (f32 * 3) * 2), ("torso_R", f32 * 9),
        ("heights", f32 * MAX_HEIGHT_POINTS), ("base_heights_offset", f32),
        ("episode_length", i64), ("term_contact", i32),
    ]


class LayoutInfo(C.Structure):
    """grx_layout_info: what grx_step launches for a handle (include/grx.h grx_layout)."""
    _fields_ = [("lanes_per_env", i32), ("waves_per_block", i32), ("envs_per_block", i32), ("num_blocks", i32), ("kernel", C.c_char * 64)]


def bind(lib, prefix="grx_"):
    """Declare argtypes/restypes of every entry point of include/grx.h on a loaded CDLL."""
    H = C.c_void_p

    def fn(name, restype, *argtypes):
        f = getattr(lib, prefix + name)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f

    api = {
        "create": fn("create", C.c_int, C.POINTER(Config), C.c_int, C.POINTER(H)),
        "destroy": fn("destroy", C.c_int, H),
        "reset_all": fn("reset_all", C.c_int, H, C.c_void_p),
        "step": fn("step", C.c_int, H, C.POINTER(StepArgs), C.c_void_p),
        "tensor": fn("tensor", C.c_int, H, C.c_int, C.POINTER(TensorDesc)),
        "refresh": fn("refresh", C.c_int, H, C.c_int, C.c_void_p),
        "set_state": fn("set_state", C.c_int, H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p),
        "episode_stats": fn("episode_stats", C.c_int, H, C.POINTER(C.c_float), C.c_void_p),
        "flush_stats": fn("flush_stats", C.c_int, H, C.c_void_p),
        "reset_idx": fn("reset_idx", C.c_int, H, C.c_void_p, C.c_int32, C.c_void_p),
        "set_state_indexed": fn("set_state_indexed", C.c_int, H, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p),
        "last_error": fn("last_error", C.c_char_p),
        "abi_version": fn("abi_version", C.c_int),
        "reward_term_name": fn("reward_term_name", C.c_char_p, C.c_int),
    }
    if hasattr(lib, prefix + "debug_post_physics") and prefix == "grx_":   # the oracle's entry of that name is per-env (oracle/binding.py)
        api["debug_post_physics"] = fn("debug_post_physics", C.c_int, H, C.POINTER(PipelineState), C.c_int, C.POINTER(StepArgs), C.c_void_p)
    if hasattr(lib, prefix + "wait_idle"):
        api["wait_idle"] = fn("wait_idle", C.c_int, H)
    if hasattr(lib, prefix + "stats_seq"):
        api["stats_seq"] = fn("stats_seq", C.c_int, H, C.POINTER(i64))
    if hasattr(lib, prefix + "debug_spin_report"):
        api["debug_spin_report"] = fn("debug_spin_report", C.c_int, H, C.POINTER(C.c_uint64), C.POINTER(C.c_int))
    if hasattr(lib, prefix + "layout"):
        api["layout"] = fn("layout", C.c_int, H, C.POINTER(LayoutInfo))
    if hasattr(lib, prefix + "debug_terrain") and prefix == "grx_":   # (the oracle's entry of that name takes one point: oracle/binding.py)
        api["debug_terrain"] = fn("debug_terrain", C.c_int, H, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.c_void_p)
    if hasattr(lib, prefix + "debug_wall") and prefix == "grx_":
        api["debug_wall"] = fn("debug_wall", C.c_int, H, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.c_void_p)
    if hasattr(lib, prefix + "debug_trimesh_tables") and prefix == "grx_":
        api["debug_trimesh_tables"] = fn("debug_trimesh_tables", C.c_int, C.POINTER(Config), C.POINTER(C.c_int16), C.POINTER(C.c_int16))
    if hasattr(lib, prefix + "sizeof"):
        api["sizeof"] = fn("sizeof", C.c_int, C.c_int)
    if hasattr(lib, prefix + "kernel_time_ms"):
        api["kernel_time_ms"] = fn("kernel_time_ms", C.c_int, H, C.c_int, C.POINTER(C.c_float), C.POINTER(i64))
    return api


EXPORTED_SYMBOLS = (
    "grx_create", "grx_destroy", "grx_reset_all", "grx_step", "grx_tensor", "grx_set_state",
    "grx_episode_stats", "grx_flush_stats", "grx_reset_idx", "grx_set_state_indexed", "grx_kernel_time_ms", "grx_wait_idle", "grx_last_error", "grx_abi_version",
    "grx_reward_term_name", "grx_debug_post_physics", "grx_layout", "grx_stats_seq", "grx_debug_spin_report", "grx_sizeof", "grx_refresh", "grx_debug_terrain", "grx_debug_wall", "grx_debug_trimesh_tables",
)
# grx_struct_id (include/grx.h): grx_sizeof(id) must equal ctypes.sizeof of the mirror -- checked once per process by sim.load_hip_library
STRUCT_IDS = {"CONFIG": (0, Config), "STEP_ARGS": (1, StepArgs), "TENSOR_DESC": (2, TensorDesc), "PIPELINE_STATE": (3, PipelineState),
              "LAYOUT_INFO": (4, LayoutInfo), "MODEL": (5, Model)}
