/*
 * grx.h -- C ABI of the MI355X-native GRx legged-locomotion environment step.
 *
 * This library replaces, for ONE path, the closed Isaac Gym / PhysX tensor API that
 * FFTAI/Wiki-GRx-Gym drives from Python.  The reference crosses its native boundary ~7 times
 * per physics sub-step (70x per policy step):
 *
 *   gym.set_dof_actuation_force_tensor   legged_robot_fftai.py:67   (legged_robot.py:258)
 *   gym.simulate                         legged_robot_fftai.py:68   (legged_robot.py:259)
 *   gym.fetch_results                    legged_robot_fftai.py:70-71
 *   gym.refresh_dof_state_tensor         legged_robot_fftai.py:73
 *   gym.refresh_actor_root_state_tensor  legged_robot_fftai.py:74
 *   gym.refresh_net_contact_force_tensor legged_robot_fftai.py:75
 *   gym.refresh_rigid_body_state_tensor  legged_robot_fftai.py:76
 *   gym.set_dof_state_tensor_indexed     legged_robot.py:737
 *   gym.set_actor_root_state_tensor_indexed  legged_robot.py:782
 *   gym.set_actor_root_state_tensor      legged_robot.py:796
 *   gym.acquire_*_tensor + gymtorch.wrap_tensor   legged_robot.py:110-135, gymtorch.py:61-73,
 *                                                 gymtorch.cpp:33-158
 *
 * and runs the whole of LeggedRobot.step() (legged_robot.py:222-246) as O(400) small torch
 * kernels.  Here the boundary is coarser: ONE call per policy step (grx_step) runs clip-actions,
 * the 10 PD+dynamics+contact sub-steps, state update, termination, the reward terms, masked
 * in-kernel reset and the observation build as hand-written HIP kernels for gfx950.
 *
 * Conventions
 *   - every function returns 0 on success, a negative grx_status otherwise; the message is
 *     available from grx_last_error() (thread-local).
 *   - the library owns every device buffer for the life of the handle; grx_tensor() hands out
 *     NON-OWNING device pointers (the gymtorch.wrap_tensor model, gymtorch.py:61-73).
 *   - caller-provided device pointers (actions, injected noise) are borrowed for the call.
 *   - all work is enqueued on the caller's stream (void* = hipStream_t); no internal threads.
 *   - one handle per process/GPU; a handle is not thread-safe.
 *   - no torch types, no C++ types: plain pointers, sizes and PODs only.
 */
#ifndef GRX_H_
#define GRX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRX_ABI_VERSION 6

#define GRX_MAX_BODIES 36   /* moving bodies after merging fixed joints (base + DOFs) */
#define GRX_MAX_DOFS 32
#define GRX_MAX_SPHERES 48  /* collision spheres (primitive URDF shapes -> sphere sets) */
#define GRX_MAX_PAIRS 192  /* collision-sphere pairs of links that can touch each other (self-collision) */
#define GRX_MAX_LINKS 40          /* URDF links (the full GR1T1 has 37): second dimension of GRX_T_CONTACT_FORCES */
#define GRX_NUM_FEET 2
#define GRX_NUM_CMD 3
#define GRX_MAX_HEIGHT_POINTS 128
#define GRX_STATS_HISTORY 128     /* policy steps of episode statistics kept in GRX_T_EPISODE_STATS_HISTORY */

/* cfg.control.control_type (legged_robot.py:693-707): torques = p_gains (a s + q0 - q) - d_gains qd  |  p_gains (a s - qd) - d_gains (qd -
   last_dof_vel) / sim_dt  |  a s   (a = clipped action, s = action_scale; then x motor strength, clipped to the torque limits) */
typedef enum { GRX_CONTROL_P = 0, GRX_CONTROL_V = 1, GRX_CONTROL_T = 2 } grx_control_type;

typedef enum grx_status {
    GRX_OK = 0,
    GRX_ERR_INVALID_ARGUMENT = -1,
    GRX_ERR_UNSUPPORTED_MODEL = -2, /* tree topology the HIP kernels are not specialised for */
    GRX_ERR_HIP = -3,               /* a HIP runtime call or kernel launch failed */
    GRX_ERR_NO_DEVICE = -4,
    GRX_ERR_OUT_OF_MEMORY = -5,
    GRX_ERR_ABI_MISMATCH = -6
} grx_status;

/* Reward terms, in ALPHABETICAL order: the reference iterates class_to_dict(cfg.rewards.scales)
 * (helpers.py:42-57, dir() order) in legged_robot.py:845-866, so rew_buf is accumulated in
 * this order (legged_robot.py:362-366).  Formulas: legged_robot_fftai.py:180-352, gr1t1.py:338-589. */
typedef enum grx_reward_term {
    GRX_REW_ACTION_DIFF = 0,
    GRX_REW_ACTION_DIFF_DIFF,
    GRX_REW_ACTION_DIFF_KNEE,
    GRX_REW_CMD_DIFF_ANG_VEL_PITCH,
    GRX_REW_CMD_DIFF_ANG_VEL_ROLL,
    GRX_REW_CMD_DIFF_ANG_VEL_YAW,
    GRX_REW_CMD_DIFF_BASE_HEIGHT,
    GRX_REW_CMD_DIFF_BASE_ORIENT,
    GRX_REW_CMD_DIFF_FOREHEAD_ORIENT,
    GRX_REW_CMD_DIFF_LIN_VEL_X,
    GRX_REW_CMD_DIFF_LIN_VEL_Y,
    GRX_REW_CMD_DIFF_LIN_VEL_Z,
    GRX_REW_CMD_DIFF_TORSO_ORIENT,
    GRX_REW_COLLISION,
    GRX_REW_DOF_ACC_NEW,
    GRX_REW_DOF_TOR_ANKLE_FEET_LIFT_UP,
    GRX_REW_DOF_TOR_NEW,
    GRX_REW_DOF_TOR_NEW_HIP_ROLL,
    GRX_REW_DOF_VEL_NEW,
    GRX_REW_DOF_VEL_NEW_KNEE,
    GRX_REW_FEET_AIR_FORCE,
    GRX_REW_FEET_AIR_HEIGHT,
    GRX_REW_FEET_AIR_TIME,
    GRX_REW_FEET_LAND_TIME,
    GRX_REW_FEET_SPEED_XY_CLOSE_TO_GROUND,
    GRX_REW_FEET_SPEED_Z_CLOSE_TO_HEIGHT_TARGET,
    GRX_REW_FEET_STUMBLE,
    GRX_REW_LIMITS_ACTIONS,
    GRX_REW_LIMITS_DOF_POS,
    GRX_REW_LIMITS_DOF_TOR,
    GRX_REW_LIMITS_DOF_VEL,
    GRX_REW_ON_THE_AIR,
    GRX_REW_POSE_OFFSET,
    GRX_REW_POSE_OFFSET_HIP_YAW,
    GRX_REW_STAND_STILL,
    GRX_REW_TERMINATION,
    GRX_NUM_REWARD_TERMS
} grx_reward_term;

/* sphere flags */
#define GRX_SPH_FOOT_LEFT 0x1u   /* belongs to feet_indices[0] (first body whose name contains foot_name) */
#define GRX_SPH_FOOT_RIGHT 0x2u
#define GRX_SPH_TERMINATE 0x4u   /* body is in termination_contact_indices (legged_robot.py:1145-1161) */
#define GRX_SPH_PENALISE 0x8u    /* body is in penalised_contact_indices  (legged_robot.py:1127-1143) */

/* How a tensor that nobody needs on every step is published (grx_config.publish_rigid_body_states / publish_measured_heights).
 * ON_REFRESH tensors are brought up to date by grx_refresh() -- from the state the last step left, and for the envs that step RESET from
 * the state before the reset (and before _push_robots): exactly what the step-written tensors show (the reference's rigid_body_states
 * is not refreshed by reset_idx, its measured_heights are taken before it: legged_robot.py:284-296). */
typedef enum grx_publish_mode { GRX_PUBLISH_NEVER = 0, GRX_PUBLISH_EVERY_STEP = 1, GRX_PUBLISH_ON_REFRESH = 2 } grx_publish_mode;

typedef enum grx_terrain_type { GRX_TERRAIN_PLANE = 0, GRX_TERRAIN_HEIGHTFIELD = 1 } grx_terrain_type;   /* 'trimesh' = heightfield + vertical_faces */

/* Robot model after merging fixed-joint subtrees into their moving ancestor.  Body 0 is the
 * free-floating base (fix_base_link=False, legged_robot_config.py:119); body i>0 hangs from
 * parent[i] < i by ONE revolute joint; DOF index of body i is i-1.  Replaces gym.load_asset +
 * create_actor (legged_robot.py:966, 1022-1028). */
typedef struct grx_model {
    int32_t num_bodies;                      /* nb = 1 + num_dofs */
    int32_t parent[GRX_MAX_BODIES];          /* parent[0] = -1 */
    float joint_axis[GRX_MAX_BODIES][3];     /* unit axis in the child frame */
    float joint_rot0[GRX_MAX_BODIES][9];     /* row-major R0: child(q=0)->parent rotation (URDF rpy) */
    float joint_pos[GRX_MAX_BODIES][3];      /* child origin in parent frame (URDF xyz) */
    float mass[GRX_MAX_BODIES];
    float com[GRX_MAX_BODIES][3];            /* centre of mass, body frame */
    float inertia[GRX_MAX_BODIES][6];        /* about COM, body axes: xx xy xz yy yz zz */
    /* base_link alone (props[0] in legged_robot.py:618-648) for mass / COM randomisation:
     * base lump = base_rest + base_link(randomised) */
    float base_link_mass, base_link_com[3], base_link_inertia[6];
    float base_rest_mass, base_rest_com[3], base_rest_inertia[6];
    /* per-DOF properties (URDF <limit>, legged_robot.py:582-616) */
    float dof_lower[GRX_MAX_DOFS], dof_upper[GRX_MAX_DOFS];
    float dof_vel_limit[GRX_MAX_DOFS], dof_effort[GRX_MAX_DOFS];
    float dof_armature[GRX_MAX_DOFS];        /* joint-space armature [kg m^2], added to the diagonal of the joint-space inertia: asset_options.armature
                                                with use_physx_armature (legged_robot.py:958, legged_robot_config.py:125: 0 for the registered
                                                tasks; isaacgym docs struct_py.html AssetOptions.armature, dof_props['armature']) */
    /* collision spheres (URDF primitives -> spheres, DESIGN.md "contact geometry") */
    int32_t num_spheres;
    int32_t sph_body[GRX_MAX_SPHERES];
    float sph_pos[GRX_MAX_SPHERES][3];       /* centre, body frame */
    float sph_radius[GRX_MAX_SPHERES];
    uint32_t sph_flags[GRX_MAX_SPHERES];
    int32_t sph_link[GRX_MAX_SPHERES];       /* index of the URDF link (0..36) the shape belongs to: contact
                                                forces are netted per LINK (contact_forces (N, nb, 3)) */
    float sph_damp_max[GRX_MAX_SPHERES];     /* cap of the normal damping coefficient [N s/m]: alpha * m_eff / dt,
                                                m_eff = effective mass of the carrying body at the sphere along
                                                its z axis; keeps the explicit contact damping stable */
    /* self-collision (legged_robot_config.py:121 self_collisions = 0 = enabled, legged_robot.py:1022-1028): sphere pairs
     * (indices into sph_*) of URDF links on different, non-adjacent moving bodies that can reach each other within the
     * joint limits (tools/self_collision_pairs.py).  Every sphere pair of such a link pair is listed. */
    int32_t num_pairs;
    int16_t pair_a[GRX_MAX_PAIRS], pair_b[GRX_MAX_PAIRS];
    /* named frames the env pipeline reads from rigid_body_states */
    int32_t foot_body[GRX_NUM_FEET];         /* moving body carrying *_foot_roll_link */
    float foot_pos[GRX_NUM_FEET][3];         /* link-frame origin in that body's frame */
    int32_t torso_body;                      /* -1: no body name contains torso_name */
    float torso_rot[9];                      /* torso link -> body rotation, row-major */
    int32_t forehead_body;
    float forehead_rot[9];
    /* every URDF link frame (gym.acquire_rigid_body_state_tensor rows, legged_robot.py:113,134): the moving body that
     * carries it after the fixed-joint merge and its pose in that body's frame */
    int32_t num_links;
    int32_t link_body[GRX_MAX_LINKS];
    float link_pos[GRX_MAX_LINKS][3];
    float link_rot[GRX_MAX_LINKS][9];        /* link -> body rotation, row-major */
} grx_model;

typedef struct grx_contact_params {
    float kn;        /* normal stiffness per sphere [N/m] */
    float dn;        /* Hunt-Crossley damping factor [s/m]: fn = kn*d*(1 + dn*ddot) */
    float kt;        /* tangential (anchor spring) stiffness, foot spheres [N/m] */
    float ct;        /* tangential damping [N s/m] */
    float cv;        /* viscous friction coefficient of non-foot spheres [N s/m] */
    float k_limit;   /* joint-limit spring: torque = k_limit*effort*(violation) [1/rad] */
    float c_limit;   /* joint-limit damper, relative to the spring [s] */
    float damp_alpha; /* sph_damp_max = damp_alpha * m_eff / sim_dt */
    float terrain_friction;      /* legged_robot_config.py:77-78 */
} grx_contact_params;

typedef struct grx_config {
    int32_t abi_version;         /* must be GRX_ABI_VERSION */
    int32_t struct_size;         /* sizeof(grx_config) */
    int32_t num_envs;            /* envs owned by THIS handle (one rank) */
    int32_t env_offset;          /* global index of local env 0 (rank * num_envs) */
    int32_t total_envs;          /* global env count (terrain_types uses the global index, legged_robot.py:1177-1180) */
    uint64_t seed;

    grx_model model;
    grx_contact_params contact;

    /* sim (legged_robot_config.py:35-52, gr1t1_config.py:11-12,185) */
    float sim_dt;                /* 0.002 */
    int32_t decimation;          /* 10 */
    float gravity[3];

    /* control (legged_robot.py:679-715, gr1t1_lower_limb_config.py:20-35) */
    float kp[GRX_MAX_DOFS], kd[GRX_MAX_DOFS];
    float default_dof_pos[GRX_MAX_DOFS];
    float action_scale;
    float clip_actions_min[GRX_MAX_DOFS], clip_actions_max[GRX_MAX_DOFS]; /* legged_robot_fftai.py:171-177 */

    /* episode / commands (legged_robot.py:91-104, 650-677) */
    float max_episode_length;    /* ceil(episode_length_s / dt) */
    float max_episode_length_s;
    int32_t resample_command_interval;
    float cmd_lin_vel_x[2], cmd_lin_vel_y[2], cmd_ang_vel_yaw[2];

    /* initial state (gr1t1_config.py:88-92) */
    float init_pos[3], init_rot[4], init_lin_vel[3], init_ang_vel[3];

    /* domain randomisation (legged_robot_config.py:177-206) */
    int32_t randomize_friction;      float friction_range[2];
    int32_t randomize_restitution;   float restitution_range[2];   /* per-env shape restitution (legged_robot.py:565-575) */
    float terrain_restitution;       /* legged_robot_config.py:79; combined with the shape's by averaging */
    float bounce_threshold_velocity; /* legged_robot_config.py:48: slower impacts do not bounce */
    int32_t self_collisions;         /* 1: links collide with each other (the reference's self_collisions = 0) */
    int32_t randomize_base_mass;     float base_mass_range[2];
    int32_t randomize_base_com;      float base_com_range[3][2];
    int32_t randomize_motor_strength; float motor_strength_range[2];
    int32_t push_robots;             int32_t push_interval; float max_push_vel_xy;
    int32_t randomize_init_dof_pos;
    int32_t randomize_init_base_velocity;

    /* rewards (gr1t1_config.py:187-259, gr1t1_lower_limb_config.py:40-80) */
    float reward_scale[GRX_NUM_REWARD_TERMS];   /* raw cfg scale; the library multiplies by dt (legged_robot.py:850) */
    float reward_sigma[GRX_NUM_REWARD_TERMS];
    int32_t only_positive_rewards;
    float base_height_target, swing_feet_height_target, feet_stumble_ratio;
    float feet_air_time_target, feet_land_time_max;
    float soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit;
    uint32_t knee_mask, hip_roll_mask, hip_yaw_mask, ankle_left_mask, ankle_right_mask; /* DOF bitmasks, gr1t1.py:127-279 */

    /* observations (gr1t1.py:281-336, gr1t1_config.py:261-283) */
    int32_t num_obs, num_pri_obs;
    float obs_scale_action, obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_gravity;
    float obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    int32_t add_noise; float noise_level;
    float noise_action, noise_lin_vel, noise_ang_vel, noise_gravity, noise_dof_pos, noise_dof_vel, noise_height;
    float clip_observations;

    /* termination (legged_robot.py:336-353) */
    float termination_force;     /* 1.0 N */
    float termination_gravity_z; /* 0.33 */

    /* terrain (legged_robot_config.py:65-110, terrain.py:38-164, legged_robot.py:1163-1274) */
    int32_t terrain_type;        /* grx_terrain_type */
    int32_t measure_heights;
    int32_t num_height_points;   /* 121 */
    float height_points[GRX_MAX_HEIGHT_POINTS][2]; /* base-frame xy, meshgrid(x,y) order (legged_robot.py:1219-1233) */
    const int16_t* height_samples; /* HOST pointer, (hf_rows, hf_cols) row-major; copied at create */
    int32_t hf_rows, hf_cols;
    float horizontal_scale, vertical_scale, border_size;
    int32_t vertical_faces;      /* mesh_type 'trimesh': the contact surface is the reference's slope-corrected triangle mesh of the raster (isaacgym
                                    terrain_utils.py:286-350 convert_heightfield_to_trimesh with slope_threshold, the call of legged_robot.py:903-921) -- a ground
                                    plane per triangle half of every cell and the mesh's vertical faces as contacts of their own */
    float slope_threshold;       /* legged_robot_config.py:99 (0.75) */
    int32_t curriculum;
    int32_t num_terrain_rows, num_terrain_cols; /* levels, types */
    int32_t max_init_terrain_level;
    const float* terrain_origins;  /* HOST pointer, (rows, cols, 3); copied at create */
    float terrain_length;          /* env_length: curriculum move-up threshold */
    float env_spacing;             /* plane grid (legged_robot.py:1187-1195) */

    int32_t publish_reward_terms;  /* 1: also write GRX_T_REWARD_TERMS, the per-term reward table (a debugging tensor with no
                                      reference counterpart; the parity tests read it).  Every other tensor is always current. */
    int32_t publish_rigid_body_states; /* grx_publish_mode of GRX_T_RIGID_BODY_STATES.  GRX_PUBLISH_EVERY_STEP: written by the step kernel after
                                      its last sub-step (the reference refreshes it every sub-step, legged_robot_fftai.py:76, and reads the
                                      last one: 1.9 KB per env-step); GRX_PUBLISH_ON_REFRESH: materialised by grx_refresh() when somebody
                                      reads it -- the gym.refresh_rigid_body_state_tensor model; GRX_PUBLISH_NEVER: no such tensor */

    int32_t publish_measured_heights;  /* ABI 6: grx_publish_mode of GRX_T_MEASURED_HEIGHTS (the raw 121-point scan, legged_robot.py:289: a reference
                                      attribute no reward or observation reads back -- they use the scan inside the step).  0 is read as
                                      GRX_PUBLISH_EVERY_STEP (484 B per env-step); GRX_PUBLISH_ON_REFRESH: grx_refresh() */

    /* ABI 5: the reference's options that the GRx tasks leave off.  A handle with either of them set runs the general one-wave
       (lower-limb robots) or tree / generic layout: the wave pipelines keep the registered tasks' code path. */
    int32_t control_type;        /* grx_control_type: how _compute_torques reads the action (legged_robot.py:693-707) */
    int32_t heading_command;     /* 1: commands[:, 2] = clip(0.5 wrap_to_pi(commands_heading - heading), ang_vel_yaw range) after every
                                    time-based resample (legged_robot.py:320-326); the yaw command is then not drawn at resample
                                    (legged_robot.py:668-676).  commands_heading is the reference's all-zero buffer (gr1t1.py:124: it is
                                    never written; the draw of legged_robot.py:669 lands in commands[:, 3], which nothing reads) */
} grx_config;

typedef enum grx_tensor_id {
    /* step outputs, row-major (N, k) */
    GRX_T_OBS = 0,            /* f32 (N, num_obs) */
    GRX_T_PRI_OBS,            /* f32 (N, num_pri_obs) */
    GRX_T_REW,                /* f32 (N) */
    GRX_T_RESET,              /* u8  (N)  bool */
    GRX_T_TIME_OUT,           /* u8  (N)  bool */
    GRX_T_EPISODE_LENGTH,     /* i64 (N)  caller-writable (on_policy_runner.py:126) */
    /* simulation state; (N, k) views with strides (1, N) over SoA storage */
    GRX_T_DOF_POS,            /* f32 (N, nd) */
    GRX_T_DOF_VEL,            /* f32 (N, nd) */
    GRX_T_TORQUES,            /* f32 (N, nd)  torque of the LAST sub-step */
    GRX_T_ACTIONS,            /* f32 (N, nd)  clipped */
    GRX_T_LAST_ACTIONS,       /* f32 (N, nd) */
    GRX_T_LAST_DOF_VEL,       /* f32 (N, nd) */
    GRX_T_COMMANDS,           /* f32 (N, 3) */
    GRX_T_ROOT_STATES,        /* f32 (N, 13)  p3 q4(xyzw) v3 w3, world frame */
    GRX_T_BASE_LIN_VEL,       /* f32 (N, 3)   base frame */
    GRX_T_BASE_ANG_VEL,       /* f32 (N, 3) */
    GRX_T_PROJECTED_GRAVITY,  /* f32 (N, 3) */
    GRX_T_FEET_CONTACT_FORCE, /* f32 (N, 2, 3) net contact force on the foot links, last sub-step */
    GRX_T_FEET_POS,           /* f32 (N, 2, 3) world position of the foot link origins */
    GRX_T_FEET_HEIGHT,        /* f32 (N, 2) */
    GRX_T_FEET_AIR_TIME,      /* f32 (N, 2) */
    GRX_T_FEET_LAND_TIME,     /* f32 (N, 2) */
    GRX_T_FEET_CONTACT,       /* u8  (N, 2) */
    GRX_T_AVG_FEET_FORCE,     /* f32 (N, 2)   sub-step averaged |F| (legged_robot_fftai.py:79,86) */
    GRX_T_AVG_FEET_SPEED,     /* f32 (N, 2, 3) sub-step averaged |v| */
    GRX_T_MEASURED_HEIGHTS,   /* f32 (N, nh); per grx_config.publish_measured_heights every step, or current after grx_refresh() */
    GRX_T_BASE_HEIGHTS_OFFSET,/* f32 (N) */
    GRX_T_EPISODE_SUMS,       /* f32 (GRX_NUM_REWARD_TERMS, N) */
    GRX_T_REWARD_TERMS,       /* f32 (GRX_NUM_REWARD_TERMS, N) last step's r_i*scale_i*dt */
    GRX_T_TERRAIN_LEVELS,     /* i32 (N) */
    GRX_T_TERRAIN_TYPES,      /* i32 (N) */
    GRX_T_ENV_ORIGINS,        /* f32 (N, 3) */
    GRX_T_MOTOR_STRENGTH,     /* f32 (N, nd) */
    GRX_T_FRICTION,           /* f32 (N) */
    GRX_T_BASE_MASS_COM,      /* f32 (N, 4)  randomised base_link mass, com xyz */
    GRX_T_TERM_CONTACT,       /* u8  (N) any terminating body touched the ground, last sub-step */
    GRX_T_EPISODE_STATS,      /* f32 (GRX_NUM_REWARD_TERMS + 2): mean episode sums of the envs reset by the last step that
                                 reset any (legged_robot.py:420-424), [NT] = their count, [NT + 1] = mean terrain level after
                                 that step's curriculum update (legged_robot.py:427-428).  Current after grx_flush_stats /
                                 grx_episode_stats; otherwise it may lag the last step by one (see grx_flush_stats) */
    GRX_T_ANCHORS,            /* f32 (N, 8, 3) foot-sphere friction anchors xy + active flag */
    GRX_T_CONTACT_FORCES,     /* f32 (N, GRX_MAX_LINKS, 3) net contact force per URDF link (index = grx_model.sph_link), last
                                 sub-step: the reference's contact_forces (legged_robot.py:117,266); links without shapes stay 0 */
    GRX_T_EPISODE_STATS_HISTORY, /* f32 (GRX_STATS_HISTORY, GRX_NUM_REWARD_TERMS + 2): row (slot) = GRX_T_EPISODE_STATS as of the step
                                 that returned grx_step_args.stats_slot = slot; a row stays valid for GRX_STATS_HISTORY - 1 later steps */
    GRX_T_RIGID_BODY_STATES,  /* f32 (N, GRX_MAX_LINKS, 13) p3 q4(xyzw) v3 w3 of every URDF link frame, world axes, after the last
                                 sub-step: gym.acquire_rigid_body_state_tensor (legged_robot.py:113,134); per
                                 grx_config.publish_rigid_body_states every step, or current after grx_refresh() */
    GRX_T_AVG_FEET_SPEED_RPY, /* f32 (N, 2, 3) sub-step averaged |angular velocity| of the foot links, world axes: avg_feet_speed_rpy
                                 (legged_robot_fftai.py:34, 81, 88, 144); no active reward term reads it */
    GRX_NUM_TENSORS
} grx_tensor_id;

typedef enum grx_dtype { GRX_F32 = 0, GRX_U8 = 1, GRX_I32 = 2, GRX_I64 = 3 } grx_dtype;

typedef struct grx_tensor_desc {
    void* data;           /* device pointer (host pointer for the CPU oracle build) */
    int32_t dtype;        /* grx_dtype */
    int32_t ndim;
    int64_t shape[4];
    int64_t stride[4];    /* in elements */
} grx_tensor_desc;

typedef struct grx_sim* grx_handle;

/* Optional per-step inputs; all pointers may be NULL. */
typedef struct grx_step_args {
    const float* actions;      /* device, (N, nd) row-major contiguous -- legged_robot.py:222 */
    float delay_substeps;      /* sub-steps (real valued) that still use last_actions: "deci < delay"
                                  (legged_robot_fftai.py:53-61); the host draws max(0, N(5,2)) */
    int64_t common_step_counter; /* value AFTER increment (legged_robot.py:281); pushes fire when
                                  counter % push_interval == 0 (legged_robot.py:333) */
    const float* noise_uniform;/* device (N, num_obs) uniforms in [0,1) replacing the internal
                                  Philox stream for obs noise (parity tests); NULL = internal */
    float* obs_out;            /* device (N, num_obs) row-major: this step's observations go HERE instead of GRX_T_OBS */
    float* pri_obs_out;        /* device (N, num_pri_obs): likewise for GRX_T_PRI_OBS.  NULL = the library's buffer.  The env
                                  wrapper alternates between two buffers: the reference hands out a FRESH obs tensor every
                                  step (torch.cat / torch.clip, gr1t1.py:282, legged_robot.py:241) and rsl_rl keeps the one it
                                  acted on until after env.step() (ppo.py:160-161, 194) -- without a copy per step */
    int64_t stats_slot;        /* OUT: row of GRX_T_EPISODE_STATS_HISTORY that holds extras["episode"] of THIS step (complete once
                                  a later step has been enqueued, or after grx_flush_stats) */
    int64_t stats_seq;         /* OUT: this step's place in the handle's launch sequence (steps, resets and debug steps each take
                                  one): stats_slot = stats_seq % GRX_STATS_HISTORY, and the row is overwritten once
                                  grx_stats_seq() - stats_seq exceeds GRX_STATS_HISTORY */
} grx_step_args;

/* create / destroy.  device_id: HIP device ordinal. */
int grx_create(const grx_config* cfg, int device_id, grx_handle* out);
int grx_destroy(grx_handle h);

/* reset every env (BaseTask.reset() first half, base_task.py:117-119) without stepping */
int grx_reset_all(grx_handle h, void* stream);

/* one policy step = LeggedRobot.step() (legged_robot.py:222-246) */
int grx_step(grx_handle h, grx_step_args* args, void* stream);   /* writes args->stats_slot */

/* non-owning view of a library buffer */
int grx_tensor(grx_handle h, int tensor_id, grx_tensor_desc* out);

/* Bring an ON_REFRESH tensor (grx_publish_mode) up to date in stream order: one small kernel, launched at most once per step however
 * often it is asked for; a no-op for every tensor the step keeps current.  The role of gym.refresh_rigid_body_state_tensor /
 * refresh_net_contact_force_tensor (legged_robot_fftai.py:75-76, legged_robot.py:275-278) for a caller that reads the tensor now and
 * then.  GRX_ERR_INVALID_ARGUMENT for a tensor this handle does not publish at all.  "Up to date" means: as of the last launch of this handle (step,
 * reset, grx_set_state*); a caller that writes simulation state THROUGH the zero-copy views (the reference's set_*_tensor role is grx_set_state) sees it in an
 * on-refresh tensor after the next such launch, exactly as the reference's tensors change at the next simulate.
 * Graph capture: the "at most once per step" saving is keyed on the HOST's count of launches, which a replayed graph does not pass through --
 * from the first launch a handle records into a graph on, grx_refresh launches its kernel on every call (eagerly or recorded) and is correct
 * after any number of replays (tests/test_env_gpu.py::test_refresh_after_graph_replays). */
int grx_refresh(grx_handle h, int tensor_id, void* stream);

/* overwrite simulation state of ALL envs from device buffers (any may be NULL = keep):
 * the set_dof_state_tensor / set_actor_root_state_tensor role (legged_robot.py:737, 796).
 * Layouts: root (N,13) row-major, dof_pos/dof_vel (N,nd) row-major. */
int grx_set_state(grx_handle h, const float* root_states, const float* dof_pos,
                  const float* dof_vel, void* stream);

/* reset the listed envs (LeggedRobot.reset_idx(env_ids), legged_robot.py:377-440) outside a step: curriculum move, dof / root /
 * command draws, history and timers zeroed, episode sums folded into the episode statistics.  env_ids: DEVICE int32[n]. */
int grx_reset_idx(grx_handle h, const int32_t* env_ids, int32_t n, void* stream);

/* indexed variant of grx_set_state (set_dof_state_tensor_indexed / set_actor_root_state_tensor_indexed, legged_robot.py:737-740,
 * 782-784): rows env_ids[i] of the FULL (N, k) buffers are copied; env_ids: DEVICE int32[n]. */
int grx_set_state_indexed(grx_handle h, const int32_t* env_ids, int32_t n, const float* root_states, const float* dof_pos,
                          const float* dof_vel, void* stream);

/* The episode statistics of a step (per-block partial sums) are reduced by the NEXT step's kernel -- one launch per policy
 * step, no host synchronisation.  grx_flush_stats reduces those of the LAST enqueued step now (one small kernel), so that
 * GRX_T_EPISODE_STATS and that step's row of GRX_T_EPISODE_STATS_HISTORY are current in stream order.
 * Steps and resets RECORDED into a HIP graph (stream capture) carry that reduction with them as a second small kernel: a
 * replayed launch cannot rely on its successor, so every replay leaves the statistics current. */
int grx_flush_stats(grx_handle h, void* stream);

/* launch number of the last launch of this handle that wrote statistics rows (see grx_step_args.stats_seq) */
int grx_stats_seq(grx_handle h, int64_t* out);

/* grx_flush_stats, then copy GRX_T_EPISODE_STATS (GRX_NUM_REWARD_TERMS + 2 floats) to host (synchronises the stream) */
int grx_episode_stats(grx_handle h, float* host_out, void* stream);

/* average duration [ms] of the fused step kernel over the timed launches since the last call, measured
 * with HIP events recorded on the launch stream (bench.py roofline leg); resets the window.
 * enable: 0 = stop, 1 = time every launch, n > 1 = time every n-th launch (an event pair costs the
 * stream several microseconds of serialisation, which matters next to an 80 us kernel). */
int grx_kernel_time_ms(grx_handle h, int enable, float* avg_ms, int64_t* launches);

/* What grx_step launches for this handle -- picked at grx_create from the LOCAL batch size or pinned by
 * GRX_WAVES_PER_BLOCK / GRX_LANES_PER_ENV / GRX_QUAD_WAVES / GRX_TREE: reports name the kernel that ran instead of guessing it
 * (bench.py's roofline.kernel), and rank-count invariance can be pinned to one layout. */
typedef struct grx_layout_info {
    int32_t lanes_per_env;     /* 2: a lane per leg; 4: a lane pair per leg (grx_quad.hip); 8 / 16: the tree kernel's lane group (grx_tree.h / grx_tree16.hip); 1: generic */
    int32_t waves_per_block;   /* 1, 2, 4 or 8 (the pipelines of grx_wavepipe.h); tree kernel: 1, 2 or 4 (GRX_TREE_WAVES); generic: 1 */
    int32_t envs_per_block;
    int32_t num_blocks;
    char kernel[64];           /* symbol as rocprofv3 --kernel-trace prints it, e.g. "grx_step_kernel_quad<true, 8>" */
} grx_layout_info;
int grx_layout(grx_handle h, grx_layout_info* out);

/* Spin (no blocking system call) until every step enqueued through this handle has finished on the GPU.
 * The library also bounds the host's run-ahead to 256 policy steps with the same progress word (host-pinned: every step
 * kernel stores, when it STARTS, the ticket of the step before it -- complete by stream order), see grx_capi.cpp. */
int grx_wait_idle(grx_handle h);

/* ---- TEST-ONLY entry (not part of the drop-in surface; the reference has no counterpart) ----------------------
 * State injected into the post-physics half of the step: the same record the CPU oracle's gro_debug_post_physics
 * takes, so that tests/ can present the reference's golden fixtures (tools/gen_golden.py: legged_robot.py:269-481,
 * legged_robot_fftai.py:90-167, gr1t1.py:281-589 evaluated on synthetic state) to the HIP kernels DIRECTLY. */
typedef struct grx_pipeline_state {
    float q[GRX_MAX_DOFS], qd[GRX_MAX_DOFS], root[13];
    float actions[GRX_MAX_DOFS], last_actions[GRX_MAX_DOFS], last_last_actions[GRX_MAX_DOFS], last_dof_vel[GRX_MAX_DOFS], torques[GRX_MAX_DOFS];
    float commands[3];
    float air_time[2], land_time[2];
    int32_t contact_last[2];
    float feet_force[2][3], feet_pos[2][3];   /* contact_forces[feet], rigid_body_states[feet, 0:3] */
    float avg_force[2], avg_speed[2][3];      /* sub-step averages (legged_robot_fftai.py:79-88) */
    float torso_R[9];                         /* ignored by the HIP path: the torso rides on the base lump */
    float heights[GRX_MAX_HEIGHT_POINTS];     /* ignored by the HIP path: plane = zeros, heightfield = own scan */
    float base_heights_offset;                /* the STALE value the reward reads */
    int64_t episode_length;
    int32_t term_contact;                     /* a terminating link carries |F| > termination_force */
} grx_pipeline_state;

/* Run post_physics_step (everything after the sub-step loop: state update, timers, termination, rewards, reset,
 * observations, history) of ALL envs on `states` (HOST array, num_envs entries).  args->actions is ignored (the
 * injected actions are used), args->common_step_counter and args->noise_uniform are honoured.  apply_reset == 0:
 * resets are reported in GRX_T_RESET but not applied.  Any model the fused kernels or the tree kernel run (round 6: the
 * 32-DOF full body too -- its torso / forehead orientations come from the kernel's own walk over the injected joint positions). */
int grx_debug_post_physics(grx_handle h, const grx_pipeline_state* states, int apply_reset, const grx_step_args* args, void* stream);

/* TEST-ONLY: the step kernels' PHYSICS terrain query -- height and gradient (dh/dx, dh/dy) of the contact surface under n points: xy HOST
 * float[n][2] (world coordinates), out HOST float[n][3].  The oracle's twin is gro_debug_terrain; tests/test_terrain_golden.py ray-casts the
 * reference's slope-corrected triangle mesh (isaacgym terrain_utils.py:286-350) against both. */
int grx_debug_terrain(grx_handle h, const float* xy, int32_t n, float* out, void* stream);

/* TEST-ONLY: mesh_type 'trimesh' -- the step kernels' contact of a sphere at rest with the VERTICAL FACES of the reference's slope-corrected mesh next to
 * it: xyzr HOST float[n][4] (centre in world coordinates, radius), out HOST float[n][3] = overlap times the unit direction from the face to the
 * centre (0 0 0: no face within reach).  The oracle's twin is gro_debug_wall; tests/test_terrain_golden.py measures both against the mesh's own
 * vertical triangles. */
int grx_debug_wall(grx_handle h, const float* xyzr, int32_t n, float* out, void* stream);

/* TEST-ONLY, host only (needs no device): the two per-cell tables grx_create derives from the raster for mesh_type 'trimesh' -- the reference's slope-corrected
 * mesh as ground corner heights per triangle half (ground: int16[hf_rows * hf_cols][6] = (e00, e01, e11) where ty >= tx, (e00, e10, e11) where tx > ty) and
 * the tops of its vertical faces on a cell's sides x-, x+, y-, y+ and at its corners 00, 10, 01, 11 (walls: int16[hf_rows * hf_cols][8], INT16_MIN = none);
 * DESIGN.md 3.  The oracle's twin is gro_debug_trimesh_tables. */
int grx_debug_trimesh_tables(const grx_config* cfg, int16_t* ground, int16_t* walls);

/* TEST-ONLY: the wave pipelines of the step kernels hand over through LDS flags and spin on them.  A library built
 * with -DGRX_SPIN_LIMIT (csrc/variants/libgrx_spinlimit.so) bounds every spin; an expired one stores 'SP' << 48 | block << 32 | LDS
 * address of the flag << 16 | value waited for in a host-pinned word and traps.  *code = that word (0: none expired; readable after
 * the trap), *bounded = whether this library is such a build. */
int grx_debug_spin_report(grx_handle h, uint64_t* code, int* bounded);

const char* grx_last_error(void);
int grx_abi_version(void);
/* sizeof() of the structs that cross this boundary, as THIS library was compiled: a binding checks its own mirror against it before the
 * first call (a short grx_step_args would be overrun by grx_step's OUT fields: INTEGRATION.md 1a, tests/test_capi.py).  -1: unknown id. */
typedef enum grx_struct_id { GRX_STRUCT_CONFIG = 0, GRX_STRUCT_STEP_ARGS = 1, GRX_STRUCT_TENSOR_DESC = 2, GRX_STRUCT_PIPELINE_STATE = 3,
                             GRX_STRUCT_LAYOUT_INFO = 4, GRX_STRUCT_MODEL = 5 } grx_struct_id;
int grx_sizeof(int struct_id);
const char* grx_reward_term_name(int term);

#ifdef __cplusplus
}
#endif
#endif /* GRX_H_ */
