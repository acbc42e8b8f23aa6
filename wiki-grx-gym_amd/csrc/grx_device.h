// grx_device.h -- device-side parameter blocks shared by the kernels and the C-ABI host code.
#pragma once
#include <stdint.h>

#include "../../include/grx.h"

#define GRX_LEG 5        // joints per leg chain
#define GRX_ND 10        // DOFs of the supported (lower-limb) topology
#define GRX_NUM_OBS 39   // 9 + 3*GRX_ND (gr1t1.py:281-295)
#define GRX_MAX_PRI 168  // 39 + 3 + 1 + 2 + 2 + 121 (gr1t1.py:297-313)
#define GRX_MAXSPH_SIDE 16
#define GRX_PROF_SLOTS 96   // GRX_PROFILE_SECTIONS builds: clock stamps per block (tools/gpu_sections.py)
#define GRX_COARSE 8     // raster cells per coarse max-map cell (0.8 m)
// rows of the episode-statistics tables: the reward terms, [NT] the number of episodes that ended, [NT + 1] the sum of the terrain levels
#define GRX_NSTAT (GRX_NUM_REWARD_TERMS + 2)

// what every kernel that writes per-block statistics rows is told about its place in the handle's launch sequence
struct StepSeq {
    long long seq;          // launch number of this handle (step, reset, debug step alike): parity seq & 1 of the statistics tables, history row seq % GRX_STATS_HISTORY
    long long* progress;    // host-pinned progress word (nullptr while a graph is being recorded) ...
    long long ticket_done;  // ... and the ticket of the work that preceded this launch on the stream: complete when this kernel starts
    int fold_prev;          // 1: this launch reduces the statistics rows of launch seq - 1 (the kernel boundary orders them).  0: a launch recorded
                            // into a graph -- on a replay its predecessor in EXECUTION order is not launch seq - 1, so it leaves the tables alone and
                            // grx_step records a grx_finalize_stats of its own rows behind it
    int pad;
};

struct SphC {   // 32 bytes
    float x, y, z, r;    // centre (body frame), radius
    uint32_t flags;      // GRX_SPH_*
    int32_t slot;        // 0..3: anchored foot sphere index within this lane's foot, -1 otherwise
    int32_t link_last;   // bit 0: last sphere of its URDF link in this lane's list (per-link force netting);
                         // bits 8..: URDF link index + 1 (row of GRX_T_CONTACT_FORCES; 0 = unused slot) -- see sph_link()
    float dmax;          // cap of the normal damping coefficient (grx_model.sph_damp_max)
};

__host__ __device__ inline int sph_link(const SphC& s) { return (s.link_last >> 8) - 1; }
static_assert(sizeof(SphC) == 32, "SphC must stay two 16-byte LDS words");

// Constants of one chain body + its joint, packed so that a body's dynamics constants are four adjacent
// 16-byte LDS words (ds_read_b128: one exposed LDS latency per body instead of ~13 narrow reads).
struct alignas(16) BodyC {
    float r[3]; float mass;        // joint origin in the parent body frame | mass
    float com[3]; float Ic[6];     // centre of mass | inertia about COM: xx xy xz yy yz zz
    float kp, kd, q0;              // PD gains, default angle
    float effort, vlim, qlo, qhi;  // URDF limits
    float Klim, Clim, amin, amax;  // joint-limit spring/damper | clip_actions
    float slo, shi, pad0, pad1;    // soft dof position limits (legged_robot.py:606-610)
};

// self-collision between a base-lump shape and this lane's two thigh shapes (grx_self.h): one entry per BASE-LUMP sphere (round 4;
// round 3 kept one per sphere pair, 64 registers on the wave that evaluates them -- the spill of the eight-wave kernels)
struct alignas(16) BaseChainPair {
    float x, y, z, r;    // the base-lump sphere, base frame
    float dmax;          // its damping cap
    int32_t tmask;       // bit t: it can touch thigh shape t of this lane (0 / 1)
    int32_t link;        // URDF link of the base-lump sphere (row of GRX_T_CONTACT_FORCES)
    int32_t pad;
};
#define GRX_MAX_BC 4

// Per-side (left leg / right leg lane) robot constants; staged into LDS by every block.
struct alignas(16) SideConst {
    BodyC body[GRX_LEG];
    float foot_pos[3];
    int32_t nbc;                  // base-lump / thigh self-collision pairs of this side
    SphC sph[GRX_MAXSPH_SIDE];
    float bs[3][4];               // bounding spheres (centre in the body frame, radius) of the shapes of chain bodies 2, 3, 4
    BaseChainPair bc[GRX_MAX_BC];
};

// GRX_T_RIGID_BODY_STATES: the URDF link frames a lane publishes -- those riding on its own chain bodies and its share of the
// base lump's -- grouped by carrying body (level 0 = base, 1 + k = chain body k)
#define GRX_RBS_MAX 24
struct RbsEntry {   // 32 bytes
    float px, py, pz;        // link origin in the carrying body's frame
    int32_t link;            // URDF link index (row of the tensor)
    float qx, qy, qz, qw;    // link -> body rotation
};
struct RbsTables {
    int32_t off[2][GRX_LEG + 2];   // entries [off[side][lvl], off[side][lvl + 1]) ride on level lvl
    RbsEntry e[2][GRX_RBS_MAX];
};

// ---- tree kernel (grx_tree.h): any robot tree whose chains fit a lane group -- the 32-DOF full body of BASELINE.json config 5 ----
// An env is a GROUP of GRX_TREE_G lanes; every lane owns one CHAIN of the tree (a path: a body's first child continues its
// chain, further children start new ones) and at global step g works on its chain's body of depth g.
#define GRX_TREE_G 8        // lanes per env of the default build of grx_tree.h
#define GRX_TREE_GMAX 16    // ... and of csrc/grx_tree16.hip (the same source with 16: four envs per wave -- twice the waves, for batches that leave half the
                            // SIMDs idle with eight lanes per env); the tables are sized for it, TreeTab.g says which one a table was built for
#define GRX_TREE_MAXSTEP 16
#define GRX_TREE_LEVELS 10   // depth levels the tree kernel's passes are unrolled for (deeper trees run on the generic kernel)
#define GRX_TREE_MAXCS 8     // bodies of one chain that carry collision shapes (or a foot frame): rounds of the contact pass
// (LDS tables indexed by a lane's OWN body / joint / shape: an odd stride in words, so that the lanes of a group -- on different entries at the
//  same field -- fall on different banks.  Round 4's TreeDof (16 words) and TreeSph (8) put them in 2 and 4 bank classes.)
struct TreeBody {            // 37 words
    float axis[3], mass;     // joint axis (child frame), mass
    float rot0[9];           // child(q = 0) -> parent rotation, row-major
    float jpos[3];           // joint origin in the parent frame
    float com[3];
    float Ic[6];             // inertia about the COM: xx xy xz yy yz zz
    int32_t parent;
    int32_t sph_begin, sph_end;
    int32_t nhc;             // children that START a chain (they hand their articulated inertia up through the chain's LDS slot) ...
    int32_t hc[4];           // ... the lanes of those chains
    int32_t lane, step;      // where this body is processed
    int32_t rot0_identity;   // the joint frame is not rotated against the parent's (rot0 = 1): skips a 3 x 3 product
    int32_t pad;
};
struct TreeDof { float kp, kd, q0, effort, vlim, qlo, qhi, slo, shi, amin, amax, Klim, Clim; int32_t lane; float arm; int32_t pad[2]; };   // 17 words (arm: joint-space armature)
struct TreeSph { float x, y, z, r, dmax; int32_t slot, link, pad[2]; };   // 9 words
static_assert(sizeof(TreeBody) == 37 * 4 && sizeof(TreeDof) == 17 * 4 && sizeof(TreeSph) == 9 * 4, "odd strides");
struct TreeTab {
    int32_t nb, nd, nsph, nlc, nchain, nstep, nh0, g;   // g: lanes per env this table was built for (work list, link / pair rounds)
    int32_t heads0[GRX_TREE_GMAX];                       // lanes whose chain hangs from the base
    int32_t first[GRX_TREE_GMAX], last[GRX_TREE_GMAX];      // step range of each lane's chain (first > last: no chain)
    int8_t sched[GRX_TREE_GMAX][GRX_TREE_MAXSTEP];       // body at (lane, step), -1: none
    TreeBody body[GRX_MAX_BODIES];
    TreeDof dof[GRX_MAX_DOFS];
    TreeSph sph[GRX_MAX_SPHERES];
    uint32_t link_flags[24];
    int32_t link_urdf[24];
    int32_t foot_body[2], foot_link[2];
    float foot_pos[2][3];
    int32_t torso_body, forehead_body;
    float torso_rot[9], forehead_rot[9];
    int32_t sph_begin0, sph_end0;                     // the base's own shapes
    int32_t nlp, pad1;
    // self-collision link pairs (grx_generic.h GenTables.lp_*): bodies, compact links
    int16_t lp_ba[48], lp_bb[48], lp_a[48], lp_b[48];
    // ... and their sphere pairs (round 6): the broad phase tests EVERY sphere pair of the model (grx_model.pair_a/b) on the sphere centres the
    // contact pass leaves in LDS -- centre distance against (ra + rb + margin)^2 -- and raises the bit of the pair's LINK pair; the narrow phase
    // then runs on the raised link pairs only, i.e. on links that really touch (rounds 2-5 tested the links' bounding spheres, which overlap in
    // every pose for neighbours like upper arm x torso: the 800-instruction narrow phase ran in every round of every sub-step)
    int32_t nsp, nsp_batches;                         // pairs; batches of 4 rounds of the group's lanes (the table is padded with pairs that never pass)
    struct { uint32_t ab; float r2; } sp[GRX_MAX_PAIRS];   // ab: sphere of link a | sphere of link b << 8 | link pair << 16 (positions in sph[]); r2: (ra + rb + margin)^2
    int32_t lc_begin[25];
    int32_t ncs;                                      // rounds of the contact pass
    int32_t nturn;                                    // items of one round that share a body add their forces in turns 0 .. nturn - 1
    int32_t nstep_kin;                                // depth levels that hold a foot, the torso or the forehead: all the final-frames walk needs without GRX_T_RIGID_BODY_STATES
    // the contact pass's work list: round r, lane c evaluates shapes [s0, s1) (at most two) of `body` (-1: nothing; 0: the base) on
    // the frame in LDS -- ANY lane may take any body's shapes, so the three lanes of a group that own no chain work too and a foot's
    // four spheres go to two lanes (round 4: four sphere-slots per sub-step for the full body instead of eight)
    struct { int8_t body, s0, s1, turn; } cw[GRX_TREE_MAXCS][GRX_TREE_GMAX];
};

// every URDF link frame by carrying body (the tree kernel's GRX_T_RIGID_BODY_STATES)
struct LinkTab { int32_t n, pad[3]; int32_t body[GRX_MAX_LINKS]; float pos[GRX_MAX_LINKS][3]; float rot[GRX_MAX_LINKS][9]; };

// grx_refresh (include/grx.h): the joint tree as grx_model holds it, for the kernels that materialise a tensor on demand from the state a
// step left behind -- path[b]: the bodies from the base's child down to b (depth[b] of them); any model, fused or tree layout
#define GRX_REFRESH_MAXDEPTH 16
struct RefreshTab {
    int32_t nb, nlinks;
    int32_t depth[GRX_MAX_BODIES];
    int8_t path[GRX_MAX_BODIES][GRX_REFRESH_MAXDEPTH];
    float axis[GRX_MAX_BODIES][3], jpos[GRX_MAX_BODIES][3], rot0[GRX_MAX_BODIES][9];
    int32_t rot0_identity[GRX_MAX_BODIES];
};

// Large read-only tables, in device memory.
struct KTables {
    SideConst side[2];
    float height_points[GRX_MAX_HEIGHT_POINTS][2];
};

// Launch parameters.  Passed BY VALUE: the kernarg segment is constant address space, so the compiler may
// hoist / merge the scalar loads across the kernel's global stores (as a device-memory struct every store forced
// a reload + s_waitcnt: 421 waits in the post-physics section, measured).
struct KParams {
    int32_t N, env_offset, total_envs, publish_debug;
    uint64_t seed;
    float sim_dt; int32_t decimation; float gravity[3];
    float action_scale;
    int32_t control_type, heading_command;   // grx_control_type; legged_robot.py:320-326 (one-wave / tree / generic layouts only)
    float kn, dn, kt, ct, cv, terrain_friction, inv_kt;
    float termination_force, termination_gravity_z;
    float max_episode_length, max_episode_length_s;
    int32_t resample_command_interval;
    float cmd_lin_vel_x[2], cmd_lin_vel_y[2], cmd_ang_vel_yaw[2];
    float init_pos[3];
    int32_t randomize_init_dof_pos, randomize_init_base_velocity;
    int32_t push_robots, push_interval; float max_push_vel_xy;
    float reward_scale_dt[GRX_NUM_REWARD_TERMS], reward_sigma[GRX_NUM_REWARD_TERMS];
    int32_t only_positive_rewards;
    float base_height_target, swing_feet_height_target, feet_stumble_ratio, feet_air_time_target, feet_land_time_max;
    float soft_dof_vel_limit, soft_torque_limit;
    uint32_t knee_mask, hip_roll_mask, hip_yaw_mask, ankle_left_mask, ankle_right_mask;
    int32_t num_pri_obs;
    float obs_scale_action, obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_gravity, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    int32_t add_noise; float noise_level, noise_action, noise_ang_vel, noise_gravity, noise_dof_pos, noise_dof_vel;
    float clip_observations;
    int32_t terrain_type, measure_heights, nh;
    const int16_t* hf; int32_t hf_rows, hf_cols;
    const float* coarse_max; int32_t coarse_rows, coarse_cols;   // dilated block-max of the raster [m]: sphere culling
    int32_t vertical_faces; int32_t tm_off;        // mesh_type 'trimesh': the surface is the reference's slope-corrected mesh, held behind hf_cells (below):
                                                   // hf_cells[tm_off + 2 * cell + half]: ground corners under triangle half `half` of the cell (0: ty >= tx (e00, e01, e11),
                                                   // 1: tx > ty (e00, e10, e11); 4th int16 unused); (const uint4*)(hf_cells + 3 * tm_off)[cell]: tops of the vertical faces
                                                   // on the cell's sides x-, x+, y-, y+ and of the posts at its corners 00, 10, 01, 11 (int16 each, INT16_MIN = none)
    float hv_scale;                                // vertical_scale / horizontal_scale (terrain gradient)
    float bounce_threshold, terrain_restitution;   // legged_robot_config.py:48, :79
    int32_t self_collisions;     // links collide with each other (legged_robot_config.py:121)
    uint64_t sp_mask;            // bit (a * 8 + b): shape 8 + a of the LEFT lane and shape 8 + b of the RIGHT lane can touch (self-collision)
    uint32_t ll_mask;            // bit (i * 3 + j): the shapes of LEFT chain body 2 + i can touch those of RIGHT chain body 2 + j
    float* restitution;          // [N] per-env shape restitution (legged_robot.py:565-575)
    const uint2* hf_cells;    // [hf_rows][hf_cols]: the four raster corners of cell (i, j), (h00 | h01 << 16, h10 | h11 << 16), edges clamped:
                              // ONE 8-byte gather per terrain lookup instead of three or four 2-byte ones
    const int16_t* hf_max4;   // [hf_rows][hf_cols]: max of the four raster corners of cell (i, j) = upper bound of the bilinear
                              // height anywhere in the cell: the exact reach test of the lane-compacted contacts (grx_rare.h)
    float horizontal_scale, vertical_scale, border_size, inv_hscale;
    int32_t curriculum, num_terrain_rows, num_terrain_cols;
    const float* terrain_origins; float terrain_length;
    float torso_rot[9], forehead_rot[9];
    int32_t has_torso, has_forehead;
    const struct KTables* tables;   // per-side robot tables + height-scan points (staged into LDS by every block)
    // state (SoA [k][N])
    float *q, *qd, *root, *anchors, *last_actions, *last_dof_vel, *actions, *torques, *motor_strength;
    float *base_m, *base_c, *base_I, *friction, *commands, *origins;
    int32_t *levels, *types;
    float *air_time, *land_time;
    uint8_t* feet_contact;   // also the "contact_last" state of the next step (legged_robot_fftai.py:113,131)
    float *feet_height, *avg_force, *feet_force, *feet_pos, *avg_speed, *base_heights_offset;
    float* avg_speed_rpy;    // [(foot * 3 + c)][N]: sub-step averaged |angular velocity| of the foot links (legged_robot_fftai.py:81, 88)
    float* contact_forces;   // [(link * 3 + c)][N]: net contact force per URDF link, last sub-step
    long long* ep_len;
    float* rew;
    uint8_t *reset, *time_out, *term_contact;
    float *base_lin_vel, *base_ang_vel, *proj_grav, *episode_sums, *reward_terms, *heights;
    float *obs, *pri_obs, *stat_partial, *stats;
    int32_t stat_stride;   // stat_partial is [2][GRX_NSTAT][stat_stride]: launch parity x statistics row x one column per block of the writing kernel
    int32_t* stat_nblocks; // [2]: columns the writing kernel of that parity filled
    float* stat_hist;      // [GRX_STATS_HISTORY][GRX_NSTAT]: GRX_T_EPISODE_STATS as of every step (extras["episode"] without a copy per step)
    float* rbs;            // GRX_T_RIGID_BODY_STATES [(link * 13 + c)][N], written when publish_rbs
    const RbsTables* rbs_tab;
    const LinkTab* link_tab;
    int32_t publish_rbs, num_links;   // publish_rbs: the STEP KERNEL writes GRX_T_RIGID_BODY_STATES (grx_config.publish_rigid_body_states == 1)
    // tensors published ON DEMAND (grx_refresh: grx_config.publish_* == 2): nothing is written per step except, by the lanes that RESET an
    // env, its state before the reset -- the reference's tensors show that state (rigid_body_states is not refreshed by a reset,
    // measured_heights is taken before reset_idx: legged_robot.py:284-296)
    int32_t publish_heights;          // the step kernel writes GRX_T_MEASURED_HEIGHTS every step (== 1)
    int32_t stash_pre_reset;          // a resetting lane stores (q, qd, root) into pre_* first
    float *pre_q, *pre_qd, *pre_root; // [nd][N], [nd][N], [13][N]: valid where `reset` is set
    float* pre_push_vel;              // [2][N]: the base's vx, vy before _push_robots overwrote them (valid after a push step)
    const RefreshTab* refresh_tab;
    int32_t nd;        // dofs of the model (10 on the fast path)
    long long* prof;   // GRX_PROFILE_SECTIONS builds only: [nblocks][16] s_memtime stamps
};
