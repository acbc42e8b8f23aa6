// grx_capi.cpp -- host side of the C ABI declared in include/grx.h (compiled with hipcc into
// libgrx_hip.so together with grx_kernels.hip).  No torch, no Python: plain C entry points.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <memory>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/grx.h"
#include "grx_device.h"
#include "grx_rng.h"

extern "C" {
void grx_launch_step(const KParams* dP, int N, int heightfield, int waves, const float* actions, float delay, long long common_step,
                     const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream);
void grx_launch_step_quad(const KParams* dP, int N, int heightfield, int waves, const float* actions, float delay, long long common_step,
                          const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream);
int grx_envs_per_block_quad(void);
void grx_launch_finalize(const KParams* dP, long long seq, long long* progress, long long ticket, hipStream_t stream);
void grx_launch_ticket(long long* progress, long long ticket, hipStream_t stream);
int grx_launch_step_generic(const KParams* dP, const void* tables, float* ws, int N, int epb, int lds_bytes, int heightfield, const float* actions,
                            float delay, long long common_step, const float* noise, float* obs_out, float* pri_out, long long seq, hipStream_t stream);
void grx_launch_reset_all_generic(const KParams* dP, const void* tables, int N, int epb, uint32_t step, long long seq, uint8_t* mask, hipStream_t stream);
int grx_generic_tables_size(void);
int grx_tree_lds_bytes(int nb, int nlc, int nchain, int nsph, int waves);
int grx_tree_envs_per_wave(void);
int grx_launch_step_tree(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions, float delay,
                         long long common_step, const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream);
int grx_launch_step_tree_debug(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions,
                               long long common_step, const float* noise, const float* dbg, const StepSeq* sq, hipStream_t stream);
int grx_generic_ws_floats_per_env(int nb, int nlc);
// csrc/grx_tree16.hip: the tree kernel with a 16-lane group per env (four envs per wave)
int grx_tree_lds_bytes16(int nb, int nlc, int nchain, int nsph, int waves);
int grx_tree_envs_per_wave16(void);
int grx_launch_step_tree16(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions, float delay,
                           long long common_step, const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream);
int grx_launch_step_tree_debug16(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions,
                                 long long common_step, const float* noise, const float* dbg, const StepSeq* sq, hipStream_t stream);
void grx_launch_reset_all(const KParams* dP, int N, uint32_t step, const StepSeq* sq, uint8_t* mask, hipStream_t stream);
void grx_launch_mark(const int32_t* env_ids, int n, int N, uint8_t* mask, hipStream_t stream);
void grx_launch_set_state(const KParams* dP, int N, const float* root, const float* q, const float* qd, const int32_t* env_ids, int n, hipStream_t stream);
int grx_envs_per_block(void);
void grx_launch_refresh_heights(const KParams* dP, int N, int nh, hipStream_t stream);
void grx_launch_debug_terrain(const KParams* dP, const float* xy, int n, float* out, hipStream_t stream);
void grx_launch_debug_wall(const KParams* dP, const float* xyzr, int n, float* out, hipStream_t stream);
void grx_launch_refresh_rbs(const KParams* dP, int N, int nlinks, int pushed, hipStream_t stream);
void grx_launch_step_debug(const KParams* dP, int N, int heightfield, int waves, const float* actions, long long common_step, const float* noise,
                           const float* dbg, const StepSeq* sq, hipStream_t stream);
void grx_launch_step_debug_quad(const KParams* dP, int N, int heightfield, int waves, const float* actions, long long common_step, const float* noise,
                                const float* dbg, const StepSeq* sq, hipStream_t stream);
int grx_debug_rows(void);
int grx_debug_row_of(int what);   // 0: torques, 1: last_last_actions, 2: termination contact, 3: apply_reset
int grx_set_spin_word(unsigned long long* p);
int grx_set_spin_word_quad(unsigned long long* p);
}

namespace {
constexpr int NT = GRX_NUM_REWARD_TERMS;
constexpr int NSTAT = GRX_NSTAT;
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(GRX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct Timing {
    bool enabled = false;
    std::deque<std::pair<hipEvent_t, hipEvent_t>> pending;   // at most kMaxPending launches in flight
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double total_ms = 0;   // launches already drained from `pending`
    int64_t count = 0;
    static constexpr size_t kMaxPending = 64;
    int stride = 1;        // record every stride-th launch (an event pair costs the stream ~7 us of serialisation)
    int64_t tick = 0;
};

// Host run-ahead pacing and completion visibility.  A caller that never synchronises (rollout loops, bench.py) can
// enqueue hundreds of policy steps in a few milliseconds.  On the shared MI355X boxes the HIP runtime's view of
// completed work (hipEventQuery / stream synchronisation) was measured to lag the GPU by 10-80 ms, sporadically, for
// this low-occupancy workload (rocprofv3 kernel traces show the kernels themselves at a steady 75 us).  The library
// therefore keeps its own progress word in host-pinned memory: every step kernel stores, when it STARTS, the ticket of
// the work enqueued before it (complete by stream order: one launch per step, no trailing kernel), and the host reads
// that word directly -- to bound its
// run-ahead (kPaceAhead steps: about 20 ms of queued work, enough to ride out a descheduled host thread) and to let callers spin until everything enqueued so far has really finished
// (grx_wait_idle) without going through the runtime's signal machinery.
struct Pace {
    static constexpr int64_t kPaceAhead = 256;
    volatile int64_t* progress = nullptr;   // host-pinned, device-visible
    long long* d_progress = nullptr;        // device view of the same word
    int64_t issued = 0;                     // ticket of the last step enqueued
    hipStream_t last_stream = nullptr;      // stream of the last ticketed launch (grx_wait_idle sends the closing ticket there)
};
}  // namespace

struct grx_sim {
    grx_config cfg;
    int device = 0;
    int N = 0;
    int waves = 1;         // waves per 32-env block of the step kernel (1, 2 or 4)
    bool quad = false;     // four waves, a lane quad per env, 16 envs per block (grx_quad.hip): while the blocks fit the CUs in one round
    bool generic = false;  // model outside the fast kernel's lower-limb topology: generic-tree kernel (grx_generic.h)
    int rbs_mode = GRX_PUBLISH_NEVER, heights_mode = GRX_PUBLISH_EVERY_STEP;   // grx_publish_mode of GRX_T_RIGID_BODY_STATES / GRX_T_MEASURED_HEIGHTS
    int nd = GRX_ND;
    void* d_gen = nullptr; // GenTables (device)
    void* d_tree = nullptr; // TreeTab (device): the lane-group tree kernel (grx_tree.h) runs this model
    int tree_lds = 0;      // its dynamic LDS
    int tree_g = GRX_TREE_G;   // lanes per env of the tree kernel this handle launches: 8, or 16 (grx_tree16.hip) while 16-lane groups still fit the SIMDs in one round
    int tree_waves = 2;    // 8-env waves per block of the tree kernel: 2 while those blocks fit the CUs in one round, else 4 (a whole CU's LDS)
    float* d_ws = nullptr; // generic workspace
    int64_t seq = 0;       // launches of this handle that write statistics rows (steps, resets, debug steps; recorded ones too)
    int64_t eager_seq = 0; // ... the last of them that was launched eagerly (its rows are what grx_flush_stats reduces)
    bool stats_current = true;   // GRX_T_EPISODE_STATS already holds the statistics of the last EAGER launch (grx_flush_stats)
    bool has_recorded = false;   // a launch of this handle was recorded into a graph: replays advance the simulation without passing through the host
                                 // counters below, so grx_refresh no longer trusts them and always launches
    int64_t rbs_seq = -1, heights_seq = -1;   // launch number (seq) the on-demand tensors were last materialised for (grx_refresh)
    int64_t state_epoch = 0, rbs_epoch = -1, heights_epoch = -1;   // ... and the count of state writes outside steps (grx_set_state*, grx_reset_*)
    bool last_pushed = false;    // the last step was a _push_robots step (the base's vx, vy were overwritten after the sub-steps)
    bool prev_recorded = false;  // the last launch in host order was recorded into a graph: its successor must not fold "launch seq - 1"
    uint8_t* d_mask = nullptr;   // grx_reset_idx: per-env flags
    int gen_epb = 64;      // generic kernel: envs per (single-wave) block
    int gen_lds = 0;       // generic kernel: bytes of dynamic LDS when the workspace lives there (0: global memory)
    KParams hp;            // launch parameters: host image ...
    KParams* d_hp = nullptr;   // ... and the device copy every kernel reads through the constant address space
    KTables tab;           // host image of the device tables
    std::vector<void*> allocs;
    uint32_t reset_count = 0;
    Timing timing;
    Pace pace;
    // tensor table
    grx_tensor_desc desc[GRX_NUM_TENSORS];
    long long* prof_host = nullptr; int prof_blocks = 0;
    bool spin_bounded = false;    // a -DGRX_SPIN_LIMIT build: LDS spins are bounded and report through pace.progress[1]
    float* d_dbg = nullptr;       // grx_debug_post_physics: injected quantities [grx_debug_rows()][N]
    float* d_dbg_actions = nullptr;   // ... and the injected (already clipped) actions, (N, nd) row-major
};

namespace {

template <typename T>
int dalloc(grx_sim* s, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return fail(GRX_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemset(p, 0, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return fail(GRX_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    s->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return GRX_OK;
}

void set_desc(grx_tensor_desc* d, void* p, int dtype, int ndim, const int64_t* shape, const int64_t* stride) {
    d->data = p;
    d->dtype = dtype;
    d->ndim = ndim;
    for (int i = 0; i < 4; ++i) { d->shape[i] = i < ndim ? shape[i] : 1; d->stride[i] = i < ndim ? stride[i] : 1; }
}
// SoA [k][N] exposed as (N, k) with strides (1, N)
void desc_soa(grx_sim* s, int id, void* p, int dtype, int64_t k) {
    int64_t shape[2] = {s->N, k}, stride[2] = {1, s->N};
    set_desc(&s->desc[id], p, dtype, 2, shape, stride);
}
void desc_soa3(grx_sim* s, int id, void* p, int64_t a, int64_t b) {  // [a*b][N] as (N, a, b)
    int64_t shape[3] = {s->N, a, b}, stride[3] = {1, b * (int64_t)s->N, s->N};
    set_desc(&s->desc[id], p, GRX_F32, 3, shape, stride);
}
void desc_vec(grx_sim* s, int id, void* p, int dtype, int64_t n) {
    int64_t shape[1] = {n}, stride[1] = {1};
    set_desc(&s->desc[id], p, dtype, 1, shape, stride);
}
void desc_rows(grx_sim* s, int id, void* p, int64_t rows, int64_t cols) {  // row-major
    int64_t shape[2] = {rows, cols}, stride[2] = {cols, 1};
    set_desc(&s->desc[id], p, GRX_F32, 2, shape, stride);
}

// the fused kernel is specialised for the GR1 lower-limb tree: base + two 5-joint chains with
// axes x, z, y, y, y and unrotated joint frames (GR1T1_lower_limb.urdf / GR1T2_lower_limb.urdf)
int check_topology(const grx_model& m) {
    if (m.num_bodies != 1 + GRX_ND) return fail(GRX_ERR_UNSUPPORTED_MODEL, "HIP path supports 10-DOF lower-limb models (2 chains x 5 joints); got num_bodies=" + std::to_string(m.num_bodies));
    static const int axes[GRX_LEG] = {0, 2, 1, 1, 1};
    for (int side = 0; side < 2; ++side)
        for (int k = 0; k < GRX_LEG; ++k) {
            int b = 1 + side * GRX_LEG + k;
            int want_parent = k == 0 ? 0 : b - 1;
            if (m.parent[b] != want_parent) return fail(GRX_ERR_UNSUPPORTED_MODEL, "unsupported tree: body " + std::to_string(b) + " parent " + std::to_string(m.parent[b]));
            for (int a = 0; a < 3; ++a) {
                float want = a == axes[k] ? 1.f : 0.f;
                if (fabsf(m.joint_axis[b][a] - want) > 1e-6f) return fail(GRX_ERR_UNSUPPORTED_MODEL, "unsupported joint axis on body " + std::to_string(b));
            }
            static const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            for (int a = 0; a < 9; ++a)
                if (fabsf(m.joint_rot0[b][a] - I9[a]) > 1e-6f) return fail(GRX_ERR_UNSUPPORTED_MODEL, "rotated joint frame on body " + std::to_string(b));
        }
    for (int f = 0; f < 2; ++f)
        if (m.foot_body[f] != (f + 1) * GRX_LEG) return fail(GRX_ERR_UNSUPPORTED_MODEL, "feet must be the chain leaves");
    if (m.torso_body > 0 || m.forehead_body > 0) return fail(GRX_ERR_UNSUPPORTED_MODEL, "torso/forehead must ride on the base lump");
    return GRX_OK;
}

int build_side_tables(const grx_config& c, KTables& P, uint32_t* ll_mask, uint64_t* sp_mask) {
    const grx_model& m = c.model;
    memset(P.side, 0, sizeof P.side);
    *ll_mask = 0;
    *sp_mask = 0;
    for (int side = 0; side < 2; ++side) {
        SideConst& S = P.side[side];
        for (int k = 0; k < GRX_LEG; ++k) {
            int b = 1 + side * GRX_LEG + k, j = b - 1;
            for (int a = 0; a < 3; ++a) { S.body[k].r[a] = m.joint_pos[b][a]; S.body[k].com[a] = m.com[b][a]; }
            for (int a = 0; a < 6; ++a) S.body[k].Ic[a] = m.inertia[b][a];
            S.body[k].mass = m.mass[b];
            S.body[k].kp = c.kp[j]; S.body[k].kd = c.kd[j]; S.body[k].q0 = c.default_dof_pos[j];
            S.body[k].effort = m.dof_effort[j]; S.body[k].vlim = m.dof_vel_limit[j];
            S.body[k].qlo = m.dof_lower[j]; S.body[k].qhi = m.dof_upper[j];
            S.body[k].Klim = c.contact.k_limit * m.dof_effort[j];
            S.body[k].Clim = c.contact.c_limit * S.body[k].Klim;
            S.body[k].amin = c.clip_actions_min[j]; S.body[k].amax = c.clip_actions_max[j];
            float mid = (m.dof_lower[j] + m.dof_upper[j]) / 2, rng = m.dof_upper[j] - m.dof_lower[j];
            S.body[k].slo = mid - 0.5f * rng * c.soft_dof_pos_limit;
            S.body[k].shi = mid + 0.5f * rng * c.soft_dof_pos_limit;
        }
        for (int a = 0; a < 3; ++a) S.foot_pos[a] = m.foot_pos[side][a];
    }
    // spheres: chain spheres go to their side; base-lump spheres are split between the two lanes
    // at a LINK boundary (per-link force netting must see a whole link on one lane)
    std::vector<int> base_idx;
    for (int i = 0; i < m.num_spheres; ++i) {
        int b = m.sph_body[i];
        if (b < 0 || b >= m.num_bodies) return fail(GRX_ERR_INVALID_ARGUMENT, "sphere body out of range");
        if (m.sph_link[i] < 0 || m.sph_link[i] >= GRX_MAX_LINKS) return fail(GRX_ERR_INVALID_ARGUMENT, "sph_link out of range");
        if (b == 0) base_idx.push_back(i);
        else if (m.sph_flags[i] & (GRX_SPH_TERMINATE | GRX_SPH_PENALISE))
            return fail(GRX_ERR_UNSUPPORTED_MODEL, "terminating/penalised shapes must ride on the base lump");
    }
    // base_idx is sorted by link (model.py emits spheres sorted by (body, link)); cut near the middle
    size_t cut = base_idx.size() / 2;
    while (cut > 0 && cut < base_idx.size() && m.sph_link[base_idx[cut]] == m.sph_link[base_idx[cut - 1]]) ++cut;
    // fixed table layout per lane: [0..7] base-lump share, [8,9] chain body 2 (thigh_pitch), [10,11] body 3 (shank),
    // [12..15] body 4 (foot, anchored).  Unused slots are parked far above any terrain (r = -1e30).
    static const int cnt[GRX_LEG] = {0, 0, 2, 2, 4}, off[GRX_LEG] = {8, 8, 8, 10, 12};
    auto put = [&](SphC& o, int i, int slot) {
        o.x = m.sph_pos[i][0]; o.y = m.sph_pos[i][1]; o.z = m.sph_pos[i][2]; o.r = m.sph_radius[i];
        o.flags = m.sph_flags[i]; o.slot = slot; o.link_last = (m.sph_link[i] + 1) << 8; o.dmax = m.sph_damp_max[i];
    };
    for (int side = 0; side < 2; ++side) {
        SideConst& S = P.side[side];
        for (int i = 0; i < GRX_MAXSPH_SIDE; ++i) { S.sph[i] = SphC{0.f, 0.f, 0.f, -1e30f, 0u, -1, 0, 0.f}; }
        size_t lo = side == 0 ? 0 : cut, hi = side == 0 ? cut : base_idx.size();
        if (hi - lo > 8) return fail(GRX_ERR_UNSUPPORTED_MODEL, "more than 8 base-lump collision spheres per lane");
        for (size_t n = lo; n < hi; ++n) {
            SphC& o = S.sph[n - lo];
            put(o, base_idx[n], -1);
            bool last = (n + 1 == hi) || m.sph_link[base_idx[n + 1]] != m.sph_link[base_idx[n]];
            o.link_last |= last ? 1 : 0;
        }
        for (int k = 0; k < GRX_LEG; ++k) {
            int b = 1 + side * GRX_LEG + k, n = 0;
            for (int i = 0; i < m.num_spheres; ++i) {
                if (m.sph_body[i] != b) continue;
                bool foot = m.sph_flags[i] & (side == 0 ? GRX_SPH_FOOT_LEFT : GRX_SPH_FOOT_RIGHT);
                if (m.sph_flags[i] & (side == 0 ? GRX_SPH_FOOT_RIGHT : GRX_SPH_FOOT_LEFT))
                    return fail(GRX_ERR_UNSUPPORTED_MODEL, "foot shape on the wrong chain");
                if (n >= cnt[k]) return fail(GRX_ERR_UNSUPPORTED_MODEL, "collision shapes on chain body " + std::to_string(k) + " exceed the kernel's table (0,0,2,2,4)");
                if (foot != (k == GRX_LEG - 1)) return fail(GRX_ERR_UNSUPPORTED_MODEL, "anchored foot shapes must sit on the chain leaf");
                put(S.sph[off[k] + n], i, foot ? n : -1);
                if (n > 0 && m.sph_link[i] != sph_link(S.sph[off[k]]))   // GRX_T_CONTACT_FORCES nets a chain body's shapes into one row
                    return fail(GRX_ERR_UNSUPPORTED_MODEL, "the collision shapes of a chain body must belong to one URDF link");
                ++n;
            }
        }
        // bounding sphere of the shapes of chain bodies 2, 3, 4 (body frame): broad phase of the leg-vs-leg self-collision
        for (int bi = 0; bi < 3; ++bi) {
            const int k = 2 + bi;
            float cx = 0, cy = 0, cz = 0;
            for (int n = 0; n < cnt[k]; ++n) { const SphC& q = S.sph[off[k] + n]; cx += q.x; cy += q.y; cz += q.z; }
            cx /= cnt[k]; cy /= cnt[k]; cz /= cnt[k];
            float rad = 0;
            for (int n = 0; n < cnt[k]; ++n) {
                const SphC& q = S.sph[off[k] + n];
                if (q.r < 0) continue;
                rad = std::max(rad, sqrtf((q.x - cx) * (q.x - cx) + (q.y - cy) * (q.y - cy) + (q.z - cz) * (q.z - cz)) + q.r);
            }
            S.bs[bi][0] = cx; S.bs[bi][1] = cy; S.bs[bi][2] = cz; S.bs[bi][3] = rad;
        }
    }
    // self-collision pairs (grx_model.pair_a / pair_b) -> the fused kernel's tables: left-leg x right-leg body pairs
    // (every sphere pair of a listed link pair is in the list, so a 3 x 3 body mask carries it) and base-lump x thigh pairs
    for (int pi = 0; pi < m.num_pairs; ++pi) {
        int ia = m.pair_a[pi], ib = m.pair_b[pi];
        if (ia < 0 || ib < 0 || ia >= m.num_spheres || ib >= m.num_spheres) return fail(GRX_ERR_INVALID_ARGUMENT, "self-collision pair out of range");
        int ba = m.sph_body[ia], bb = m.sph_body[ib];
        if (ba > bb) { std::swap(ia, ib); std::swap(ba, bb); }
        if (ba == 0 && bb == 0) continue;
        auto side_of = [](int b) { return (b - 1) / GRX_LEG; };
        auto k_of = [](int b) { return (b - 1) % GRX_LEG; };
        if (ba == 0) {   // base lump x chain shape: thigh shapes only
            const int side = side_of(bb), k = k_of(bb);
            if (k != 2) return fail(GRX_ERR_UNSUPPORTED_MODEL, "base-lump self-collision with a chain body other than the thigh");
            SideConst& S = P.side[side];
            int tsel = -1;
            for (int n = 0; n < cnt[2]; ++n) {
                const SphC& q = S.sph[off[2] + n];
                if (q.x == m.sph_pos[ib][0] && q.y == m.sph_pos[ib][1] && q.z == m.sph_pos[ib][2]) tsel = n;
            }
            if (tsel < 0 || tsel > 1) return fail(GRX_ERR_UNSUPPORTED_MODEL, "base-lump / thigh self-collision: unknown thigh shape");
            int at = -1;   // one entry per base-lump sphere (entries of one link stay adjacent: the pairs arrive sorted by sphere)
            for (int n = 0; n < S.nbc; ++n)
                if (S.bc[n].x == m.sph_pos[ia][0] && S.bc[n].y == m.sph_pos[ia][1] && S.bc[n].z == m.sph_pos[ia][2] && S.bc[n].link == m.sph_link[ia] &&
                    S.bc[n].r == m.sph_radius[ia] && S.bc[n].dmax == m.sph_damp_max[ia]) at = n;   // (coincident spheres of different size or damping stay apart)
            if (at < 0) {
                if (S.nbc >= GRX_MAX_BC) return fail(GRX_ERR_UNSUPPORTED_MODEL, "base-lump / thigh self-collision table overflow");
                at = S.nbc++;
                BaseChainPair& e = S.bc[at];
                e.x = m.sph_pos[ia][0]; e.y = m.sph_pos[ia][1]; e.z = m.sph_pos[ia][2]; e.r = m.sph_radius[ia];
                e.dmax = m.sph_damp_max[ia]; e.tmask = 0; e.link = m.sph_link[ia]; e.pad = 0;
            }
            S.bc[at].tmask |= 1 << tsel;
        } else {
            if (side_of(ba) == side_of(bb)) return fail(GRX_ERR_UNSUPPORTED_MODEL, "self-collision within one leg chain");
            const int kl = side_of(ba) == 0 ? k_of(ba) : k_of(bb), kr = side_of(ba) == 0 ? k_of(bb) : k_of(ba);
            if (kl < 2 || kr < 2) return fail(GRX_ERR_UNSUPPORTED_MODEL, "self-collision shapes on a chain body without a shape table");
            *ll_mask |= 1u << ((kl - 2) * 3 + (kr - 2));
            // the sphere pair itself: table slots of the left-lane and the right-lane shape
            const int il = side_of(ba) == 0 ? ia : ib, ir = side_of(ba) == 0 ? ib : ia;
            auto slot_of = [&](int side, int k, int isph) {
                const SideConst& S = P.side[side];
                for (int n = 0; n < cnt[k]; ++n) {
                    const SphC& q = S.sph[off[k] + n];
                    if (q.x == m.sph_pos[isph][0] && q.y == m.sph_pos[isph][1] && q.z == m.sph_pos[isph][2]) return off[k] + n;
                }
                return -1;
            };
            const int sl = slot_of(0, kl, il), sr = slot_of(1, kr, ir);
            if (sl < 0 || sr < 0) return fail(GRX_ERR_UNSUPPORTED_MODEL, "leg x leg self-collision shape not in the kernel's tables");
            *sp_mask |= 1ull << ((sl - 8) * 8 + (sr - 8));
        }
    }
    return GRX_OK;
}

// GRX_T_RIGID_BODY_STATES tables of the fused kernel: chain links go to their leg's lane, the base lump's links alternate
void rot_to_quat(const float R[9], float q[4]) {   // row-major rotation -> xyzw, largest-component form
    const float t0 = 1 + R[0] - R[4] - R[8], t1 = 1 - R[0] + R[4] - R[8], t2 = 1 - R[0] - R[4] + R[8], t3 = 1 + R[0] + R[4] + R[8];
    if (t3 >= t0 && t3 >= t1 && t3 >= t2) { q[0] = R[7] - R[5]; q[1] = R[2] - R[6]; q[2] = R[3] - R[1]; q[3] = t3; }
    else if (t0 >= t1 && t0 >= t2) { q[0] = t0; q[1] = R[1] + R[3]; q[2] = R[2] + R[6]; q[3] = R[7] - R[5]; }
    else if (t1 >= t2) { q[0] = R[1] + R[3]; q[1] = t1; q[2] = R[5] + R[7]; q[3] = R[2] - R[6]; }
    else { q[0] = R[2] + R[6]; q[1] = R[5] + R[7]; q[2] = t2; q[3] = R[3] - R[1]; }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
int build_rbs_tables(const grx_model& m, RbsTables& T) {
    memset(&T, 0, sizeof T);
    if (m.num_links < 0 || m.num_links > GRX_MAX_LINKS) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_model.num_links out of range");
    int nbase = 0;
    std::vector<int> lists[2][GRX_LEG + 1];
    for (int l = 0; l < m.num_links; ++l) {
        const int b = m.link_body[l];
        if (b < 0 || b >= m.num_bodies) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_model.link_body out of range");
        if (b == 0) lists[(nbase++) & 1][0].push_back(l);
        else lists[(b - 1) / GRX_LEG][1 + (b - 1) % GRX_LEG].push_back(l);
    }
    for (int side = 0; side < 2; ++side) {
        int n = 0;
        for (int lvl = 0; lvl <= GRX_LEG; ++lvl) {
            T.off[side][lvl] = n;
            for (int l : lists[side][lvl]) {
                if (n >= GRX_RBS_MAX) return fail(GRX_ERR_UNSUPPORTED_MODEL, "more link frames per lane than the rigid-body-state table holds");
                RbsEntry& E = T.e[side][n++];
                E.px = m.link_pos[l][0]; E.py = m.link_pos[l][1]; E.pz = m.link_pos[l][2]; E.link = l;
                float q[4];
                rot_to_quat(m.link_rot[l], q);
                E.qx = q[0]; E.qy = q[1]; E.qz = q[2]; E.qw = q[3];
            }
        }
        T.off[side][GRX_LEG + 1] = n;
    }
    return GRX_OK;
}

/* c10::div_floor_floating (what torch.div(..., rounding_mode='floor') evaluates in float32) */
float torch_div_floor(float a, float b) {
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    if (div == 0.0f) return copysignf(0.0f, a / b);
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.0f;
    return fl;
}

// randomised base lump (oracle base_lump(); legged_robot.py:618-648)
void base_lump(const grx_model& m, float link_mass, const float link_com[3], float* M_out, float c_out[3], float I_out[6]) {
    float m1 = m.base_rest_mass, m2 = link_mass;
    float scale = m.base_link_mass > 0 ? m2 / m.base_link_mass : 1.f;
    float M = m1 + m2, c[3], I[6];
    for (int i = 0; i < 3; ++i) c[i] = (m1 * m.base_rest_com[i] + m2 * link_com[i]) / M;
    for (int i = 0; i < 6; ++i) I[i] = m.base_rest_inertia[i] + scale * m.base_link_inertia[i];
    const float* cs[2] = {m.base_rest_com, link_com};
    float ms[2] = {m1, m2};
    for (int k = 0; k < 2; ++k) {
        float d[3] = {cs[k][0] - c[0], cs[k][1] - c[1], cs[k][2] - c[2]};
        float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        I[0] += ms[k] * (dd - d[0] * d[0]); I[1] -= ms[k] * d[0] * d[1]; I[2] -= ms[k] * d[0] * d[2];
        I[3] += ms[k] * (dd - d[1] * d[1]); I[4] -= ms[k] * d[1] * d[2];
        I[5] += ms[k] * (dd - d[2] * d[2]);
    }
    *M_out = M;
    for (int i = 0; i < 3; ++i) c_out[i] = c[i];
    for (int i = 0; i < 6; ++i) I_out[i] = I[i];
}

}  // namespace

// ---- mesh_type 'trimesh': the reference's slope-corrected triangle mesh as per-cell tables
// legged_robot.py:903-921 hands PhysX convert_heightfield_to_trimesh(raster, slope_threshold) (isaacgym terrain_utils.py:286-350): the raster's
// heights on vertices that were MOVED by whole cells -- a vertex whose +x / -x / +y / -y (or, where those do not move it, diagonal) neighbour
// stands more than the threshold above it goes under that neighbour (:313-325), which turns the steep cell into a vertical face.  All vertices
// stay on grid points, so the mesh above one raster cell is: a plane per triangle half of the cell (the halves of :335-347; at a concave corner
// the two halves can sit on different levels) and vertical faces on grid lines.  Tables (grx_device.h KParams::tm_off, read by terrain_eval /
// wall_contact in grx_kernels.hip; the oracle builds its own in trimesh_build):
//   ground[cell][6]: corner heights of the top surface under half 0 (ty >= tx: e00, e01, e11) and half 1 (tx > ty: e00, e10, e11) -- the plane
//     a vertical ray hits at the half's centroid, evaluated at the cell's corners, in raster units (rounded: a sloped neighbour stretched over
//     two cells leaves half units);
//   walls[cell][8]: tops of the vertical faces on the sides x-, x+, y-, y+ (the rectangle both of whose ends the faces reach) and of the posts at
//     the corners 00, 10, 01, 11 (the end of a face that runs away from the corner), where they rise above the cell's own ground; else TM_NONE.
namespace {
constexpr int16_t TM_NONE = INT16_MIN;
struct TrimeshTables { std::vector<int16_t> ground, walls; };
struct TmVertex { double x, y, z; };
TrimeshTables build_trimesh_tables(const grx_config& c) {
    const int R = c.hf_rows, C = c.hf_cols;
    const int16_t* H = c.height_samples;
    const size_t n = (size_t)R * C;
    const double thr = (double)c.slope_threshold * ((double)c.horizontal_scale / (double)c.vertical_scale);   // raster units, in double like numpy (:310)
    auto at = [&](int i, int j) { return (int)H[(size_t)i * C + j]; };
    auto above = [&](int i, int j, int di, int dj) {   // the neighbour stands more than the threshold above (i, j)
        const int a = i + di, b = j + dj;
        return a >= 0 && a < R && b >= 0 && b < C && at(a, b) - at(i, j) > thr ? 1 : 0;
    };
    std::vector<int8_t> mx(n), my(n);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int sx = above(i, j, 1, 0) - above(i, j, -1, 0), sy = above(i, j, 0, 1) - above(i, j, 0, -1), sc = above(i, j, 1, 1) - above(i, j, -1, -1);
            mx[(size_t)i * C + j] = (int8_t)(sx != 0 ? sx : sc);   // xx += move_x + move_corners * (move_x == 0)   (:324)
            my[(size_t)i * C + j] = (int8_t)(sy != 0 ? sy : sc);
        }
    auto vertex = [&](int i, int j) { const size_t k = (size_t)i * C + j; return TmVertex{(double)(i + mx[k]), (double)(j + my[k]), (double)H[k]}; };
    auto triangle = [&](int a, int b, int second, TmVertex t[3]) {   // (ind0, ind3, ind1) and (ind0, ind2, ind3) of :339-347
        t[0] = vertex(a, b);
        t[1] = second ? vertex(a + 1, b) : vertex(a + 1, b + 1);
        t[2] = second ? vertex(a + 1, b + 1) : vertex(a, b + 1);
    };
    auto round16 = [](double z) { return (int16_t)lrint(std::min(std::max(z, -32767.0), 32767.0)); };
    // the plane of the highest triangle over the raster point (px, py): z there and its gradient
    auto plane_at = [&](double px, double py, double pl[3]) {
        const int ci = (int)floor(px), cj = (int)floor(py);
        bool found = false;
        for (int a = std::max(ci - 1, 0); a <= std::min(ci + 1, R - 2); ++a)
            for (int b = std::max(cj - 1, 0); b <= std::min(cj + 1, C - 2); ++b)
                for (int k = 0; k < 2; ++k) {
                    TmVertex t[3];
                    triangle(a, b, k, t);
                    const double ux = t[1].x - t[0].x, uy = t[1].y - t[0].y, vx = t[2].x - t[0].x, vy = t[2].y - t[0].y;
                    const double den = ux * vy - vx * uy;
                    if (fabs(den) < 1e-9) continue;   // projects to a segment: a vertical face
                    const double qx = px - t[0].x, qy = py - t[0].y;
                    const double w1 = (qx * vy - vx * qy) / den, w2 = (ux * qy - qx * uy) / den;
                    if (w1 < -1e-9 || w2 < -1e-9 || 1 - w1 - w2 < -1e-9) continue;
                    const double uz = t[1].z - t[0].z, vz = t[2].z - t[0].z, z = t[0].z + w1 * uz + w2 * vz;
                    if (!found || z > pl[0]) { pl[0] = z; pl[1] = (uz * vy - vz * uy) / den; pl[2] = (ux * vz - vx * uz) / den; found = true; }
                }
        return found;
    };
    TrimeshTables out;
    out.ground.resize(6 * n);
    out.walls.assign(8 * n, TM_NONE);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int i1 = std::min(i + 1, R - 1), j1 = std::min(j + 1, C - 1);
            int16_t* e = &out.ground[6 * ((size_t)i * C + j)];
            e[0] = e[3] = (int16_t)at(i, j); e[1] = (int16_t)at(i, j1); e[4] = (int16_t)at(i1, j); e[2] = e[5] = (int16_t)at(i1, j1);
            if (i > R - 2 || j > C - 2) continue;
            bool touched = false;   // only a moved vertex within the 3 x 3 cells around this one can change what lies over it
            for (int a = std::max(i - 1, 0); a <= std::min(i + 2, R - 1) && !touched; ++a)
                for (int b = std::max(j - 1, 0); b <= std::min(j + 2, C - 1) && !touched; ++b) touched = mx[(size_t)a * C + b] != 0 || my[(size_t)a * C + b] != 0;
            if (!touched) continue;
            for (int half = 0; half < 2; ++half) {
                const double px = i + (half ? 2.0 : 1.0) / 3, py = j + (half ? 1.0 : 2.0) / 3;   // the half's centroid
                double pl[3];
                if (!plane_at(px, py, pl)) continue;
                auto corner = [&](int ci, int cj) { return round16(pl[0] + pl[1] * (ci - px) + pl[2] * (cj - py)); };
                e[3 * half] = corner(i, j);
                e[3 * half + 1] = half ? corner(i + 1, j) : corner(i, j + 1);
                e[3 * half + 2] = corner(i + 1, j + 1);
            }
        }
    // the vertical faces, per unit segment of a grid line and per END of the segment: line x = X, y in [k, k + 1] -> fx[2 * (X * C + k) + end];
    // line y = Y, x in [k, k + 1] -> fy[2 * (Y * R + k) + end].  (Where three levels meet, the vertices slid along a face leave it triangular.)
    std::vector<int16_t> fx(2 * n, TM_NONE), fy(2 * n, TM_NONE);
    for (int a = 0; a < R - 1; ++a)
        for (int b = 0; b < C - 1; ++b)
            for (int k = 0; k < 2; ++k) {
                TmVertex t[3];
                triangle(a, b, k, t);
                if (fabs((t[1].x - t[0].x) * (t[2].y - t[0].y) - (t[2].x - t[0].x) * (t[1].y - t[0].y)) > 1e-9) continue;
                if (std::max({t[0].z, t[1].z, t[2].z}) <= std::min({t[0].z, t[1].z, t[2].z})) continue;
                const bool on_x_line = t[0].x == t[1].x && t[0].x == t[2].x, on_y_line = t[0].y == t[1].y && t[0].y == t[2].y;
                if (on_x_line == on_y_line) continue;   // a needle, or a face across the grid (axis-aligned steps make none)
                double pos[3];
                for (int q = 0; q < 3; ++q) pos[q] = on_x_line ? t[q].y : t[q].x;
                const int line = (int)(on_x_line ? t[0].x : t[0].y), lo = (int)std::min({pos[0], pos[1], pos[2]}), hi = (int)std::max({pos[0], pos[1], pos[2]});
                const int nlines = on_x_line ? R : C, nseg = on_x_line ? C - 1 : R - 1, stride = on_x_line ? C : R;
                if (line < 0 || line >= nlines) continue;
                std::vector<int16_t>& f = on_x_line ? fx : fy;
                for (int q = std::max(lo, 0); q < std::min(hi, nseg); ++q)
                    for (int end = 0; end < 2; ++end) {
                        const double where = q + end;
                        double top = -1e30;   // the triangle's highest point over `where`
                        for (int m0 = 0; m0 < 3; ++m0) {
                            const int m1 = (m0 + 1) % 3;
                            if (where < std::min(pos[m0], pos[m1]) || where > std::max(pos[m0], pos[m1])) continue;
                            top = std::max(top, pos[m0] == pos[m1] ? std::max(t[m0].z, t[m1].z) : t[m0].z + (t[m1].z - t[m0].z) * (where - pos[m0]) / (pos[m1] - pos[m0]));
                        }
                        int16_t& o = f[2 * ((size_t)line * stride + q) + end];
                        if (top > -1e29) o = std::max(o, round16(top));
                    }
            }
    auto face_x = [&](int X, int k, int end) { return X >= 0 && X < R && k >= 0 && k < C - 1 ? fx[2 * ((size_t)X * C + k) + end] : TM_NONE; };
    auto face_y = [&](int Y, int k, int end) { return Y >= 0 && Y < C && k >= 0 && k < R - 1 ? fy[2 * ((size_t)Y * R + k) + end] : TM_NONE; };
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int16_t* e = &out.ground[6 * ((size_t)i * C + j)];
            int16_t* w = &out.walls[8 * ((size_t)i * C + j)];
            const int16_t side[4] = {std::min(face_x(i, j, 0), face_x(i, j, 1)), std::min(face_x(i + 1, j, 0), face_x(i + 1, j, 1)),
                                     std::min(face_y(j, i, 0), face_y(j, i, 1)), std::min(face_y(j + 1, i, 0), face_y(j + 1, i, 1))};
            // the cell's own ground along the side (x- and y+ bound half 0, x+ and y- half 1)
            const int16_t ground[4] = {std::max(e[0], e[1]), std::max(e[4], e[5]), std::max(e[3], e[4]), std::max(e[1], e[2])};
            for (int q = 0; q < 4; ++q) w[q] = side[q] > ground[q] ? side[q] : TM_NONE;
            const int16_t away_x[4] = {face_x(i, j - 1, 1), face_x(i + 1, j - 1, 1), face_x(i, j + 1, 0), face_x(i + 1, j + 1, 0)};
            const int16_t away_y[4] = {face_y(j, i - 1, 1), face_y(j, i + 1, 0), face_y(j + 1, i - 1, 1), face_y(j + 1, i + 1, 0)};
            const int16_t corner[4] = {std::max(e[0], e[3]), e[4], e[1], std::max(e[2], e[5])};   // 00, 10, 01, 11
            for (int q = 0; q < 4; ++q) { const int16_t top = std::max(away_x[q], away_y[q]); w[4 + q] = top > corner[q] ? top : TM_NONE; }
        }
    return out;
}
}   // namespace

// ---- generic-tree path: device tables mirroring grx_generic.h's GenTables (kept in sync by the size check below)
namespace {
constexpr int GEN_MAXLC_H = 24;
struct GenTablesH {
    int32_t nb, nd, nsph, nlc;
    int32_t parent[GRX_MAX_BODIES];
    float axis[GRX_MAX_BODIES][3], rot0[GRX_MAX_BODIES][9], jpos[GRX_MAX_BODIES][3], mass[GRX_MAX_BODIES], com[GRX_MAX_BODIES][3], Ic[GRX_MAX_BODIES][6];
    float kp[GRX_MAX_DOFS], kd[GRX_MAX_DOFS], q0[GRX_MAX_DOFS], effort[GRX_MAX_DOFS], vlim[GRX_MAX_DOFS], qlo[GRX_MAX_DOFS], qhi[GRX_MAX_DOFS];
    float slo[GRX_MAX_DOFS], shi[GRX_MAX_DOFS], amin[GRX_MAX_DOFS], amax[GRX_MAX_DOFS], Klim[GRX_MAX_DOFS], Clim[GRX_MAX_DOFS];
    float arm[GRX_MAX_DOFS];
    int32_t sph_begin[GRX_MAX_BODIES + 1];
    float sx[GRX_MAX_SPHERES], sy[GRX_MAX_SPHERES], sz[GRX_MAX_SPHERES], sr[GRX_MAX_SPHERES], sdmax[GRX_MAX_SPHERES];
    int32_t sslot[GRX_MAX_SPHERES];
    int32_t slink[GRX_MAX_SPHERES];
    uint32_t link_flags[GEN_MAXLC_H];
    int32_t link_urdf[GEN_MAXLC_H];
    int32_t foot_body[2], foot_link[2];
    float foot_pos[2][3];
    int32_t torso_body, forehead_body;
    float torso_rot[9], forehead_rot[9];
    int32_t nlp;
    int32_t lp_a[48], lp_b[48], lp_ba[48], lp_bb[48];
    float lp_ca[48][4], lp_cb[48][4];
    int32_t lc_begin[GEN_MAXLC_H + 1];
};

int build_generic(grx_sim* s, const grx_config& c) {
    const grx_model& m = c.model;
    if ((int)sizeof(GenTablesH) != grx_generic_tables_size()) return fail(GRX_ERR_HIP, "GenTables layout mismatch between grx_capi.cpp and grx_generic.h");
    if (m.num_spheres > GRX_MAX_SPHERES) return fail(GRX_ERR_UNSUPPORTED_MODEL, "too many collision spheres");
    std::vector<GenTablesH> tv(1);
    GenTablesH& T = tv[0];
    memset(&T, 0, sizeof T);
    T.nb = m.num_bodies; T.nd = m.num_bodies - 1; T.nsph = m.num_spheres;
    for (int b = 0; b < m.num_bodies; ++b) {
        T.parent[b] = m.parent[b];
        if (b > 0 && (m.parent[b] < 0 || m.parent[b] >= b)) return fail(GRX_ERR_UNSUPPORTED_MODEL, "bodies must be listed parents first");
        for (int a = 0; a < 3; ++a) { T.axis[b][a] = m.joint_axis[b][a]; T.jpos[b][a] = m.joint_pos[b][a]; T.com[b][a] = m.com[b][a]; }
        for (int a = 0; a < 9; ++a) T.rot0[b][a] = m.joint_rot0[b][a];
        for (int a = 0; a < 6; ++a) T.Ic[b][a] = m.inertia[b][a];
        T.mass[b] = m.mass[b];
    }
    for (int j = 0; j < T.nd; ++j) {
        T.kp[j] = c.kp[j]; T.kd[j] = c.kd[j]; T.q0[j] = c.default_dof_pos[j];
        T.effort[j] = m.dof_effort[j]; T.vlim[j] = m.dof_vel_limit[j]; T.qlo[j] = m.dof_lower[j]; T.qhi[j] = m.dof_upper[j];
        T.Klim[j] = c.contact.k_limit * m.dof_effort[j];
        T.Clim[j] = c.contact.c_limit * T.Klim[j];
        T.arm[j] = m.dof_armature[j];
        T.amin[j] = c.clip_actions_min[j]; T.amax[j] = c.clip_actions_max[j];
        const float mid = (m.dof_lower[j] + m.dof_upper[j]) / 2, rng = m.dof_upper[j] - m.dof_lower[j];
        T.slo[j] = mid - 0.5f * rng * c.soft_dof_pos_limit;
        T.shi[j] = mid + 0.5f * rng * c.soft_dof_pos_limit;
    }
    // spheres sorted by carrying body (stable: model.py emits them sorted by (body, link) already)
    std::vector<int> order(m.num_spheres);
    for (int i = 0; i < m.num_spheres; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return m.sph_body[a] < m.sph_body[b]; });
    std::vector<int> link_ids;   // compact ids of the URDF links that carry shapes
    int foot_slots[2] = {0, 0};
    for (int f = 0; f < 2; ++f) T.foot_link[f] = -1;
    for (int k = 0; k < m.num_spheres; ++k) {
        const int i = order[k], b = m.sph_body[i];
        if (b < 0 || b >= m.num_bodies) return fail(GRX_ERR_INVALID_ARGUMENT, "sphere body out of range");
        T.sx[k] = m.sph_pos[i][0]; T.sy[k] = m.sph_pos[i][1]; T.sz[k] = m.sph_pos[i][2]; T.sr[k] = m.sph_radius[i]; T.sdmax[k] = m.sph_damp_max[i];
        int lc = -1;
        for (size_t t = 0; t < link_ids.size(); ++t) if (link_ids[t] == m.sph_link[i]) lc = (int)t;
        if (lc < 0) { lc = (int)link_ids.size(); link_ids.push_back(m.sph_link[i]); }
        if (lc >= GEN_MAXLC_H) return fail(GRX_ERR_UNSUPPORTED_MODEL, "too many links carry collision shapes");
        if (m.sph_link[i] < 0 || m.sph_link[i] >= GRX_MAX_LINKS) return fail(GRX_ERR_INVALID_ARGUMENT, "sph_link out of range");
        T.link_urdf[lc] = m.sph_link[i];
        T.slink[k] = lc;
        T.link_flags[lc] |= m.sph_flags[i] & (GRX_SPH_TERMINATE | GRX_SPH_PENALISE);
        T.sslot[k] = -1;
        for (int f = 0; f < 2; ++f)
            if (m.sph_flags[i] & (f == 0 ? GRX_SPH_FOOT_LEFT : GRX_SPH_FOOT_RIGHT)) {
                if (foot_slots[f] >= 4) return fail(GRX_ERR_UNSUPPORTED_MODEL, "more than 4 anchored spheres on a foot");
                T.sslot[k] = f * 4 + foot_slots[f]++;
                T.foot_link[f] = lc;
            }
    }
    T.nlc = (int)link_ids.size();
    {   // a compact link's shapes are contiguous (spheres sorted by (body, link)): ranges + bounding spheres (body frame)
        std::vector<int> pos_of(m.num_spheres);   // model sphere index -> position in the sorted tables
        for (int k = 0; k < m.num_spheres; ++k) pos_of[order[k]] = k;
        for (int l = 0; l <= T.nlc; ++l) T.lc_begin[l] = m.num_spheres;
        for (int k = m.num_spheres - 1; k >= 0; --k) T.lc_begin[T.slink[k]] = k;
        for (int l = T.nlc - 1; l >= 0; --l) if (T.lc_begin[l] > T.lc_begin[l + 1]) return fail(GRX_ERR_UNSUPPORTED_MODEL, "collision shapes of a link are not contiguous");
        auto bound = [&](int l, float out4[4]) {
            float c[3] = {0, 0, 0};
            const int b0 = T.lc_begin[l], b1 = T.lc_begin[l + 1];
            for (int k = b0; k < b1; ++k) { c[0] += T.sx[k]; c[1] += T.sy[k]; c[2] += T.sz[k]; }
            for (int a = 0; a < 3; ++a) c[a] /= (float)(b1 - b0);
            float rad = 0;
            for (int k = b0; k < b1; ++k)
                rad = std::max(rad, sqrtf((T.sx[k] - c[0]) * (T.sx[k] - c[0]) + (T.sy[k] - c[1]) * (T.sy[k] - c[1]) + (T.sz[k] - c[2]) * (T.sz[k] - c[2])) + T.sr[k]);
            out4[0] = c[0]; out4[1] = c[1]; out4[2] = c[2]; out4[3] = rad;
        };
        T.nlp = 0;
        for (int pi = 0; pi < m.num_pairs; ++pi) {
            int ka = pos_of[m.pair_a[pi]], kb = pos_of[m.pair_b[pi]];
            int la = T.slink[ka], lb = T.slink[kb];
            int ba = m.sph_body[order[ka]], bb = m.sph_body[order[kb]];
            if (ba > bb) { std::swap(la, lb); std::swap(ba, bb); }   // body bb is never the base (workspace addressing)
            bool seen = false;
            for (int q = 0; q < T.nlp; ++q) seen = seen || (T.lp_a[q] == la && T.lp_b[q] == lb);
            if (seen) continue;
            if (ba == bb) return fail(GRX_ERR_INVALID_ARGUMENT, "self-collision pair within one body");
            if (T.nlp >= 48) return fail(GRX_ERR_UNSUPPORTED_MODEL, "too many self-collision link pairs");
            T.lp_a[T.nlp] = la; T.lp_b[T.nlp] = lb; T.lp_ba[T.nlp] = ba; T.lp_bb[T.nlp] = bb;
            bound(la, T.lp_ca[T.nlp]); bound(lb, T.lp_cb[T.nlp]);
            ++T.nlp;
        }
    }
    for (int f = 0; f < 2; ++f) if (T.foot_link[f] < 0) return fail(GRX_ERR_UNSUPPORTED_MODEL, "a foot carries no collision shape");
    {
        int k = 0;
        for (int b = 0; b <= m.num_bodies; ++b) {
            while (k < m.num_spheres && m.sph_body[order[k]] < b) ++k;
            T.sph_begin[b] = k;
        }
    }
    for (int f = 0; f < 2; ++f) { T.foot_body[f] = m.foot_body[f]; for (int a = 0; a < 3; ++a) T.foot_pos[f][a] = m.foot_pos[f][a]; }
    T.torso_body = m.torso_body; T.forehead_body = m.forehead_body;
    memcpy(T.torso_rot, m.torso_rot, sizeof T.torso_rot);
    memcpy(T.forehead_rot, m.forehead_rot, sizeof T.forehead_rot);
    GenTablesH* d = nullptr;
    int rc = dalloc(s, &d, 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(d, &T, sizeof T, hipMemcpyHostToDevice));
    s->d_gen = d;
    rc = dalloc(s, &s->d_ws, (size_t)grx_generic_ws_floats_per_env(T.nb, T.nlc) * (size_t)s->N);
    if (rc) return rc;
    // ---- the lane-group tree kernel (grx_tree.h): chains of the tree -> lanes, depth levels -> steps
    if (const char* tv_ = getenv("GRX_TREE")) if (atoi(tv_) == 0) return GRX_OK;
    std::vector<TreeTab> tt(1);
    TreeTab& K = tt[0];
    memset(&K, 0, sizeof K);
    // lanes per env: 8 (eight envs per wave), or 16 (four envs per wave: the passes are bound by the tree's depth levels whatever the group
    // size, but twice the waves) while those waves still have a SIMD each -- 4096 envs on an MI355X, BASELINE.json config 5's per-GPU size
    int G = GRX_TREE_G;
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        if ((long long)s->N * GRX_TREE_GMAX <= 64ll * 4 * prop.multiProcessorCount) G = GRX_TREE_GMAX;
        if (const char* g_ = getenv("GRX_TREE_G")) { const int v = atoi(g_); if (v == GRX_TREE_G || v == GRX_TREE_GMAX) G = v; }
    }
    K.g = G; s->tree_g = G;
    K.nb = T.nb; K.nd = T.nd; K.nsph = T.nsph; K.nlc = T.nlc;
    memset(K.sched, 0xff, sizeof K.sched);
    std::vector<int> depth(T.nb, -1), lane_of(T.nb, -1), cont(T.nb, 0);
    int nchain = 0, nstep = 0;
    bool fits = true;
    for (int b = 1; b < T.nb && fits; ++b) {
        const int p = T.parent[b];
        depth[b] = p == 0 ? 0 : depth[p] + 1;
        TreeBody& tb = K.body[b];
        bool head = false;
        if (p != 0 && !cont[p]) { lane_of[b] = lane_of[p]; cont[p] = 1; }   // a body's first child continues its chain
        else { head = true; lane_of[b] = nchain++; }
        if (nchain > GRX_TREE_G || depth[b] >= GRX_TREE_LEVELS) { fits = false; break; }
        if (head && p == 0) K.heads0[K.nh0++] = lane_of[b];
        if (head && p != 0) { if (K.body[p].nhc >= 4) { fits = false; break; } K.body[p].hc[K.body[p].nhc++] = lane_of[b]; }
        K.sched[lane_of[b]][depth[b]] = (int8_t)b;
        nstep = std::max(nstep, depth[b] + 1);
        for (int a = 0; a < 3; ++a) { tb.axis[a] = T.axis[b][a]; tb.jpos[a] = T.jpos[b][a]; tb.com[a] = T.com[b][a]; }
        for (int a = 0; a < 9; ++a) tb.rot0[a] = T.rot0[b][a];
        for (int a = 0; a < 6; ++a) tb.Ic[a] = T.Ic[b][a];
        tb.mass = T.mass[b]; tb.parent = p; tb.sph_begin = T.sph_begin[b]; tb.sph_end = T.sph_begin[b + 1];
        tb.lane = lane_of[b]; tb.step = depth[b];
        tb.rot0_identity = 1;
        for (int a = 0; a < 9; ++a) if (tb.rot0[a] != ((a % 4 == 0) ? 1.f : 0.f)) tb.rot0_identity = 0;
    }
    if (!fits) return GRX_OK;   // more chains / levels than a lane group holds: the one-lane generic kernel runs it
    K.nchain = nchain; K.nstep = nstep;
    K.nstep_kin = 0;   // (the arms of the full body hang four levels deeper than anything the env pipeline reads)
    for (int b : {T.foot_body[0], T.foot_body[1], T.torso_body, T.forehead_body}) if (b >= 1) K.nstep_kin = std::max(K.nstep_kin, depth[b] + 1);
    for (int c = 0; c < GRX_TREE_GMAX; ++c) {
        K.first[c] = 1; K.last[c] = 0;
        bool any = false;
        for (int g = 0; g < nstep; ++g) if (K.sched[c][g] >= 0) { if (!any) K.first[c] = g; K.last[c] = g; any = true; }
    }
    for (int j = 0; j < T.nd; ++j) {
        TreeDof& d = K.dof[j];
        d.kp = T.kp[j]; d.kd = T.kd[j]; d.q0 = T.q0[j]; d.effort = T.effort[j]; d.vlim = T.vlim[j]; d.qlo = T.qlo[j]; d.qhi = T.qhi[j];
        d.slo = T.slo[j]; d.shi = T.shi[j]; d.amin = T.amin[j]; d.amax = T.amax[j]; d.Klim = T.Klim[j]; d.Clim = T.Clim[j]; d.lane = lane_of[j + 1]; d.arm = T.arm[j];
    }
    for (int k = 0; k < T.nsph; ++k) { TreeSph& q = K.sph[k]; q.x = T.sx[k]; q.y = T.sy[k]; q.z = T.sz[k]; q.r = T.sr[k]; q.dmax = T.sdmax[k]; q.slot = T.sslot[k]; q.link = T.slink[k]; }
    for (int l = 0; l < T.nlc; ++l) { K.link_flags[l] = T.link_flags[l]; K.link_urdf[l] = T.link_urdf[l]; }
    for (int f = 0; f < 2; ++f) { K.foot_body[f] = T.foot_body[f]; K.foot_link[f] = T.foot_link[f]; for (int a = 0; a < 3; ++a) K.foot_pos[f][a] = T.foot_pos[f][a]; }
    K.torso_body = T.torso_body; K.forehead_body = T.forehead_body;
    memcpy(K.torso_rot, T.torso_rot, sizeof K.torso_rot); memcpy(K.forehead_rot, T.forehead_rot, sizeof K.forehead_rot);
    K.sph_begin0 = T.sph_begin[0]; K.sph_end0 = T.sph_begin[1];
    {   // the contact pass's work list (TreeTab.cw): every body's shapes in chunks of two, the largest bodies first, dealt to the eight
        // lanes round by round; chunks of one body that land in the same round take turns at its accumulators.  A foot body is listed
        // even without shapes (its frame gives the sub-step averaged foot speed).
        struct Item { int body, s0, s1, turn; };
        std::vector<Item> items;
        std::vector<int> order;
        for (int b = 0; b < T.nb; ++b) order.push_back(b);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b_) { return (T.sph_begin[a + 1] - T.sph_begin[a]) > (T.sph_begin[b_ + 1] - T.sph_begin[b_]); });
        for (int b : order) {
            const int n0 = T.sph_begin[b], n1 = T.sph_begin[b + 1];
            const bool foot = b == T.foot_body[0] || b == T.foot_body[1];
            if (n1 == n0 && foot) items.push_back({b, n0, n0, 0});
            for (int i = n0; i < n1; i += 2) items.push_back({b, i, std::min(i + 2, n1), 0});
        }
        if (T.foot_body[0] < 1 || T.foot_body[1] < 1) return GRX_OK;   // (feet on the base: not this kernel's layout)
        const int rounds = ((int)items.size() + G - 1) / G;
        if (rounds > GRX_TREE_MAXCS) return GRX_OK;   // (the generic kernel runs it)
        memset(K.cw, 0xff, sizeof K.cw);
        K.ncs = rounds; K.nturn = 1;
        for (int r = 0; r < rounds; ++r)
            for (int ln = 0; ln < G; ++ln) {
                const size_t k = (size_t)r * G + ln;
                if (k >= items.size()) continue;
                Item it = items[k];
                for (int l2 = 0; l2 < ln; ++l2) if (K.cw[r][l2].body == it.body) it.turn = std::max(it.turn, K.cw[r][l2].turn + 1);
                K.cw[r][ln].body = (int8_t)it.body; K.cw[r][ln].s0 = (int8_t)it.s0; K.cw[r][ln].s1 = (int8_t)it.s1; K.cw[r][ln].turn = (int8_t)it.turn;
                K.nturn = std::max(K.nturn, it.turn + 1);
            }
    }
    K.nlp = T.nlp;
    for (int q = 0; q < T.nlp; ++q) { K.lp_ba[q] = (int16_t)T.lp_ba[q]; K.lp_bb[q] = (int16_t)T.lp_bb[q]; K.lp_a[q] = (int16_t)T.lp_a[q]; K.lp_b[q] = (int16_t)T.lp_b[q]; }
    {   // the broad phase's sphere pairs: every pair of grx_model.pair_a/b with its link pair; (ra + rb + margin)^2 -- the margin (0.1 mm) keeps the
        // test on the contact pass's centres a superset of sphere_pair's own (the two form a centre with differently rounded products)
        if (T.nsph > 255) return GRX_OK;
        std::vector<int> pos_of(m.num_spheres);
        {
            std::vector<int> ord(m.num_spheres);
            for (int i = 0; i < m.num_spheres; ++i) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return m.sph_body[a] < m.sph_body[b]; });
            for (int k = 0; k < m.num_spheres; ++k) pos_of[ord[k]] = k;
        }
        K.nsp = 0;
        for (int pi = 0; pi < m.num_pairs; ++pi) {
            int ka = pos_of[m.pair_a[pi]], kb = pos_of[m.pair_b[pi]];
            int la = T.slink[ka], lb = T.slink[kb];
            int lp = -1;
            for (int q = 0; q < T.nlp; ++q) if ((T.lp_a[q] == la && T.lp_b[q] == lb) || (T.lp_a[q] == lb && T.lp_b[q] == la)) lp = q;
            if (lp < 0) return fail(GRX_ERR_INVALID_ARGUMENT, "self-collision sphere pair without a link pair");
            const float rs = T.sr[ka] + T.sr[kb] + 1e-4f;
            K.sp[K.nsp].ab = (uint32_t)ka | ((uint32_t)kb << 8) | ((uint32_t)lp << 16);
            K.sp[K.nsp].r2 = rs * rs;
            ++K.nsp;
        }
        // the broad phase takes the table four rounds of the group's lanes at a time, without a bounds test: padded with pairs that never pass
        K.nsp_batches = (K.nsp + 4 * G - 1) / (4 * G);
        if (K.nsp_batches * 4 * G > GRX_MAX_PAIRS) return GRX_OK;
        for (int k = K.nsp; k < K.nsp_batches * 4 * G; ++k) { K.sp[k].ab = 0u; K.sp[k].r2 = -1.f; }
    }
    for (int l = 0; l <= GEN_MAXLC_H; ++l) K.lc_begin[l] = T.lc_begin[l];
    // waves per block: two while those blocks fit the device's CUs in one round (every wave still has a SIMD to itself and the CU's
    // LDS bandwidth is shared by two), else four -- a whole CU's LDS, one wave on every SIMD.  GRX_TREE_WAVES overrides (A/B runs).
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        const int epw = G == GRX_TREE_GMAX ? grx_tree_envs_per_wave16() : grx_tree_envs_per_wave();
        s->tree_waves = (s->N + 2 * epw - 1) / (2 * epw) <= prop.multiProcessorCount ? 2 : 4;
        if (const char* tw_ = getenv("GRX_TREE_WAVES")) { const int v = atoi(tw_); if (v == 1 || v == 2 || v == 4) s->tree_waves = v; }
    }
    auto lds_of = [&](int waves) { return G == GRX_TREE_GMAX ? grx_tree_lds_bytes16(T.nb, T.nlc, nchain, T.nsph, waves) : grx_tree_lds_bytes(T.nb, T.nlc, nchain, T.nsph, waves); };
    int lds = lds_of(s->tree_waves);
    while (lds > 160 * 1024 - 1024 && s->tree_waves > 1) { s->tree_waves /= 2; lds = lds_of(s->tree_waves); }
    if (lds > 160 * 1024 - 1024) return GRX_OK;   // the workspace of one wave does not fit a CU's LDS
    TreeTab* dk = nullptr;
    rc = dalloc(s, &dk, 1);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dk, &K, sizeof K, hipMemcpyHostToDevice));
    s->d_tree = dk; s->tree_lds = lds;
    return GRX_OK;
}
}  // namespace

extern "C" {

int grx_create(const grx_config* cfg, int device_id, grx_handle* out) {
    if (!cfg || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: null argument");
    if (cfg->abi_version != GRX_ABI_VERSION || cfg->struct_size != (int)sizeof(grx_config))
        return fail(GRX_ERR_ABI_MISMATCH, "grx_create: grx_config ABI mismatch (header " + std::to_string(sizeof(grx_config)) + " B, caller " + std::to_string(cfg->struct_size) + " B)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GRX_ERR_NO_DEVICE, "grx_create: no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: bad device id");
    HIP_TRY(hipSetDevice(device_id));
    const grx_config& c = *cfg;
    const grx_model& m = c.model;
    if (c.num_envs < 1) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: num_envs < 1");
    // (the step kernels form a column element's byte offset in 32 bits -- GCOL, grx_kernels.hip: rows * N * 4 < 4 GiB for the tallest column
    //  table, the height scan's GRX_MAX_HEIGHT_POINTS rows: 8.4 M envs, ~30 x what fits one MI355X)
    if ((long long)c.num_envs * GRX_MAX_HEIGHT_POINTS * 4 >= (1ll << 32)) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: num_envs too large for 32-bit column offsets");
    // the lower-limb topology runs on the fused lane-pair kernel; every other tree on the generic-tree kernel
    int rc = check_topology(m);
    bool armature = false;   // (the fused lower-limb kernels carry no armature term: such a model runs on the tree kernel)
    for (int j = 0; j + 1 < m.num_bodies && j < GRX_MAX_DOFS; ++j) {
        if (!(m.dof_armature[j] >= 0.f)) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: negative joint armature");
        armature = armature || m.dof_armature[j] != 0.f;
    }
    bool generic = rc != GRX_OK || armature || getenv("GRX_FORCE_GENERIC") != nullptr;
    // control_type outside {P, V, T}: the reference raises (legged_robot.py:707); a C caller gets the status code (the Python host raises NameError before)
    if (c.control_type != GRX_CONTROL_P && c.control_type != GRX_CONTROL_V && c.control_type != GRX_CONTROL_T)
        return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: control_type must be GRX_CONTROL_P, _V or _T");
    // the fused lower-limb kernels hold fixed-size shape / pair tables: a lower-limb model that overflows them (say, more base-lump spheres near
    // the thighs than GRX_MAX_BC) is not refused -- it runs on the tree kernel like every other model
    std::unique_ptr<KTables> side_tab(new KTables());
    uint32_t side_ll = 0; uint64_t side_sp = 0;
    if (!generic) {
        const int rs = build_side_tables(c, *side_tab, &side_ll, &side_sp);
        if (rs == GRX_ERR_UNSUPPORTED_MODEL) generic = true;
        else if (rs) return rs;
    }
    g_err.clear();
    const int nd = m.num_bodies - 1;
    if (nd < 1 || m.num_bodies > GRX_MAX_BODIES) return fail(GRX_ERR_UNSUPPORTED_MODEL, "grx_create: unsupported body count");
    const int nh = c.measure_heights ? c.num_height_points : 0;
    if (c.num_obs != 9 + 3 * nd) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: num_obs must be 9 + 3 * num_dofs");
    if (c.num_pri_obs != c.num_obs + 8 + nh || (!generic && c.num_pri_obs > GRX_MAX_PRI)) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: num_pri_obs mismatch");
    if (nh > GRX_MAX_HEIGHT_POINTS) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: too many height points");
    if (!generic && ((c.ankle_left_mask >> GRX_LEG) || (c.ankle_right_mask & 31u))) return fail(GRX_ERR_UNSUPPORTED_MODEL, "ankle masks must be split left/right");
    if (c.terrain_type == GRX_TERRAIN_HEIGHTFIELD && (!c.height_samples || !c.terrain_origins))
        return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: heightfield terrain needs height_samples and terrain_origins");

    grx_sim* s = new grx_sim();
    s->cfg = c;
    s->cfg.height_samples = nullptr;
    s->cfg.terrain_origins = nullptr;
    s->device = device_id;
    s->N = c.num_envs;
    s->generic = generic; s->nd = nd;
    const size_t N = (size_t)c.num_envs;
    KParams& P = s->hp;
    memset(&P, 0, sizeof P);
    P.N = c.num_envs; P.env_offset = c.env_offset; P.total_envs = c.total_envs; P.nd = nd;
    {   // waves per block: the step kernel needs a SIMD per wave (512 registers/lane).  Four waves per block while the
        // blocks fit the device's SIMDs in at most TWO rounds (measured on MI355X, rough terrain: 16384 envs = 2 rounds of
        // the four-wave layout 131 us, two-wave layout 152 us, one-wave layout 170 us; 20480 envs: 192 / 276 / 171 us),
        // one wave per block beyond; the two-wave layout is never the fastest any more (kept: GRX_WAVES_PER_BLOCK, tests)
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        const int simds = prop.multiProcessorCount * 4;
        const int nblocks = (c.num_envs + grx_envs_per_block() - 1) / grx_envs_per_block();
        s->waves = nblocks * 4 <= 2 * simds ? 8 : 1;   // (the pipeline with eight waves per block, two per SIMD: +3 % over four at 8192 envs, +1 % at 16384)
        if (const char* w = getenv("GRX_WAVES_PER_BLOCK")) {
            const int v = atoi(w);
            if (v == 1 || v == 2 || v == 4 || v == 8) s->waves = v;
        }
        // a lane QUAD per env, 16 envs per block: one block per CU needs all of the CU (four SIMDs' registers, 100+ KB of LDS),
        // so this layout pays exactly while its blocks fit the device in ONE round -- half the CUs would otherwise idle
        const int qblocks = (c.num_envs + grx_envs_per_block_quad() - 1) / grx_envs_per_block_quad();
        s->quad = !generic && s->waves >= 4 && qblocks <= prop.multiProcessorCount && !getenv("GRX_WAVES_PER_BLOCK");
        if (const char* q = getenv("GRX_LANES_PER_ENV")) s->quad = !generic && atoi(q) == 4;   // tests / A-B runs: 2 or 4
        if (s->quad) { s->waves = 8; if (const char* w = getenv("GRX_QUAD_WAVES")) s->waves = atoi(w) == 4 ? 4 : 8; }   // eight waves (two per SIMD) unless a test asks for the four-role pipeline
        // control_type 'V' / 'T' and heading_command (ABI 5; off in every registered task) live in the general one-wave layout only:
        // the wave pipelines keep the registered tasks' code path
        if (c.control_type != GRX_CONTROL_P || c.heading_command) { s->quad = false; s->waves = 1; }
    }
    const char* dbg = getenv("GRX_PUBLISH_DEBUG");   // (tools/: overrides the config either way)
    P.publish_debug = dbg ? atoi(dbg) : c.publish_reward_terms;
    P.seed = c.seed;
    P.sim_dt = c.sim_dt; P.decimation = c.decimation;
    for (int i = 0; i < 3; ++i) { P.gravity[i] = c.gravity[i]; P.init_pos[i] = c.init_pos[i]; }
    P.action_scale = c.action_scale;
    P.control_type = c.control_type; P.heading_command = c.heading_command ? 1 : 0;
    P.kn = c.contact.kn; P.dn = c.contact.dn; P.kt = c.contact.kt; P.ct = c.contact.ct; P.cv = c.contact.cv;
    P.terrain_friction = c.contact.terrain_friction;
    P.inv_kt = 1.0f / c.contact.kt;
    P.bounce_threshold = c.bounce_threshold_velocity; P.terrain_restitution = c.terrain_restitution;
    P.self_collisions = c.self_collisions;
    if (const char* sc = getenv("GRX_SELF_COLLISIONS")) P.self_collisions = atoi(sc);   // A/B runs (tools/)
    if (getenv("GRX_NO_RESTITUTION")) P.bounce_threshold = 1e30f;   // A/B runs: no contact ever bounces (tools/train_ab.py)
    P.termination_force = c.termination_force; P.termination_gravity_z = c.termination_gravity_z;
    P.max_episode_length = c.max_episode_length; P.max_episode_length_s = c.max_episode_length_s;
    P.resample_command_interval = c.resample_command_interval;
    for (int i = 0; i < 2; ++i) { P.cmd_lin_vel_x[i] = c.cmd_lin_vel_x[i]; P.cmd_lin_vel_y[i] = c.cmd_lin_vel_y[i]; P.cmd_ang_vel_yaw[i] = c.cmd_ang_vel_yaw[i]; }
    P.randomize_init_dof_pos = c.randomize_init_dof_pos; P.randomize_init_base_velocity = c.randomize_init_base_velocity;
    P.push_robots = c.push_robots; P.push_interval = c.push_interval; P.max_push_vel_xy = c.max_push_vel_xy;
    const float dtp = c.sim_dt * (float)c.decimation;
    for (int t = 0; t < NT; ++t) { P.reward_scale_dt[t] = c.reward_scale[t] * dtp; P.reward_sigma[t] = c.reward_sigma[t]; }
    P.only_positive_rewards = c.only_positive_rewards;
    P.base_height_target = c.base_height_target; P.swing_feet_height_target = c.swing_feet_height_target;
    P.feet_stumble_ratio = c.feet_stumble_ratio; P.feet_air_time_target = c.feet_air_time_target; P.feet_land_time_max = c.feet_land_time_max;
    P.soft_dof_vel_limit = c.soft_dof_vel_limit; P.soft_torque_limit = c.soft_torque_limit;
    P.knee_mask = c.knee_mask; P.hip_roll_mask = c.hip_roll_mask; P.hip_yaw_mask = c.hip_yaw_mask;
    P.ankle_left_mask = c.ankle_left_mask; P.ankle_right_mask = c.ankle_right_mask;
    P.num_pri_obs = c.num_pri_obs;
    P.obs_scale_action = c.obs_scale_action; P.obs_scale_lin_vel = c.obs_scale_lin_vel; P.obs_scale_ang_vel = c.obs_scale_ang_vel;
    P.obs_scale_gravity = c.obs_scale_gravity; P.obs_scale_dof_pos = c.obs_scale_dof_pos; P.obs_scale_dof_vel = c.obs_scale_dof_vel;
    P.obs_scale_height = c.obs_scale_height;
    P.add_noise = c.add_noise; P.noise_level = c.noise_level; P.noise_action = c.noise_action; P.noise_ang_vel = c.noise_ang_vel;
    P.noise_gravity = c.noise_gravity; P.noise_dof_pos = c.noise_dof_pos; P.noise_dof_vel = c.noise_dof_vel;
    P.clip_observations = c.clip_observations;
    P.terrain_type = c.terrain_type; P.measure_heights = c.measure_heights; P.nh = nh;
    memcpy(s->tab.height_points, c.height_points, sizeof s->tab.height_points);
    P.hf_rows = c.hf_rows; P.hf_cols = c.hf_cols;
    P.horizontal_scale = c.horizontal_scale; P.vertical_scale = c.vertical_scale; P.border_size = c.border_size;
    P.inv_hscale = 1.0f / c.horizontal_scale;
    P.vertical_faces = c.terrain_type == GRX_TERRAIN_HEIGHTFIELD && c.vertical_faces; P.tm_off = 0;   // (tm_off: with the cell tables below)
    P.hv_scale = c.vertical_scale / c.horizontal_scale;
    P.curriculum = c.curriculum; P.num_terrain_rows = c.num_terrain_rows; P.num_terrain_cols = c.num_terrain_cols;
    P.terrain_length = c.terrain_length;
    memcpy(P.torso_rot, m.torso_rot, sizeof P.torso_rot);
    memcpy(P.forehead_rot, m.forehead_rot, sizeof P.forehead_rot);
    P.has_torso = m.torso_body >= 0; P.has_forehead = m.forehead_body >= 0;
    if (!generic) { memcpy(s->tab.side, side_tab->side, sizeof s->tab.side); P.ll_mask = side_ll; P.sp_mask = side_sp; }

#define DA(field, count) do { rc = dalloc(s, &P.field, (count)); if (rc) { grx_destroy(s); return rc; } } while (0)
    DA(q, nd * N); DA(qd, nd * N); DA(root, 13 * N); DA(anchors, 24 * N);
    DA(last_actions, nd * N); DA(last_dof_vel, nd * N); DA(actions, nd * N); DA(torques, nd * N);
    DA(motor_strength, nd * N); DA(base_m, N); DA(base_c, 3 * N); DA(base_I, 6 * N); DA(friction, N); DA(restitution, N);
    DA(commands, 3 * N); DA(origins, 3 * N); DA(levels, N); DA(types, N);
    DA(air_time, 2 * N); DA(land_time, 2 * N); DA(feet_contact, 2 * N);
    DA(feet_height, 2 * N); DA(avg_force, 2 * N); DA(feet_force, 6 * N); DA(contact_forces, 3 * GRX_MAX_LINKS * N); DA(feet_pos, 6 * N); DA(avg_speed, 6 * N); DA(avg_speed_rpy, 6 * N);
    DA(base_heights_offset, N); DA(ep_len, N); DA(rew, N); DA(reset, N); DA(time_out, N); DA(term_contact, N);
    DA(base_lin_vel, 3 * N); DA(base_ang_vel, 3 * N); DA(proj_grav, 3 * N);
    DA(episode_sums, NT * N); DA(reward_terms, NT * N); DA(heights, (size_t)(nh > 0 ? nh : 1) * N);
    DA(obs, (size_t)c.num_obs * N + 64); DA(pri_obs, (size_t)c.num_pri_obs * N + 64);
    const int nblocks = (c.num_envs + grx_envs_per_block() - 1) / grx_envs_per_block();
    // a column per block of the writing kernel: 16-env blocks at the least for the fused kernels (lane quads) and the one-lane generic kernel;
    // the tree kernel goes down to one four-env wave per block (16 lanes per env, GRX_TREE_WAVES=1)
    P.stat_stride = (generic ? 8 : 2) * nblocks + 1;
    DA(stat_partial, (size_t)2 * NSTAT * P.stat_stride); DA(stat_nblocks, 2); DA(stat_hist, (size_t)GRX_STATS_HISTORY * NSTAT);
    DA(stats, NSTAT); DA(prof, (size_t)std::max(2 * nblocks, 64) * GRX_PROF_SLOTS);   // (16-env blocks in the quad layout; the tree kernel stamps blocks 0..63 whatever their size)
    rc = dalloc(s, &s->d_mask, N);
    if (rc) { grx_destroy(s); return rc; }
    if (c.publish_rigid_body_states < 0 || c.publish_rigid_body_states > 2 || c.publish_measured_heights < 0 || c.publish_measured_heights > 2) {
        grx_destroy(s); return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: publish_* must be a grx_publish_mode");
    }
    s->rbs_mode = c.publish_rigid_body_states;
    s->heights_mode = c.publish_measured_heights == GRX_PUBLISH_ON_REFRESH ? GRX_PUBLISH_ON_REFRESH : GRX_PUBLISH_EVERY_STEP;
    P.publish_rbs = s->rbs_mode == GRX_PUBLISH_EVERY_STEP;   // (the one-lane generic fallback does not publish link frames: cleared below)
    P.publish_heights = s->heights_mode == GRX_PUBLISH_EVERY_STEP;
    P.stash_pre_reset = s->rbs_mode == GRX_PUBLISH_ON_REFRESH || s->heights_mode == GRX_PUBLISH_ON_REFRESH;
    P.num_links = m.num_links;
    DA(rbs, s->rbs_mode != GRX_PUBLISH_NEVER ? (size_t)13 * GRX_MAX_LINKS * N : 1);
    DA(pre_q, P.stash_pre_reset ? nd * N : 1); DA(pre_qd, P.stash_pre_reset ? nd * N : 1); DA(pre_root, P.stash_pre_reset ? 13 * N : 1);
    DA(pre_push_vel, P.stash_pre_reset ? 2 * N : 1);
    if (s->rbs_mode == GRX_PUBLISH_ON_REFRESH) {   // grx_refresh: the joint tree as the model holds it (built only for the handles that refresh link
        // frames; a tree deeper than the refresh kernel walks -- such models run on the one-lane generic kernel -- loses the tensor, not the handle)
        std::unique_ptr<RefreshTab> rt(new RefreshTab());
        memset(rt.get(), 0, sizeof(RefreshTab));
        rt->nb = m.num_bodies; rt->nlinks = m.num_links;
        bool too_deep = false;
        for (int b = 1; b < m.num_bodies; ++b) {
            int n = 0;
            for (int x = b; x > 0 && n <= GRX_MAX_BODIES; x = m.parent[x]) ++n;
            too_deep = too_deep || n > GRX_REFRESH_MAXDEPTH;
        }
        for (int b = 1; b < m.num_bodies && !too_deep; ++b) {
            int chain[GRX_MAX_BODIES], n = 0;
            for (int x = b; x > 0 && n < GRX_MAX_BODIES; x = m.parent[x]) chain[n++] = x;
            rt->depth[b] = n;
            for (int d = 0; d < n; ++d) rt->path[b][d] = (int8_t)chain[n - 1 - d];
            bool ident = true;
            for (int a = 0; a < 9; ++a) { rt->rot0[b][a] = m.joint_rot0[b][a]; ident = ident && m.joint_rot0[b][a] == (a % 4 == 0 ? 1.f : 0.f); }
            rt->rot0_identity[b] = ident ? 1 : 0;
            for (int a = 0; a < 3; ++a) { rt->axis[b][a] = m.joint_axis[b][a]; rt->jpos[b][a] = m.joint_pos[b][a]; }
        }
        if (too_deep) s->rbs_mode = GRX_PUBLISH_NEVER;   // (desc[GRX_T_RIGID_BODY_STATES].data is cleared with the other NEVER handles below)
        else {
            RefreshTab* drt = nullptr;
            rc = dalloc(s, &drt, 1);
            if (rc) { grx_destroy(s); return rc; }
            HIP_TRY(hipMemcpy(drt, rt.get(), sizeof(RefreshTab), hipMemcpyHostToDevice));
            P.refresh_tab = drt;
        }
    }
    {
        std::vector<LinkTab> lt(1);
        memset(&lt[0], 0, sizeof(LinkTab));
        if (m.num_links < 0 || m.num_links > GRX_MAX_LINKS) { grx_destroy(s); return fail(GRX_ERR_INVALID_ARGUMENT, "grx_model.num_links out of range"); }
        lt[0].n = m.num_links;
        for (int l = 0; l < m.num_links; ++l) {
            if (m.link_body[l] < 0 || m.link_body[l] >= m.num_bodies) { grx_destroy(s); return fail(GRX_ERR_INVALID_ARGUMENT, "grx_model.link_body out of range"); }
            lt[0].body[l] = m.link_body[l];
            for (int a = 0; a < 3; ++a) lt[0].pos[l][a] = m.link_pos[l][a];
            for (int a = 0; a < 9; ++a) lt[0].rot[l][a] = m.link_rot[l][a];
        }
        LinkTab* dl = nullptr;
        rc = dalloc(s, &dl, 1);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(dl, &lt[0], sizeof(LinkTab), hipMemcpyHostToDevice));
        P.link_tab = dl;
    }
    if (!generic) {
        RbsTables rt;
        rc = build_rbs_tables(m, rt);
        if (rc) { grx_destroy(s); return rc; }
        RbsTables* drt = nullptr;
        rc = dalloc(s, &drt, 1);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(drt, &rt, sizeof rt, hipMemcpyHostToDevice));
        P.rbs_tab = drt;
    }
    if (c.num_terrain_rows > 255) { grx_destroy(s); return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: more than 255 terrain levels"); }
    float* base_mass_com = nullptr;
    rc = dalloc(s, &base_mass_com, 4 * N);
    if (rc) { grx_destroy(s); return rc; }
    if (c.terrain_type == GRX_TERRAIN_HEIGHTFIELD) {
        int16_t* dhf = nullptr;
        size_t n = (size_t)c.hf_rows * c.hf_cols;
        rc = dalloc(s, &dhf, n);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(dhf, c.height_samples, n * sizeof(int16_t), hipMemcpyHostToDevice));
        P.hf = dhf;
        {   // per-cell max of the four corners the bilinear terrain query blends (terrain_height in grx_kernels.hip)
            std::vector<int16_t> m4(n);
            for (int i = 0; i < c.hf_rows; ++i)
                for (int j = 0; j < c.hf_cols; ++j) {
                    const int i1 = std::min(i + 1, c.hf_rows - 1), j1 = std::min(j + 1, c.hf_cols - 1);
                    const int16_t* H = c.height_samples;
                    m4[(size_t)i * c.hf_cols + j] = std::max(std::max(H[(size_t)i * c.hf_cols + j], H[(size_t)i1 * c.hf_cols + j]),
                                                             std::max(H[(size_t)i * c.hf_cols + j1], H[(size_t)i1 * c.hf_cols + j1]));
                }
            {   // the four corners of every cell, packed (grx_device.h hf_cells); mesh_type 'trimesh': + the corrected mesh's tables behind them
                const size_t tm_off = P.vertical_faces ? ((n + 1) & ~(size_t)1) : 0;
                std::vector<uint32_t> cells(2 * (P.vertical_faces ? 3 * tm_off + 2 * n : n), 0u);
                for (int i = 0; i < c.hf_rows; ++i)
                    for (int j = 0; j < c.hf_cols; ++j) {
                        const int i1 = std::min(i + 1, c.hf_rows - 1), j1 = std::min(j + 1, c.hf_cols - 1);
                        const int16_t* H = c.height_samples;
                        const uint32_t h00 = (uint16_t)H[(size_t)i * c.hf_cols + j], h01 = (uint16_t)H[(size_t)i * c.hf_cols + j1];
                        const uint32_t h10 = (uint16_t)H[(size_t)i1 * c.hf_cols + j], h11 = (uint16_t)H[(size_t)i1 * c.hf_cols + j1];
                        cells[2 * ((size_t)i * c.hf_cols + j)] = h00 | (h01 << 16);
                        cells[2 * ((size_t)i * c.hf_cols + j) + 1] = h10 | (h11 << 16);
                    }
                if (P.vertical_faces) {
                    if (n > (size_t)0x0fffffff) { grx_destroy(s); return fail(GRX_ERR_INVALID_ARGUMENT, "grx_create: trimesh raster too large"); }
                    TrimeshTables tm = build_trimesh_tables(c);
                    for (size_t k = 0; k < n; ++k) {
                        const int16_t* e = &tm.ground[6 * k];
                        uint32_t* t0 = &cells[2 * (tm_off + 2 * k)];
                        t0[0] = (uint16_t)e[0] | ((uint32_t)(uint16_t)e[1] << 16); t0[1] = (uint16_t)e[2];
                        t0[2] = (uint16_t)e[3] | ((uint32_t)(uint16_t)e[4] << 16); t0[3] = (uint16_t)e[5];
                        const int16_t* w = &tm.walls[8 * k];
                        uint32_t* w0 = &cells[2 * (3 * tm_off + 2 * k)];
                        for (int q = 0; q < 4; ++q) w0[q] = (uint16_t)w[2 * q] | ((uint32_t)(uint16_t)w[2 * q + 1] << 16);
                        int16_t top = m4[k];   // the contact reach test (grx_rare.h) must see what the cell can touch: its ground corners and its faces' tops
                        for (int q = 0; q < 6; ++q) top = std::max(top, e[q]);
                        for (int q = 0; q < 8; ++q) top = std::max(top, w[q]);
                        m4[k] = top;
                    }
                    P.tm_off = (int32_t)tm_off;
                }
                uint32_t* dc = nullptr;
                rc = dalloc(s, &dc, cells.size());
                if (rc) { grx_destroy(s); return rc; }
                HIP_TRY(hipMemcpy(dc, cells.data(), cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                P.hf_cells = reinterpret_cast<const uint2*>(dc);
            }
            int16_t* dm4 = nullptr;
            rc = dalloc(s, &dm4, n);
            if (rc) { grx_destroy(s); return rc; }
            HIP_TRY(hipMemcpy(dm4, m4.data(), n * sizeof(int16_t), hipMemcpyHostToDevice));
            P.hf_max4 = dm4;
        }
        {   // coarse max map: max height over each 8x8-cell block dilated by 3 blocks (>= 2.4 m: robot reach 1.1 m
            // + travel within a policy step + bilinear support), used only to cull spheres that cannot touch
            int cr = (c.hf_rows + GRX_COARSE - 1) / GRX_COARSE, cc = (c.hf_cols + GRX_COARSE - 1) / GRX_COARSE;
            std::vector<int16_t> blk((size_t)cr * cc, INT16_MIN);
            for (int i = 0; i < c.hf_rows; ++i)
                for (int j = 0; j < c.hf_cols; ++j) {
                    int16_t& b = blk[(size_t)(i / GRX_COARSE) * cc + j / GRX_COARSE];
                    int16_t v = c.height_samples[(size_t)i * c.hf_cols + j];
                    if (v > b) b = v;
                }
            std::vector<float> cm((size_t)cr * cc);
            for (int i = 0; i < cr; ++i)
                for (int j = 0; j < cc; ++j) {
                    int16_t m = INT16_MIN;
                    for (int di = -3; di <= 3; ++di)
                        for (int dj = -3; dj <= 3; ++dj) {
                            int ii = i + di, jj = j + dj;
                            if (ii < 0 || jj < 0 || ii >= cr || jj >= cc) continue;
                            if (blk[(size_t)ii * cc + jj] > m) m = blk[(size_t)ii * cc + jj];
                        }
                    cm[(size_t)i * cc + j] = (float)m * c.vertical_scale;
                }
            float* dcm = nullptr;
            rc = dalloc(s, &dcm, cm.size());
            if (rc) { grx_destroy(s); return rc; }
            HIP_TRY(hipMemcpy(dcm, cm.data(), cm.size() * sizeof(float), hipMemcpyHostToDevice));
            P.coarse_max = dcm; P.coarse_rows = cr; P.coarse_cols = cc;
        }
        float* dor = nullptr;
        size_t no = (size_t)c.num_terrain_rows * c.num_terrain_cols * 3;
        rc = dalloc(s, &dor, no);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(dor, c.terrain_origins, no * sizeof(float), hipMemcpyHostToDevice));
        P.terrain_origins = dor;
    }
    // ---- per-env constants on the host (same arithmetic as the oracle's gro_create)
    {
        std::vector<float> h_strength(nd * N), h_bm(N), h_bc(3 * N), h_bI(6 * N), h_fr(N), h_rs(N), h_or(3 * N), h_q(nd * N), h_root(13 * N, 0.f), h_bmc(4 * N);
        std::vector<int32_t> h_lv(N, 0), h_ty(N, 0);
        std::vector<uint8_t> h_reset(N, 1);
        for (size_t i = 0; i < N; ++i) {
            uint32_t ge = (uint32_t)(c.env_offset + (int)i);
            float origin[3] = {0, 0, 0};
            if (c.terrain_type == GRX_TERRAIN_HEIGHTFIELD) {
                int max_init = c.curriculum ? c.max_init_terrain_level : c.num_terrain_rows - 1;
                float u = grx_rand(c.seed, ge, 0, GRX_RNG_INIT_LEVEL, 0);
                int lv = (int)(u * (float)(max_init + 1));
                if (lv > max_init) lv = max_init;
                // torch.div(arange(N), N / num_cols, rounding_mode='floor') evaluates in float32 (legged_robot.py:1177-1180)
                float per = (float)((double)c.total_envs / c.num_terrain_cols);
                int ty = (int)torch_div_floor((float)ge, per);
                if (ty > c.num_terrain_cols - 1) ty = c.num_terrain_cols - 1;
                h_lv[i] = lv; h_ty[i] = ty;
                const float* o = c.terrain_origins + ((size_t)lv * c.num_terrain_cols + ty) * 3;
                origin[0] = o[0]; origin[1] = o[1]; origin[2] = o[2];
            } else {
                int ncols = (int)floor(sqrt((double)c.total_envs));
                if (ncols < 1) ncols = 1;
                origin[0] = c.env_spacing * (float)(ge / (uint32_t)ncols);
                origin[1] = c.env_spacing * (float)(ge % (uint32_t)ncols);
            }
            for (int k = 0; k < 3; ++k) h_or[k * N + i] = origin[k];
            float fr = 1.f;
            if (c.randomize_friction) {
                uint32_t b = (uint32_t)(grx_rand(c.seed, ge, 0, GRX_RNG_INIT_DR, 0) * 64);
                if (b > 63) b = 63;
                fr = c.friction_range[0] + (c.friction_range[1] - c.friction_range[0]) * grx_rand(c.seed, b, 1, GRX_RNG_INIT_DR, 0);
            }
            h_fr[i] = fr;
            float rs = 0.f;
            if (c.randomize_restitution) {
                uint32_t b = (uint32_t)(grx_rand(c.seed, ge, 0, GRX_RNG_INIT_DR, 1) * 64);
                if (b > 63) b = 63;
                rs = c.restitution_range[0] + (c.restitution_range[1] - c.restitution_range[0]) * grx_rand(c.seed, b, 1, GRX_RNG_INIT_DR, 1);
            }
            h_rs[i] = rs;
            float lm = m.base_link_mass, lc[3] = {m.base_link_com[0], m.base_link_com[1], m.base_link_com[2]};
            if (c.randomize_base_mass) lm *= c.base_mass_range[0] + (c.base_mass_range[1] - c.base_mass_range[0]) * grx_rand(c.seed, ge, 0, GRX_RNG_INIT_DR, 2);
            if (c.randomize_base_com)
                for (int k = 0; k < 3; ++k) lc[k] += c.base_com_range[k][0] + (c.base_com_range[k][1] - c.base_com_range[k][0]) * grx_rand(c.seed, ge, 0, GRX_RNG_INIT_DR, 3 + k);
            float M, cc[3], I6[6];
            base_lump(m, lm, lc, &M, cc, I6);
            h_bm[i] = M;
            for (int k = 0; k < 3; ++k) h_bc[k * N + i] = cc[k];
            for (int k = 0; k < 6; ++k) h_bI[k * N + i] = I6[k];
            h_bmc[4 * i] = lm;
            for (int k = 0; k < 3; ++k) h_bmc[4 * i + 1 + k] = lc[k];
            for (int j = 0; j < nd; ++j) {
                float st = 1.f;
                if (c.randomize_motor_strength) st = c.motor_strength_range[0] + (c.motor_strength_range[1] - c.motor_strength_range[0]) * grx_rand(c.seed, ge, 0, GRX_RNG_INIT_DR, 8 + j);
                h_strength[j * N + i] = st;
                h_q[j * N + i] = c.default_dof_pos[j];
            }
            for (int k = 0; k < 3; ++k) h_root[k * N + i] = c.init_pos[k] + origin[k];
            h_root[6 * N + i] = 1.f;
        }
#define UP(dst, vec) HIP_TRY(hipMemcpy(P.dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice))
        UP(motor_strength, h_strength); UP(base_m, h_bm); UP(base_c, h_bc); UP(base_I, h_bI); UP(friction, h_fr); UP(restitution, h_rs);
        UP(origins, h_or); UP(levels, h_lv); UP(types, h_ty); UP(q, h_q); UP(root, h_root); UP(reset, h_reset);
        HIP_TRY(hipMemcpy(base_mass_com, h_bmc.data(), h_bmc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    {
        KTables* dt = nullptr;
        rc = dalloc(s, &dt, 1);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(dt, &s->tab, sizeof(KTables), hipMemcpyHostToDevice));
        P.tables = dt;
    }
    static_assert(sizeof(KParams) <= 4096, "KParams must fit the kernarg segment");

    // ---- tensor table
    const int64_t Ni = c.num_envs;
    desc_rows(s, GRX_T_OBS, P.obs, Ni, c.num_obs);
    desc_rows(s, GRX_T_PRI_OBS, P.pri_obs, Ni, c.num_pri_obs);
    desc_vec(s, GRX_T_REW, P.rew, GRX_F32, Ni);
    desc_vec(s, GRX_T_RESET, P.reset, GRX_U8, Ni);
    desc_vec(s, GRX_T_TIME_OUT, P.time_out, GRX_U8, Ni);
    desc_vec(s, GRX_T_EPISODE_LENGTH, P.ep_len, GRX_I64, Ni);
    desc_soa(s, GRX_T_DOF_POS, P.q, GRX_F32, nd);
    desc_soa(s, GRX_T_DOF_VEL, P.qd, GRX_F32, nd);
    desc_soa(s, GRX_T_TORQUES, P.torques, GRX_F32, nd);
    desc_soa(s, GRX_T_ACTIONS, P.actions, GRX_F32, nd);
    desc_soa(s, GRX_T_LAST_ACTIONS, P.last_actions, GRX_F32, nd);
    desc_soa(s, GRX_T_LAST_DOF_VEL, P.last_dof_vel, GRX_F32, nd);
    desc_soa(s, GRX_T_COMMANDS, P.commands, GRX_F32, 3);
    desc_soa(s, GRX_T_ROOT_STATES, P.root, GRX_F32, 13);
    desc_soa(s, GRX_T_BASE_LIN_VEL, P.base_lin_vel, GRX_F32, 3);
    desc_soa(s, GRX_T_BASE_ANG_VEL, P.base_ang_vel, GRX_F32, 3);
    desc_soa(s, GRX_T_PROJECTED_GRAVITY, P.proj_grav, GRX_F32, 3);
    desc_soa3(s, GRX_T_FEET_CONTACT_FORCE, P.feet_force, 2, 3);
    desc_soa3(s, GRX_T_FEET_POS, P.feet_pos, 2, 3);
    desc_soa(s, GRX_T_FEET_HEIGHT, P.feet_height, GRX_F32, 2);
    desc_soa(s, GRX_T_FEET_AIR_TIME, P.air_time, GRX_F32, 2);
    desc_soa(s, GRX_T_FEET_LAND_TIME, P.land_time, GRX_F32, 2);
    desc_soa(s, GRX_T_FEET_CONTACT, P.feet_contact, GRX_U8, 2);
    desc_soa(s, GRX_T_AVG_FEET_FORCE, P.avg_force, GRX_F32, 2);
    desc_soa3(s, GRX_T_AVG_FEET_SPEED, P.avg_speed, 2, 3);
    desc_soa3(s, GRX_T_AVG_FEET_SPEED_RPY, P.avg_speed_rpy, 2, 3);
    desc_soa(s, GRX_T_MEASURED_HEIGHTS, P.heights, GRX_F32, nh);
    desc_vec(s, GRX_T_BASE_HEIGHTS_OFFSET, P.base_heights_offset, GRX_F32, Ni);
    desc_rows(s, GRX_T_EPISODE_SUMS, P.episode_sums, NT, Ni);
    desc_rows(s, GRX_T_REWARD_TERMS, P.reward_terms, NT, Ni);
    desc_vec(s, GRX_T_TERRAIN_LEVELS, P.levels, GRX_I32, Ni);
    desc_vec(s, GRX_T_TERRAIN_TYPES, P.types, GRX_I32, Ni);
    desc_soa(s, GRX_T_ENV_ORIGINS, P.origins, GRX_F32, 3);
    desc_soa(s, GRX_T_MOTOR_STRENGTH, P.motor_strength, GRX_F32, nd);
    desc_vec(s, GRX_T_FRICTION, P.friction, GRX_F32, Ni);
    desc_rows(s, GRX_T_BASE_MASS_COM, base_mass_com, Ni, 4);
    desc_vec(s, GRX_T_TERM_CONTACT, P.term_contact, GRX_U8, Ni);
    desc_vec(s, GRX_T_EPISODE_STATS, P.stats, GRX_F32, NSTAT);
    desc_rows(s, GRX_T_EPISODE_STATS_HISTORY, P.stat_hist, GRX_STATS_HISTORY, NSTAT);
    desc_soa3(s, GRX_T_ANCHORS, P.anchors, 8, 3);
    desc_soa3(s, GRX_T_CONTACT_FORCES, P.contact_forces, GRX_MAX_LINKS, 3);
    desc_soa3(s, GRX_T_RIGID_BODY_STATES, P.rbs, GRX_MAX_LINKS, 13);
    if (s->rbs_mode == GRX_PUBLISH_NEVER) s->desc[GRX_T_RIGID_BODY_STATES].data = nullptr;
    s->prof_host = P.prof; s->prof_blocks = s->quad ? (c.num_envs + grx_envs_per_block_quad() - 1) / grx_envs_per_block_quad() : nblocks;
    // generic kernel: the per-body workspace goes to LDS when 16 envs' rows fit (155 KB for the 33-body robot: LDS round
    // trips are ~5x shorter than global ones and the kernel is bound by exactly those); else 64 envs per block over the
    // global workspace.  GRX_GENERIC_EPB = 16 / 32 / 64 forces a block size over the GLOBAL workspace (A/B runs).
    if (generic) {
        const size_t per_env = (size_t)grx_generic_ws_floats_per_env(m.num_bodies, 24) * sizeof(float);
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        // one such block fills a CU's LDS, so this only pays while all blocks run in ONE round (<= 16 envs per CU:
        // 4096 envs on an MI355X; measured 1.9 ms vs 3.2 ms at 4096 envs, but 7.5 ms vs 3.7 ms at 16384)
        if (per_env * 16 + 1024 <= 159 * 1024 && (c.num_envs + 15) / 16 <= prop.multiProcessorCount) { s->gen_epb = 16; s->gen_lds = (int)(per_env * 16); }
        if (const char* ev = getenv("GRX_GENERIC_EPB")) { const int v = atoi(ev); if (v == 16 || v == 32 || v == 64) { s->gen_epb = v; s->gen_lds = 0; } }
    }
    if (generic) {
        rc = build_generic(s, c);
        if (rc) { grx_destroy(s); return rc; }
        if (!s->d_tree) {   // the one-lane generic kernel (trees with more than eight chains): no link frames, neither every step nor on refresh (it does not stash the state before a reset)
            s->hp.publish_rbs = 0;
            if (s->rbs_mode == GRX_PUBLISH_EVERY_STEP) { s->rbs_mode = GRX_PUBLISH_NEVER; s->desc[GRX_T_RIGID_BODY_STATES].data = nullptr; }
            // (it reads the raw heights back from memory: always materialised there; it neither stashes the state before a reset)
            s->hp.publish_heights = 1; s->heights_mode = GRX_PUBLISH_EVERY_STEP;
            if (s->rbs_mode == GRX_PUBLISH_ON_REFRESH) { s->rbs_mode = GRX_PUBLISH_NEVER; s->desc[GRX_T_RIGID_BODY_STATES].data = nullptr; }
            s->hp.stash_pre_reset = 0;
        }
    }
    {
        void* hp_ = nullptr;
        HIP_TRY(hipHostMalloc(&hp_, 2 * sizeof(int64_t), hipHostMallocMapped | hipHostMallocCoherent));   // [0] progress, [1] spin report
        s->pace.progress = static_cast<volatile int64_t*>(hp_);
        s->pace.progress[0] = 0; s->pace.progress[1] = 0;
        void* dp_ = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp_, hp_, 0));
        s->pace.d_progress = static_cast<long long*>(dp_);
        // -DGRX_SPIN_LIMIT builds: an expired LDS spin stores its code in word [1] before it traps (grx_flags.h)
        s->spin_bounded = grx_set_spin_word(reinterpret_cast<unsigned long long*>(s->pace.d_progress + 1)) == 1;
        grx_set_spin_word_quad(reinterpret_cast<unsigned long long*>(s->pace.d_progress + 1));
    }
    {   // the parameter block is immutable from here on: upload it once
        rc = dalloc(s, &s->d_hp, 1);
        if (rc) { grx_destroy(s); return rc; }
        HIP_TRY(hipMemcpy(s->d_hp, &s->hp, sizeof(KParams), hipMemcpyHostToDevice));
    }
    *out = s;
    return GRX_OK;
}

int grx_destroy(grx_handle s) {
    if (!s) return GRX_OK;
    hipSetDevice(s->device);
    for (void* p : s->allocs) hipFree(p);
    for (auto& pr : s->timing.pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& pr : s->timing.pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (s->pace.progress) hipHostFree((void*)s->pace.progress);
    delete s;
    return GRX_OK;
}

// Spin on the pinned progress word until it reaches `target`, WITH a way out: a GPU fault (or steps that were recorded into
// a graph and never run) would otherwise hang the host forever.  The HIP runtime is polled every few thousand reads and a
// deadline (GRX_SPIN_TIMEOUT_S, default 60 s -- a step takes microseconds) bounds the wait.
static int spin_until(grx_sim* s, int64_t target, const char* who) {
    static const double limit_s = [] { const char* e = getenv("GRX_SPIN_TIMEOUT_S"); return e ? atof(e) : 60.0; }();
    if (*s->pace.progress >= target) return GRX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 1;; ++it) {
        if (*s->pace.progress >= target) return GRX_OK;
        if ((it & 0x3fff) == 0) {
            const hipError_t e = hipPeekAtLastError();
            if (e != hipSuccess && e != hipErrorNotReady)
                return fail(GRX_ERR_HIP, std::string(who) + ": device error while waiting for enqueued steps: " + hipGetErrorString(e));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
                return fail(GRX_ERR_HIP, std::string(who) + ": enqueued steps did not finish within GRX_SPIN_TIMEOUT_S");
        }
    }
}

// true while `st` records into a graph: a recorded step runs later (or never), so it neither paces nor takes a ticket
static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

// place of the next launch in the handle's sequence: statistics parity / history row, and -- unless the stream is recording a
// graph -- the ticket of the work before it, which that launch publishes when it starts
static StepSeq next_seq(grx_sim* s, hipStream_t st, bool capturing) {
    StepSeq q;
    q.seq = ++s->seq;
    q.progress = capturing ? nullptr : s->pace.d_progress;
    q.ticket_done = s->pace.issued;
    // a recorded launch runs later, any number of times, or never: neither it nor the first eager launch behind it may fold the rows of
    // "launch seq - 1" (an eager step after a captured-but-never-replayed one folded partials of a launch that had not run)
    q.fold_prev = (capturing || s->prev_recorded) ? 0 : 1;
    q.pad = 0;
    s->prev_recorded = capturing;
    if (!capturing) { ++s->pace.issued; s->pace.last_stream = st; s->stats_current = false; s->eager_seq = q.seq; }   // (recorded work changes nothing until it is replayed)
    return q;
}

// Stream capture starts from reduced statistics: the rows of the last EAGER launch cannot be reduced by a recorded kernel (replayed later,
// between other launches, it would reduce some other launch's rows of that parity and overwrite a history row that is not its own).
static int capture_needs_flushed_stats(grx_sim* s, const char* who) {
    if (s->stats_current) return GRX_OK;
    return fail(GRX_ERR_INVALID_ARGUMENT, std::string(who) + ": call grx_flush_stats() on this stream BEFORE stream capture begins (the episode statistics of the last eager "
                "launch are still unreduced, and a recorded kernel cannot reduce them)");
}

int grx_reset_all(grx_handle s, void* stream) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_reset_all: null handle");
    hipStream_t st = (hipStream_t)stream;
    // extras["episode"] of a full reset: mean of the running episode sums over all envs
    // (legged_robot.py:420-424); computed by the stats path with every env flagged.
    const bool capturing = stream_is_capturing(st);
    if (capturing) s->has_recorded = true;
    if (capturing) if (int rc = capture_needs_flushed_stats(s, "grx_reset_all")) return rc;
    ++s->state_epoch;
    uint32_t step = 0x80000000u + (s->reset_count++);
    // (the generic reset kernel does not fold its predecessor's statistics: reduce them now)
    if (s->generic && !s->stats_current) grx_launch_finalize(s->d_hp, s->eager_seq, nullptr, 0, st);
    const StepSeq q = next_seq(s, st, capturing);
    if (s->generic) {
        grx_launch_reset_all_generic(s->d_hp, s->d_gen, s->N, s->gen_epb, step, q.seq, nullptr, st);
        grx_launch_finalize(s->d_hp, q.seq, q.progress, q.progress ? s->pace.issued : 0, st);
        if (!capturing) s->stats_current = true;
    } else {
        grx_launch_reset_all(s->d_hp, s->N, step, &q, nullptr, st);
        if (capturing) grx_launch_finalize(s->d_hp, q.seq, nullptr, 0, st);   // recorded into a graph: carries its own reduction, see grx_step
    }
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

int grx_reset_idx(grx_handle s, const int32_t* env_ids, int32_t n, void* stream) {
    if (!s || (n > 0 && !env_ids)) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_reset_idx: null argument");
    if (n <= 0) return GRX_OK;   // legged_robot.py:387-388
    hipStream_t st = (hipStream_t)stream;
    const bool capturing = stream_is_capturing(st);
    if (capturing) s->has_recorded = true;
    if (capturing) if (int rc = capture_needs_flushed_stats(s, "grx_reset_idx")) return rc;
    ++s->state_epoch;
    uint32_t step = 0x80000000u + (s->reset_count++);
    if (s->generic && !s->stats_current) grx_launch_finalize(s->d_hp, s->eager_seq, nullptr, 0, st);
    const StepSeq q = next_seq(s, st, capturing);
    grx_launch_mark(env_ids, n, s->N, s->d_mask, st);
    if (s->generic) {
        grx_launch_reset_all_generic(s->d_hp, s->d_gen, s->N, s->gen_epb, step, q.seq, s->d_mask, st);
        grx_launch_finalize(s->d_hp, q.seq, q.progress, q.progress ? s->pace.issued : 0, st);
        if (!capturing) s->stats_current = true;
    } else {
        grx_launch_reset_all(s->d_hp, s->N, step, &q, s->d_mask, st);
        if (capturing) grx_launch_finalize(s->d_hp, q.seq, nullptr, 0, st);
    }
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

// the launchers' `heightfield` argument: 0 plane, 1 the raster as a heightfield, 2 mesh_type 'trimesh' (the *_trimesh kernels: the reference's corrected mesh)
static int terrain_mode(const grx_sim* s) { return s->cfg.terrain_type == GRX_TERRAIN_HEIGHTFIELD ? (s->cfg.vertical_faces ? 2 : 1) : 0; }

int grx_step(grx_handle s, grx_step_args* a, void* stream) {
    if (!s || !a) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_step: null argument");
    hipStream_t st = (hipStream_t)stream;
    static const bool no_pace = getenv("GRX_DEBUG_NO_PACE") != nullptr;
    const bool capturing = stream_is_capturing(st);
    if (capturing) s->has_recorded = true;
    if (!no_pace && !capturing)   // (the progress word trails the GPU by one launch: a step publishes its predecessor's ticket)
        if (int rc = spin_until(s, s->pace.issued - Pace::kPaceAhead, "grx_step")) return rc;
    std::pair<hipEvent_t, hipEvent_t> ev;
    const bool timed = !capturing && s->timing.enabled && (s->timing.tick++ % s->timing.stride) == 0;
    if (timed) {
        if (s->timing.pending.size() >= Timing::kMaxPending) {   // bound the event population (and the host's run-ahead)
            auto pr = s->timing.pending.front();
            s->timing.pending.pop_front();
            hipError_t qe;
            while ((qe = hipEventQuery(pr.second)) == hipErrorNotReady) {}   // spin: see Pace (any other error ends it)
            if (qe != hipSuccess) return fail(GRX_ERR_HIP, std::string("grx_step: timing event: ") + hipGetErrorString(qe));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            s->timing.total_ms += ms; ++s->timing.count;
            s->timing.pool.push_back(pr);
        }
        if (!s->timing.pool.empty()) { ev = s->timing.pool.back(); s->timing.pool.pop_back(); }
        else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
        HIP_TRY(hipEventRecord(ev.first, st));
    }
    // A step recorded into a graph may be replayed any number of times: its predecessor in execution order is then not launch
    // seq - 1, so it must not fold that launch's statistics rows (it would re-publish stale means and lose its own partials:
    // ADVICE r3).  Recorded steps therefore carry their statistics reduction with them, right behind the step: every replay leaves
    // GRX_T_EPISODE_STATS and its row of the history ring current.  The rows of the last EAGER launch must have been reduced before
    // the capture began (grx_flush_stats; ADVICE r4: recorded, that reduction ran at replay time on whatever launch then held the parity).
    if (capturing) if (int rc = capture_needs_flushed_stats(s, "grx_step")) return rc;
    const StepSeq q = next_seq(s, st, capturing);
    a->stats_slot = q.seq & (GRX_STATS_HISTORY - 1);
    a->stats_seq = q.seq;
    s->last_pushed = s->cfg.push_robots && s->cfg.push_interval > 0 && ((uint32_t)a->common_step_counter % (uint32_t)s->cfg.push_interval) == 0;
    if (s->generic)
    {
        if (s->d_tree) {
            if ((s->tree_g == GRX_TREE_GMAX ? grx_launch_step_tree16 : grx_launch_step_tree)(s->d_hp, s->d_tree, s->d_gen, s->N, s->tree_waves, s->tree_lds, terrain_mode(s), a->actions, a->delay_substeps,
                                     (long long)a->common_step_counter, a->noise_uniform, a->obs_out, a->pri_obs_out, &q, st))
                return fail(GRX_ERR_HIP, "grx_step: cannot raise the dynamic LDS limit of the tree kernel");
        } else if (grx_launch_step_generic(s->d_hp, s->d_gen, s->d_ws, s->N, s->gen_epb, s->gen_lds, terrain_mode(s), a->actions,
                                    a->delay_substeps, (long long)a->common_step_counter, a->noise_uniform, a->obs_out, a->pri_obs_out, q.seq, st))
            return fail(GRX_ERR_HIP, "grx_step: cannot raise the dynamic LDS limit of the generic kernel");
    }
    else
    {
        if (s->quad) grx_launch_step_quad(s->d_hp, s->N, terrain_mode(s), s->waves, a->actions, a->delay_substeps,
                                          (long long)a->common_step_counter, a->noise_uniform, a->obs_out, a->pri_obs_out, &q, st);
        else grx_launch_step(s->d_hp, s->N, terrain_mode(s), s->waves, a->actions, a->delay_substeps,
                             (long long)a->common_step_counter, a->noise_uniform, a->obs_out, a->pri_obs_out, &q, st);
    }
    if (timed) {
        HIP_TRY(hipEventRecord(ev.second, st));
        s->timing.pending.push_back(ev);
    }
    if ((s->generic && !s->d_tree) || capturing) {   // the one-lane generic kernel does not fold its predecessor's statistics: its own small kernel, with the step's ticket
        grx_launch_finalize(s->d_hp, q.seq, q.progress, q.progress ? s->pace.issued : 0, st);
        if (!capturing) s->stats_current = true;
    }
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

int grx_flush_stats(grx_handle s, void* stream) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_flush_stats: null handle");
    if (s->stats_current) return GRX_OK;
    if (stream_is_capturing((hipStream_t)stream)) return capture_needs_flushed_stats(s, "grx_flush_stats");
    // (the last launch in host order may be a recorded one that has not run: the rows to reduce are those of the last EAGER launch)
    grx_launch_finalize(s->d_hp, s->eager_seq, nullptr, 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    s->stats_current = true;
    return GRX_OK;
}

int grx_refresh(grx_handle s, int id, void* stream) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_refresh: null handle");
    if (id < 0 || id >= GRX_NUM_TENSORS) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_refresh: unknown tensor id");
    if (!s->desc[id].data) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_refresh: this handle does not publish that tensor (grx_config.publish_*)");
    hipStream_t st = (hipStream_t)stream;
    if (id == GRX_T_RIGID_BODY_STATES && s->rbs_mode == GRX_PUBLISH_ON_REFRESH) {
        if (s->rbs_seq == s->seq && s->rbs_epoch == s->state_epoch && !s->has_recorded && !stream_is_capturing(st)) return GRX_OK;   // current
        grx_launch_refresh_rbs(s->d_hp, s->N, s->cfg.model.num_links, s->last_pushed ? 1 : 0, st);
        HIP_TRY(hipGetLastError());
        if (!stream_is_capturing(st)) { s->rbs_seq = s->seq; s->rbs_epoch = s->state_epoch; }
    } else if (id == GRX_T_MEASURED_HEIGHTS && s->heights_mode == GRX_PUBLISH_ON_REFRESH) {
        if (s->heights_seq == s->seq && s->heights_epoch == s->state_epoch && !s->has_recorded && !stream_is_capturing(st)) return GRX_OK;
        grx_launch_refresh_heights(s->d_hp, s->N, s->hp.nh, st);
        HIP_TRY(hipGetLastError());
        if (!stream_is_capturing(st)) { s->heights_seq = s->seq; s->heights_epoch = s->state_epoch; }
    }
    return GRX_OK;
}

int grx_tensor(grx_handle s, int id, grx_tensor_desc* out) {
    if (!s || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_tensor: null argument");
    if (id < 0 || id >= GRX_NUM_TENSORS) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_tensor: unknown tensor id");
    *out = s->desc[id];
    return GRX_OK;
}

int grx_set_state(grx_handle s, const float* root, const float* q, const float* qd, void* stream) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_set_state: null handle");
    if (stream_is_capturing((hipStream_t)stream)) s->has_recorded = true;
    ++s->state_epoch;
    grx_launch_set_state(s->d_hp, s->N, root, q, qd, nullptr, 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

int grx_set_state_indexed(grx_handle s, const int32_t* env_ids, int32_t n, const float* root, const float* q, const float* qd, void* stream) {
    if (!s || (n > 0 && !env_ids)) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_set_state_indexed: null argument");
    if (n <= 0) return GRX_OK;
    if (stream_is_capturing((hipStream_t)stream)) s->has_recorded = true;
    ++s->state_epoch;
    grx_launch_set_state(s->d_hp, s->N, root, q, qd, env_ids, n, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

int grx_episode_stats(grx_handle s, float* host_out, void* stream) {
    if (!s || !host_out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_episode_stats: null argument");
    if (int rc = grx_flush_stats(s, stream)) return rc;
    HIP_TRY(hipMemcpyAsync(host_out, s->hp.stats, NSTAT * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GRX_OK;
}

// tests / tools: 0 = no spin of the wave pipelines has expired; else 'SP' << 48 | block << 32 | LDS address of the flag << 16 | value
// waited for (-DGRX_SPIN_LIMIT builds; *bounded = whether this library is one).  Readable after the kernel has trapped: the word
// is host memory.
int grx_debug_spin_report(grx_handle s, uint64_t* code, int* bounded) {
    if (!s || !code) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_spin_report: null argument");
    *code = (uint64_t)s->pace.progress[1];
    if (bounded) *bounded = s->spin_bounded ? 1 : 0;
    return GRX_OK;
}

int grx_stats_seq(grx_handle s, int64_t* out) {
    if (!s || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_stats_seq: null argument");
    *out = s->seq;
    return GRX_OK;
}

int grx_layout(grx_handle s, grx_layout_info* out) {
    if (!s || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_layout: null argument");
    memset(out, 0, sizeof *out);
    const char* hf = s->cfg.terrain_type == GRX_TERRAIN_HEIGHTFIELD ? "true" : "false";
    const bool tm = terrain_mode(s) == 2;   // mesh_type 'trimesh': the *_trimesh entries of the same kernels
    if (s->generic && s->d_tree) {
        out->lanes_per_env = s->tree_g; out->waves_per_block = s->tree_waves; out->envs_per_block = (s->tree_g == GRX_TREE_GMAX ? grx_tree_envs_per_wave16() : grx_tree_envs_per_wave()) * s->tree_waves;
        if (tm) snprintf(out->kernel, sizeof out->kernel, s->tree_g == GRX_TREE_GMAX ? "grx_step_tree16_trimesh<false>" : "grx_step_tree_trimesh<false>");
        else snprintf(out->kernel, sizeof out->kernel, s->tree_g == GRX_TREE_GMAX ? "grx_step_tree16<%s, false>" : "grx_step_tree<%s, false>", hf);
    } else if (s->generic) {
        out->lanes_per_env = 1; out->waves_per_block = 1; out->envs_per_block = s->gen_epb;
        if (tm) snprintf(out->kernel, sizeof out->kernel, "grx_step_generic_trimesh");
        else snprintf(out->kernel, sizeof out->kernel, "grx_step_generic<%s>", hf);
    } else if (s->quad) {
        out->lanes_per_env = 4; out->waves_per_block = s->waves; out->envs_per_block = grx_envs_per_block_quad();
        if (tm) snprintf(out->kernel, sizeof out->kernel, "grx_step_kernel_quad_trimesh<%d, false>", s->waves);
        else snprintf(out->kernel, sizeof out->kernel, "grx_step_kernel_quad<%s, %d, false>", hf, s->waves);
    } else {
        out->lanes_per_env = 2; out->waves_per_block = s->waves; out->envs_per_block = grx_envs_per_block();
        if (tm) snprintf(out->kernel, sizeof out->kernel, "grx_step_kernel_trimesh<%d, false>", s->waves);
        else snprintf(out->kernel, sizeof out->kernel, "grx_step_kernel<%s, %d, false>", hf, s->waves);
    }
    out->num_blocks = (s->N + out->envs_per_block - 1) / out->envs_per_block;
    return GRX_OK;
}

int grx_kernel_time_ms(grx_handle s, int enable, float* avg_ms, int64_t* launches) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_kernel_time_ms: null handle");
    double tot = s->timing.total_ms;
    int64_t n = s->timing.count;
    s->timing.total_ms = 0; s->timing.count = 0;
    for (auto& pr : s->timing.pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        ++n;
        s->timing.pool.push_back(pr);
    }
    s->timing.pending.clear();
    s->timing.enabled = enable != 0;
    s->timing.stride = enable > 1 ? enable : 1;
    s->timing.tick = 0;
    if (avg_ms) *avg_ms = n ? (float)(tot / n) : 0.f;
    if (launches) *launches = n;
    return GRX_OK;
}

// tools only (GRX_PROFILE_SECTIONS builds): copy the per-block section stamps to the host
int grx_debug_profile(grx_handle s, long long* out, int max_blocks) {
    if (!s || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_profile: null argument");
    int nb = s->prof_blocks < max_blocks ? s->prof_blocks : max_blocks;
    HIP_TRY(hipMemcpy(out, s->prof_host, (size_t)nb * GRX_PROF_SLOTS * sizeof(long long), hipMemcpyDeviceToHost));
    return nb;
}

// TEST-ONLY: post_physics_step of every env on injected state (include/grx.h).  Uploads the records into the SoA state
// buffers + the debug rows, then launches the DBG instantiation (no sub-steps) of the step kernel this handle runs.
int grx_debug_post_physics(grx_handle s, const grx_pipeline_state* ps, int apply_reset, const grx_step_args* a, void* stream) {
    if (!s || !ps || !a) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_post_physics: null argument");
    if (s->generic && !s->d_tree) return fail(GRX_ERR_UNSUPPORTED_MODEL, "grx_debug_post_physics: the fused kernels or the tree kernel only (this model runs on the one-lane generic kernel)");
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)s->N;
    const int nd = s->nd, rows = grx_debug_rows(), r_tor = grx_debug_row_of(0), r_lla = grx_debug_row_of(1), r_term = grx_debug_row_of(2), r_apply = grx_debug_row_of(3);
    int rc;
    if (!s->d_dbg) {
        if ((rc = dalloc(s, &s->d_dbg, (size_t)rows * N))) return rc;
        if ((rc = dalloc(s, &s->d_dbg_actions, (size_t)nd * N))) return rc;
    }
    std::vector<float> q(nd * N), qd(nd * N), root(13 * N), la(nd * N), lqd(nd * N), cmd(3 * N), air(2 * N), land(2 * N), bho(N), dbg((size_t)rows * N), act(nd * N);
    std::vector<uint8_t> fc(2 * N);
    std::vector<long long> ep(N);
    for (size_t i = 0; i < N; ++i) {
        const grx_pipeline_state& p = ps[i];
        for (int j = 0; j < nd; ++j) {
            q[j * N + i] = p.q[j]; qd[j * N + i] = p.qd[j]; la[j * N + i] = p.last_actions[j]; lqd[j * N + i] = p.last_dof_vel[j];
            act[i * nd + j] = p.actions[j];
            dbg[(r_tor + j) * N + i] = p.torques[j]; dbg[(r_lla + j) * N + i] = p.last_last_actions[j];
        }
        for (int k = 0; k < 13; ++k) root[k * N + i] = p.root[k];
        for (int k = 0; k < 3; ++k) cmd[k * N + i] = p.commands[k];
        for (int f = 0; f < 2; ++f) {
            air[f * N + i] = p.air_time[f]; land[f * N + i] = p.land_time[f]; fc[f * N + i] = p.contact_last[f] ? 1 : 0;
            dbg[(12 + f) * N + i] = p.avg_force[f];
            for (int k = 0; k < 3; ++k) {
                dbg[(0 + f * 3 + k) * N + i] = p.feet_force[f][k]; dbg[(6 + f * 3 + k) * N + i] = p.feet_pos[f][k];
                dbg[(14 + f * 3 + k) * N + i] = p.avg_speed[f][k];
            }
        }
        bho[i] = p.base_heights_offset; ep[i] = p.episode_length;
        dbg[(size_t)r_term * N + i] = p.term_contact ? 1.f : 0.f; dbg[(size_t)r_apply * N + i] = apply_reset ? 1.f : 0.f;
    }
    const KParams& P = s->hp;
#define UPS(dst, vec) HIP_TRY(hipMemcpyAsync(dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, st))
    UPS(P.q, q); UPS(P.qd, qd); UPS(P.root, root); UPS(P.last_actions, la); UPS(P.last_dof_vel, lqd); UPS(P.commands, cmd);
    UPS(P.air_time, air); UPS(P.land_time, land); UPS(P.feet_contact, fc); UPS(P.base_heights_offset, bho); UPS(P.ep_len, ep);
    UPS(s->d_dbg, dbg); UPS(s->d_dbg_actions, act);
#undef UPS
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors go out of scope
    const StepSeq sq = next_seq(s, st, false);
    // the post-physics half of the kernel this handle steps with (lane pairs: 1 / 4 / 8 waves; lane quads: 4 / 8; GRX_FORCE_GENERIC: the tree kernel)
    if (s->generic) {
        if ((s->tree_g == GRX_TREE_GMAX ? grx_launch_step_tree_debug16 : grx_launch_step_tree_debug)(s->d_hp, s->d_tree, s->d_gen, s->N, s->tree_waves, s->tree_lds, terrain_mode(s), s->d_dbg_actions,
                                       (long long)a->common_step_counter, a->noise_uniform, s->d_dbg, &sq, st))
            return fail(GRX_ERR_HIP, "grx_debug_post_physics: cannot raise the dynamic LDS limit of the tree kernel");
    } else if (s->quad) grx_launch_step_debug_quad(s->d_hp, s->N, terrain_mode(s), s->waves, s->d_dbg_actions,
                                            (long long)a->common_step_counter, a->noise_uniform, s->d_dbg, &sq, st);
    else grx_launch_step_debug(s->d_hp, s->N, terrain_mode(s), s->waves, s->d_dbg_actions, (long long)a->common_step_counter,
                               a->noise_uniform, s->d_dbg, &sq, st);
    HIP_TRY(hipGetLastError());
    return GRX_OK;
}

// TEST-ONLY: the step kernels' physics terrain query at n host points (x, y) -> host (height, dh/dx, dh/dy) each (include/grx.h)
int grx_debug_terrain(grx_handle s, const float* xy, int32_t n, float* out, void* stream) {
    if (!s || (n > 0 && (!xy || !out))) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_terrain: null argument");
    if (n <= 0) return GRX_OK;
    hipStream_t st = (hipStream_t)stream;
    float *dxy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dxy, (size_t)n * 2 * sizeof(float)));
    if (hipMalloc(&dout, (size_t)n * 3 * sizeof(float)) != hipSuccess) { hipFree(dxy); return fail(GRX_ERR_OUT_OF_MEMORY, "grx_debug_terrain: out of device memory"); }
    hipError_t e = hipMemcpyAsync(dxy, xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { grx_launch_debug_terrain(s->d_hp, dxy, n, dout, st); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(dxy); hipFree(dout);
    if (e != hipSuccess) return fail(GRX_ERR_HIP, std::string("grx_debug_terrain: ") + hipGetErrorString(e));
    return GRX_OK;
}

// TEST-ONLY, host only (no device needed): the per-cell tables grx_create builds for mesh_type 'trimesh' (build_trimesh_tables) -- ground int16[rows * cols][6],
// walls int16[rows * cols][8] (include/grx.h); the CPU test tier compares them with the oracle's own (gro_debug_trimesh_tables)
int grx_debug_trimesh_tables(const grx_config* cfg, int16_t* ground, int16_t* walls) {
    if (!cfg || !ground || !walls) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_trimesh_tables: null argument");
    if (cfg->terrain_type != GRX_TERRAIN_HEIGHTFIELD || !cfg->height_samples || cfg->hf_rows < 2 || cfg->hf_cols < 2)
        return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_trimesh_tables: a heightfield raster is needed");
    const TrimeshTables tm = build_trimesh_tables(*cfg);
    memcpy(ground, tm.ground.data(), tm.ground.size() * sizeof(int16_t));
    memcpy(walls, tm.walls.data(), tm.walls.size() * sizeof(int16_t));
    return GRX_OK;
}

// TEST-ONLY: mesh_type 'trimesh', spheres at rest (x, y, z, r) against the vertical faces next to them -> host (overlap * unit direction) each (include/grx.h)
int grx_debug_wall(grx_handle s, const float* xyzr, int32_t n, float* out, void* stream) {
    if (!s || (n > 0 && (!xyzr || !out))) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_debug_wall: null argument");
    if (n <= 0) return GRX_OK;
    hipStream_t st = (hipStream_t)stream;
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, (size_t)n * 4 * sizeof(float)));
    if (hipMalloc(&dout, (size_t)n * 3 * sizeof(float)) != hipSuccess) { hipFree(din); return fail(GRX_ERR_OUT_OF_MEMORY, "grx_debug_wall: out of device memory"); }
    hipError_t e = hipMemcpyAsync(din, xyzr, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { grx_launch_debug_wall(s->d_hp, din, n, dout, st); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) return fail(GRX_ERR_HIP, std::string("grx_debug_wall: ") + hipGetErrorString(e));
    return GRX_OK;
}

// spin until every step enqueued through this handle has finished on the GPU (reads the pinned progress word: the last
// step's own ticket is published by a one-thread kernel behind it)
int grx_wait_idle(grx_handle s) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "grx_wait_idle: null handle");
    if ((int64_t)*s->pace.progress < s->pace.issued) {
        grx_launch_ticket(s->pace.d_progress, (long long)s->pace.issued, s->pace.last_stream);
        HIP_TRY(hipGetLastError());
    }
    return spin_until(s, s->pace.issued, "grx_wait_idle");
}

const char* grx_last_error(void) { return g_err.c_str(); }
int grx_abi_version(void) { return GRX_ABI_VERSION; }

int grx_sizeof(int id) {
    switch (id) {
        case GRX_STRUCT_CONFIG: return (int)sizeof(grx_config);
        case GRX_STRUCT_STEP_ARGS: return (int)sizeof(grx_step_args);
        case GRX_STRUCT_TENSOR_DESC: return (int)sizeof(grx_tensor_desc);
        case GRX_STRUCT_PIPELINE_STATE: return (int)sizeof(grx_pipeline_state);
        case GRX_STRUCT_LAYOUT_INFO: return (int)sizeof(grx_layout_info);
        case GRX_STRUCT_MODEL: return (int)sizeof(grx_model);
    }
    return -1;
}

const char* grx_reward_term_name(int t) {
    static const char* names[NT] = {
        "action_diff", "action_diff_diff", "action_diff_knee", "cmd_diff_ang_vel_pitch", "cmd_diff_ang_vel_roll",
        "cmd_diff_ang_vel_yaw", "cmd_diff_base_height", "cmd_diff_base_orient", "cmd_diff_forehead_orient",
        "cmd_diff_lin_vel_x", "cmd_diff_lin_vel_y", "cmd_diff_lin_vel_z", "cmd_diff_torso_orient", "collision",
        "dof_acc_new", "dof_tor_ankle_feet_lift_up", "dof_tor_new", "dof_tor_new_hip_roll", "dof_vel_new",
        "dof_vel_new_knee", "feet_air_force", "feet_air_height", "feet_air_time", "feet_land_time",
        "feet_speed_xy_close_to_ground", "feet_speed_z_close_to_height_target", "feet_stumble", "limits_actions",
        "limits_dof_pos", "limits_dof_tor", "limits_dof_vel", "on_the_air", "pose_offset", "pose_offset_hip_yaw",
        "stand_still", "termination"};
    return (t >= 0 && t < NT) ? names[t] : "";
}

}  // extern "C"
