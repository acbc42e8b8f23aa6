// grx_kernels.hip -- the fused GRx environment-step kernel for MI355X (gfx950, wave64).
//
// One launch = one LeggedRobot.step() (reference legged_robot.py:222-246) for every env:
//   clip_actions -> 10 x [PD torque -> articulated-body dynamics + contact -> integrate]
//   -> state update -> termination -> 36 reward terms -> masked in-kernel reset
//   -> observations (+noise) -> history.
//
// MI355X mapping:
//   * ONE ENV PER LANE PAIR: lane 2e holds the left leg chain (5 joints), lane 2e+1 the right
//     leg.  The two chains only meet in the floating base, so the articulated inertia / bias of
//     each chain is combined with ONE DPP quad_perm exchange (27 values) per sub-step; no LDS,
//     no barrier.  A 64-lane wave advances 32 envs; all state stays in VGPRs across the fused
//     decimation loop (no per-sub-step tensor refresh: the reference moves >= 25 KB/env-step
//     through gym.refresh_*_tensor).
//   * dynamics in WORLD axes about the base origin: all spatial quantities share one frame, so
//     the articulated-body recursion needs no 6x6 frame transforms (the CPU oracle uses the
//     textbook body-frame form: the two implementations are independent).
//   * SoA state in HBM ([k][N]): a wave's loads are 128-B contiguous segments; AoS outputs
//     (obs (N,39), pri_obs (N,168), consumed row-major by the PPO GEMMs) are staged through LDS
//     and written as 16-B-per-lane coalesced rows.
//   * per-side robot constants (joint tree, inertias, gains, sphere tables) staged once per
//     block into LDS; per-launch scalars come through the scalar cache (s_load) from KParams.
//   * resets are masked in-kernel (counter-based Philox): no nonzero()/host sync
//     (legged_robot.py:292, 317 each force one in the reference).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/grx.h"
#include "grx_device.h"
#include "grx_math.h"
#include "grx_rng.h"
// A per-env column element with the byte offset formed in 32 bits (base: a uniform pointer of KParams): `global_load v, v_off, s[base]`
// instead of a 64-bit address per lane.  Valid while rows * N * sizeof(element) < 4 GiB (grx_create checks N).
#if !defined(GRX_NO_GCOL) && !(defined(GRX_LPE) && GRX_LPE == 4)   // (the lane-quad pipelines: measured 1 % slower with it -- their wave 0 keeps no addresses, see the env index behind the sub-steps)
#define GCOL(base, idx) (*reinterpret_cast<std::remove_reference_t<decltype(*(base))>*>(reinterpret_cast<char*>(const_cast<std::remove_const_t<std::remove_reference_t<decltype(*(base))>>*>(base)) + (uint32_t)((uint32_t)(idx) * (uint32_t)sizeof(*(base)))))
#else
#define GCOL(base, idx) ((base)[idx])
#endif

// Launch parameters live in device memory, uploaded once per handle, and are read through the CONSTANT address space
// (like the kernarg segment: scalar loads the compiler may hoist and merge across global stores).  Passing the ~1.5 KB
// struct by value instead exhausted the HIP runtime's kernarg pool every ~55 launches; its refill is a blocking
// wait whose wake-up was measured at 10-60 ms on a loaded host.
#define GRX_AS4 __attribute__((address_space(4)))
typedef const GRX_AS4 KParams& KP;
#define GRX_PARAMS(Pg) (*reinterpret_cast<const GRX_AS4 KParams*>(reinterpret_cast<uintptr_t>(Pg)))

#ifdef GRX_PROFILE_SECTIONS
#define GRX_TICK(i) do { __builtin_amdgcn_sched_barrier(0); long long t_ = clock64(); if (threadIdx.x == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + (i)] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
// sub-step sections accumulate in registers (g_tacc is a kernel-scope local); sched_barrier pins the code motion
#define GRX_TICKW(i) do { __builtin_amdgcn_sched_barrier(0); long long t_ = clock64(); if ((threadIdx.x & 63) == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + (i)] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define GRX_TICK2(i) do { __builtin_amdgcn_sched_barrier(0); long long t_ = clock64(); tacc[(i) - 16] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GRX_TICK2(i) do {} while (0)
#define GRX_TICK(i) do {} while (0)
#define GRX_TICKW(i) do {} while (0)
#endif

#ifndef GRX_W1_SCAN_BATCH
#define GRX_W1_SCAN_BATCH 31   // one-wave layout: heightfield gathers in flight per batch of the 121-point scan -- a lane's 61 points in two batches (8: -0.4 % at 32768 envs, -0.8 % at 131072; 16: -1 %; 61, one batch: 28 spilled dwords, -4 %; round 4, 458 registers)
#endif
#ifndef GRX_WPE
#define GRX_WPE 1   // waves per SIMD the step kernel's register budget is sized for
#endif
namespace {

constexpr int NT = GRX_NUM_REWARD_TERMS;
constexpr int NSTAT = GRX_NSTAT;
constexpr int LEG = GRX_LEG;
constexpr int EPB = 64 / LPE;  // envs per block: one wave64 = 32 lane pairs (16 lane quads in grx_quad.hip)

// ---- episode statistics (extras["episode"], legged_robot.py:387-388, 420-428) without a kernel of their own -----------------
// Every kernel that finishes episodes (step, reset, debug step) leaves per-block partial sums in the table of its launch parity
// (statistics row major, one column per block); the NEXT kernel of the handle reduces them -- the kernel boundary is the
// ordering, no fence, no atomics -- row t in block t, keeps the previous means when nobody reset, and files the result as
// the finished launch's row of the history ring.  grx_finalize_stats does the same on demand (grx_flush_stats).
GRX_DEV float* stat_row(KP P, long long seq, int t) { return P.stat_partial + ((size_t)(seq & 1) * NSTAT + t) * P.stat_stride; }
// one wave: (finished episodes, sum of row t) over the nb columns of the launch `seq`, same summation order every time
GRX_DEV void stat_reduce(KP P, long long seq, int t, int nb, int lane, float& cnt, float& s) {
    const float* const crow = stat_row(P, seq, NT);
    const float* const row = stat_row(P, seq, t);
    cnt = 0.f; s = 0.f;
    int b = lane;
    for (; b + 7 * 64 < nb; b += 8 * 64) {   // (summed in the same order as one by one)
        float c_[8], s_[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c_[u] = crow[b + u * 64]; s_[u] = row[b + u * 64]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { cnt += c_[u]; s += s_[u]; }
    }
    for (; b < nb; b += 64) { cnt += crow[b]; s += row[b]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cnt += __shfl_xor(cnt, off); s += __shfl_xor(s, off); }
}
GRX_DEV void stat_publish(KP P, long long seq, int t, float cnt, float s) {   // one lane
    float v = P.stats[t];   // nobody reset: the reference keeps the previous dict (reset_idx returns early, legged_robot.py:387-388)
    if (cnt > 0.f) {
        v = t == NT ? cnt : (t == NT + 1 ? s / (float)P.N : s / cnt / P.max_episode_length_s);
        P.stats[t] = v;
    }
    P.stat_hist[(size_t)(seq & (GRX_STATS_HISTORY - 1)) * NSTAT + t] = v;
}
// Which 32- / 16-env group of the batch a block of the STEP kernel works on.  Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md
// "Workgroup dispatch"): with the identity mapping every XCD's L2 sees envs from all over the batch -- i.e., with the terrain types assigned by env
// index (legged_robot.py:1177-1180), robots from all 20 columns of the 21.8 MB cell table.  Mapped like this an XCD works on ONE contiguous eighth
// of the envs (two or three terrain columns) and its 4 MiB L2 keeps that part of the raster.  A bijection of the block indices: results do not
// depend on it (the statistics rows are indexed by the group, as before).
#ifndef GRX_XCD_REMAP
#define GRX_XCD_REMAP 1
#endif
GRX_DEV int step_group() {
    const int b = blockIdx.x, n = gridDim.x;
    return (GRX_XCD_REMAP && (n & 7) == 0) ? (b & 7) * (n >> 3) + (b >> 3) : b;
}
// called by ONE full wave of every block at the start of a kernel: the statistics of launch sq.seq - 1, and its ticket
GRX_DEV void stats_fold_previous(KP P, const StepSeq& sq, int lane) {
    if (blockIdx.x == 0 && lane == 0 && sq.progress) __hip_atomic_store(sq.progress, sq.ticket_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!sq.fold_prev) return;   // (recorded into a graph: see StepSeq)
    const long long prev = sq.seq - 1;
    const int nb = P.stat_nblocks[prev & 1];
    for (int t = blockIdx.x; t < NSTAT; t += gridDim.x) {
        float cnt, s;
        stat_reduce(P, prev, t, nb, lane, cnt, s);
        if (lane == 0) stat_publish(P, prev, t, cnt, s);
    }
}
// sum of the terrain levels of the block's envs (row NT + 1): ballots over the bits of the level (levels < 256), no LDS
GRX_DEV float level_sum(int level, bool counted) {
    int sum = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) sum += __popcll(__ballot(counted && ((level >> bit) & 1))) << bit;
    return (float)sum;
}

// joint axes of a GR1 leg chain: hip_roll(x) hip_yaw(z) hip_pitch(y) knee_pitch(y) ankle_pitch(y)
__device__ constexpr int kAxis[LEG] = {0, 2, 1, 1, 1};

GRX_DEV R3 joint_rot_k(const R3& P, float c, float s, int ax) {
    if (ax == 0) return joint_rot<0>(P, c, s);
    if (ax == 1) return joint_rot<1>(P, c, s);
    return joint_rot<2>(P, c, s);
}
GRX_DEV V3 axis_k(const R3& R, int ax) { return ax == 0 ? R.cx : (ax == 1 ? R.cy : R.cz); }

// sin/cos of a joint angle: hardware v_sin/v_cos (|q| < pi: abs error < 1e-6, far below the contact noise)
GRX_DEV void grx_sincos(float x, float& s, float& c) {
    s = __sinf(x);
    c = __cosf(x);
}

// physics terrain query (oracle: terrain_query): height and gradient (gx, gy) = (dh/dx, dh/dy) of the surface under (x, y).
//  * heightfield: bilinear patch of the int16 raster (hf_cells: the four corners of a cell in one 8-byte gather);
//  * mesh_type 'trimesh' (HF == 2: kernels of their own, grx_step_*_trimesh, so that the heightfield's code is what it was): the reference's slope-corrected mesh (isaacgym terrain_utils.py:286-350 moves
//    vertices by whole cells, so it is still described per raster cell: grx_capi.cpp build_trimesh_tables): the plane of the
//    triangle half under the point, from that half's three corner heights -- again ONE 8-byte gather; the mesh's vertical
//    faces are a second contact (wall_contact below).
// In two halves: the gather (~1.2 us of memory latency on this part) and the interpolation that first USES it -- a caller
// with independent work puts it between the two.
#ifndef GRX_TE_BARRIER
#define GRX_TE_BARRIER (GRX_LPE == 4)   // see terrain_eval
#endif
struct TerrainRaw { int h00, h01, h10, h11; float tx, ty; };
// raster cell under (x, y) (index of its low corner) and the position inside it
GRX_DEV int terrain_locate(KP P, float x, float y, float& tx, float& ty) {
    float fx = (x + P.border_size) * P.inv_hscale;
    float fy = (y + P.border_size) * P.inv_hscale;
    fx = fminf(fmaxf(fx, 0.0f), (float)(P.hf_rows - 1));
    fy = fminf(fmaxf(fy, 0.0f), (float)(P.hf_cols - 1));
    const int ix = min((int)fx, P.hf_rows - 2), iy = min((int)fy, P.hf_cols - 2);
    tx = fx - (float)ix; ty = fy - (float)iy;
    return ix * P.hf_cols + iy;
}
// HF template parameter of everything below: the terrain the kernel is compiled for
enum { GRX_HF_PLANE = 0, GRX_HF_RASTER = 1, GRX_HF_TRIMESH = 2 };
// index into hf_cells of the corner record the physics reads for a point of `cell` (trimesh: the record of its triangle half)
template <int HF>
GRX_DEV int terrain_record(KP P, int cell, float tx, float ty) {
    return HF == GRX_HF_TRIMESH ? P.tm_off + 2 * cell + (ty >= tx ? 0 : 1) : cell;
}
GRX_DEV void terrain_unpack(uint2 cc, TerrainRaw& r) {
    r.h00 = (int16_t)(cc.x & 0xffffu); r.h01 = (int16_t)(cc.x >> 16);
    r.h10 = (int16_t)(cc.y & 0xffffu); r.h11 = (int16_t)(cc.y >> 16);
}
template <int HF>
GRX_DEV void terrain_gather(KP P, float x, float y, TerrainRaw& r) {
    if (!HF) return;
    const int cell = terrain_locate(P, x, y, r.tx, r.ty);
    terrain_unpack(P.hf_cells[terrain_record<HF>(P, cell, r.tx, r.ty)], r);   // one gather
}
template <int HF>
GRX_DEV float terrain_eval(KP P, const TerrainRaw& r, float& gx, float& gy) {
    gx = 0.0f; gy = 0.0f;
    if (!HF) return 0.0f;
#if GRX_TE_BARRIER
    // (round 6) nothing is scheduled across the start of a heightfield evaluation: the lane-quad kernels' waves keep the work their roles put between the
    // gathers and this first use of them there (4096 envs: 45.4 -> 44.9 us; the lane-pair kernels lose 1 % with it: profiles/r06_experiments.md)
    if (HF == GRX_HF_RASTER) __builtin_amdgcn_sched_barrier(0);
#endif
    const float tx = r.tx, ty = r.ty;
    const float h00 = (float)r.h00, h01 = (float)r.h01, h10 = (float)r.h10, h11 = (float)r.h11;
    float h;
    if (HF == GRX_HF_TRIMESH) {   // the record: (e00, e01 | e10, e11, -) of the half the point is in
        const bool up = ty >= tx;
        gx = up ? h10 - h01 : h01 - h00;
        gy = up ? h01 - h00 : h10 - h01;
        h = h00 + gx * tx + gy * ty;
        gx *= P.hv_scale; gy *= P.hv_scale;
    } else {
        h = (h00 * (1.0f - tx) + h10 * tx) * (1.0f - ty) + (h01 * (1.0f - tx) + h11 * tx) * ty;
        gx = ((h10 - h00) * (1.0f - ty) + (h11 - h01) * ty) * P.hv_scale;
        gy = ((h01 - h00) * (1.0f - tx) + (h11 - h10) * tx) * P.hv_scale;
    }
    return h * P.vertical_scale;
}
template <int HF>
GRX_DEV float terrain_height(KP P, float x, float y, float& gx, float& gy) {
    TerrainRaw r;
    terrain_gather<HF>(P, x, y, r);
    return terrain_eval<HF>(P, r, gx, gy);
}
// mesh_type 'trimesh': the sphere (centre c = (wx, wy, wz), radius r) against the vertical faces of the corrected mesh on the four sides
// of the raster cell under its centre and the posts (ends of faces that run away) at its four corners -- oracle wall_overlap + the wall
// branch of contact_forces.  A face spans its side from the ground to its top; the closest point on it is level with the centre,
// or on its upper edge.  ONE contact, the deepest; Hunt-Crossley normal force like the ground's, friction viscous and capped by the
// cone in the face's tangent plane.  ww: the cell's record (wall_gather), (tx, ty): the centre's position in the cell.
GRX_DEV uint4 wall_gather(KP P, float x, float y, float& tx, float& ty) {
    const int cell = terrain_locate(P, x, y, tx, ty);
    return reinterpret_cast<const uint4*>(P.hf_cells + 3 * (size_t)P.tm_off)[cell];
}
GRX_DEV bool wall_none(uint4 ww) { return (ww.x & ww.y & ww.z & ww.w) == 0x80008000u && (ww.x | ww.y | ww.z | ww.w) == 0x80008000u; }   // all eight tops INT16_MIN: no face at this cell
GRX_DEV V3 wall_contact(KP P, uint4 ww, float tx, float ty, float wz, float r, float dmax, V3 u, float mu) {
    V3 F = v3(0.f, 0.f, 0.f);
    if (wall_none(ww)) return F;   // (most cells)
    const float dx[2] = {tx * P.horizontal_scale, (tx - 1.0f) * P.horizontal_scale}, dy[2] = {ty * P.horizontal_scale, (ty - 1.0f) * P.horizontal_scale};
    const uint32_t w32[4] = {ww.x, ww.y, ww.z, ww.w};
    float best = r * r;   // squared distance of the closest face within reach
    V3 d = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int top = (int16_t)((q & 1) ? (w32[q >> 1] >> 16) : (w32[q >> 1] & 0xffffu));
        const float ex = q < 2 ? dx[q] : (q < 4 ? 0.0f : dx[(q - 4) & 1]);
        const float ey = q < 2 ? 0.0f : (q < 4 ? dy[q - 2] : dy[(q - 4) >> 1]);
        const float ez = fmaxf(wz - (float)top * P.vertical_scale, 0.0f);
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (top != -32768 && d2 < best && d2 > 1e-18f) { best = d2; d = v3(ex, ey, ez); }
    }
    if (best < r * r) {
        const float dist = grx_sqrt(best);
        const V3 n = d * grx_rcp(dist);
        const float pen = r - dist;
        const float un = dot(u, n);
        const float cd = fminf(P.kn * pen * P.dn, dmax);
        const float fn = fmaxf(P.kn * pen - cd * un, 0.0f);
        const V3 ut = u - n * un;
        const float sp = grx_sqrt(dot(ut, ut));
        const float ft = fminf(P.cv * sp, mu * fn);
        F = n * fn;
        if (sp > 1e-9f) F = F - ut * (ft * grx_rcp(sp));
    }
    return F;
}

// the vertical faces next to CNT spheres of one body (S[i]: radius r, damping cap dmax), added to the body's wrench about O
template <int CNT, typename SphT>
GRX_DEV void wall_pass(KP P, const SphT* S, V3 w, V3 v, V3 O, float mu, float hmax, const V3* xr, const uint4* ww, const TerrainRaw* raw, V3& fa, V3& fl) {
    // ww: the cells' face records, gathered with the ground records (foot_probe): the position inside the cell is the ground lookup's
    bool some = false;
#pragma unroll
    for (int i = 0; i < CNT; ++i) some = some || !wall_none(ww[i]);
    if (!__any(some)) return;   // nobody in the wave stands next to a face
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const float wz = O.z + xr[i].z;
        if (wz - S[i].r <= hmax) {
            const V3 F = wall_contact(P, ww[i], raw[i].tx, raw[i].ty, wz, S[i].r, S[i].dmax, v + cross(w, xr[i]), mu);
            fa = fa + cross(xr[i], F); fl = fl + F;
        }
    }
}

// per-lane persistent simulation state
struct LaneState {
    float q[LEG], qd[LEG];
    V3 pos, vel, ang;
    float qx, qy, qz, qw;
    float ax[4], ay[4];  // friction anchors of the 4 spheres of this lane's foot
    float vimp[4];       // normal approach speed at the first touch of the current contact (restitution)
    uint32_t anchor_on;  // bit i: anchor i active
};

struct LaneConst {  // per-env constants held in registers
    float strength[LEG];
    float base_m;
    V3 base_c;
    S3 base_I;
    float mu;
    float om_e;   // 1 - restitution of the foot / terrain pair (legged_robot.py:565-575; PhysX combines by averaging)
    float hmax;   // upper bound of the terrain height within reach during this policy step
};

struct SubstepOut {
    V3 foot_force;   // net contact force on this lane's foot link (world)
    bool term;       // a terminating link handled by this lane carries |F| > threshold
    bool pen;        // a penalised link carries |F| > 0.1 (count in pen_count)
    float pen_count;
};

struct FootKin { V3 pos, vel, ang; };  // foot link origin (world), its velocity, body angular velocity

// Fixed per-lane sphere table layout (grx_capi.cpp build_side_tables): slots 0..7 = this lane's share of the
// base-lump shapes, then the chain bodies' shapes.  Unused slots carry r = -1e30 (never within reach).
__device__ constexpr int kSphCnt[LEG] = {0, 0, 2, 2, 4};
__device__ constexpr int kSphOff[LEG] = {8, 8, 8, 10, 12};

// Centre of a sphere relative to O and the terrain height under it.  The heightfield gathers are issued for every
// lane, unconditionally (indices are clamped), so that a group's lookups are all in flight together instead of one
// exposed memory latency per sphere inside divergent branches.
struct TerrainAt { float h, gx, gy; };   // height and gradient of the terrain under a sphere centre
template <int HF>
GRX_DEV void sphere_probe(KP P, const SphC& S, const R3& R, V3 rho, V3 O, V3& xr, TerrainAt& th) {
    xr = rho + rot(R, v3(S.x, S.y, S.z));
    th.h = terrain_height<HF>(P, O.x + xr.x, O.y + xr.y, th.gx, th.gy);
}

// One sphere against the terrain.  R/rho/w/v: rotation, origin (relative to the base origin O), angular
// velocity and O-referenced linear velocity of the carrying body.  SLOT: friction-anchor slot of a foot
// sphere (compile time), -1 for the other shapes.  xr = sphere centre relative to O, th = terrain height under it
// (sphere_probe).  Returns the world-frame force.
template <int HF, int SLOT>
GRX_DEV V3 sphere_contact(KP P, const SphC& S, V3 w, V3 v, V3 O, float mu, float hmax, LaneState& st, V3 xr, const TerrainAt& th, float om_e = 1.0f) {
    V3 F = v3(0.f, 0.f, 0.f);
    const float wz = O.z + xr.z;
    // cull: hmax bounds the terrain height anywhere the robot can reach during this policy step
    // (plane: 0; heightfield: dilated coarse max map): above it the sphere cannot touch (exactly d <= 0)
    bool touching = false;
    if (wz - S.r <= hmax) {
        const float wx = O.x + xr.x, wy = O.y + xr.y;
        const float dv = th.h + S.r - wz;   // vertical overlap
        if (dv > 0.0f) {
            touching = true;
            // surface normal from the gradient of the patch; overlap along it (locally planar terrain).  Plane: n = (0, 0, 1)
            const float nn = HF ? grx_rsq(1.0f + th.gx * th.gx + th.gy * th.gy) : 1.0f;
            const V3 n = HF ? v3(-th.gx * nn, -th.gy * nn, nn) : v3(0.f, 0.f, 1.f);
            const float d = dv * nn;
            V3 u = v + cross(w, xr);
            const float un = HF ? dot(u, n) : u.z;   // > 0: separating
            float cd = fminf(P.kn * d * P.dn, S.dmax);  // Hunt-Crossley damping, mass-aware cap (oracle contact_forces())
            if (SLOT >= 0) {   // restitution: a contact that began faster than the bounce threshold keeps (1 - e) of its damping while separating
                constexpr int sl = SLOT < 0 ? 0 : SLOT;
                if (!(st.anchor_on & (1u << sl))) st.vimp[sl] = fmaxf(-un, 0.0f);
                if (un > 0.0f && st.vimp[sl] > P.bounce_threshold) cd *= om_e;
            }
            float fn = fmaxf(P.kn * d - cd * un, 0.0f);
            F = n * fn;
            float fmax = mu * fn;
            // friction: in the horizontal plane (anchored stick/slip on the foot spheres, viscous-capped elsewhere)
            if (SLOT >= 0) {
                float axx = st.ax[SLOT < 0 ? 0 : SLOT], ayy = st.ay[SLOT < 0 ? 0 : SLOT];
                if (!(st.anchor_on & (1u << (SLOT < 0 ? 0 : SLOT)))) { axx = wx; ayy = wy; }
                float ftx = -P.kt * (wx - axx) - P.ct * u.x;
                float fty = -P.kt * (wy - ayy) - P.ct * u.y;
                float ft = grx_sqrt(ftx * ftx + fty * fty);
                if (ft > fmax) {  // slip: clamp to the cone, drag the anchor along
                    float sc = fmax * grx_rcp(ft);
                    ftx *= sc; fty *= sc;
                    axx = wx + ftx * P.inv_kt;
                    ayy = wy + fty * P.inv_kt;
                }
                st.ax[SLOT < 0 ? 0 : SLOT] = axx; st.ay[SLOT < 0 ? 0 : SLOT] = ayy;
                F.x += ftx; F.y += fty;
            } else {
                float sp = grx_sqrt(u.x * u.x + u.y * u.y);
                float ft = fminf(P.cv * sp, fmax);
                if (sp > 1e-9f) { float k = -ft * grx_rcp(sp); F.x += k * u.x; F.y += k * u.y; }
            }
        }
    }
    if (SLOT >= 0) {
        const uint32_t bit = 1u << (SLOT < 0 ? 0 : SLOT);
        st.anchor_on = touching ? (st.anchor_on | bit) : (st.anchor_on & ~bit);
    }
    return F;
}

// Wave-uniform pre-check of a group of CNT spheres on one body: can ANY lane's sphere be within reach of the
// terrain?  All LDS reads of the group are issued together (one exposed latency instead of one per sphere);
// when no lane of the wave qualifies the whole contact block is skipped (upright robot: every group but the feet).
template <int CNT>
GRX_DEV bool group_within_reach(const SphC* S, const R3& R, V3 rho, V3 O, float hmax) {
    bool cand = false;
    const float base_z = O.z + rho.z;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        float z = base_z + fmaf(R.cx.z, S[i].x, fmaf(R.cy.z, S[i].y, R.cz.z * S[i].z));
        cand = cand || (z - S[i].r <= hmax);
    }
    return __any(cand);
}

// parent rotation from the child rotation: inverse of joint_rot_k (R_parent = R_child * Rot_axis(-q))
GRX_DEV R3 joint_unrot_k(const R3& R, float c, float s, int ax) {
    R3 P;
    if (ax == 0) { P.cx = R.cx; P.cy = fma3(R.cy, c, R.cz * (-s)); P.cz = fma3(R.cz, c, R.cy * s); }
    else if (ax == 1) { P.cy = R.cy; P.cx = fma3(R.cx, c, R.cz * s); P.cz = fma3(R.cz, c, R.cx * (-s)); }
    else { P.cz = R.cz; P.cx = fma3(R.cx, c, R.cy * (-s)); P.cy = fma3(R.cy, c, R.cx * s); }
    return P;
}

// GRX_T_CONTACT_FORCES: net contact force of one URDF link.  The tensor shows the LAST sub-step (like the reference's after
// its last gym.simulate): `last` is wave-uniform, so the nine other sub-steps pay one scalar branch per call site.
// cf = the env's column of the tensor, nullptr on an inactive lane.
struct LinkForceOut { bool last; float* cf; size_t N; };
GRX_DEV void put_link_force(const LinkForceOut& o_, const SphC& S, V3 F) {   // S: any shape of the link (table read only when last)
    if (o_.last) {
        const int link = sph_link(S);
        if (o_.cf && link >= 0) { float* o = o_.cf + (size_t)(link * 3) * o_.N; o[0] = F.x; o[o_.N] = F.y; o[2 * o_.N] = F.z; }
    }
}

// running frame of a chain body during an outward walk: rotation, origin relative to the base origin O,
// angular velocity, O-referenced linear velocity (world axes)
struct ChainKin { R3 R; V3 rho, w, v; };

GRX_DEV void chain_step(const SideConst& C, int k, float q, float qd, ChainKin& K) {
    K.rho = K.rho + rot(K.R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
    float sn, cs;
    grx_sincos(q, sn, cs);
    K.R = joint_rot_k(K.R, cs, sn, kAxis[k]);
    const V3 a = axis_k(K.R, kAxis[k]);
    const V3 s = cross(K.rho, a);
    K.w = fma3(a, qd, K.w); K.v = fma3(s, qd, K.v);
}

// the four anchored spheres of this lane's foot (chain body LEG-1): wrench about O + anchor update
// in two halves, so that a caller with other work at hand (wave 2 of the four-wave layout: the bias forces) can put it
// between the heightfield gathers and their first use
struct FootProbe { bool reach; V3 xr[4]; TerrainRaw raw[4]; uint4 ww[4]; };   // (ww: mesh_type 'trimesh' only -- the face records of the cells)
template <int HF>
GRX_DEV void foot_probe(KP P, const SideConst& C, const ChainKin& K, V3 O, float hmax, FootProbe& fp) {
    constexpr int o = kSphOff[LEG - 1];
    fp.reach = group_within_reach<4>(&C.sph[o], K.R, K.rho, O, hmax);
    if (fp.reach) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fp.xr[i] = K.rho + rot(K.R, v3(C.sph[o + i].x, C.sph[o + i].y, C.sph[o + i].z));
            terrain_gather<HF>(P, O.x + fp.xr[i].x, O.y + fp.xr[i].y, fp.raw[i]);
            if (HF == GRX_HF_TRIMESH) { float tx_, ty_; fp.ww[i] = wall_gather(P, O.x + fp.xr[i].x, O.y + fp.xr[i].y, tx_, ty_); }
        }
    }
}
template <int HF>
GRX_DEV void foot_contacts(KP P, const SideConst& C, const ChainKin& K, V3 O, float mu, float hmax, LaneState& st,
                           V3& fa, V3& fl, float om_e, const FootProbe& fp) {
    fa = v3(0.f, 0.f, 0.f); fl = v3(0.f, 0.f, 0.f);
    constexpr int o = kSphOff[LEG - 1];
    if (fp.reach) {
        V3 F;
        const V3* xr = fp.xr;
        TerrainAt th[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) th[i].h = terrain_eval<HF>(P, fp.raw[i], th[i].gx, th[i].gy);
        F = sphere_contact<HF, 0>(P, C.sph[o + 0], K.w, K.v, O, mu, hmax, st, xr[0], th[0], om_e); fa = fa + cross(xr[0], F); fl = fl + F;
        F = sphere_contact<HF, 1>(P, C.sph[o + 1], K.w, K.v, O, mu, hmax, st, xr[1], th[1], om_e); fa = fa + cross(xr[1], F); fl = fl + F;
        F = sphere_contact<HF, 2>(P, C.sph[o + 2], K.w, K.v, O, mu, hmax, st, xr[2], th[2], om_e); fa = fa + cross(xr[2], F); fl = fl + F;
        F = sphere_contact<HF, 3>(P, C.sph[o + 3], K.w, K.v, O, mu, hmax, st, xr[3], th[3], om_e); fa = fa + cross(xr[3], F); fl = fl + F;
        if (HF == GRX_HF_TRIMESH) wall_pass<4>(P, &C.sph[o], K.w, K.v, O, mu, hmax, xr, fp.ww, fp.raw, fa, fl);
    } else st.anchor_on = 0;   // nobody in the wave can touch: all four anchors released
}
template <int HF>
GRX_DEV void foot_contacts(KP P, const SideConst& C, const ChainKin& K, V3 O, float mu, float hmax, LaneState& st,
                           V3& fa, V3& fl, float om_e) {
    FootProbe fp;
    foot_probe<HF>(P, C, K, O, hmax, fp);
    foot_contacts<HF>(P, C, K, O, mu, hmax, st, fa, fl, om_e, fp);
}

// LPE == 4: the two lanes of a leg take two foot spheres each -- lane `half` the spheres 2 half, 2 half + 1 of the foot's table,
// read from the LDS copy of the table (lane-dependent index); their friction anchors live in the lane's slots 0, 1
struct FootProbeQ { bool reach; V3 xr[2]; TerrainRaw raw[2]; };
template <int HF>
GRX_DEV void foot_probe_q(KP P, const SideConst& C, const SideConst& Clds, int half, const ChainKin& K, V3 O, float hmax, FootProbeQ& fp) {
    constexpr int o = kSphOff[LEG - 1];
    fp.reach = group_within_reach<4>(&C.sph[o], K.R, K.rho, O, hmax);
    if (fp.reach) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const SphC& S = Clds.sph[o + 2 * half + j];
            fp.xr[j] = K.rho + rot(K.R, v3(S.x, S.y, S.z));
            terrain_gather<HF>(P, O.x + fp.xr[j].x, O.y + fp.xr[j].y, fp.raw[j]);
        }
    }
}
template <int HF>
GRX_DEV void foot_contacts_q(KP P, const SideConst& Clds, int half, const ChainKin& K, V3 O, float mu, float hmax, LaneState& st,
                             V3& fa, V3& fl, float om_e, const FootProbeQ& fp) {
    fa = v3(0.f, 0.f, 0.f); fl = v3(0.f, 0.f, 0.f);
    constexpr int o = kSphOff[LEG - 1];
    if (fp.reach) {
        TerrainAt th[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) th[j].h = terrain_eval<HF>(P, fp.raw[j], th[j].gx, th[j].gy);
        V3 F;
        F = sphere_contact<HF, 0>(P, Clds.sph[o + 2 * half + 0], K.w, K.v, O, mu, hmax, st, fp.xr[0], th[0], om_e); fa = fa + cross(fp.xr[0], F); fl = fl + F;
        F = sphere_contact<HF, 1>(P, Clds.sph[o + 2 * half + 1], K.w, K.v, O, mu, hmax, st, fp.xr[1], th[1], om_e); fa = fa + cross(fp.xr[1], F); fl = fl + F;
        if (HF == GRX_HF_TRIMESH) {   // (the face records gathered HERE: with the ground records in the probe the lane-quad kernel is 0.4 % slower, the lane-pair one 0.7 % faster)
            uint4 ww[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { float tx_, ty_; ww[j] = wall_gather(P, O.x + fp.xr[j].x, O.y + fp.xr[j].y, tx_, ty_); }
            wall_pass<2>(P, &Clds.sph[o + 2 * half], K.w, K.v, O, mu, hmax, fp.xr, ww, fp.raw, fa, fl);
        }
        fa = half_sum(fa); fl = half_sum(fl);   // the foot's wrench: both halves
    } else st.anchor_on = 0;
}

#include "grx_rare.h"
#include "grx_self.h"

// One physics sub-step (gym.simulate(dt), legged_robot_fftai.py:68) for this lane's half of the env.
// tau: motor torques of this lane's 5 joints.  fk_only: just the kinematics pass (foot frames).
// W = waves per block: 1 = everything inline; 2 = the base-lump contacts come from the helper wave through `wr`
// (W == 4 uses the producer/consumer pipeline of grx_wavepipe.h instead of this function).
template <int HF, int W>
GRX_DEV void substep(KP P, const KTables& T, const SideConst& C, const LaneConst& LC, LaneState& st, const float tau_m[LEG],
                     SubstepOut& out, FootKin& fk_before, const float* wr, long long* tacc, const LinkForceOut& lfo,
                     const RareBuf& RB, int lane, int el, int side, SelfNear& sn, bool first) {
    static_assert(SELF_BYTES <= RC_RES_BYTES, "the self-collision staging reuses the rare contacts' result table");
#ifndef GRX_W1_FOLDC
#define GRX_W1_FOLDC 1   // one- and two-wave layouts: the velocity-product accelerations folded into the bias forces (+1 % at >= 32768 envs, measured)
#endif
    constexpr bool kFoldC = GRX_W1_FOLDC != 0;
    const SelfBuf SB = self_carve(reinterpret_cast<char*>(RB.res));   // the rare contacts' result table is free again by then
    const float dt = P.sim_dt;
    R3 R0 = quat_to_R(st.qx, st.qy, st.qz, st.qw);
    V3 O = st.pos;
#ifdef GRX_PROFILE_SECTIONS
    long long tprev = clock64();
#endif
    // ---- pass 1: kinematics, rigid inertias, bias forces, contacts (root -> leaf)
    V3 Sa[LEG], Ss[LEG];       // joint motion subspace S = (a; rho x a)
    S3 IAk[LEG]; V3 Ih[LEG];   // rigid inertia about O: A and h = m*kappa
    V3 pA[LEG], pL[LEG];       // bias force
    R3 Rp = R0;
    V3 rho_p = v3(0.f, 0.f, 0.f);
    V3 w = st.ang, v = st.vel;
    V3 za = v3(0.f, 0.f, 0.f), zl = v3(0.f, 0.f, 0.f);
    out.foot_force = v3(0.f, 0.f, 0.f);
    out.term = false;
    out.pen_count = 0.f;
    ChainKin K2, K3, K4;   // thigh / shank (/ foot) frames: rare terrain contacts (grx_rare.h) and self-collision (grx_self.h) after the walk
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        V3 rho = rho_p + rot(Rp, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
        float sn, cs;
        grx_sincos(st.q[k], sn, cs);
        R3 R = joint_rot_k(Rp, cs, sn, kAxis[k]);
        V3 a = axis_k(R, kAxis[k]);
        V3 s = cross(rho, a);
        V3 wk = fma3(a, st.qd[k], w), vk = fma3(s, st.qd[k], v);
        float m = C.body[k].mass;
        V3 kap = rho + rot(R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
        S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
        S3 A = rot_sym(R, Ic);
        float kk = dot(kap, kap);
        A.xx += m * (kk - kap.x * kap.x); A.xy -= m * kap.x * kap.y; A.xz -= m * kap.x * kap.z;
        A.yy += m * (kk - kap.y * kap.y); A.yz -= m * kap.y * kap.z; A.zz += m * (kk - kap.z * kap.z);
        V3 h = kap * m;
        V3 hl = fma3(vk, m, cross(wk, h));
        V3 ha = mul(A, wk) + cross(h, vk);
        V3 pa = cross(wk, ha) + cross(vk, hl);
        V3 pl = cross(wk, hl);
        if (kFoldC) {   // velocity-product accelerations folded into the bias forces (grx_wavepipe.h, "eight waves"): with zeta_k = sum_{j <= k} c_j
            // and a_k = a^_k + zeta_k the recursion in a^ has no c terms; body k's bias force gains I_k zeta_k (rigid inertia about O)
            za = za + cross(w, a) * st.qd[k];
            zl = zl + (cross(v, a) + cross(w, s)) * st.qd[k];
            pa = pa + mul(A, za) + cross(h, zl);
            pl = pl + zl * m - cross(h, za);
        }
        // contacts of the shapes carried by chain body k (thigh_pitch: 2, shank: 2, foot: 4 anchored spheres)
        if (kSphCnt[k] > 0) {
            ChainKin K = {R, rho, wk, vk};
            if (kSphCnt[k] == 2) { if (k == 2) K2 = K; else K3 = K; }
            else {
                V3 fa, fl;
                foot_contacts<HF>(P, C, K, O, LC.mu, LC.hmax, st, fa, fl, LC.om_e); out.foot_force = fl;
                pa = pa - fa; pl = pl - fl;
                K4 = K;
            }
        }
        if (k == LEG - 1) {  // foot link frame BEFORE this sub-step's integration
            V3 fr = rho + rot(R, v3(C.foot_pos[0], C.foot_pos[1], C.foot_pos[2]));
            fk_before.pos = O + fr;
            fk_before.vel = vk + cross(wk, fr);
            fk_before.ang = wk;
        }
        Sa[k] = a; Ss[k] = s; IAk[k] = A; Ih[k] = h; pA[k] = pa; pL[k] = pl;
        Rp = R; rho_p = rho; w = wk; v = vk;
    }
    GRX_TICK2(16);
    // thigh / shank shapes (W == 1: and the base-lump shapes), compacted over the wave
    RareOut ro;
#ifdef GRX_PROFILE_SECTIONS
    long long* const racc_ = tacc + 6; long long* const sacc_ = tacc + 14;   // (one-wave layout: the phases of the two compacted evaluations)
#else
    long long* const racc_ = nullptr; long long* const sacc_ = nullptr;
#endif
    rare_contacts<HF, (W == 1 ? 0 : 8), RC_NS>(P, T, C, RB, lane, el, side, R0, O, st.ang, st.vel, K2, K3, LC.mu, LC.hmax, ro, racc_, RareNoWait(), lfo.last);
    SelfOut sc;
    {   // self-collision: leg against leg, thigh against base-lump shapes
        const ChainKin KS[3] = {K2, K3, K4};
        if (first) sn = self_broad_phase(P, C, side, R0, KS);   // wave-uniform
        self_collision(P, T, C, SB, lane, side, R0, st.ang, st.vel, KS, 2.0f * LC.mu - P.terrain_friction, sn, sc, sacc_);
    }
    pA[2] = pA[2] - ro.fa2 - sc.fa[0]; pL[2] = pL[2] - ro.fl2 - sc.fl[0];
    pA[3] = pA[3] - ro.fa3 - sc.fa[1]; pL[3] = pL[3] - ro.fl3 - sc.fl[1];
    pA[4] = pA[4] - sc.fa[2]; pL[4] = pL[4] - sc.fl[2];
    const V3 foot_terrain = out.foot_force;
    out.foot_force = out.foot_force + sc.fl[2];   // net contact force on the foot link: terrain + self-collision
    GRX_TICK2(17);
    // ---- pass 2: articulated inertias (leaf -> root).  w, v currently = velocity of body LEG-1.
    S3 A = IAk[LEG - 1];
    V3 h4 = Ih[LEG - 1];
    float m4 = C.body[LEG - 1].mass;
    M3 B = {0.f, -h4.z, h4.y, h4.z, 0.f, -h4.x, -h4.y, h4.x, 0.f};
    S3 D = {m4, 0.f, 0.f, m4, 0.f, m4};
    V3 pa = pA[LEG - 1], pl = pL[LEG - 1];
    V3 Ua[LEG], Ul[LEG], ca[LEG], cl[LEG];
    float dinv[LEG], uu[LEG];
#ifndef GRX_PK_INERTIA
#define GRX_PK_INERTIA 0   // 1: hand-packed fp32 (v_pk_fma_f32) in the articulated-inertia recursion.  MEASURED SLOWER (round 5): 179.3 against 164.1 us per step at
                           // 32768 envs (512 registers + 17-22 spilled; see grx_wavepipe.h substep_p) -- kept as the record of the experiment, off
#endif
    constexpr bool kPk = GRX_PK_INERTIA && kFoldC;
    typedef float f2_t __attribute__((ext_vector_type(2)));
    struct C6 { f2_t r01, r23, r45; };
    C6 c6[6];   // the 6 x 6 [A B; B^T D] as six columns of three float2 (rows 01 | 23 | 45)
    auto pk_add_rigid = [&](const S3& K, V3 h, float m) {
        c6[0].r01 += f2_t{K.xx, K.xy}; c6[0].r23.x += K.xz;            c6[0].r45 += f2_t{-h.z, h.y};
        c6[1].r01 += f2_t{K.xy, K.yy}; c6[1].r23 += f2_t{K.yz, h.z};   c6[1].r45.y += -h.x;
        c6[2].r01 += f2_t{K.xz, K.yz}; c6[2].r23 += f2_t{K.zz, -h.y};  c6[2].r45.x += h.x;
        c6[3].r01.y += h.z;            c6[3].r23 += f2_t{-h.y, m};
        c6[4].r01.x += -h.z;           c6[4].r23.x += h.x;             c6[4].r45.x += m;
        c6[5].r01 += f2_t{h.y, -h.x};                                  c6[5].r45.y += m;
    };
    if (kPk) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { c6[j].r01 = f2_t{0.f, 0.f}; c6[j].r23 = f2_t{0.f, 0.f}; c6[j].r45 = f2_t{0.f, 0.f}; }
        pk_add_rigid(IAk[LEG - 1], h4, m4);
    }
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        V3 a = Sa[k], s = Ss[k];
        float qdk = st.qd[k];
        if (kPk) {
            const float sj[6] = {a.x, a.y, a.z, s.x, s.y, s.z};
            f2_t u01 = c6[0].r01 * f2_t{sj[0], sj[0]}, u23 = c6[0].r23 * f2_t{sj[0], sj[0]}, u45 = c6[0].r45 * f2_t{sj[0], sj[0]};
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                u01 = __builtin_elementwise_fma(c6[j].r01, f2_t{sj[j], sj[j]}, u01);
                u23 = __builtin_elementwise_fma(c6[j].r23, f2_t{sj[j], sj[j]}, u23);
                u45 = __builtin_elementwise_fma(c6[j].r45, f2_t{sj[j], sj[j]}, u45);
            }
            f2_t t_ = f2_t{a.x, a.y} * u01;
            t_ = __builtin_elementwise_fma(f2_t{a.z, s.x}, u23, t_);
            t_ = __builtin_elementwise_fma(f2_t{s.y, s.z}, u45, t_);
            const float di = grx_rcp(t_.x + t_.y);
            const V3 ua = v3(u01.x, u01.y, u23.x), ul = v3(u23.y, u45.x, u45.y);
            const float qk = st.q[k], qlo = C.body[k].qlo, qhi = C.body[k].qhi;
            const float viol = qk < qlo ? qlo - qk : (qk > qhi ? qhi - qk : 0.f);
            const float t = tau_m[k] + (C.body[k].Klim * viol - (viol != 0.f ? C.body[k].Clim * qdk : 0.f));
            const float u = t - (dot(a, pa) + dot(s, pl));
            const f2_t w01 = u01 * f2_t{di, di}, w23 = u23 * f2_t{di, di}, w45 = u45 * f2_t{di, di};
            const float wj[6] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                c6[j].r01 = __builtin_elementwise_fma(-u01, f2_t{wj[j], wj[j]}, c6[j].r01);
                c6[j].r23 = __builtin_elementwise_fma(-u23, f2_t{wj[j], wj[j]}, c6[j].r23);
                c6[j].r45 = __builtin_elementwise_fma(-u45, f2_t{wj[j], wj[j]}, c6[j].r45);
            }
            const float ud = u * di;
            Ua[k] = ua; Ul[k] = ul; dinv[k] = di; uu[k] = u; ca[k] = v3(0.f, 0.f, 0.f); cl[k] = ca[k];
            pa = fma3(ua, ud, pa); pl = fma3(ul, ud, pl);
            if (k > 0) {
                pk_add_rigid(IAk[k - 1], Ih[k - 1], C.body[k - 1].mass);
                pa = pa + pA[k - 1]; pl = pl + pL[k - 1];
            }
            continue;
        }
        V3 cak = v3(0.f, 0.f, 0.f), clk = cak;
        if (!kFoldC) {
            w = fma3(a, -qdk, w); v = fma3(s, -qdk, v);  // parent velocity
            cak = cross(w, a) * qdk;
            clk = (cross(v, a) + cross(w, s)) * qdk;
        }
        V3 ua = mul(A, a) + mul(B, s);
        V3 ul = mulT(B, a) + mul(D, s);
        float d = dot(a, ua) + dot(s, ul);
        float di = grx_rcp(d);
        // joint-limit spring/damper (oracle substep()): added to the motor torque
        const float qk = st.q[k], qlo = C.body[k].qlo, qhi = C.body[k].qhi;   // (branch-free: see grx_wavepipe.h)
        const float viol = qk < qlo ? qlo - qk : (qk > qhi ? qhi - qk : 0.f);
        const float t = tau_m[k] + (C.body[k].Klim * viol - (viol != 0.f ? C.body[k].Clim * qdk : 0.f));
        float u = t - (dot(a, pa) + dot(s, pl));
        syr(A, ua, di); ger(B, ua, ul, di); syr(D, ul, di);
        float ud = u * di;
        V3 npa = kFoldC ? fma3(ua, ud, pa) : pa + mul(A, cak) + mul(B, clk) + ua * ud;
        V3 npl = kFoldC ? fma3(ul, ud, pl) : pl + mulT(B, cak) + mul(D, clk) + ul * ud;
        Ua[k] = ua; Ul[k] = ul; dinv[k] = di; uu[k] = u; ca[k] = cak; cl[k] = clk;
        pa = npa; pl = npl;
        if (k > 0) {  // add the parent's rigid inertia: [A B; B^T D] += rigid(k-1)
            V3 hp = Ih[k - 1];
            float mp = C.body[k - 1].mass;
            A = A + IAk[k - 1];
            B.a01 -= hp.z; B.a02 += hp.y; B.a10 += hp.z; B.a12 -= hp.x; B.a20 -= hp.y; B.a21 += hp.x;
            D.xx += mp; D.yy += mp; D.zz += mp;
            pa = pa + pA[k - 1]; pl = pl + pL[k - 1];
        }
    }
    GRX_TICK2(18);
    // ---- base: combine both chains (DPP pair exchange), add the base lump, solve the 6x6
    if (W >= 2) {
        __syncthreads();   // #3: the helper wave's base-lump contact wrench of this sub-step is in LDS (wr: this lane's column)
        const V3 f0a = v3(wr[0 * 64], wr[1 * 64], wr[2 * 64]), f0l = v3(wr[3 * 64], wr[4 * 64], wr[5 * 64]);
        out.term = wr[6 * 64] != 0.f;
        out.pen_count = wr[7 * 64];
        pa = pa - f0a - sc.f0a; pl = pl - f0l - sc.f0l;
        if (lfo.last) {   // the helper wave parked the per-link forces of the base-lump shapes behind its wrench
#pragma unroll
            for (int i = 0; i < 8; ++i) ro.lf[i] = v3(wr[(8 + 3 * i) * 64], wr[(9 + 3 * i) * 64], wr[(10 + 3 * i) * 64]);
        }
    } else {
        out.term = ro.term; out.pen_count = ro.pen_count;
        pa = pa - ro.f0a - sc.f0a; pl = pl - ro.f0l - sc.f0l;
    }
    write_link_rows(P, lfo, C, ro.lf, ro.fl2, ro.fl3, foot_terrain, sc, out.term, out.pen_count);   // (last sub-step: the flags from the NET link forces)
    if (kPk) {
        A.xx = c6[0].r01.x; A.xy = c6[0].r01.y; A.xz = c6[0].r23.x; A.yy = c6[1].r01.y; A.yz = c6[1].r23.x; A.zz = c6[2].r23.x;
        B.a00 = c6[3].r01.x; B.a10 = c6[3].r01.y; B.a20 = c6[3].r23.x; B.a01 = c6[4].r01.x; B.a11 = c6[4].r01.y; B.a21 = c6[4].r23.x;
        B.a02 = c6[5].r01.x; B.a12 = c6[5].r01.y; B.a22 = c6[5].r23.x;
        D.xx = c6[3].r23.y; D.xy = c6[4].r23.y; D.xz = c6[5].r23.y; D.yy = c6[4].r45.x; D.yz = c6[5].r45.x; D.zz = c6[5].r45.y;
    }
    A = pair_sum(A); B = pair_sum(B); D = pair_sum(D);
    pa = pair_sum(pa); pl = pair_sum(pl);
    {
        V3 kap = rot(R0, LC.base_c);
        float m = LC.base_m;
        S3 A0 = rot_sym(R0, LC.base_I);
        float kk = dot(kap, kap);
        A0.xx += m * (kk - kap.x * kap.x); A0.xy -= m * kap.x * kap.y; A0.xz -= m * kap.x * kap.z;
        A0.yy += m * (kk - kap.y * kap.y); A0.yz -= m * kap.y * kap.z; A0.zz += m * (kk - kap.z * kap.z);
        V3 h = kap * m;
        V3 w0 = st.ang, v0 = st.vel;
        V3 hl = fma3(v0, m, cross(w0, h));
        V3 ha = mul(A0, w0) + cross(h, v0);
        pa = pa + cross(w0, ha) + cross(v0, hl);
        pl = pl + cross(w0, hl);
        A = A + A0;
        B.a01 -= h.z; B.a02 += h.y; B.a10 += h.z; B.a12 -= h.x; B.a20 -= h.y; B.a21 += h.x;
        D.xx += m; D.yy += m; D.zz += m;
    }
    // [A B; B^T D][alpha; acc] = -[pa; pl]:  acc = -Dinv (pl + B^T alpha);  (A - B Dinv B^T) alpha = -pa + B Dinv pl
    S3 Di = inv(D);
    V3 Dipl = mul(Di, pl);
    V3 rhs = mul(B, Dipl) - pa;
    // Schur complement S = A - B Dinv B^T (symmetric)
    V3 b0 = v3(B.a00, B.a01, B.a02), b1 = v3(B.a10, B.a11, B.a12), b2 = v3(B.a20, B.a21, B.a22);
    V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
    S3 Sc = {A.xx - dot(b0, d0), A.xy - dot(b0, d1), A.xz - dot(b0, d2), A.yy - dot(b1, d1), A.yz - dot(b1, d2), A.zz - dot(b2, d2)};
    V3 alpha = mul(inv(Sc), rhs);
    V3 acc = neg(mul(Di, pl + mulT(B, alpha)));
    GRX_TICK2(19);
    // ---- pass 3: accelerations (root -> leaf)
    float qdd[LEG];
    V3 aa = alpha, al = acc;
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        V3 pa_ = kFoldC ? aa : aa + ca[k], pl_ = kFoldC ? al : al + cl[k];
        float qd2 = (uu[k] - (dot(Ua[k], pa_) + dot(Ul[k], pl_))) * dinv[k];
        qdd[k] = qd2;
        aa = fma3(Sa[k], qd2, pa_);
        al = fma3(Ss[k], qd2, pl_);
    }
    GRX_TICK2(20);
    // ---- integrate (semi-implicit Euler)
    V3 lin = acc + cross(st.ang, st.vel);  // classical acceleration of the base origin
    st.vel = v3(st.vel.x + (lin.x + P.gravity[0]) * dt, st.vel.y + (lin.y + P.gravity[1]) * dt, st.vel.z + (lin.z + P.gravity[2]) * dt);
    st.ang = fma3(alpha, dt, st.ang);
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        float vq = fmaf(qdd[k], dt, st.qd[k]);
        vq = fminf(fmaxf(vq, -C.body[k].vlim), C.body[k].vlim);
        st.qd[k] = vq;
        st.q[k] = fmaf(vq, dt, st.q[k]);
    }
    st.pos = fma3(st.vel, dt, st.pos);
    float hx = 0.5f * dt * st.ang.x, hy = 0.5f * dt * st.ang.y, hz = 0.5f * dt * st.ang.z;
    float x = st.qx, y = st.qy, z = st.qz, ww = st.qw;
    float nx = x + hx * ww + hy * z - hz * y;
    float ny = y - hx * z + hy * ww + hz * x;
    float nz = z + hx * y - hy * x + hz * ww;
    float nw = ww - hx * x - hy * y - hz * z;
    float n = grx_rsq(nx * nx + ny * ny + nz * nz + nw * nw);
    st.qx = nx * n; st.qy = ny * n; st.qz = nz * n; st.qw = nw * n;
    GRX_TICK2(21);
}

// rigid inertia of a body about O in world axes: A (rotational 3x3) and h = m kap  (= grx_wavepipe.h rigid_inertia, which is declared further down)
GRX_DEV void rigid_inertia_lean(const R3& R, V3 kap, float m, const S3& Ic, S3& A, V3& h) {
    A = rot_sym(R, Ic);
    const float kk = dot(kap, kap);
    A.xx += m * (kk - kap.x * kap.x); A.xy -= m * kap.x * kap.y; A.xz -= m * kap.x * kap.z;
    A.yy += m * (kk - kap.y * kap.y); A.yz -= m * kap.y * kap.z; A.zz += m * (kk - kap.z * kap.z);
    h = kap * m;
}

// ---------------------------------------------------------------------------------------------------------------
// The same sub-step, REGISTER-LEAN (round 4; the one-wave layout at two waves per SIMD, GRX_W1_DUO): substep() keeps ~105 values of
// its outward pass (motion subspaces, rigid inertias, bias forces of five bodies) and the three contact frames live across the two
// wave-cooperative contact evaluations -- 360 spilled dwords when the kernel is held to 256 registers.  Here the contacts come
// FIRST, on a walk that keeps nothing but the three frames; the inward pass then starts from the LEAF frame and recomputes every
// body's frame, velocity and accumulated velocity-product acceleration going DOWN the chain (R_{k-1} = R_k Rot(-q_k), rho_{k-1} =
// rho_k - R_{k-1} r_k, w_{k-1} = w_k - a_k qd_k, zeta_{k-1} = zeta_k - c_k), forming the rigid inertia and the bias force of a body
// where they are consumed: per joint only S, U = I^A S, 1/d and u survive for the acceleration pass.  ~45 instructions more per
// joint, a third of the live values.  Same physics; results agree with substep() to rounding (the recomputed frames differ in the
// last bit).
// EXPERIMENT, compiled only with -DGRX_W1_LEAN (profiles/r04_experiments_session3.md): at the one-wave kernel's 512-register budget it
// needs 448 registers and no scratch (substep(): 512 + 8-15 spilled dwords) and runs 1.7 % SLOWER (190.3 against 193.5 M env-steps/s
// at 32768 envs: the extra instructions); held to 256 registers (-DGRX_WPE=2 with a small arena, compile only) the kernel around it
// still spills 224 dwords (substep(): 360) -- the post-physics state carried across the sub-step loop and the contact evaluations'
// own temporaries are the next ~150.  Two waves per SIMD at > 16384 envs per GPU need that AND 20 KB of LDS per wave (37 KB now):
// not reached this round.
template <int HF>
GRX_DEV void substep_lean(KP P, const KTables& T, const SideConst& C, const LaneConst& LC, LaneState& st, const float tau_m[LEG],
                          SubstepOut& out, FootKin& fk_before, const LinkForceOut& lfo,
                          const RareBuf& RB, int lane, int el, int side, SelfNear& sn, bool first) {
    const SelfBuf SB = self_carve(reinterpret_cast<char*>(RB.res));   // the rare contacts' result table is free again by then
    const float dt = P.sim_dt;
    const R3 R0 = quat_to_R(st.qx, st.qy, st.qz, st.qw);
    const V3 O = st.pos;
    const V3 zero = v3(0.f, 0.f, 0.f);
    // ---- walk 1 (root -> leaf): frames with velocities of the three shape-carrying bodies; zeta of the leaf
    float cs[LEG], sn_[LEG];
    V3 za = zero, zl = zero;
    R3 R = R0;
    V3 rho = zero, w = st.ang, v = st.vel;
    V3 ca2, cl2, ca3, cl3, ca4, cl4, c0a, c0l;   // contact wrenches about O on thigh, shank, foot and on the base lump
    {
        ChainKin K2, K3;
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            rho = rho + rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
            grx_sincos(st.q[k], sn_[k], cs[k]);
            R = joint_rot_k(R, cs[k], sn_[k], kAxis[k]);
            const V3 a = axis_k(R, kAxis[k]);
            const V3 s = cross(rho, a);
            const float qdk = st.qd[k];
            za = za + cross(w, a) * qdk;
            zl = zl + (cross(v, a) + cross(w, s)) * qdk;
            w = fma3(a, qdk, w); v = fma3(s, qdk, v);
            if (k == 2) K2 = ChainKin{R, rho, w, v};
            if (k == 3) K3 = ChainKin{R, rho, w, v};
        }
        const ChainKin K4 = {R, rho, w, v};
        {   // foot link frame BEFORE this sub-step's integration
            const V3 fr = rho + rot(R, v3(C.foot_pos[0], C.foot_pos[1], C.foot_pos[2]));
            fk_before.pos = O + fr;
            fk_before.vel = v + cross(w, fr);
            fk_before.ang = w;
        }
        foot_contacts<HF>(P, C, K4, O, LC.mu, LC.hmax, st, ca4, cl4, LC.om_e);
        const V3 foot_terrain = cl4;
        RareOut ro;
        rare_contacts<HF, 0, RC_NS>(P, T, C, RB, lane, el, side, R0, O, st.ang, st.vel, K2, K3, LC.mu, LC.hmax, ro, nullptr, RareNoWait(), lfo.last);
        SelfOut sc;
        {
            const ChainKin KS[3] = {K2, K3, K4};
            if (first) sn = self_broad_phase(P, C, side, R0, KS);   // wave-uniform
            self_collision(P, T, C, SB, lane, side, R0, st.ang, st.vel, KS, 2.0f * LC.mu - P.terrain_friction, sn, sc);
        }
        out.term = ro.term; out.pen_count = ro.pen_count;
        out.foot_force = cl4 + sc.fl[2];   // net contact force on the foot link: terrain + self-collision
        write_link_rows(P, lfo, C, ro.lf, ro.fl2, ro.fl3, foot_terrain, sc, out.term, out.pen_count);   // (last sub-step: the flags from the NET link forces)
        ca2 = ro.fa2 + sc.fa[0]; cl2 = ro.fl2 + sc.fl[0];
        ca3 = ro.fa3 + sc.fa[1]; cl3 = ro.fl3 + sc.fl[1];
        ca4 = ca4 + sc.fa[2]; cl4 = cl4 + sc.fl[2];
        c0a = ro.f0a + sc.f0a; c0l = ro.f0l + sc.f0l;
    }
    // ---- inward pass from the leaf frame: frames recomputed downwards, rigid inertia + bias force formed where consumed
    S3 A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    M3 B = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    V3 pa = zero, pl = zero;
    V3 Sa[LEG], Ss[LEG], Ua[LEG], Ul[LEG];
    float dinv[LEG], uu[LEG];
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        const V3 a = axis_k(R, kAxis[k]);
        const V3 s = cross(rho, a);
        const float m = C.body[k].mass;
        {
            const V3 kap = rho + rot(R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            const S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
            S3 Ak; V3 h;
            rigid_inertia_lean(R, kap, m, Ic, Ak, h);
            const V3 hl = fma3(v, m, cross(w, h));
            const V3 ha = mul(Ak, w) + cross(h, v);
            V3 bpa = cross(w, ha) + cross(v, hl) + mul(Ak, za) + cross(h, zl);   // velocity-product bias + I_k zeta_k (substep(): kFoldC)
            V3 bpl = cross(w, hl) + zl * m - cross(h, za);
            if (k == 2) { bpa = bpa - ca2; bpl = bpl - cl2; }
            if (k == 3) { bpa = bpa - ca3; bpl = bpl - cl3; }
            if (k == 4) { bpa = bpa - ca4; bpl = bpl - cl4; }
            A = A + Ak;
            B.a01 -= h.z; B.a02 += h.y; B.a10 += h.z; B.a12 -= h.x; B.a20 -= h.y; B.a21 += h.x;
            D.xx += m; D.yy += m; D.zz += m;
            pa = pa + bpa; pl = pl + bpl;
        }
        const V3 ua = mul(A, a) + mul(B, s);
        const V3 ul = mulT(B, a) + mul(D, s);
        const float di = grx_rcp(dot(a, ua) + dot(s, ul));
        const float qk = st.q[k], qdk = st.qd[k], qlo = C.body[k].qlo, qhi = C.body[k].qhi;
        const float viol = qk < qlo ? qlo - qk : (qk > qhi ? qhi - qk : 0.f);
        const float t = tau_m[k] + (C.body[k].Klim * viol - (viol != 0.f ? C.body[k].Clim * qdk : 0.f));   // joint-limit spring/damper on top of the motor torque
        const float u = t - (dot(a, pa) + dot(s, pl));
        syr(A, ua, di); ger(B, ua, ul, di); syr(D, ul, di);
        const float ud = u * di;
        pa = fma3(ua, ud, pa); pl = fma3(ul, ud, pl);
        Sa[k] = a; Ss[k] = s; Ua[k] = ua; Ul[k] = ul; dinv[k] = di; uu[k] = u;
        if (k > 0) {   // down to the parent's frame
            w = fma3(a, -qdk, w); v = fma3(s, -qdk, v);
            za = za - cross(w, a) * qdk;
            zl = zl - (cross(v, a) + cross(w, s)) * qdk;
            R = joint_rot_k(R, cs[k], -sn_[k], kAxis[k]);
            rho = rho - rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
        }
    }
    // ---- base: both chains (DPP pair exchange), the base lump, the 6 x 6
    pa = pa - c0a; pl = pl - c0l;
    A = pair_sum(A); B = pair_sum(B); D = pair_sum(D);
    pa = pair_sum(pa); pl = pair_sum(pl);
    {
        const V3 kap = rot(R0, LC.base_c);
        const float m = LC.base_m;
        S3 A0; V3 h;
        rigid_inertia_lean(R0, kap, m, LC.base_I, A0, h);
        const V3 w0 = st.ang, v0 = st.vel;
        const V3 hl = fma3(v0, m, cross(w0, h));
        const V3 ha = mul(A0, w0) + cross(h, v0);
        pa = pa + cross(w0, ha) + cross(v0, hl);
        pl = pl + cross(w0, hl);
        A = A + A0;
        B.a01 -= h.z; B.a02 += h.y; B.a10 += h.z; B.a12 -= h.x; B.a20 -= h.y; B.a21 += h.x;
        D.xx += m; D.yy += m; D.zz += m;
    }
    const S3 Di = inv(D);
    const V3 b0 = v3(B.a00, B.a01, B.a02), b1 = v3(B.a10, B.a11, B.a12), b2 = v3(B.a20, B.a21, B.a22);
    const V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
    const S3 Sc = {A.xx - dot(b0, d0), A.xy - dot(b0, d1), A.xz - dot(b0, d2), A.yy - dot(b1, d1), A.yz - dot(b1, d2), A.zz - dot(b2, d2)};
    const V3 alpha = mul(inv(Sc), mul(B, mul(Di, pl)) - pa);
    const V3 acc = neg(mul(Di, pl + mulT(B, alpha)));
    // ---- accelerations (root -> leaf), integration (semi-implicit Euler)
    V3 aa = alpha, al = acc;
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        const float qd2 = (uu[k] - (dot(Ua[k], aa) + dot(Ul[k], al))) * dinv[k];
        aa = fma3(Sa[k], qd2, aa);
        al = fma3(Ss[k], qd2, al);
        float vq = fmaf(qd2, dt, st.qd[k]);
        vq = fminf(fmaxf(vq, -C.body[k].vlim), C.body[k].vlim);
        st.qd[k] = vq;
        st.q[k] = fmaf(vq, dt, st.q[k]);
    }
    const V3 lin = acc + cross(st.ang, st.vel);  // classical acceleration of the base origin
    st.vel = v3(st.vel.x + (lin.x + P.gravity[0]) * dt, st.vel.y + (lin.y + P.gravity[1]) * dt, st.vel.z + (lin.z + P.gravity[2]) * dt);
    st.ang = fma3(alpha, dt, st.ang);
    st.pos = fma3(st.vel, dt, st.pos);
    const float hx = 0.5f * dt * st.ang.x, hy = 0.5f * dt * st.ang.y, hz = 0.5f * dt * st.ang.z;
    const float x = st.qx, y = st.qy, z = st.qz, ww = st.qw;
    const float nx = x + hx * ww + hy * z - hz * y;
    const float ny = y - hx * z + hy * ww + hz * x;
    const float nz = z + hx * y - hy * x + hz * ww;
    const float nw = ww - hx * x - hy * y - hz * z;
    const float n = grx_rsq(nx * nx + ny * ny + nz * nz + nw * nw);
    st.qx = nx * n; st.qy = ny * n; st.qz = nz * n; st.qw = nw * n;
}
#include "grx_wavepipe.h"

// kinematics only: this lane's foot link frame in the current state
GRX_DEV FootKin foot_kinematics(const SideConst& C, const LaneState& st) {
    R3 Rp = quat_to_R(st.qx, st.qy, st.qz, st.qw);
    V3 rho = v3(0.f, 0.f, 0.f), w = st.ang, v = st.vel;
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        rho = rho + rot(Rp, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
        float sn, cs;
        grx_sincos(st.q[k], sn, cs);
        R3 R = joint_rot_k(Rp, cs, sn, kAxis[k]);
        V3 a = axis_k(R, kAxis[k]);
        V3 s = cross(rho, a);
        w = fma3(a, st.qd[k], w);
        v = fma3(s, st.qd[k], v);
        Rp = R;
    }
    V3 fr = rho + rot(Rp, v3(C.foot_pos[0], C.foot_pos[1], C.foot_pos[2]));
    FootKin f;
    f.pos = st.pos + fr;
    f.vel = v + cross(w, fr);
    f.ang = w;
    return f;
}

// GRX_T_RIGID_BODY_STATES (gym.acquire_rigid_body_state_tensor, legged_robot.py:113,134; refreshed after every gym.simulate,
// legged_robot_fftai.py:76): position, orientation (xyzw), linear and angular velocity of every URDF link frame in the state
// AFTER the last sub-step and BEFORE reset_idx (the reference's tensor is not refreshed by a reset either).  One lane walks
// its leg once more -- rotations as matrices for the offsets, as quaternions for the orientations -- and stores the frames
// of the links its tables list (grx_capi.cpp build_rbs_tables).  Only with grx_config.publish_rigid_body_states.
GRX_DEV void quat_mul(const float a[4], const float b[4], float o[4]) {   // xyzw
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// part / nparts: the table entries go round the nparts waves that share the work (each walks the chain: ~150 instructions)
GRX_DEV void publish_rigid_body_states(KP P, const SideConst& C, int side, V3 pos, const float rootq[4], V3 vel, V3 ang,
                                       const float q[LEG], const float qd[LEG], int e, int N, bool act, int part = 0, int nparts = 1) {
    const RbsTables& T = *P.rbs_tab;
    ChainKin K = {quat_to_R(rootq[0], rootq[1], rootq[2], rootq[3]), v3(0.f, 0.f, 0.f), ang, vel};
    float bq[4] = {rootq[0], rootq[1], rootq[2], rootq[3]};
#pragma unroll
    for (int lvl = 0; lvl <= LEG; ++lvl) {
        if (lvl > 0) {
            const int k = lvl - 1;
            chain_step(C, k, q[k], qd[k], K);
            float sh, ch;
            grx_sincos(0.5f * q[k], sh, ch);
            const float jq[4] = {kAxis[k] == 0 ? sh : 0.f, kAxis[k] == 1 ? sh : 0.f, kAxis[k] == 2 ? sh : 0.f, ch};
            float nq[4];
            quat_mul(bq, jq, nq);
            bq[0] = nq[0]; bq[1] = nq[1]; bq[2] = nq[2]; bq[3] = nq[3];
        }
        const int i0 = T.off[side][lvl], n = T.off[side][lvl + 1] - i0;
        const int nmax = max(T.off[0][lvl + 1] - T.off[0][lvl], T.off[1][lvl + 1] - T.off[1][lvl]);   // uniform trip count
        for (int j = 0; j < nmax; ++j) {
            if ((T.off[0][lvl] + j) % nparts != part) continue;   // (wave-uniform)
            if (j < n && act) {
                const RbsEntry E = T.e[side][i0 + j];
                const V3 r = K.rho + rot(K.R, v3(E.px, E.py, E.pz));
                const V3 vl = K.v + cross(K.w, r);
                const float eq[4] = {E.qx, E.qy, E.qz, E.qw};
                float lq[4];
                quat_mul(bq, eq, lq);
                float* o = P.rbs + (size_t)(E.link * 13) * N + e;
                const size_t n_ = (size_t)N;
                o[0] = pos.x + r.x; o[n_] = pos.y + r.y; o[2 * n_] = pos.z + r.z;
                o[3 * n_] = lq[0]; o[4 * n_] = lq[1]; o[5 * n_] = lq[2]; o[6 * n_] = lq[3];
                o[7 * n_] = vl.x; o[8 * n_] = vl.y; o[9 * n_] = vl.z;
                o[10 * n_] = K.w.x; o[11 * n_] = K.w.y; o[12 * n_] = K.w.z;
            }
        }
    }
}

GRX_DEV float urand(KP P, uint32_t genv, uint32_t step, uint32_t stream, uint32_t i, float lo, float hi) {
    return (hi - lo) * grx_rand(P.seed, genv, step, stream, i) + lo;
}

// _compute_torques before the motor-strength ratio and the clip (legged_robot.py:693-707), any control type: the tree / generic kernels
// (the fused kernels spell the 'P' law out in their sub-step loop; 'V' and 'T' run in the one-wave layout only)
GRX_DEV float control_torque(KP P, float kp, float kd, float q0, float a, float q, float qd, const float* qd_last) {
    if (P.control_type == GRX_CONTROL_T) return a * P.action_scale;
    if (P.control_type == GRX_CONTROL_V) return kp * (a * P.action_scale - qd) - kd * (qd - *qd_last) / P.sim_dt;
    return kp * (a * P.action_scale + q0 - q) - kd * qd;
}
// legged_robot.py:650-677
GRX_DEV void resample_commands(KP P, uint32_t genv, uint32_t step, uint32_t stream, float cmd[3]) {
    float c0 = urand(P, genv, step, stream, 0, P.cmd_lin_vel_x[0], P.cmd_lin_vel_x[1]);
    float c1 = urand(P, genv, step, stream, 1, P.cmd_lin_vel_y[0], P.cmd_lin_vel_y[1]);
    float keep = sqrtf(c0 * c0 + c1 * c1) > 0.1f ? 1.0f : 0.0f;
    cmd[0] = c0 * keep;
    cmd[1] = c1 * keep;
    // heading mode: the reference's draw goes to commands[:, 3], which nothing reads (legged_robot.py:668-671)
    if (!P.heading_command) cmd[2] = urand(P, genv, step, stream, 2, P.cmd_ang_vel_yaw[0], P.cmd_ang_vel_yaw[1]);
}
// legged_robot.py:320-326: the yaw command from the heading error; commands_heading is the reference's all-zero buffer (gr1t1.py:124)
GRX_DEV float heading_yaw_command(KP P, V3 qv, float qw) {
    const V3 fwd = quat_apply(qv, qw, v3(1.f, 0.f, 0.f));
    const float tp = 6.283185307179586f;
    float r = fmodf(0.f - atan2f(fwd.y, fwd.x), tp);   // wrap_to_pi (math.py:38-41): Python's %, then -2 pi above pi
    if (r < 0.f) r += tp;
    if (r > 3.14159265358979f) r -= tp;
    return fminf(fmaxf(0.5f * r, P.cmd_ang_vel_yaw[0]), P.cmd_ang_vel_yaw[1]);
}

struct EnvAux {  // per-env (replicated in both lanes) pipeline state touched by reset
    float cmd[3];
    float origin[3];
    int level, type;
};

// The uniform draws of one lane's reset_idx: they depend on (seed, env, step) only, not on the state, so an idle
// helper wave can have them ready before wave 0 knows who resets.  Same counters and words as grx_rand(stream, i)
// item by item; six Philox blocks instead of one per draw.
struct ResetRand { float dof[LEG], root[9], cmd[3]; };
GRX_DEV ResetRand reset_rand(KP P, uint32_t genv, uint32_t step, int side) {
    const uint32_t k0 = (uint32_t)P.seed, k1 = (uint32_t)(P.seed >> 32);
    ResetRand r;
    {   // RESET_DOF items side*5 + k: side 0 -> block 0 words 0..3, block 1 word 0; side 1 -> block 1 words 1..3, block 2 words 0, 1
        const U4 a = grx_philox4x32_10(genv, step, GRX_RNG_RESET_DOF, (uint32_t)side, k0, k1);
        const U4 b = grx_philox4x32_10(genv, step, GRX_RNG_RESET_DOF, (uint32_t)side + 1u, k0, k1);
        r.dof[0] = grx_u01(side ? a.y : a.x); r.dof[1] = grx_u01(side ? a.z : a.y); r.dof[2] = grx_u01(side ? a.w : a.z);
        r.dof[3] = grx_u01(side ? b.x : a.w); r.dof[4] = grx_u01(side ? b.y : b.x);
    }
    {
        const U4 a = grx_philox4x32_10(genv, step, GRX_RNG_RESET_ROOT, 0u, k0, k1);
        const U4 b = grx_philox4x32_10(genv, step, GRX_RNG_RESET_ROOT, 1u, k0, k1);
        const U4 c = grx_philox4x32_10(genv, step, GRX_RNG_RESET_ROOT, 2u, k0, k1);
        r.root[0] = grx_u01(a.x); r.root[1] = grx_u01(a.y); r.root[2] = grx_u01(a.z); r.root[3] = grx_u01(a.w);
        r.root[4] = grx_u01(b.x); r.root[5] = grx_u01(b.y); r.root[6] = grx_u01(b.z); r.root[7] = grx_u01(b.w);
        r.root[8] = grx_u01(c.x);
    }
    {
        const U4 a = grx_philox4x32_10(genv, step, GRX_RNG_CMD_RESET, 0u, k0, k1);
        r.cmd[0] = grx_u01(a.x); r.cmd[1] = grx_u01(a.y); r.cmd[2] = grx_u01(a.z);
    }
    return r;
}
GRX_DEV float lerp_u(float u, float lo, float hi) { return (hi - lo) * u + lo; }   // urand() on a ready draw

// reset_idx for one env (legged_robot.py:377-440, 717-826; legged_robot_fftai.py:137-146):
// each lane resets its own leg, the root state is computed redundantly (same counters -> same values)
GRX_DEV void reset_env(KP P, const SideConst& C, int side, uint32_t genv, uint32_t step, bool init_done,
                       LaneState& st, EnvAux& ea, const ResetRand& rr) {
    if (P.curriculum && P.terrain_type != GRX_TERRAIN_PLANE && init_done) {  // legged_robot.py:799-826
        float dx = st.pos.x - ea.origin[0], dy = st.pos.y - ea.origin[1];
        float dist = sqrtf(dx * dx + dy * dy);
        int up = dist > P.terrain_length * 0.5f;
        float cn = sqrtf(ea.cmd[0] * ea.cmd[0] + ea.cmd[1] * ea.cmd[1]);
        int down = (dist < cn * P.max_episode_length_s * 0.5f) && !up;
        ea.level += up - down;
        if (ea.level >= P.num_terrain_rows) {
            float u = grx_rand(P.seed, genv, step, GRX_RNG_CURRICULUM, 0);
            ea.level = min((int)(u * (float)P.num_terrain_rows), P.num_terrain_rows - 1);
        } else if (ea.level < 0)
            ea.level = 0;
        const float* o = P.terrain_origins + ((size_t)ea.level * P.num_terrain_cols + ea.type) * 3;
        ea.origin[0] = o[0]; ea.origin[1] = o[1]; ea.origin[2] = o[2];
    }
#pragma unroll
    for (int k = 0; k < LEG; ++k) {  // _reset_dofs
        float f = P.randomize_init_dof_pos ? lerp_u(rr.dof[k], 0.5f, 1.5f) : 1.0f;
        st.q[k] = f * C.body[k].q0;
        st.qd[k] = 0.0f;
    }
    st.pos = v3(P.init_pos[0] + ea.origin[0], P.init_pos[1] + ea.origin[1], P.init_pos[2] + ea.origin[2]);
    if (P.terrain_type != GRX_TERRAIN_PLANE) {
        st.pos.x += lerp_u(rr.root[0], -1.0f, 1.0f);
        st.pos.y += lerp_u(rr.root[1], -1.0f, 1.0f);
    }
    float yaw = lerp_u(rr.root[2], -6.283185307179586f, 6.283185307179586f);
    float sy, cy;
    sincosf(yaw * 0.5f, &sy, &cy);
    st.qx = 0.f; st.qy = 0.f; st.qz = sy; st.qw = cy;  // quat_from_euler_xyz(0,0,yaw)
    if (P.randomize_init_base_velocity) {
        st.vel = v3(lerp_u(rr.root[3], -0.5f, 0.5f), lerp_u(rr.root[4], -0.5f, 0.5f), lerp_u(rr.root[5], -0.5f, 0.5f));
        st.ang = v3(lerp_u(rr.root[6], -0.5f, 0.5f), lerp_u(rr.root[7], -0.5f, 0.5f), lerp_u(rr.root[8], -0.5f, 0.5f));
    } else {
        st.vel = v3(0.f, 0.f, 0.f);
        st.ang = v3(0.f, 0.f, 0.f);
    }
    {   // _resample_commands (legged_robot.py:650-677) on the ready draws
        const float c0 = lerp_u(rr.cmd[0], P.cmd_lin_vel_x[0], P.cmd_lin_vel_x[1]);
        const float c1 = lerp_u(rr.cmd[1], P.cmd_lin_vel_y[0], P.cmd_lin_vel_y[1]);
        const float keep = sqrtf(c0 * c0 + c1 * c1) > 0.1f ? 1.0f : 0.0f;
        ea.cmd[0] = c0 * keep;
        ea.cmd[1] = c1 * keep;
        if (!P.heading_command) ea.cmd[2] = lerp_u(rr.cmd[2], P.cmd_ang_vel_yaw[0], P.cmd_ang_vel_yaw[1]);
    }
    st.anchor_on = 0;
}

// legged_robot.py:1235-1274 _get_heights, one point.  The cell index is a discrete decision (truncation, then the
// min of three raster corners), so this function is compiled with reassociation OFF and spells out
// quat_apply_yaw (math.py:38-42 -> torch_utils.py:48-55) in the reference's operation order; zn/wn are the
// yaw-only quaternion's z and w (the x, y terms of the two cross products are exact zeros and are dropped).
GRX_DEV float height_sample(KP P, const KTables& T, float zn, float wn, V3 pos, int k) {
#pragma clang fp reassociate(off)
    const float bx = T.height_points[k][0], by = T.height_points[k][1];
    const float tx = -(zn * by) * 2.0f, ty = (zn * bx) * 2.0f;   // t = 2 * cross(qv, b)
    const float ux = -(zn * ty), uy = zn * tx;                   // cross(qv, t)
    const float qx_ = bx + wn * tx + ux, qy_ = by + wn * ty + uy;
    float px = (qx_ + pos.x + P.border_size) / P.horizontal_scale;
    float py = (qy_ + pos.y + P.border_size) / P.horizontal_scale;
    int ix = min(max((int)px, 0), P.hf_rows - 2), iy = min(max((int)py, 0), P.hf_cols - 2);
    const uint2 cc = P.hf_cells[(size_t)ix * P.hf_cols + iy];   // one gather: heights[ix, iy], [ix + 1, iy], [ix, iy + 1]
    const int h1 = (int16_t)(cc.x & 0xffffu), h3 = (int16_t)(cc.x >> 16), h2 = (int16_t)(cc.y & 0xffffu);
    const int h = min(min(h1, h2), h3);
    return (float)h * P.vertical_scale;
}

// One lane's share of the height scan: points k = first, first + LPE*NW, ... (NW waves x LPE lanes per env); raw heights
// parked in the env's pri_obs staging row; returns the lane's partial sum.  Batches of 8 independent gathers.
template <int NW, int B = 8>   // B: gathers per batch (the reference's 121 points over 7 x 4 lanes: 5 per lane)
GRX_DEV float height_scan_share(KP P, const KTables& T, float zn, float wn, V3 pos, int first, int nh, float* prow) {
    float hsum = 0.f;
    for (int k0 = first; k0 < nh; k0 += B * LPE * NW) {
        float hb[B];
#pragma unroll
        for (int j = 0; j < B; ++j) hb[j] = height_sample(P, T, zn, wn, pos, min(k0 + LPE * NW * j, nh - 1));
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int k = k0 + LPE * NW * j;
            if (k < nh) { prow[GRX_NUM_OBS + 8 + k] = hb[j]; hsum += hb[j]; }
        }
    }
    return hsum;
}

// LDS row stride of the pri_obs staging rows: 2 x odd, so that the two lanes of an env (adjacent columns) and the
// 32 envs of a wave land in 64 distinct banks (the natural stride, 168 = 8 x 21, is a 4-way conflict on every access).
constexpr int PRS = GRX_MAX_PRI + 2;
static_assert(PRS % 2 == 0 && (PRS / 2) % 2 == 1, "PRS must be 2 x odd");

// compute_observations' height block (legged_robot.py:449-451 -> gr1t1.py:305-313): one lane's share of an env's
// points, k = first, first + NL, ...: raw height (parked in the staging row by the scan) -> clipped, scaled offset,
// written back in place; returns the lane's partial sum of the clipped offsets (base_heights_offset numerator).
template <int NL>
GRX_DEV float obs_heights_share(KP P, float posz, int first, int nh, float* prow, bool have_raw, bool act, int e, int N) {
    float sum = 0.f;
    for (int k0 = first; k0 < nh; k0 += 8 * NL) {   // batches of 8: one exposed LDS latency per batch, not per point
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = have_raw ? prow[GRX_NUM_OBS + 8 + min(k0 + NL * j, nh - 1)] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + NL * j;
            if (k < nh) {
                float d = posz - P.base_height_target - hv[j];
                d = fminf(fmaxf(d, -1.f), 1.f) * P.obs_scale_height;
                if (act && P.publish_heights) P.heights[(size_t)k * N + e] = hv[j];   // env.measured_heights (a reference attribute): every step, or on demand by grx_refresh
                prow[GRX_NUM_OBS + 8 + k] = fminf(fmaxf(d * P.obs_scale_height, -P.clip_observations), P.clip_observations);
                sum += d;
            }
        }
    }
    return sum;
}

GRX_DEV float sum_abs_mask(const float a[LEG], uint32_t mask) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LEG; ++k) if (mask & (1u << k)) s += fabsf(a[k]);
    return s;
}


// Observation-noise Philox blocks of one lane.  Streams are keyed so that every lane indexes its blocks statically:
// base terms (ang vel, gravity; obs 3..8) = stream NOISE item i-3; dof terms = stream NOISE_DOF_L/R item
// group*5 + k (group 0 pos, 1 vel, 2 action).  The six 10-round chains are advanced together, round by round, so
// the 64-bit multiplies of independent chains interleave (a serial chain per value cost ~18k cycles/step, measured).
constexpr int NZB = 6;   // slots 0..3: dof stream blocks 0..3; slots 4,5: base stream blocks 0,1 (left lane)
GRX_DEV void noise_blocks(KP P, uint32_t genv, uint32_t step, int side, U4 nzb[NZB]) {
    uint32_t c0[NZB], c1[NZB], c2[NZB], c3[NZB];
#pragma unroll
    for (int b = 0; b < NZB; ++b) {
        c0[b] = genv; c1[b] = step;
        c2[b] = b < 4 ? (uint32_t)(side == 0 ? GRX_RNG_NOISE_DOF_L : GRX_RNG_NOISE_DOF_R) : (uint32_t)GRX_RNG_NOISE;
        c3[b] = b < 4 ? b : b - 4;
    }
    uint32_t k0 = (uint32_t)P.seed, k1 = (uint32_t)(P.seed >> 32);
#pragma unroll
    for (int rnd = 0; rnd < 10; ++rnd) {
#pragma unroll
        for (int b = 0; b < NZB; ++b) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0[b], p1 = (uint64_t)0xCD9E8D57u * c2[b];
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1[b] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3[b] ^ k1;
            c1[b] = (uint32_t)p1; c3[b] = (uint32_t)p0; c0[b] = n0; c2[b] = n2;
        }
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int b = 0; b < NZB; ++b) { nzb[b].x = c0[b]; nzb[b].y = c1[b]; nzb[b].z = c2[b]; nzb[b].w = c3[b]; }
}

// rows of the debug-injection table ([DBG_ROWS][N], grx_debug_post_physics): see grx_step_kernel's DBG instantiations
enum DbgRow { DBG_FEET_FORCE = 0, DBG_FEET_POS = 6, DBG_AVG_FORCE = 12, DBG_AVG_SPEED = 14, DBG_TORQUES = 20, DBG_LAST_LAST_ACTIONS = DBG_TORQUES + GRX_MAX_DOFS,
              DBG_TERM_CONTACT = DBG_LAST_LAST_ACTIONS + GRX_MAX_DOFS, DBG_APPLY_RESET = DBG_TERM_CONTACT + 1, DBG_ROWS = DBG_APPLY_RESET + 1 };   // (a row per dof of ANY model: the 32-DOF full body's fixture goes through the tree kernel)
#ifndef GRX_QUAD_TU
#include "grx_generic.h"
#include "grx_tree.h"
#endif

}  // namespace

// ------------------------------------------------------------------------------------------
// Inputs of compute_reward for one lane (one leg of one env): everything the reward terms and the episode sums read.
struct RewIn {
    float a_last[LEG], a_cur[LEG], q[LEG], qd[LEG], qd_last[LEG], torque[LEG];
    float feet_height, air_time, land_time, avg_force;
    V3 avg_speed, foot_force;
    float contact, first_contact;   // 0 / 1
    float cmd[3];
    V3 blv, bav, pg;
    float bho_stale, qx, qy, qz, qw, pen_count, reset, time_out;
};
constexpr int REWIN_FLOATS = 6 * LEG + 4 + 6 + 2 + 3 + 9 + 8;
template <class F>
GRX_DEV void rewin_fields(RewIn& r, F&& f) {
#pragma unroll
    for (int k = 0; k < LEG; ++k) { f(r.a_last[k]); f(r.a_cur[k]); f(r.q[k]); f(r.qd[k]); f(r.qd_last[k]); f(r.torque[k]); }
    f(r.feet_height); f(r.air_time); f(r.land_time); f(r.avg_force);
    f(r.avg_speed.x); f(r.avg_speed.y); f(r.avg_speed.z); f(r.foot_force.x); f(r.foot_force.y); f(r.foot_force.z);
    f(r.contact); f(r.first_contact);
    f(r.cmd[0]); f(r.cmd[1]); f(r.cmd[2]);
    f(r.blv.x); f(r.blv.y); f(r.blv.z); f(r.bav.x); f(r.bav.y); f(r.bav.z); f(r.pg.x); f(r.pg.y); f(r.pg.z);
    f(r.bho_stale); f(r.qx); f(r.qy); f(r.qz); f(r.qw); f(r.pen_count); f(r.reset); f(r.time_out);
}

// Which reward terms a call of reward_and_sums owns.  PART 0: all of them (one wave).  With four waves per block the
// terms are split over two helper waves: PART 1 = the joint-space terms (action differences, dof acc / torque /
// velocity, limits, pose), PART 2 = the base, feet, collision and termination terms.
template <int PART>
__host__ __device__ constexpr bool rew_in_part(int t) {
    if (PART == 0) return true;
    const bool joint = t == GRX_REW_ACTION_DIFF || t == GRX_REW_ACTION_DIFF_DIFF || t == GRX_REW_ACTION_DIFF_KNEE ||
                       t == GRX_REW_DOF_ACC_NEW || t == GRX_REW_DOF_TOR_NEW || t == GRX_REW_DOF_TOR_NEW_HIP_ROLL ||
                       t == GRX_REW_DOF_VEL_NEW || t == GRX_REW_DOF_VEL_NEW_KNEE || t == GRX_REW_LIMITS_ACTIONS ||
                       t == GRX_REW_LIMITS_DOF_POS || t == GRX_REW_LIMITS_DOF_TOR || t == GRX_REW_LIMITS_DOF_VEL ||
                       t == GRX_REW_POSE_OFFSET || t == GRX_REW_POSE_OFFSET_HIP_YAW || t == GRX_REW_STAND_STILL;
    return PART == 1 ? joint : !joint;
}

// the env's running episode sums of the terms a part owns (issued early so the HBM latency overlaps other work)
template <int PART>
GRX_DEV void load_episode_sums(KP P, int e, int N, float es[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) if (rew_in_part<PART>(t)) es[t] = (P.reward_scale_dt[t] != 0.f) ? P.episode_sums[(size_t)t * N + e] : 0.f;
}

// compute_reward (legged_robot.py:355-375) + episode sums + the block's finished-episode statistics for one lane.
// Runs on wave 0 (PART 0), or -- four waves per block -- as PART 1 on wave 1 and PART 2 on wave 3 while wave 0 goes on
// with reset and observations: PART 2 parks its partial reward in *rew_part and raises rew_flag, PART 1 adds it,
// clips, adds the termination term and writes rew_buf.
// es_pre: the env's running episode sums if the caller loaded them early, else nullptr.
template <int PART>
GRX_DEV void reward_and_sums(KP P, const SideConst& C, const RewIn& in, int lane, int side, int e, int N, bool act,
                             float* s_stat, const float* es_pre, float* rew_part = nullptr, int* rew_flag = nullptr,
                             const float* a_ll = nullptr, bool keep_sums = false) {   // keep_sums: debug entry, reset reported but not applied
    const int j0 = side * LEG;
    const float dtp = P.sim_dt * (float)P.decimation;
    const bool reset = in.reset != 0.f, time_out = in.time_out != 0.f;
    float es_raw[NT];   // running episode sums
#pragma unroll
    for (int t = 0; t < NT; ++t)
        es_raw[t] = !rew_in_part<PART>(t) ? 0.f : (es_pre ? es_pre[t] : ((P.reward_scale_dt[t] != 0.f) ? P.episode_sums[(size_t)t * N + e] : 0.f));
    // ---- compute_reward (legged_robot.py:355-375): per-lane partial sums, pair-combined
    float r[NT];
    {
        const float as = P.action_scale, H = P.swing_feet_height_target, T = P.feet_air_time_target;
        const GRX_AS4 float* sg = P.reward_sigma;
        const uint32_t knee = (P.knee_mask >> j0) & 31u, hiproll = (P.hip_roll_mask >> j0) & 31u, hipyaw = (P.hip_yaw_mask >> j0) & 31u;
        const uint32_t ankle = ((side ? P.ankle_right_mask : P.ankle_left_mask) >> j0) & 31u;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, sacc = 0.f, stor = 0.f, svel = 0.f, spose = 0.f, sla = 0.f, slp = 0.f, slt = 0.f, slv = 0.f, shy = 0.f;
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            float d1 = (in.a_last[k] - in.a_cur[k]) * as;
            s1 += fabsf(d1);  // last_last_actions == last_actions (legged_robot_fftai.py:94 copies after legged_robot.py:299)
            if (a_ll) s2 += fabsf(d1 - (a_ll[k] - in.a_last[k]) * as);   // grx_debug_post_physics only: injected last_last_actions
            if (knee & (1u << k)) s3 += fabsf((in.a_cur[k] - in.a_last[k]) * as);
            sacc += fabsf((in.qd[k] - in.qd_last[k]) / dtp);
            stor += fabsf(in.torque[k]);
            svel += fabsf(in.qd[k]);
            float po = fabsf(in.q[k] - C.body[k].q0);
            spose += po;
            if (hipyaw & (1u << k)) shy += po;
            float a = in.a_cur[k] * as, oa = 0.f, op = 0.f;
            if (a - C.body[k].slo < 0.f) oa += -(a - C.body[k].slo);
            if (a - C.body[k].shi > 0.f) oa += (a - C.body[k].shi);
            sla += oa * oa;
            if (in.q[k] - C.body[k].slo < 0.f) op += -(in.q[k] - C.body[k].slo);
            if (in.q[k] - C.body[k].shi > 0.f) op += (in.q[k] - C.body[k].shi);
            slp += fabsf(op);
            slv += fminf(fmaxf(fabsf(in.qd[k]) - C.body[k].vlim * P.soft_dof_vel_limit, 0.f), 1.f);
            slt += fmaxf(fabsf(in.torque[k]) - C.body[k].effort * P.soft_torque_limit, 0.f);
        }
        float tor_hr = sum_abs_mask(in.torque, hiproll), vel_kn = sum_abs_mask(in.qd, knee);
        float h = in.feet_height;
        float lift = sum_abs_mask(in.torque, ankle) * fabsf(h) * (h > H * 0.5f ? 1.f : 0.f);
        float hmin = fminf(h, pair_swap(h));
        float mid = fabsf(in.air_time - T * 0.5f);
        float af = mid * in.avg_force;
        float ah = mid * fabsf(h - hmin - H);
        float at = expf(sg[GRX_REW_FEET_AIR_TIME] * fabsf(in.air_time - T)) * in.first_contact;
        float le = (in.land_time - P.feet_land_time_max) * (in.land_time > P.feet_land_time_max ? 1.f : 0.f);
        float lt = 1.f - expf(sg[GRX_REW_FEET_LAND_TIME] * le);
        float close = fabsf(h - H * 0.25f) * (h < H * 0.25f ? 1.f : 0.f) / (H * 0.25f);
        float exy = sqrtf(in.avg_speed.x * in.avg_speed.x + in.avg_speed.y * in.avg_speed.y) * close;
        float far = fabsf(h - H * 3.f / 4.f) * (h > H * 3.f / 4.f ? 1.f : 0.f) / (H * 1.f / 4.f);
        float ez = fabsf(in.avg_speed.z) * far;
        V3 F = in.foot_force;
        float serr = sqrtf(F.x * F.x + F.y * F.y) - P.feet_stumble_ratio * fabsf(F.z);
        serr = serr * (serr > 0.f ? 1.f : 0.f);
        float stum = 1.f - expf(sg[GRX_REW_FEET_STUMBLE] * serr);
        float ncontact = in.contact;
        s1 = pair_sum(s1); s3 = pair_sum(s3);
        s2 = a_ll ? pair_sum(s2) : s1; sacc = pair_sum(sacc); stor = pair_sum(stor); svel = pair_sum(svel);
        spose = pair_sum(spose); sla = pair_sum(sla); slp = pair_sum(slp); slt = pair_sum(slt); slv = pair_sum(slv);
        shy = pair_sum(shy); tor_hr = pair_sum(tor_hr); vel_kn = pair_sum(vel_kn); lift = pair_sum(lift);
        af = pair_sum(af); ah = pair_sum(ah); at = pair_sum(at); lt = pair_sum(lt); exy = pair_sum(exy); ez = pair_sum(ez);
        stum = pair_sum(stum); ncontact = pair_sum(ncontact);
        const float cmd_n = sqrtf(in.cmd[0] * in.cmd[0] + in.cmd[1] * in.cmd[1]);
        const float moving = cmd_n > 0.1f ? 1.f : 0.f;
        r[GRX_REW_ACTION_DIFF] = 1.f - expf(sg[GRX_REW_ACTION_DIFF] * s1);
        r[GRX_REW_ACTION_DIFF_DIFF] = 1.f - expf(sg[GRX_REW_ACTION_DIFF_DIFF] * s2);
        r[GRX_REW_ACTION_DIFF_KNEE] = 1.f - expf(sg[GRX_REW_ACTION_DIFF_KNEE] * s3);
        r[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] * fabsf(0.f - in.bav.y));
        r[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] * fabsf(0.f - in.bav.x));
        r[GRX_REW_CMD_DIFF_ANG_VEL_YAW] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_YAW] * fabsf(in.cmd[2] - in.bav.z));
        r[GRX_REW_CMD_DIFF_BASE_HEIGHT] = expf(sg[GRX_REW_CMD_DIFF_BASE_HEIGHT] * (fabsf(in.bho_stale) * (in.bho_stale < 0.f ? 1.f : 0.f)));
        r[GRX_REW_CMD_DIFF_BASE_ORIENT] = expf(sg[GRX_REW_CMD_DIFF_BASE_ORIENT] * (fabsf(in.pg.x) + fabsf(in.pg.y)));
        R3 R0 = quat_to_R(in.qx, in.qy, in.qz, in.qw);
        {   // torso / forehead links ride on the base lump: R_link = R0 * rot; R^T(0,0,-1) = -(third row)
            float tx = -(R0.cx.z * P.torso_rot[0] + R0.cy.z * P.torso_rot[3] + R0.cz.z * P.torso_rot[6]);
            float ty = -(R0.cx.z * P.torso_rot[1] + R0.cy.z * P.torso_rot[4] + R0.cz.z * P.torso_rot[7]);
            r[GRX_REW_CMD_DIFF_TORSO_ORIENT] = P.has_torso ? expf(sg[GRX_REW_CMD_DIFF_TORSO_ORIENT] * (fabsf(tx) + fabsf(ty))) : 0.f;
            float fx = -(R0.cx.z * P.forehead_rot[0] + R0.cy.z * P.forehead_rot[3] + R0.cz.z * P.forehead_rot[6]);
            float fy = -(R0.cx.z * P.forehead_rot[1] + R0.cy.z * P.forehead_rot[4] + R0.cz.z * P.forehead_rot[7]);
            r[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] = P.has_forehead ? expf(sg[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] * (fabsf(fx) + fabsf(fy))) : 0.f;
        }
        r[GRX_REW_CMD_DIFF_LIN_VEL_X] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_X] * fabsf(in.cmd[0] - in.blv.x));
        r[GRX_REW_CMD_DIFF_LIN_VEL_Y] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_Y] * fabsf(in.cmd[1] - in.blv.y));
        r[GRX_REW_CMD_DIFF_LIN_VEL_Z] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_Z] * fabsf(0.f - in.blv.z));
        r[GRX_REW_COLLISION] = 1.f - expf(sg[GRX_REW_COLLISION] * in.pen_count);
        r[GRX_REW_DOF_ACC_NEW] = 1.f - expf(sg[GRX_REW_DOF_ACC_NEW] * sacc);
        r[GRX_REW_DOF_TOR_ANKLE_FEET_LIFT_UP] = 1.f - expf(sg[GRX_REW_DOF_TOR_ANKLE_FEET_LIFT_UP] * lift);
        r[GRX_REW_DOF_TOR_NEW] = 1.f - expf(sg[GRX_REW_DOF_TOR_NEW] * stor);
        r[GRX_REW_DOF_TOR_NEW_HIP_ROLL] = 1.f - expf(sg[GRX_REW_DOF_TOR_NEW_HIP_ROLL] * tor_hr);
        r[GRX_REW_DOF_VEL_NEW] = 1.f - expf(sg[GRX_REW_DOF_VEL_NEW] * svel);
        r[GRX_REW_DOF_VEL_NEW_KNEE] = 1.f - expf(sg[GRX_REW_DOF_VEL_NEW_KNEE] * vel_kn);
        r[GRX_REW_FEET_AIR_FORCE] = expf(sg[GRX_REW_FEET_AIR_FORCE] * af) * moving;
        r[GRX_REW_FEET_AIR_HEIGHT] = expf(sg[GRX_REW_FEET_AIR_HEIGHT] * ah) * moving;
        r[GRX_REW_FEET_AIR_TIME] = at * moving;
        r[GRX_REW_FEET_LAND_TIME] = lt * moving;
        r[GRX_REW_FEET_SPEED_XY_CLOSE_TO_GROUND] = expf(sg[GRX_REW_FEET_SPEED_XY_CLOSE_TO_GROUND] * exy);
        r[GRX_REW_FEET_SPEED_Z_CLOSE_TO_HEIGHT_TARGET] = expf(sg[GRX_REW_FEET_SPEED_Z_CLOSE_TO_HEIGHT_TARGET] * ez);
        r[GRX_REW_FEET_STUMBLE] = stum;
        r[GRX_REW_LIMITS_ACTIONS] = 1.f - expf(sg[GRX_REW_LIMITS_ACTIONS] * sla);
        r[GRX_REW_LIMITS_DOF_POS] = 1.f - expf(sg[GRX_REW_LIMITS_DOF_POS] * slp);
        r[GRX_REW_LIMITS_DOF_TOR] = 1.f - expf(sg[GRX_REW_LIMITS_DOF_TOR] * slt);
        r[GRX_REW_LIMITS_DOF_VEL] = 1.f - expf(sg[GRX_REW_LIMITS_DOF_VEL] * slv);
        r[GRX_REW_ON_THE_AIR] = ncontact == 0.f ? 1.f : 0.f;
        r[GRX_REW_POSE_OFFSET] = expf(sg[GRX_REW_POSE_OFFSET] * spose);
        r[GRX_REW_POSE_OFFSET_HIP_YAW] = 1.f - expf(sg[GRX_REW_POSE_OFFSET_HIP_YAW] * shy);
        r[GRX_REW_STAND_STILL] = expf(sg[GRX_REW_STAND_STILL] * spose) * (cmd_n < 0.1f ? 1.f : 0.f);
        r[GRX_REW_TERMINATION] = (reset && !time_out) ? 1.f : 0.f;
    }
    float rew = 0.f;
    const bool writer = act && side == 0 && lane_half(lane) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float rt = 0.f;
        if (rew_in_part<PART>(t)) {
            float sc_t = P.reward_scale_dt[t];
            if (t != GRX_REW_TERMINATION && sc_t != 0.f) { rt = r[t] * sc_t; rew += rt; }
        }
        r[t] = rt;
    }
    if (PART == 2) {   // partial reward -> the PART 1 wave
        rew_part[lane] = rew;
        flag_set(rew_flag, 1, lane);
    }
    const float term_rt = ((reset && !time_out) ? 1.f : 0.f) * P.reward_scale_dt[GRX_REW_TERMINATION];
    if (rew_in_part<PART>(GRX_REW_TERMINATION) && P.reward_scale_dt[GRX_REW_TERMINATION] != 0.f) r[GRX_REW_TERMINATION] = term_rt;
    // episode sums; reset envs contribute to the block's episode statistics (legged_robot.py:420-424)
    const unsigned long long reset_mask = __ballot(reset && writer);
    float es_all[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) es_all[t] = es_raw[t] + r[t];
    if (reset_mask) {
        // finished episodes: a wave holds 0-2 of them per step, so walk the set bits (uniform loop) and pull the
        // lane's sums through v_readlane instead of 36 six-step shuffle reductions (216 ds_bpermute + waits, measured)
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = 0.f;
        unsigned long long m = reset_mask;
        while (m) {
            const int L = __ffsll((long long)m) - 1;
            m &= m - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) if (rew_in_part<PART>(t)) acc[t] += __shfl(es_all[t], L);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) if (rew_in_part<PART>(t)) s_stat[t] = acc[t];
        }
    }
    if (writer) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!rew_in_part<PART>(t) || P.reward_scale_dt[t] == 0.f) continue;  // uniform
            P.episode_sums[(size_t)t * N + e] = (reset && !keep_sums) ? 0.f : es_all[t];
            if (P.publish_debug) P.reward_terms[(size_t)t * N + e] = r[t];
        }
    }
    if (PART != 2 && lane == 0) s_stat[NT] = (float)__popcll(reset_mask);

    if (PART != 2) {
        if (PART == 1) {
            flag_wait(rew_flag, 1);
            rew += rew_part[lane];
        }
        if (P.only_positive_rewards) rew = fmaxf(rew, 0.f);
        if (P.reward_scale_dt[GRX_REW_TERMINATION] != 0.f) rew += term_rt;
        if (writer) P.rew[e] = rew;
    }
}

// Eight waves: feet_height = foot z - mean of the measured heights, formed by whoever needs it once the six scanning waves have counted in (FL_SCAN)
GRX_DEV float scan_feet_height(KP P, int* s_flag, const float* s_hsum, float foot_z, int nh, int lane) {
    if (!P.measure_heights) return foot_z;   // (feet_height was final already; heightfield kernels only)
    flag_wait(s_flag + FL_SCAN, 6);
    const float hsum = env_sum(s_hsum[1 * 64 + lane] + s_hsum[2 * 64 + lane] + ((s_hsum[4 * 64 + lane] + s_hsum[5 * 64 + lane]) + (s_hsum[6 * 64 + lane] + s_hsum[7 * 64 + lane])));
    return nh > 0 ? (foot_z * (float)nh - hsum) / (float)nh : foot_z;
}

// Block = W waves (W = 1, 2, 4 or 8, chosen at launch: up to 4 every wave has a SIMD to itself, 8 puts two on each) for the same 32 envs,
// one env per lane PAIR in each wave.  The kernel runs one wave per SIMD, i.e. at one instruction per 4 cycles, so
// a wave's instruction count IS its time:
//   W == 1: one wave does everything (large batches: every SIMD is busy with its own envs anyway);
//   W == 2: wave 1 computes the base-lump contact wrench of every sub-step (two block barriers per sub-step);
//   W == 4: the four-wave producer/consumer pipeline of grx_wavepipe.h (sequence counters in LDS);
//   W == 8: the same work re-cut into eight roles (grx_wavepipe.h, "Eight waves per block"; the default of the pipelined layouts).
// DBG (behind the test-only entry grx_debug_post_physics; W == 1 and the pipelined layouts W == 4, 8 with lane pairs and lane
// quads -- the kernels BASELINE.json's configs launch): no sub-steps; the quantities the physics would have produced (feet
// forces / positions, sub-step averages, torques, termination contact) and last_last_actions come from `dbg` ([DBG_ROWS][N],
// see DbgRow), so the post-physics half of the step can be fed the reference's golden fixtures directly.  With the pipelines
// the helper waves skip their sub-step loops and everything behind the barrier that ends the sub-steps runs as in the product
// kernel: the height scan over the waves, the reward inputs through LDS to the two reward waves (which fetch the injected
// last_last_actions themselves), reset_idx's draws from the foot wave, the termination flag through s_tp from the base-lump
// wave, the observation height block on the helper waves.
#define GRX_STEP_KERNEL_ARGS const KParams* __restrict__ Pg, const float* __restrict__ actions_in, float delay, long long common_step, const float* __restrict__ noise_in, \
                             const float* __restrict__ dbg, float* __restrict__ obs_out, float* __restrict__ pri_out, const StepSeq sq
// (the body is a file of its own, compiled under two heads: the terrain as a compile-time property without a third value in the kernels' names)
template <bool HF, int W, bool DBG = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W > 4 ? 2 : GRX_WPE, W > 4 ? 2 : GRX_WPE))) void grx_step_kernel(GRX_STEP_KERNEL_ARGS) {
#include "grx_step_kernel_body.inc"
}
// mesh_type 'trimesh': the same step against the reference's slope-corrected triangle mesh (terrain_eval's planes per triangle half, wall_contact)
template <int W, bool DBG = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W > 4 ? 2 : GRX_WPE, W > 4 ? 2 : GRX_WPE))) void grx_step_kernel_trimesh(GRX_STEP_KERNEL_ARGS) {
    constexpr int HF = GRX_HF_TRIMESH;
#include "grx_step_kernel_body.inc"
}

// -DGRX_SPIN_LIMIT builds (grx_flags.h): where an expired spin reports before it traps -- one pointer per translation unit
#if defined(GRX_TREE16_TU)
extern "C" int grx_set_spin_word_tree16(unsigned long long* p) {
#elif defined(GRX_QUAD_TU)
extern "C" int grx_set_spin_word_quad(unsigned long long* p) {
#else
extern "C" int grx_set_spin_word(unsigned long long* p) {
#endif
#ifdef GRX_SPIN_LIMIT
    return hipMemcpyToSymbol(HIP_SYMBOL(g_grx_spin_word), &p, sizeof p) == hipSuccess ? 1 : -1;
#else
    (void)p;
    return 0;   // (product build: spins are unbounded)
#endif
}

#ifdef GRX_QUAD_TU
// grx_quad.hip: this translation unit built with GRX_LPE = 4 -- the four-wave step kernel with a lane QUAD per env, 16 envs per
// block: at <= 16 envs per CU (4096 envs on an MI355X) every CU gets a block instead of every other one
// waves: 4 (the roles of grx_wavepipe.h's header) or 8 (two waves per SIMD: four more roles take work off wave 0's chain)
extern "C" void grx_launch_step_quad(const KParams* dP, int N, int heightfield, int waves, const float* actions, float delay, long long common_step,
                                     const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream) {
    const int nblocks = (N + EPB - 1) / EPB;
#define GRX_LAUNCH_QUAD(HF_, W_) hipLaunchKernelGGL((grx_step_kernel<HF_, W_>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, delay, common_step, noise, (const float*)nullptr, obs_out, pri_out, *sq)
#define GRX_LAUNCH_QUAD_TM(W_) hipLaunchKernelGGL((grx_step_kernel_trimesh<W_>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, delay, common_step, noise, (const float*)nullptr, obs_out, pri_out, *sq)
    if (heightfield == 1) { if (waves == 8) GRX_LAUNCH_QUAD(true, 8); else GRX_LAUNCH_QUAD(true, 4); }   // (heightfield: 0 plane, 1 raster, 2 trimesh)
    else if (heightfield == 0) { if (waves == 8) GRX_LAUNCH_QUAD(false, 8); else GRX_LAUNCH_QUAD(false, 4); }
    else { if (waves == 8) GRX_LAUNCH_QUAD_TM(8); else GRX_LAUNCH_QUAD_TM(4); }
#undef GRX_LAUNCH_QUAD
#undef GRX_LAUNCH_QUAD_TM
}
extern "C" int grx_envs_per_block_quad(void) { return EPB; }
// TEST-ONLY (grx_debug_post_physics): the post-physics half of the lane-quad kernels on injected state
extern "C" void grx_launch_step_debug_quad(const KParams* dP, int N, int heightfield, int waves, const float* actions, long long common_step, const float* noise,
                                           const float* dbg, const StepSeq* sq, hipStream_t stream) {
    const int nblocks = (N + EPB - 1) / EPB;
#define GRX_LAUNCH_DBGQ(HF_, W_) hipLaunchKernelGGL((grx_step_kernel<HF_, W_, true>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, 0.f, common_step, noise, dbg, (float*)nullptr, (float*)nullptr, *sq)
#define GRX_LAUNCH_DBGQ_TM(W_) hipLaunchKernelGGL((grx_step_kernel_trimesh<W_, true>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, 0.f, common_step, noise, dbg, (float*)nullptr, (float*)nullptr, *sq)
    if (heightfield == 1) { if (waves == 8) GRX_LAUNCH_DBGQ(true, 8); else GRX_LAUNCH_DBGQ(true, 4); }
    else if (heightfield == 0) { if (waves == 8) GRX_LAUNCH_DBGQ(false, 8); else GRX_LAUNCH_DBGQ(false, 4); }
    else { if (waves == 8) GRX_LAUNCH_DBGQ_TM(8); else GRX_LAUNCH_DBGQ_TM(4); }
#undef GRX_LAUNCH_DBGQ
#undef GRX_LAUNCH_DBGQ_TM
}
#else
#ifndef GRX_TREE16_TU   // (grx_tree16.hip: only the tree kernel's launchers below)
// extras["episode"] (legged_robot.py:420-428) ON DEMAND: the reduction stats_fold_previous would do in the handle's next launch,
// for the launch `seq`, now (grx_flush_stats / grx_episode_stats; the generic-tree kernel's step still ends with it).  The next
// launch repeats it with the same result.  ONE block, a wave per statistics row (round-robin); its ticket store comes after
// the block's barrier, i.e. after every row has been published.
constexpr int kFinalizeWaves = 16;
__global__ __launch_bounds__(64 * kFinalizeWaves) void grx_finalize_stats(const KParams* __restrict__ Pg, long long seq, long long* progress, long long ticket) {
    KP P = GRX_PARAMS(Pg);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nb = P.stat_nblocks[seq & 1];
    for (int t = wave; t < NSTAT; t += kFinalizeWaves) {
        float cnt, s;
        stat_reduce(P, seq, t, nb, lane, cnt, s);
        if (lane == 0) stat_publish(P, seq, t, cnt, s);
    }
    __syncthreads();
    if (threadIdx.x == 0 && progress) __hip_atomic_store(progress, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the ticket alone (grx_wait_idle: everything enqueued before it on the stream has finished when it runs)
__global__ void grx_ticket_kernel(long long* progress, long long ticket) {
    __hip_atomic_store(progress, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// BaseTask.reset() first half (base_task.py:117-119): reset_idx(all envs), no step
// mask != nullptr: LeggedRobot.reset_idx(env_ids) (legged_robot.py:377-440) for the envs flagged in mask[N] (grx_reset_idx), with
// the curriculum move of an initialised env; the flags are consumed
__global__ __launch_bounds__(64) void grx_reset_all_kernel(const KParams* __restrict__ Pg, uint32_t step, const StepSeq sq, uint8_t* __restrict__ mask) {
    KP P = GRX_PARAMS(Pg);
    stats_fold_previous(P, sq, threadIdx.x);
    __shared__ SideConst sc[2];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(P.tables->side);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sc);
        for (int i = threadIdx.x; i < (int)(2 * sizeof(SideConst) / 4); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const int N = P.N;
    const int lane = threadIdx.x, el = lane >> 1, side = lane & 1;
    const int e_raw = blockIdx.x * EPB + el;
    const bool act = e_raw < N;
    const int e = act ? e_raw : N - 1;
    const bool sel = act && (!mask || mask[e] != 0);   // this env resets
    const bool writer = sel && side == 0;
    const SideConst& C = sc[side];
    const uint32_t genv = (uint32_t)(P.env_offset + e);
    const int j0 = side * LEG;
    // extras["episode"]: every env is "finished" (legged_robot.py:420-424)
    for (int t = 0; t < NT; ++t) {
        float contrib = writer ? P.episode_sums[(size_t)t * N + e] : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) contrib += __shfl_xor(contrib, off);
        if (lane == 0) stat_row(P, sq.seq, t)[blockIdx.x] = contrib;
    }
    LaneState st;
    EnvAux ea;
    st.pos = v3(P.root[e], P.root[(size_t)N + e], P.root[2 * (size_t)N + e]);
    ea.cmd[0] = P.commands[e]; ea.cmd[1] = P.commands[(size_t)N + e]; ea.cmd[2] = P.commands[2 * (size_t)N + e];
    ea.origin[0] = P.origins[e]; ea.origin[1] = P.origins[(size_t)N + e]; ea.origin[2] = P.origins[2 * (size_t)N + e];
    ea.level = P.levels[e]; ea.type = P.types[e];
    const int level_before = ea.level;
    reset_env(P, C, side, genv, step, mask != nullptr, st, ea, reset_rand(P, genv, step, side));
    {   // statistics rows NT (episodes that ended) and NT + 1 (terrain levels after the curriculum moves, legged_robot.py:427-428)
        const unsigned long long wm = __ballot(writer);
        const float ls = level_sum(sel ? ea.level : level_before, act && side == 0);
        if (lane == 0) {
            stat_row(P, sq.seq, NT)[blockIdx.x] = (float)__popcll(wm);
            stat_row(P, sq.seq, NT + 1)[blockIdx.x] = ls;
            if (blockIdx.x == 0) P.stat_nblocks[sq.seq & 1] = (int)gridDim.x;
        }
    }
    if (!sel) return;
    if (P.stash_pre_reset) {   // on-demand tensors (grx_refresh) show the state before the reset
#pragma unroll
        for (int k = 0; k < LEG; ++k) { const size_t o = (size_t)(j0 + k) * N + e; P.pre_q[o] = P.q[o]; P.pre_qd[o] = P.qd[o]; }
        if (side == 0)
            for (int i = 0; i < 13; ++i) P.pre_root[(size_t)i * N + e] = P.root[(size_t)i * N + e];
    }
    if (mask && side == 0) {
        mask[e] = 0;
        P.origins[e] = ea.origin[0]; P.origins[(size_t)N + e] = ea.origin[1]; P.origins[2 * (size_t)N + e] = ea.origin[2];
        P.levels[e] = ea.level;
    }
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        size_t o = (size_t)(j0 + k) * N + e;
        P.q[o] = st.q[k]; P.qd[o] = 0.f; P.last_actions[o] = 0.f; P.last_dof_vel[o] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) P.anchors[(size_t)((side * 4 + i) * 3 + 2) * N + e] = 0.f;
    P.air_time[(size_t)side * N + e] = 0.f; P.land_time[(size_t)side * N + e] = 0.f;
    P.feet_contact[(size_t)side * N + e] = 0;
    if (side == 0) {
        float rs[13] = {st.pos.x, st.pos.y, st.pos.z, st.qx, st.qy, st.qz, st.qw, st.vel.x, st.vel.y, st.vel.z, st.ang.x, st.ang.y, st.ang.z};
#pragma unroll
        for (int i = 0; i < 13; ++i) P.root[(size_t)i * N + e] = rs[i];
        P.commands[e] = ea.cmd[0]; P.commands[(size_t)N + e] = ea.cmd[1]; P.commands[2 * (size_t)N + e] = ea.cmd[2];
        P.ep_len[e] = 0;
        P.reset[e] = 1;
        for (int t = 0; t < NT; ++t) P.episode_sums[(size_t)t * N + e] = 0.f;
    }
}

// set_dof_state_tensor / set_actor_root_state_tensor (legged_robot.py:737, 796): AoS rows -> SoA state; env_ids != nullptr: the
// _indexed variants (legged_robot.py:737-740, 782-784) -- rows env_ids[0..n) of the same full-size buffers
__global__ void grx_set_state_kernel(const KParams* __restrict__ Pg, const float* __restrict__ root, const float* __restrict__ q,
                                     const float* __restrict__ qd, const int32_t* __restrict__ env_ids, int n) {
    KP P = GRX_PARAMS(Pg);
    const int i_ = blockIdx.x * blockDim.x + threadIdx.x;
    if (i_ >= (env_ids ? n : P.N)) return;
    const int e = env_ids ? env_ids[i_] : i_;
    if (e < 0 || e >= P.N) return;
    size_t N = P.N;
    if (root) {
        for (int i = 0; i < 13; ++i) P.root[i * N + e] = root[(size_t)e * 13 + i];
        const float* qp = root + (size_t)e * 13 + 3;
        float n_ = sqrtf(qp[0] * qp[0] + qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
        for (int i = 0; i < 4; ++i) P.root[(3 + i) * N + e] = qp[i] / n_;
    }
    const int nd = P.nd;
    if (q) for (int j = 0; j < nd; ++j) P.q[j * N + e] = q[(size_t)e * nd + j];
    if (qd) for (int j = 0; j < nd; ++j) P.qd[j * N + e] = qd[(size_t)e * nd + j];
    for (int i = 0; i < 8; ++i) P.anchors[(size_t)(i * 3 + 2) * N + e] = 0.f;
    // grx_refresh shows the state "as of the last launch" -- this one: whatever an earlier step stashed for this env (its state before a reset,
    // the base velocity before a push: refresh_source) is replaced by the state just written
    if (P.stash_pre_reset) {
        for (int i = 0; i < 13; ++i) P.pre_root[i * N + e] = P.root[i * N + e];
        for (int j = 0; j < nd; ++j) { P.pre_q[j * N + e] = P.q[j * N + e]; P.pre_qd[j * N + e] = P.qd[j * N + e]; }
        P.pre_push_vel[e] = P.root[7 * N + e]; P.pre_push_vel[N + e] = P.root[8 * N + e];
    }
}
// grx_reset_idx: flag the listed envs for the masked reset kernel
__global__ void grx_mark_kernel(const int32_t* __restrict__ env_ids, int n, int N, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int e = env_ids[i]; if (e >= 0 && e < N) mask[e] = 1; }
}

// ---- grx_refresh (include/grx.h): tensors published ON DEMAND, from the state the last step left -- for the envs that step reset, from
// the state it stashed before the reset (pre_*: the reference's tensors show that state).  The role of gym.refresh_rigid_body_state_tensor
// (legged_robot_fftai.py:76) for callers that read the tensor now and then instead of paying 1.9 KB per env-step for it.
struct RefreshState { const float *q, *qd, *root; };
GRX_DEV RefreshState refresh_source(KP P, int e) {
    const bool pre = P.stash_pre_reset && P.reset[e] != 0;
    RefreshState r = {pre ? P.pre_q : P.q, pre ? P.pre_qd : P.qd, pre ? P.pre_root : P.root};
    return r;
}
// GRX_T_MEASURED_HEIGHTS: one thread per (scan point, env); the step kernels' own height_sample (same translation unit, same flags)
__global__ __launch_bounds__(256) void grx_refresh_heights_kernel(const KParams* __restrict__ Pg) {
    KP P = GRX_PARAMS(Pg);
    const size_t N = (size_t)P.N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * (size_t)P.nh) return;
    const int e = (int)(t % N), k = (int)(t / N);
    float h = 0.f;
    if (P.terrain_type != GRX_TERRAIN_PLANE && P.measure_heights) {
        const RefreshState src = refresh_source(P, e);
        const float qz = src.root[5 * N + e], qw = src.root[6 * N + e];
        const float yaw_n = fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);   // normalize(): torch_utils.py:43-45 (as the step kernels spell it)
        const float yaw_z = qz / yaw_n, yaw_w = qw / yaw_n;
        h = height_sample(P, *P.tables, yaw_z, yaw_w, v3(src.root[e], src.root[N + e], src.root[2 * N + e]), k);
    }
    P.heights[(size_t)k * N + e] = h;
}
// GRX_T_RIGID_BODY_STATES: one thread per (URDF link, env) walks the joints from the base down to the link's body -- any grx_model (the
// arithmetic of the tree kernel's link frames, grx_tree.h: rotation matrices, orientation in the oracle's largest-component form)
__global__ __launch_bounds__(256) void grx_refresh_rbs_kernel(const KParams* __restrict__ Pg, int pushed) {   // pushed: the last step overwrote the base's vx, vy (_push_robots) after the sub-steps
    KP P = GRX_PARAMS(Pg);
    const size_t N = (size_t)P.N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * (size_t)P.num_links) return;
    const int e = (int)(t % N), l = (int)(t / N);   // (a wave: one link, 64 consecutive envs -- uniform control flow, coalesced columns)
    const RefreshTab& T = *P.refresh_tab;
    const LinkTab& LT = *P.link_tab;
    const RefreshState src = refresh_source(P, e);
    const float* r0 = src.root + e;
    const V3 pos = v3(r0[0], r0[N], r0[2 * N]);
    R3 R = quat_to_R(r0[3 * N], r0[4 * N], r0[5 * N], r0[6 * N]);
    V3 rho = v3(0.f, 0.f, 0.f), v = v3(r0[7 * N], r0[8 * N], r0[9 * N]), w = v3(r0[10 * N], r0[11 * N], r0[12 * N]);
    if (pushed && P.stash_pre_reset) { v.x = P.pre_push_vel[e]; v.y = P.pre_push_vel[N + e]; }
    const int b_link = LT.body[l];
    const int depth = b_link > 0 ? T.depth[b_link] : 0;
    for (int d = 0; d < depth; ++d) {
        const int b = T.path[b_link][d], j = b - 1;
        const float qj = src.q[(size_t)j * N + e], qdj = src.qd[(size_t)j * N + e];
        rho = rho + rot(R, v3(T.jpos[b][0], T.jpos[b][1], T.jpos[b][2]));
        R3 J = R;
        if (!T.rot0_identity[b]) {
            J.cx = rot(R, v3(T.rot0[b][0], T.rot0[b][3], T.rot0[b][6]));
            J.cy = rot(R, v3(T.rot0[b][1], T.rot0[b][4], T.rot0[b][7]));
            J.cz = rot(R, v3(T.rot0[b][2], T.rot0[b][5], T.rot0[b][8]));
        }
        float sn, cs;
        grx_sincos(qj, sn, cs);
        const float ax = T.axis[b][0], ay = T.axis[b][1], az = T.axis[b][2], oc = 1.f - cs;
        const V3 qx_ = v3(cs + ax * ax * oc, az * sn + ax * ay * oc, -ay * sn + ax * az * oc);
        const V3 qy_ = v3(-az * sn + ax * ay * oc, cs + ay * ay * oc, ax * sn + ay * az * oc);
        const V3 qz_ = v3(ay * sn + ax * az * oc, -ax * sn + ay * az * oc, cs + az * az * oc);
        R.cx = rot(J, qx_); R.cy = rot(J, qy_); R.cz = rot(J, qz_);
        const V3 a = rot(R, v3(ax, ay, az));
        w = fma3(a, qdj, w); v = fma3(cross(rho, a), qdj, v);   // (world axes about the base origin: a body's v is the velocity of its point at O)
    }
    const V3 r_ = rho + rot(R, v3(LT.pos[l][0], LT.pos[l][1], LT.pos[l][2]));
    const V3 vl = v + cross(w, r_);
    float m[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const V3 col = rot(R, v3(LT.rot[l][k], LT.rot[l][3 + k], LT.rot[l][6 + k]));
        m[k] = col.x; m[3 + k] = col.y; m[6 + k] = col.z;
    }
    float qx, qy, qz, qw;   // largest-component form (the oracle's m3_to_quat)
    const float t0 = 1 + m[0] - m[4] - m[8], t1 = 1 - m[0] + m[4] - m[8], t2 = 1 - m[0] - m[4] + m[8], t3 = 1 + m[0] + m[4] + m[8];
    if (t3 >= t0 && t3 >= t1 && t3 >= t2) { qx = m[7] - m[5]; qy = m[2] - m[6]; qz = m[3] - m[1]; qw = t3; }
    else if (t0 >= t1 && t0 >= t2) { qx = t0; qy = m[1] + m[3]; qz = m[2] + m[6]; qw = m[7] - m[5]; }
    else if (t1 >= t2) { qx = m[1] + m[3]; qy = t1; qz = m[5] + m[7]; qw = m[2] - m[6]; }
    else { qx = m[2] + m[6]; qy = m[5] + m[7]; qz = t2; qw = m[3] - m[1]; }
    const float qn = grx_rsq(qx * qx + qy * qy + qz * qz + qw * qw);
    float* o_ = P.rbs + (size_t)(l * 13) * N + e;
    o_[0] = pos.x + r_.x; o_[N] = pos.y + r_.y; o_[2 * N] = pos.z + r_.z;
    o_[3 * N] = qx * qn; o_[4 * N] = qy * qn; o_[5 * N] = qz * qn; o_[6 * N] = qw * qn;
    o_[7 * N] = vl.x; o_[8 * N] = vl.y; o_[9 * N] = vl.z;
    o_[10 * N] = w.x; o_[11 * N] = w.y; o_[12 * N] = w.z;
}
// TEST-ONLY (grx_debug_terrain): the physics terrain query of the step kernels -- height and gradient of the surface under (x, y) -- at n points
__global__ void grx_debug_terrain_kernel(const KParams* __restrict__ Pg, const float* __restrict__ xy, int n, float* __restrict__ out) {
    KP P = GRX_PARAMS(Pg);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gx = 0.f, gy = 0.f, h = 0.f;
    if (P.terrain_type != GRX_TERRAIN_PLANE) h = P.vertical_faces ? terrain_height<GRX_HF_TRIMESH>(P, xy[2 * i], xy[2 * i + 1], gx, gy) : terrain_height<GRX_HF_RASTER>(P, xy[2 * i], xy[2 * i + 1], gx, gy);
    out[3 * i] = h; out[3 * i + 1] = gx; out[3 * i + 2] = gy;
}
extern "C" void grx_launch_debug_terrain(const KParams* dP, const float* xy, int n, float* out, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(grx_debug_terrain_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dP, xy, n, out);
}
// TEST-ONLY (grx_debug_wall): mesh_type 'trimesh', the contact of a sphere AT REST (centre x y z, radius r) with the vertical faces next to it, as the step
// kernels compute it (wall_gather / wall_contact): out = force / kn = overlap times the unit direction from the face to the centre (0 0 0: none)
__global__ void grx_debug_wall_kernel(const KParams* __restrict__ Pg, const float* __restrict__ xyzr, int n, float* __restrict__ out) {
    KP P = GRX_PARAMS(Pg);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 F = v3(0.f, 0.f, 0.f);
    if (P.terrain_type != GRX_TERRAIN_PLANE && P.vertical_faces) {
        float tx, ty;
        const uint4 ww = wall_gather(P, xyzr[4 * i], xyzr[4 * i + 1], tx, ty);
        F = wall_contact(P, ww, tx, ty, xyzr[4 * i + 2], xyzr[4 * i + 3], 0.0f, v3(0.f, 0.f, 0.f), 0.0f) * (1.0f / P.kn);
    }
    out[3 * i] = F.x; out[3 * i + 1] = F.y; out[3 * i + 2] = F.z;
}
extern "C" void grx_launch_debug_wall(const KParams* dP, const float* xyzr, int n, float* out, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(grx_debug_wall_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dP, xyzr, n, out);
}
extern "C" void grx_launch_refresh_heights(const KParams* dP, int N, int nh, hipStream_t stream) {
    const size_t n = (size_t)N * (size_t)nh;
    if (n) hipLaunchKernelGGL(grx_refresh_heights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dP);
}
extern "C" void grx_launch_refresh_rbs(const KParams* dP, int N, int nlinks, int pushed, hipStream_t stream) {
    const size_t n = (size_t)N * (size_t)nlinks;
    if (n) hipLaunchKernelGGL(grx_refresh_rbs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dP, pushed);
}

// host-callable launchers (grx_capi.cpp is compiled by hipcc too; kept separate for readability)
// waves: waves per 32-env block (1, 2 or 4; grx_capi.cpp picks the largest that still gives every wave its own SIMD)
extern "C" void grx_launch_step(const KParams* dP, int N, int heightfield, int waves, const float* actions, float delay, long long common_step,
                                const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream) {
    int nblocks = (N + EPB - 1) / EPB;
#define GRX_LAUNCH_STEP(HF_, W_) hipLaunchKernelGGL((grx_step_kernel<HF_, W_>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, delay, common_step, noise, (const float*)nullptr, obs_out, pri_out, *sq)
#define GRX_LAUNCH_STEP_TM(W_) hipLaunchKernelGGL((grx_step_kernel_trimesh<W_>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, delay, common_step, noise, (const float*)nullptr, obs_out, pri_out, *sq)
    if (heightfield == 1) { if (waves == 8) GRX_LAUNCH_STEP(true, 8); else if (waves == 4) GRX_LAUNCH_STEP(true, 4); else if (waves == 2) GRX_LAUNCH_STEP(true, 2); else GRX_LAUNCH_STEP(true, 1); }   // (heightfield: 0 plane, 1 raster, 2 trimesh)
    else if (heightfield == 0) { if (waves == 8) GRX_LAUNCH_STEP(false, 8); else if (waves == 4) GRX_LAUNCH_STEP(false, 4); else if (waves == 2) GRX_LAUNCH_STEP(false, 2); else GRX_LAUNCH_STEP(false, 1); }
    else { if (waves == 8) GRX_LAUNCH_STEP_TM(8); else if (waves == 4) GRX_LAUNCH_STEP_TM(4); else if (waves == 2) GRX_LAUNCH_STEP_TM(2); else GRX_LAUNCH_STEP_TM(1); }
#undef GRX_LAUNCH_STEP
#undef GRX_LAUNCH_STEP_TM
}
// TEST-ONLY (grx_debug_post_physics): the post-physics half of the step on injected state, in the layout the handle steps with
// (waves = 1, 4, 8; the two-wave layout shares the one-wave kernel's post-physics code and is served by it)
extern "C" void grx_launch_step_debug(const KParams* dP, int N, int heightfield, int waves, const float* actions, long long common_step, const float* noise,
                                      const float* dbg, const StepSeq* sq, hipStream_t stream) {
    int nblocks = (N + EPB - 1) / EPB;
#define GRX_LAUNCH_DBG(HF_, W_) hipLaunchKernelGGL((grx_step_kernel<HF_, W_, true>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, 0.f, common_step, noise, dbg, (float*)nullptr, (float*)nullptr, *sq)
#define GRX_LAUNCH_DBG_TM(W_) hipLaunchKernelGGL((grx_step_kernel_trimesh<W_, true>), dim3(nblocks), dim3(64 * W_), 0, stream, dP, actions, 0.f, common_step, noise, dbg, (float*)nullptr, (float*)nullptr, *sq)
    if (heightfield == 1) { if (waves == 8) GRX_LAUNCH_DBG(true, 8); else if (waves == 4) GRX_LAUNCH_DBG(true, 4); else GRX_LAUNCH_DBG(true, 1); }
    else if (heightfield == 0) { if (waves == 8) GRX_LAUNCH_DBG(false, 8); else if (waves == 4) GRX_LAUNCH_DBG(false, 4); else GRX_LAUNCH_DBG(false, 1); }
    else { if (waves == 8) GRX_LAUNCH_DBG_TM(8); else if (waves == 4) GRX_LAUNCH_DBG_TM(4); else GRX_LAUNCH_DBG_TM(1); }
#undef GRX_LAUNCH_DBG
#undef GRX_LAUNCH_DBG_TM
}
extern "C" int grx_debug_rows(void) { return DBG_ROWS; }
extern "C" int grx_debug_row_of(int what) { return what == 0 ? (int)DBG_TORQUES : what == 1 ? (int)DBG_LAST_LAST_ACTIONS : what == 2 ? (int)DBG_TERM_CONTACT : (int)DBG_APPLY_RESET; }
// epb: envs per block (= threads per block, at most 64); lds_bytes > 0: the per-body workspace lives in (dynamic) LDS
extern "C" int grx_launch_step_generic(const KParams* dP, const void* tables, float* ws, int N, int epb, int lds_bytes, int heightfield,
                                       const float* actions, float delay, long long common_step, const float* noise, float* obs_out, float* pri_out,
                                       long long seq, hipStream_t stream) {
    const int nblocks = (N + epb - 1) / epb;
    const GenTables* T = static_cast<const GenTables*>(tables);
    if (lds_bytes > 0) {
        static bool raised = false;
        if (!raised) {   // > 64 KB of dynamic LDS needs the opt-in
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_generic<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_generic<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_generic_trimesh), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
            raised = true;
        }
        ws = nullptr;
    }
    if (heightfield == 2) hipLaunchKernelGGL(grx_step_generic_trimesh, dim3(nblocks), dim3(epb), lds_bytes, stream, dP, T, ws, actions, delay, common_step, noise, obs_out, pri_out, seq);
    else if (heightfield) hipLaunchKernelGGL(grx_step_generic<true>, dim3(nblocks), dim3(epb), lds_bytes, stream, dP, T, ws, actions, delay, common_step, noise, obs_out, pri_out, seq);
    else hipLaunchKernelGGL(grx_step_generic<false>, dim3(nblocks), dim3(epb), lds_bytes, stream, dP, T, ws, actions, delay, common_step, noise, obs_out, pri_out, seq);
    return 0;
}
extern "C" void grx_launch_reset_all_generic(const KParams* dP, const void* tables, int N, int epb, uint32_t step, long long seq, uint8_t* mask, hipStream_t stream) {
    hipLaunchKernelGGL(grx_reset_all_generic, dim3((N + epb - 1) / epb), dim3(epb), 0, stream, dP, static_cast<const GenTables*>(tables), step, seq, mask);
}
#endif   // !GRX_TREE16_TU
// the tree kernel (grx_tree.h): 8 lanes per env, two or four 8-env waves per block; grx_tree16.hip: the same with 16 lanes per env (names + "16")
#ifdef GRX_TREE16_TU
#define GRX_TREE_FN(n) n##16
#else
#define GRX_TREE_FN(n) n
#endif
extern "C" int GRX_TREE_FN(grx_tree_lds_bytes)(int nb, int nlc, int nchain, int nsph, int waves) { return (int)sizeof(TreeTab) + waves * 2 * tree_half_words(tree_offsets(nb, nlc, nchain, nsph).total) * 4; }
extern "C" int GRX_TREE_FN(grx_tree_envs_per_wave)(void) { return TEPW; }
extern "C" int GRX_TREE_FN(grx_launch_step_tree)(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions, float delay,
                                    long long common_step, const float* noise, float* obs_out, float* pri_out, const StepSeq* sq, hipStream_t stream) {
    static bool raised = false;
    if (!raised) {   // > 64 KB of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree_trimesh<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        raised = true;
    }
    const int nblocks = (N + TEPW * waves - 1) / (TEPW * waves);
    const TreeTab* Tt = static_cast<const TreeTab*>(tree_tab);
    const GenTables* Tg = static_cast<const GenTables*>(gen_tab);
    if (heightfield == 2) hipLaunchKernelGGL((grx_step_tree_trimesh<false>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, delay, common_step, noise, obs_out, pri_out, *sq, (const float*)nullptr);
    else if (heightfield) hipLaunchKernelGGL((grx_step_tree<true, false>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, delay, common_step, noise, obs_out, pri_out, *sq, (const float*)nullptr);
    else hipLaunchKernelGGL((grx_step_tree<false, false>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, delay, common_step, noise, obs_out, pri_out, *sq, (const float*)nullptr);
    return 0;
}
// TEST-ONLY (grx_debug_post_physics): the post-physics half of the tree kernel on injected state (a 10-dof model forced through it)
extern "C" int GRX_TREE_FN(grx_launch_step_tree_debug)(const KParams* dP, const void* tree_tab, const void* gen_tab, int N, int waves, int lds_bytes, int heightfield, const float* actions,
                                          long long common_step, const float* noise, const float* dbg, const StepSeq* sq, hipStream_t stream) {
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grx_step_tree_trimesh<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) return -1;
        raised = true;
    }
    const int nblocks = (N + TEPW * waves - 1) / (TEPW * waves);
    const TreeTab* Tt = static_cast<const TreeTab*>(tree_tab);
    const GenTables* Tg = static_cast<const GenTables*>(gen_tab);
    if (heightfield == 2) hipLaunchKernelGGL((grx_step_tree_trimesh<true>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, 0.f, common_step, noise, (float*)nullptr, (float*)nullptr, *sq, dbg);
    else if (heightfield) hipLaunchKernelGGL((grx_step_tree<true, true>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, 0.f, common_step, noise, (float*)nullptr, (float*)nullptr, *sq, dbg);
    else hipLaunchKernelGGL((grx_step_tree<false, true>), dim3(nblocks), dim3(64 * waves), lds_bytes, stream, dP, Tt, Tg, actions, 0.f, common_step, noise, (float*)nullptr, (float*)nullptr, *sq, dbg);
    return 0;
}
#ifndef GRX_TREE16_TU
extern "C" int grx_generic_tables_size(void) { return (int)sizeof(GenTables); }
extern "C" int grx_generic_ws_floats_per_env(int nb, int nlc) { return nb * WSB + 3 * nlc; }
// the statistics of launch `seq` now (grx_flush_stats; the generic path after every step) + optionally a ticket
extern "C" void grx_launch_finalize(const KParams* dP, long long seq, long long* progress, long long ticket, hipStream_t stream) {
    hipLaunchKernelGGL(grx_finalize_stats, dim3(1), dim3(64 * kFinalizeWaves), 0, stream, dP, seq, progress, ticket);
}
extern "C" void grx_launch_ticket(long long* progress, long long ticket, hipStream_t stream) {
    hipLaunchKernelGGL(grx_ticket_kernel, dim3(1), dim3(1), 0, stream, progress, ticket);
}
// mask: nullptr = every env (BaseTask.reset()); else the envs flagged there (reset_idx(env_ids))
extern "C" void grx_launch_reset_all(const KParams* dP, int N, uint32_t step, const StepSeq* sq, uint8_t* mask, hipStream_t stream) {
    int nblocks = (N + EPB - 1) / EPB;
    hipLaunchKernelGGL(grx_reset_all_kernel, dim3(nblocks), dim3(64), 0, stream, dP, step, *sq, mask);
}
extern "C" void grx_launch_mark(const int32_t* env_ids, int n, int N, uint8_t* mask, hipStream_t stream) {
    hipLaunchKernelGGL(grx_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, env_ids, n, N, mask);
}
// env_ids: nullptr = all N envs, else the n listed rows
extern "C" void grx_launch_set_state(const KParams* dP, int N, const float* root, const float* q, const float* qd, const int32_t* env_ids, int n, hipStream_t stream) {
    const int cnt = env_ids ? n : N;
    hipLaunchKernelGGL(grx_set_state_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, dP, root, q, qd, env_ids, n);
}
extern "C" int grx_envs_per_block(void) { return EPB; }
#endif   // !GRX_TREE16_TU
#endif   // GRX_QUAD_TU
