// grx_rare.h -- lane-compacted evaluation of the SELDOM-TOUCHING collision shapes (included by grx_kernels.hip inside
// its anonymous namespace, after sphere_probe / sphere_contact).
//
// Base-lump shapes (torso, head, arms: table slots 0..7 of a lane) and the thigh / shank shapes (slots 8..11) are out
// of the terrain's reach for an upright robot.  The round-1 kernel looped over a lane's shapes whenever ANY of the
// wave's 32 envs had one within reach: one fallen robot made all 64 lanes walk through 8 + 2 + 2 full contact
// evaluations (7 k cycles per sub-step on the base-lump wave -- the heaviest item of the sub-step).
// Here the (lane, shape) pairs that pass the reach test are COMPACTED over the wave: every pair gets a slot in an LDS
// list (ballot + mbcnt prefix, shape-major), the wave evaluates the list 64 pairs at a time -- a worker lane gathers the
// carrying body's frame and the shape constants of the pair it drew from LDS -- and hands the force and its moment
// about the base origin back through an LDS table indexed [shape][owner lane], from which every owner lane sums its
// own shapes in table order.  The arithmetic of a pair does not depend on the lane that evaluates it and the owner's
// summation order is fixed, so results stay bit-identical across reruns and across shards.
#pragma once

#ifdef GRX_PROFILE_SECTIONS   // cycles per phase of rare_contacts, accumulated over the policy step (slots 32..37 of the block's row)
#define GRX_RARE_T(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); rare_acc[i] += t_ - rare_t; rare_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define GRX_RARE_T0 long long rare_t = clock64()
__device__ long long rare_dummy_acc[8];
#else
#define GRX_RARE_T(i) do {} while (0)
#define GRX_RARE_T0 do {} while (0)
#endif

constexpr int RC_NS = 12;          // candidate table slots per lane: 0..7 base lump, 8..9 thigh (chain body 2), 10..11 shank (body 3)
constexpr int RC_FR4 = 5;          // frame, in float4 units: (R.cx, R.cy.x) (R.cy.yz, R.cz.xy) (R.cz.z, rho) (w, v.x) (v.yz, hmax*, -)   *base frame only
constexpr int RC_LIST_BYTES = RC_NS * 64 * 2;
constexpr int RC_RES_BYTES = RC_NS * 3 * 64 * 8;      // [shape][3][lane] float2: (F.x F.y) (F.z T.x) (T.y T.z)
constexpr int RC_FBASE_BYTES = (RC_FR4 + 1) * EPB * 16;   // per env: base frame + (O, mu)
constexpr int RC_FCHAIN_BYTES = 2 * RC_FR4 * 64 * 16; // per lane: thigh, shank frames
constexpr int RC_BYTES = RC_LIST_BYTES + RC_RES_BYTES + RC_FBASE_BYTES + RC_FCHAIN_BYTES;

struct RareBuf {
    uint16_t* list;   // [RC_NS * 64]  lane | shape << 6, shape-major
    float2* res;      // [RC_NS][3][64]
    float4* fbase;    // [RC_FR4 + 1][EPB]
    float4* fchain;   // [2][RC_FR4][64]
};
GRX_DEV RareBuf rare_carve(char* p) {
    RareBuf b;
    b.res = reinterpret_cast<float2*>(p);
    b.fbase = reinterpret_cast<float4*>(p + RC_RES_BYTES);
    b.fchain = reinterpret_cast<float4*>(p + RC_RES_BYTES + RC_FBASE_BYTES);
    b.list = reinterpret_cast<uint16_t*>(p + RC_RES_BYTES + RC_FBASE_BYTES + RC_FCHAIN_BYTES);
    return b;
}

GRX_DEV float4 rc4(float a, float b, float c, float d) { float4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }
GRX_DEV void rare_store_frame(float4* f, int st, const R3& R, V3 rho, V3 w, V3 v, float extra = 0.f) {
    f[0 * st] = rc4(R.cx.x, R.cx.y, R.cx.z, R.cy.x);
    f[1 * st] = rc4(R.cy.y, R.cy.z, R.cz.x, R.cz.y);
    f[2 * st] = rc4(R.cz.z, rho.x, rho.y, rho.z);
    f[3 * st] = rc4(w.x, w.y, w.z, v.x);
    f[4 * st] = rc4(v.y, v.z, extra, 0.f);
}
struct RareFrame { R3 R; V3 rho, w, v; float extra; };
GRX_DEV RareFrame rare_load_frame(const float4* f, int st) {
    const float4 a = f[0 * st], b = f[1 * st], c = f[2 * st], d = f[3 * st], e = f[4 * st];
    RareFrame F;
    F.R.cx = v3(a.x, a.y, a.z); F.R.cy = v3(a.w, b.x, b.y); F.R.cz = v3(b.z, b.w, c.x);
    F.rho = v3(c.y, c.z, c.w); F.w = v3(d.x, d.y, d.z); F.v = v3(d.w, e.x, e.y); F.extra = e.z;
    return F;
}

// what a lane gets back: the wrenches (about the base origin O) of its shapes, per carrying body, and the per-link flags
struct RareOut {
    V3 f0a, f0l;        // base-lump shapes
    V3 fa2, fl2;        // thigh shapes (chain body 2)
    V3 fa3, fl3;        // shank shapes (chain body 3)
    V3 lf[8];           // net force of the base-lump URDF link that ends at table slot i (slots with link_last set)
    bool term; float pen_count;
};

struct RareNoWait { GRX_DEV void operator()() const {} };

// Shapes [S0, S1) of every lane.  R0 / O / ang / vel: base frame (used when S0 < 8); K2in, K3in: thigh / shank frames (used
// when S1 > 8).  FRAMES_IN_LDS: another wave publishes the thigh / shank frames in B.fchain (grx_wavepipe.h: wave 2's
// walk); wait_frames() is called once the base-lump shapes have been tested, right before the frames are read.
// Must be called by all 64 lanes in wave-uniform control flow.
// OWN_POS (with FRAMES_IN_LDS): the caller walked the chain itself for R, rho of K2in / K3in (the reach tests need no velocities), so
// the tests start before the other wave's frames are out; wait_frames() is then called right before the evaluation reads them.
template <int HF, int S0, int S1, bool FRAMES_IN_LDS = false, bool OWN_POS = false, class WaitFrames = RareNoWait>
GRX_DEV void rare_contacts(KP P, const KTables& T, const SideConst& C, const RareBuf& B, int lane, int el, int side, const R3& R0, V3 O, V3 ang, V3 vel,
                           const ChainKin& K2in, const ChainKin& K3in, float mu, float hmax, RareOut& out,
                           long long* rare_acc = nullptr, WaitFrames wait_frames = WaitFrames(), const bool want_links = true) {
    const V3 zero = v3(0.f, 0.f, 0.f);
#ifdef GRX_PROFILE_SECTIONS
    long long rare_scratch[8];
    if (!rare_acc) rare_acc = rare_scratch;
#endif
    GRX_RARE_T0;
    out.f0a = zero; out.f0l = zero; out.fa2 = zero; out.fl2 = zero; out.fa3 = zero; out.fl3 = zero;
    out.term = false; out.pen_count = 0.f;
    // ---- 1. reach test of the lane's own shapes.  First against hmax, the bound of the terrain height anywhere the robot
    // can be during this policy step (plane: exactly 0): z of the centre only.  On the heightfield that bound sits up to
    // a metre above the ground under a robot on a slope or stairs, so shapes that pass it are then tested against the
    // max of the four raster corners of the cell under their centre (hf_max4: an upper bound of the bilinear height the
    // contact test itself will read there; 1e-4 m covers its rounding).
    // (R and rho of the thigh / shank frames: only their z rows and rho are needed here)
    R3 R2 = K2in.R, R3_ = K3in.R;
    V3 rho2 = K2in.rho, rho3 = K3in.rho;
    uint32_t m = 0;
    // LPE == 4: the two lanes of a leg hold the same table; each tests (and later sums) the shapes of its own parity -- shape
    // i0 + half, read from the LDS copy of the table (lane-dependent index); the carrying body is the same for both (i0 even)
    constexpr int ST = LPL;
    static_assert(ST == 1 || (S0 % 2 == 0 && S1 % 2 == 0), "shape ranges split by parity");
    const int hf_ = lane_half(lane);
    const SideConst& Ct = ST == 1 ? C : T.side[side];
    auto test_range = [&](const int a, const int b) {   // shapes [a, b): cheap test, then the exact one for the wave's survivors
        uint32_t mc = 0;
#pragma unroll
        for (int i0 = S0; i0 < S1; i0 += ST) {
            if (i0 < a || i0 >= b) continue;
            const int i = ST == 1 ? i0 : i0 + hf_;
            const R3& R = i0 < 8 ? R0 : (i0 < 10 ? R2 : R3_);
            const float rz = i0 < 8 ? 0.f : (i0 < 10 ? rho2.z : rho3.z);
            const SphC& S = Ct.sph[i];
            const float z = O.z + rz + fmaf(R.cx.z, S.x, fmaf(R.cy.z, S.y, R.cz.z * S.z));
            if (z - S.r <= hmax) mc |= 1u << i;
        }
        if (HF && __any(mc != 0u)) {
            float zb[S1 - S0];   // bottom of the shape minus the local bound
#pragma unroll
            for (int i0 = S0; i0 < S1; i0 += ST) {   // all loads of the wave in flight together
                if (i0 < a || i0 >= b) continue;
                const int i = ST == 1 ? i0 : i0 + hf_;
                const R3& R = i0 < 8 ? R0 : (i0 < 10 ? R2 : R3_);
                const V3 rho = i0 < 8 ? zero : (i0 < 10 ? rho2 : rho3);
                const SphC& S = Ct.sph[i];
                const V3 xr = rho + rot(R, v3(S.x, S.y, S.z));
                float fx = (O.x + xr.x + P.border_size) * P.inv_hscale, fy = (O.y + xr.y + P.border_size) * P.inv_hscale;
                fx = fminf(fmaxf(fx, 0.0f), (float)(P.hf_rows - 1));
                fy = fminf(fmaxf(fy, 0.0f), (float)(P.hf_cols - 1));
                const int ix = min((int)fx, P.hf_rows - 2), iy = min((int)fy, P.hf_cols - 2);
                zb[i0 - S0] = O.z + xr.z - S.r - (float)P.hf_max4[(size_t)ix * P.hf_cols + iy] * P.vertical_scale;
            }
            uint32_t m2 = 0;
#pragma unroll
            for (int i0 = S0; i0 < S1; i0 += ST) if (i0 >= a && i0 < b && zb[i0 - S0] <= 1e-4f) m2 |= 1u << (ST == 1 ? i0 : i0 + hf_);
            mc &= m2;
        }
        m |= mc;
    };
    if (S0 < 8) test_range(S0, S1 < 8 ? S1 : 8);
    GRX_RARE_T(0);
    if (S1 > 8) {
        if (FRAMES_IN_LDS && !OWN_POS) {
            wait_frames();
            const RareFrame f2 = rare_load_frame(B.fchain + lane, 64), f3 = rare_load_frame(B.fchain + RC_FR4 * 64 + lane, 64);
            R2 = f2.R; rho2 = f2.rho; R3_ = f3.R; rho3 = f3.rho;
        }
        test_range(S0 > 8 ? S0 : 8, S1);
    }
    GRX_RARE_T(1);
    V3 Fs[8];   // forces of the base-lump shapes (per-link netting below)
#pragma unroll
    for (int i = 0; i < 8; ++i) Fs[i] = zero;
    // which shapes have a candidate anywhere in the wave (wave-uniform: scalar branches below)
    uint32_t act = 0;
#pragma unroll
    for (int i = S0; i < S1; ++i) if (__any((m >> i) & 1u)) act |= 1u << i;
    if (act) {
        // ---- 2. publish the frames the workers gather, and compact the candidate pairs: lane-major slots from a
        // prefix sum of the per-lane candidate counts (bit planes of the count -> ballot + mbcnt)
        if (side == 0) {
            if (S0 < 8 && (act & 0xffu)) rare_store_frame(B.fbase + el, EPB, R0, zero, ang, vel, hmax);
            else B.fbase[4 * EPB + el] = rc4(0.f, 0.f, hmax, 0.f);
            B.fbase[RC_FR4 * EPB + el] = rc4(O.x, O.y, O.z, mu);
        }
        if (S1 > 8 && !FRAMES_IN_LDS) {
            rare_store_frame(B.fchain + lane, 64, K2in.R, K2in.rho, K2in.w, K2in.v);
            rare_store_frame(B.fchain + RC_FR4 * 64 + lane, 64, K3in.R, K3in.rho, K3in.w, K3in.v);
        }
        if (S1 > 8 && FRAMES_IN_LDS && OWN_POS) wait_frames();
        const int cnt = __popc(m);
        int pos = 0, total = 0;
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {   // cnt <= 12
            const unsigned long long bm = __ballot((cnt >> bit) & 1);
            pos += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)) << bit;
            total += __popcll(bm) << bit;
        }
        {   // own candidates, ascending shape order: walk the set bits (trip count = the wave's largest count, usually 2-6)
            uint32_t mm = m;
            while (__any(mm != 0u)) {
                if (mm) {
                    const int i = __ffs((int)mm) - 1;
                    B.list[pos] = (uint16_t)(lane | (i << 6));
                    ++pos;
                    mm &= mm - 1u;
                }
            }
        }
        GRX_RARE_T(2);
#ifdef GRX_PROFILE_SECTIONS
        rare_acc[6] += total; rare_acc[7] += 1;
#endif
        // (one wave: its LDS operations complete in program order, so lane A's stores above are visible to lane B's
        //  loads below without a barrier; the fence only keeps the compiler from reordering across it)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- 3. evaluate the list, 64 pairs per pass
        for (int p0 = 0; p0 < total; p0 += 64) {
            const int idx = p0 + lane;
            if (idx < total) {
                const int rec = B.list[idx];
                const int src = rec & 63, si = rec >> 6;
                const SphC& S = T.side[lane_side(src)].sph[si];
                const float4* fb = B.fbase + lane_env(src);
                const RareFrame Fr = si < 8 ? rare_load_frame(fb, EPB) : rare_load_frame(B.fchain + (si < 10 ? 0 : RC_FR4 * 64) + src, 64);
                const float4 om = fb[RC_FR4 * EPB];
                const float hmax_ = fb[4 * EPB].z;
                const V3 Ow = v3(om.x, om.y, om.z);
                V3 xr; TerrainAt th;
                sphere_probe<HF>(P, S, Fr.R, Fr.rho, Ow, xr, th);
                LaneState dummy;
                dummy.anchor_on = 0;
                V3 F = sphere_contact<HF, -1>(P, S, Fr.w, Fr.v, Ow, om.w, hmax_, dummy, xr, th);
                if (HF == GRX_HF_TRIMESH) {   // mesh_type 'trimesh': the vertical faces next to the shape
                    float wtx, wty;
                    const uint4 ww = wall_gather(P, Ow.x + xr.x, Ow.y + xr.y, wtx, wty);
                    F = F + wall_contact(P, ww, wtx, wty, Ow.z + xr.z, S.r, S.dmax, Fr.v + cross(Fr.w, xr), om.w);
                }
                const V3 Tq = cross(xr, F);
                float2* r = B.res + (si * 3) * 64 + src;
                float2 a; a.x = F.x; a.y = F.y; r[0] = a;
                a.x = F.z; a.y = Tq.x; r[64] = a;
                a.x = Tq.y; a.y = Tq.z; r[128] = a;
            }
        }
        GRX_RARE_T(3);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- 4. every owner lane sums its own shapes, in table order (shapes nobody has a candidate for: scalar skip)
        if (ST == 2) {   // own parity: all reads in one batch (one exposed LDS latency), unselected rows dropped by the lane's mask
            float2 ra[(S1 - S0) / 2], rb[(S1 - S0) / 2], rc[(S1 - S0) / 2];
#pragma unroll
            for (int i0 = S0; i0 < S1; i0 += 2) {
                const float2* r = B.res + ((i0 + hf_) * 3) * 64 + lane;
                ra[(i0 - S0) / 2] = r[0]; rb[(i0 - S0) / 2] = r[64]; rc[(i0 - S0) / 2] = r[128];
            }
            V3 mine[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int i0 = S0; i0 < S1; i0 += 2) {
                const bool has = (m >> (i0 + hf_)) & 1u;
                const float2 a = ra[(i0 - S0) / 2], b = rb[(i0 - S0) / 2], c = rc[(i0 - S0) / 2];
                const V3 F = v3(has ? a.x : 0.f, has ? a.y : 0.f, has ? b.x : 0.f), Tq = v3(has ? b.y : 0.f, has ? c.x : 0.f, has ? c.y : 0.f);
                if (i0 < 8) { out.f0a = out.f0a + Tq; out.f0l = out.f0l + F; mine[i0 / 2] = F; }
                else if (i0 < 10) { out.fa2 = out.fa2 + Tq; out.fl2 = out.fl2 + F; }
                else { out.fa3 = out.fa3 + Tq; out.fl3 = out.fl3 + F; }
            }
            out.f0a = half_sum(out.f0a); out.f0l = half_sum(out.f0l); out.fa2 = half_sum(out.fa2); out.fl2 = half_sum(out.fl2);
            out.fa3 = half_sum(out.fa3); out.fl3 = half_sum(out.fl3);
            if (want_links && S0 < 8 && (act & 0xffu)) {   // all eight base-lump forces in both halves, for the per-link netting
#pragma unroll
                for (int q_ = 0; q_ < 4; ++q_) {
                    const V3 oth = half_swap(mine[q_]);
                    Fs[2 * q_] = v3(hf_ ? oth.x : mine[q_].x, hf_ ? oth.y : mine[q_].y, hf_ ? oth.z : mine[q_].z);
                    Fs[2 * q_ + 1] = v3(hf_ ? mine[q_].x : oth.x, hf_ ? mine[q_].y : oth.y, hf_ ? mine[q_].z : oth.z);
                }
            }
        } else
#pragma unroll
        for (int i = S0; i < S1; ++i) {
            if (!((act >> i) & 1u)) continue;   // scalar
            if ((m >> i) & 1u) {
                const float2* r = B.res + (i * 3) * 64 + lane;
                const float2 a = r[0], b = r[64], c = r[128];
                const V3 F = v3(a.x, a.y, b.x), Tq = v3(b.y, c.x, c.y);
                if (i < 8) { out.f0a = out.f0a + Tq; out.f0l = out.f0l + F; Fs[i] = F; }
                else if (i < 10) { out.fa2 = out.fa2 + Tq; out.fl2 = out.fl2 + F; }
                else { out.fa3 = out.fa3 + Tq; out.fl3 = out.fl3 + F; }
            }
        }
        // (the reference looks at the net contact forces AFTER the last sub-step only -- check_termination and the collision
        //  term read `contact_forces` of the final state, legged_robot.py:336-353, 266 -- so the callers ask for the per-link
        //  netting on that sub-step alone: want_links, wave-uniform)
        if (want_links && S0 < 8 && (act & 0xffu)) {   // per-link netting for termination / collision
            uint32_t fl[8]; int ll[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { fl[i] = C.sph[i].flags; ll[i] = C.sph[i].link_last; }
            V3 Flink = zero;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                Flink = Flink + Fs[i];
                if (ll[i] & 1) {   // uniform per side: net force of one URDF link complete
                    Fs[i] = Flink;
                    const float n2 = dot(Flink, Flink);
                    if ((fl[i] & GRX_SPH_TERMINATE) && n2 > P.termination_force * P.termination_force) out.term = true;
                    if ((fl[i] & GRX_SPH_PENALISE) && n2 > 0.01f) out.pen_count += 1.0f;
                    Flink = zero;
                }
            }
        }
    }
    GRX_RARE_T(4);
#pragma unroll
    for (int i = 0; i < 8; ++i) out.lf[i] = Fs[i];
}
