// grx_self.h -- self-collision of the lower-limb robot (included by grx_kernels.hip inside its anonymous namespace).
//
// The reference creates its actors with self_collisions = 0, i.e. ENABLED (legged_robot_config.py:121,
// legged_robot.py:1022-1028): links that are not joined by a joint collide with each other.  Within the joint limits
// that is, for the GR1 lower-limb models (tools/self_collision_pairs.py): left-leg x right-leg shapes (thigh, shank,
// foot) and the thigh against a few base-lump shapes (base_link; GR1T2: the hands).
//
// One env = one lane pair, left leg on the even lane, right leg on the odd one: the partner's sphere centres and
// velocities are ONE DPP quad_perm exchange away (pair_swap).  Both lanes evaluate every left x right pair, each for its
// own sphere; the contact law is written symmetrically in the two spheres, so the two lanes compute bit-for-bit
// opposite forces at the same point of application: the pair is an internal force of the env by construction.
//
// Contact law (identical in oracle/grx_oracle.c contact_forces()): spheres a, b overlap by pen = ra + rb - |ca - cb| > 0;
// n = (ca - cb) / |ca - cb|;  fn = max(kn pen - min(kn pen dn, dmax_a, dmax_b) (ua - ub).n, 0);  viscous friction
// min(cv |ut|, mu fn) against the tangential relative velocity, mu = the shapes' own (per-env) friction;
// F on a, -F on b, applied at the middle of the overlap 0.5 (ca + cb) + 0.5 n (rb - ra).
#pragma once

struct SelfOut {
    V3 fa[3], fl[3];   // contact wrench about O on this lane's chain bodies 2 (thigh), 3 (shank), 4 (foot)
    V3 f0a, f0l;       // ... and on the base lump (this lane's share; the pair sum of the bias force adds both)
    V3 fbase[2];       // force on the base-lump link of bc entries 0.. (by distinct link: at most 2 links per side)
};

struct SphW { V3 c, u; float r, dmax; };   // a sphere in world axes: centre relative to O, velocity of its centre

GRX_DEV SphW sph_world(const SphC& S, const ChainKin& K) {
    SphW w;
    w.c = K.rho + rot(K.R, v3(S.x, S.y, S.z));
    w.u = K.v + cross(K.w, w.c);
    w.r = S.r; w.dmax = S.dmax;
    return w;
}
GRX_DEV SphW sph_swap(const SphW& a) {
    SphW b;
    b.c = v3(pair_swap(a.c.x), pair_swap(a.c.y), pair_swap(a.c.z));
    b.u = v3(pair_swap(a.u.x), pair_swap(a.u.y), pair_swap(a.u.z));
    b.r = pair_swap(a.r); b.dmax = pair_swap(a.dmax);
    return b;
}

// force on sphere a from sphere b and its point of application (relative to O); false: no overlap
GRX_DEV bool sphere_pair(KP P, const SphW& a, const SphW& b, float mu, V3& F, V3& pw) {
    const V3 dv = a.c - b.c;
    const float d2 = dot(dv, dv), Rs = a.r + b.r;
    if (!(d2 < Rs * Rs && d2 > 1e-12f)) return false;
    const float inv = grx_rsq(d2), dist = d2 * inv, pen = Rs - dist;
    const V3 n = dv * inv;
    const V3 ur = a.u - b.u;
    const float un = dot(ur, n);
    const float cd = fminf(P.kn * pen * P.dn, fminf(a.dmax, b.dmax));
    const float fn = fmaxf(P.kn * pen - cd * un, 0.0f);
    const V3 ut = ur - n * un;
    const float sp = grx_sqrt(dot(ut, ut));
    const float ft = fminf(P.cv * sp, mu * fn);
    F = n * fn;
    if (sp > 1e-9f) F = F - ut * (ft * grx_rcp(sp));
    pw = (a.c + b.c) * 0.5f + n * (0.5f * (b.r - a.r));
    return true;
}

// ---- leg x leg: compacted over the wave ------------------------------------------------------------------------------
// Legs rarely touch, but with 32 envs per wave SOME env's legs usually are close: evaluated lane by lane, the 52 sphere
// pairs of a leg pair would run on all 64 lanes whenever one env needs them (measured: 7 k cycles per sub-step, +60 % on
// the launch).  Instead:
//   policy step:  bounding spheres of the shape-carrying bodies, with a margin for the motion over the decimation
//                 -> SelfNear (per env: can the legs meet at all; which base-lump / thigh entries are near)
//   sub-step:     separating-plane test along the base's lateral axis (left leg's innermost extent vs the right leg's):
//                 legs side by side, as in every sane gait, stop here
//   candidates:   the envs that pass, eight at a time: their two lanes stage their 8 + 8 spheres (centre, radius,
//                 velocity, damping cap) in LDS; the wave's 64 lanes then test the sphere pairs of those envs -- 8 lanes
//                 per env, 7 pairs per lane -- and OR the overlapping ones into a 64-bit mask per env (order-free);
//                 the env's two lanes walk the set bits in ascending order, each computing the force on ITS sphere
//                 (bit-for-bit opposite in the two lanes: see the contact law above)
struct SelfNear { uint32_t m; };   // bit 0: the legs' bounding spheres can meet during this policy step; bits 16..23: base-lump / thigh entries
constexpr float kSelfMargin = 0.10f;   // 5 m/s of closing speed over the 0.02 s of a policy step
constexpr int SELF_GROUP = 8;          // candidate envs per round of pair tests (8 lanes each)
// LDS staging, one row of SELF_ROW float4 per lane (env, side): its 8 sphere centres relative to O with their radii, then
// (w, v) of its three shape-carrying bodies.  Every lane stages every sub-step, in the same pass that finds the leg's
// lateral extent (the centres are computed once); the velocities of the few spheres that do overlap are formed from the
// body rows by the lanes that need them.
constexpr int SELF_ROW = 8 + 6;
constexpr int SELF_ST_BYTES = 64 * SELF_ROW * 16, SELF_BYTES = SELF_ST_BYTES + EPB * 8 + 64;
struct SelfBuf { float4* st; unsigned long long* mask; uint8_t* envs; };   // mask[env]: overlapping pairs; envs[rank]: candidate envs, compacted
GRX_DEV SelfBuf self_carve(char* p) {
    SelfBuf b;
    b.st = reinterpret_cast<float4*>(p);
    b.mask = reinterpret_cast<unsigned long long*>(p + SELF_ST_BYTES);
    b.envs = reinterpret_cast<uint8_t*>(p + SELF_ST_BYTES + EPB * 8);
    return b;
}

GRX_DEV V3 sph_centre(const SphC& S, const ChainKin& K) { return K.rho + rot(K.R, v3(S.x, S.y, S.z)); }
GRX_DEV V3 v3_swap(V3 a) { return v3(pair_swap(a.x), pair_swap(a.y), pair_swap(a.z)); }

template <int PARTS = 3>   // 1: the leg x leg bit only, 2: the thigh x base-lump bits only (eight waves: a part per wave, like self_collision)
GRX_DEV SelfNear self_broad_phase(KP P, const SideConst& C, int side, const R3& R0, const ChainKin K[3]) {
    SelfNear sn; sn.m = 0;
    if (!P.self_collisions) return sn;
    V3 bc[3], oc[3];
    float br[3], orr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bc[i] = K[i].rho + rot(K[i].R, v3(C.bs[i][0], C.bs[i][1], C.bs[i][2]));
        br[i] = C.bs[i][3];
        oc[i] = v3_swap(bc[i]);
        orr[i] = pair_swap(br[i]);
    }
    if (PARTS & 1)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int bit_l = i * 3 + j, bit_r = j * 3 + i;   // the mask is indexed (left body, right body)
            const bool feasible = (P.ll_mask >> (side == 0 ? bit_l : bit_r)) & 1u;
            const V3 d = bc[i] - oc[j];
            const float R = br[i] + orr[j] + kSelfMargin;
            if (feasible && dot(d, d) < R * R) sn.m |= 1u;   // (the same verdict in both lanes of the env: symmetric arithmetic)
        }
    if (PARTS & 2)
#pragma unroll
    for (int e = 0; e < GRX_MAX_BC; ++e) {
        if (!__any(e < C.nbc)) break;
        if (e < C.nbc) {
            const BaseChainPair& q = C.bc[e];
            const V3 d = bc[0] - rot(R0, v3(q.x, q.y, q.z));
            const float R = br[0] + q.r + kSelfMargin;
            if (dot(d, d) < R * R) sn.m |= 1u << (16 + e);
        }
    }
    return sn;
}

// K[0..2]: frames of this lane's chain bodies 2, 3, 4; sn:
// this policy step's broad phase.  Must be called by all 64 lanes in wave-uniform control flow.
struct SelfNoVel { GRX_DEV void operator()() const {} };
// velocities(): called once the sphere centres are staged and before any body velocity K[i].w / K[i].v is read (a caller that walked
// the chain for positions only fills them in there)
// PARTS: 1 = leg x leg, 2 = thigh x base-lump shapes, 3 = both (eight waves per block: a part per wave)
template <class Vel = SelfNoVel, int PARTS = 3>
GRX_DEV void self_collision(KP P, const KTables& T, const SideConst& C, const SelfBuf& SB, int lane, int side, const R3& R0, V3 ang, V3 vel,
                            const ChainKin K[3], float mu, const SelfNear& sn, SelfOut& o, long long* pacc = nullptr, Vel velocities = Vel()) {
    const V3 zero = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.fa[i] = zero; o.fl[i] = zero; }
    o.f0a = zero; o.f0l = zero; o.fbase[0] = zero; o.fbase[1] = zero;
    if (!__any((sn.m & (PARTS == 1 ? 1u : (PARTS == 2 ? 0xffff0000u : 0xffffffffu))) != 0u)) return;
#ifdef GRX_PROFILE_SECTIONS
    long long pdummy[8]; if (!pacc) pacc = pdummy;
    const long long t0_ = clock64();
#endif
    constexpr int cnt[3] = {2, 2, 4}, off[3] = {8, 10, 12};
    bool have_vel = false;   // (wave-uniform)
    auto need_vel = [&]() { if (!have_vel) { velocities(); have_vel = true; } };
    // ---- leg x leg
    if ((PARTS & 1) && __any((sn.m & 1u) != 0u)) {
        // one pass over this lane's 8 spheres: centre -> LDS row, and the leg's extent towards the other leg along the
        // base's lateral axis (the separating-plane test below)
        const int el = lane_env(lane);
        const int hf_ = lane_half(lane);
        float4* const row = SB.st + (lane - hf_) * SELF_ROW;   // the leg's row (LPE == 4: at its first lane)
        const V3 yb = R0.cy;
        float ext = side == 0 ? 1e30f : -1e30f;
        if (LPL == 2) {   // each half of the leg stages the spheres of its own parity: thigh, shank, and two of the foot's four
            const SideConst& Ct = T.side[side];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = j < 2 ? j : 2;
                const int si = 2 * j + hf_;   // 0..7
                const SphC& S = Ct.sph[8 + si];
                const V3 c = sph_centre(S, K[i]);
                row[si] = rc4(c.x, c.y, c.z, S.r);
                const float y = dot(c, yb);
                ext = side == 0 ? fminf(ext, y - S.r) : fmaxf(ext, y + S.r);
            }
            const float oe = half_swap(ext);
            ext = side == 0 ? fminf(ext, oe) : fmaxf(ext, oe);
            need_vel();
#pragma unroll
            for (int i = 0; i < 3; ++i) {   // (both halves write the same body rows)
                row[8 + 2 * i] = rc4(K[i].w.x, K[i].w.y, K[i].w.z, K[i].v.x);
                row[9 + 2 * i] = rc4(K[i].v.y, K[i].v.z, 0.f, 0.f);
            }
        } else {
        need_vel();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int a = 0; a < cnt[i]; ++a) {
                const SphC& S = C.sph[off[i] + a];
                const V3 c = sph_centre(S, K[i]);
                row[off[i] - 8 + a] = rc4(c.x, c.y, c.z, S.r);
                const float y = dot(c, yb);
                ext = side == 0 ? fminf(ext, y - S.r) : fmaxf(ext, y + S.r);   // left leg (+y side): its lowest y; right leg: its highest
            }
            row[8 + 2 * i] = rc4(K[i].w.x, K[i].w.y, K[i].w.z, K[i].v.x);
            row[9 + 2 * i] = rc4(K[i].v.y, K[i].v.z, 0.f, 0.f);
        }
        }
#ifdef GRX_PROFILE_SECTIONS
        long long tp_ = clock64(); pacc[4] += tp_ - t0_;   // centres, extents, staging
#endif
        const float oext = pair_swap(ext);
        const bool cand = (sn.m & 1u) && (side == 0 ? ext <= oext : oext <= ext);   // not separated (same verdict in both lanes)
        const unsigned long long cb = __ballot(cand && side == 0 && lane_half(lane) == 0);
#ifdef GRX_PROFILE_SECTIONS
        pacc[1] += __popcll(cb);
#endif
        if (cb) {
            const int ncand = __popcll(cb);
            if (cand && side == 0 && lane_half(lane) == 0) {   // compacted list of the candidate envs; their pair masks start empty
                SB.envs[__popcll(cb & ((1ull << lane) - 1ull))] = (uint8_t)el;
                SB.mask[el] = 0ull;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef GRX_PROFILE_SECTIONS
            { const long long t_ = clock64(); pacc[5] += t_ - tp_; tp_ = t_; }   // ballot, list
#endif
            for (int g0 = 0; g0 < ncand; g0 += SELF_GROUP) {
                // all 64 lanes: 8 lanes per candidate env; lane `sub` tests the right leg's shape `sub` against the
                // left leg's eight (pair id = left shape * 8 + right shape; P.sp_mask: the pairs the model lists)
                const int ws_ = lane >> 3, sub = lane & 7;
                if (g0 + ws_ < ncand) {
                    const int env = SB.envs[g0 + ws_];
                    const float4* stl = SB.st + (env * LPE) * SELF_ROW;   // the left leg's row; the right leg's is LPL rows on
                    const float4 b = stl[LPL * SELF_ROW + sub];
                    unsigned long long bits = 0ull;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float4 a = stl[t];
                        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, Rs = a.w + b.w;
                        if (dx * dx + dy * dy + dz * dz < Rs * Rs) bits |= 1ull << (t * 8 + sub);
                    }
                    bits &= P.sp_mask;
                    if (bits) __hip_atomic_fetch_or(SB.mask + env, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            unsigned long long hm = cand ? SB.mask[el] : 0ull;
#ifdef GRX_PROFILE_SECTIONS
            { const long long t_ = clock64(); pacc[6] += t_ - tp_; tp_ = t_; }   // pair tests
            pacc[2] += __popcll(__ballot(hm != 0ull)); pacc[3] += 1;
#endif
            while (__any(hm != 0ull)) {   // overlapping pairs of this lane's env, ascending
                if (hm) {
                    const int pid = __ffsll((long long)hm) - 1;
                    hm &= hm - 1ull;
                    const int sa = pid >> 3, sb_ = pid & 7;      // left shape, right shape
                    const int ms = side == 0 ? sa : sb_, os = side == 0 ? sb_ : sa;   // mine, the other leg's
                    const int kb = ms < 2 ? 0 : (ms < 4 ? 1 : 2), ko = os < 2 ? 0 : (os < 4 ? 1 : 2);   // carrying chain body - 2
                    const float4* rm = SB.st + (lane - hf_) * SELF_ROW;
                    const float4* ro = SB.st + ((lane - hf_) ^ LPL) * SELF_ROW;
                    const float4 m0 = rm[ms], o0 = ro[os];
                    const float4 mw = rm[8 + 2 * kb], mv = rm[9 + 2 * kb], ow = ro[8 + 2 * ko], ov = ro[9 + 2 * ko];
                    SphW ma, ob;
                    ma.c = v3(m0.x, m0.y, m0.z); ma.r = m0.w; ma.dmax = T.side[side].sph[8 + ms].dmax;
                    ob.c = v3(o0.x, o0.y, o0.z); ob.r = o0.w; ob.dmax = T.side[side ^ 1].sph[8 + os].dmax;
                    ma.u = v3(mw.w, mv.x, mv.y) + cross(v3(mw.x, mw.y, mw.z), ma.c);
                    ob.u = v3(ow.w, ov.x, ov.y) + cross(v3(ow.x, ow.y, ow.z), ob.c);
                    V3 F, pw;
                    if (sphere_pair(P, ma, ob, mu, F, pw)) {
                        const V3 Tq = cross(pw, F);
                        if (kb == 0) { o.fa[0] = o.fa[0] + Tq; o.fl[0] = o.fl[0] + F; }
                        else if (kb == 1) { o.fa[1] = o.fa[1] + Tq; o.fl[1] = o.fl[1] + F; }
                        else { o.fa[2] = o.fa[2] + Tq; o.fl[2] = o.fl[2] + F; }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the rows are rewritten by the next sub-step
#ifdef GRX_PROFILE_SECTIONS
            { const long long t_ = clock64(); pacc[7] += t_ - tp_; tp_ = t_; }   // forces
#endif
        }
    }
    // ---- base-lump shapes x this lane's thigh shapes
    if ((PARTS & 2) && __any((sn.m >> 16) != 0u)) {
        need_vel();
        const ChainKin KB = {R0, zero, ang, vel};
        const V3 c0 = sph_centre(C.sph[8], K[0]), c1 = sph_centre(C.sph[9], K[0]);
        int prev_link = -1, slot = -1;
#pragma unroll
        for (int e = 0; e < GRX_MAX_BC; ++e) {
            if (!__any(e < C.nbc)) break;
            const BaseChainPair& q = C.bc[e];
            const bool mine = e < C.nbc;
            if (mine && q.link != prev_link) { prev_link = q.link; ++slot; }
            if (!__any((sn.m >> (16 + e)) & 1u)) continue;
            SphC sb; sb.x = q.x; sb.y = q.y; sb.z = q.z; sb.r = q.r; sb.dmax = q.dmax;
            const V3 cb = rot(R0, v3(q.x, q.y, q.z));
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {   // the base-lump sphere against the lane's two thigh shapes, in table order
                const V3 d = (ts ? c1 : c0) - cb;
                const float Rs = q.r + (ts ? C.sph[9].r : C.sph[8].r);
                const bool hit = mine && ((sn.m >> (16 + e)) & 1u) && ((q.tmask >> ts) & 1) && dot(d, d) < Rs * Rs;
                if (!__any(hit)) continue;
                if (hit) {
                    const SphW b = sph_world(sb, KB);
                    const SphW t = sph_world(ts ? C.sph[9] : C.sph[8], K[0]);
                    V3 F, pw;
                    if (sphere_pair(P, t, b, mu, F, pw)) {
                        o.fa[0] = o.fa[0] + cross(pw, F); o.fl[0] = o.fl[0] + F;
                        o.f0a = o.f0a - cross(pw, F); o.f0l = o.f0l - F;
                        if (slot == 0) o.fbase[0] = o.fbase[0] - F; else o.fbase[1] = o.fbase[1] - F;
                    }
                }
            }
        }
    }
#ifdef GRX_PROFILE_SECTIONS
    pacc[0] += clock64() - t0_;
#endif
}

// URDF link of the base-lump shape behind o.fbase[s] (-1: none)
GRX_DEV int self_base_link(const SideConst& C, int s) {
    int prev = -1, slot = -1, res = -1;
#pragma unroll
    for (int e = 0; e < GRX_MAX_BC; ++e)
        if (e < C.nbc && C.bc[e].link != prev) { prev = C.bc[e].link; ++slot; if (slot == s) res = prev; }
    return res;
}

// GRX_T_CONTACT_FORCES (the reference's contact_forces, legged_robot.py:117,266): net contact force per URDF link on the
// LAST sub-step = terrain contacts + self-collision.  lf: RareOut.lf (base-lump links of this lane's table); fl2 / fl3 /
// fl4: terrain forces on the thigh / shank / foot link; so: this lane's self-collision result.
// The termination / collision flags (legged_robot.py:336-353, the `collision` reward term) are taken HERE, from the same net
// forces the tensor shows -- a thigh pressing on a hand (GR1T2) terminates like a hand on the ground: term / pen_count of this
// lane's base-lump links (terminating / penalised shapes ride on the base lump: build_side_tables).
// what write_link_rows reads from the lane's tables, gathered ahead of the forces (wave 3 fills it while it waits for them)
struct LinkPrep { int lk[4]; int link[8]; uint32_t fl[8]; int chain[3]; };
GRX_DEV LinkPrep link_prep(const SideConst& C) {
    LinkPrep p;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        p.lk[s_] = self_base_link(C, s_);
        p.lk[2 + s_] = __builtin_bit_cast(int, pair_swap(__builtin_bit_cast(float, p.lk[s_])));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { p.link[i] = (C.sph[i].link_last & 1) ? sph_link(C.sph[i]) : -2; p.fl[i] = C.sph[i].flags; }
    p.chain[0] = sph_link(C.sph[8]); p.chain[1] = sph_link(C.sph[10]); p.chain[2] = sph_link(C.sph[12]);
    return p;
}
GRX_DEV void put_link_row(const LinkForceOut& o_, int link, V3 F) {
    if (o_.cf && link >= 0) { float* o = o_.cf + (size_t)(link * 3) * o_.N; o[0] = F.x; o[o_.N] = F.y; o[2 * o_.N] = F.z; }
}
// net force per link: rows[0..7] the base-lump links (valid where lp.link[i] != -2), rows[8..10] thigh, shank, foot; flags from them
GRX_DEV void net_link_forces(KP P, const LinkPrep& lp, const V3 lf[8], V3 fl2, V3 fl3, V3 fl4, const SelfOut& so, V3 rows[11], bool& term, float& pen_count) {
    term = false; pen_count = 0.f;
    // forces the self-collision puts on base-lump links: mine and the partner lane's (its thigh against the same or another link);
    // seldom any in the whole wave
    const bool anyfb = __any(dot(so.fbase[0], so.fbase[0]) + dot(so.fbase[1], so.fbase[1]) != 0.f);
    V3 fb[4];
    if (anyfb) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            fb[s_] = so.fbase[s_];
            fb[2 + s_] = v3(pair_swap(fb[s_].x), pair_swap(fb[s_].y), pair_swap(fb[s_].z));
        }
    }
    const float tf2 = P.termination_force * P.termination_force;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        V3 f = lf[i];
        const int link = lp.link[i];
        if (anyfb) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) if (lp.lk[s_] == link) f = f + fb[s_];
        }
        rows[i] = f;
        const bool closes = link != -2;
        const float n2 = dot(f, f);
        if (closes && (lp.fl[i] & GRX_SPH_TERMINATE) && n2 > tf2) term = true;
        if (closes && (lp.fl[i] & GRX_SPH_PENALISE) && n2 > 0.01f) pen_count += 1.0f;
    }
    rows[8] = fl2 + so.fl[0]; rows[9] = fl3 + so.fl[1]; rows[10] = fl4 + so.fl[2];
}
GRX_DEV void store_link_rows(const LinkForceOut& lfo, const LinkPrep& lp, const V3 rows[11]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (lp.link[i] != -2) put_link_row(lfo, lp.link[i], rows[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) put_link_row(lfo, lp.chain[i], rows[8 + i]);
}
GRX_DEV void write_link_rows(KP P, const LinkForceOut& lfo, const SideConst& C, const V3 lf[8], V3 fl2, V3 fl3, V3 fl4, const SelfOut& so,
                             bool& term, float& pen_count) {
    if (!lfo.last) return;
    const LinkPrep lp = link_prep(C);
    V3 rows[11];
    net_link_forces(P, lp, lf, fl2, fl3, fl4, so, rows, term, pen_count);
    store_link_rows(lfo, lp, rows);
}
