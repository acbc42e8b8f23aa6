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

// K[0..2]: frames of this lane's chain bodies 2, 3, 4.  Must be called by all 64 lanes in wave-uniform control flow.
GRX_DEV void self_collision(KP P, const SideConst& C, int side, const R3& R0, V3 ang, V3 vel, const ChainKin K[3], float mu, SelfOut& o) {
    const V3 zero = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.fa[i] = zero; o.fl[i] = zero; }
    o.f0a = zero; o.f0l = zero; o.fbase[0] = zero; o.fbase[1] = zero;
    if (!P.self_collisions) return;
    // ---- broad phase: bounding spheres of the three shape-carrying bodies of either leg
    V3 bc[3], oc[3];
    float br[3], orr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bc[i] = K[i].rho + rot(K[i].R, v3(C.bs[i][0], C.bs[i][1], C.bs[i][2]));
        br[i] = C.bs[i][3];
        oc[i] = v3(pair_swap(bc[i].x), pair_swap(bc[i].y), pair_swap(bc[i].z));
        orr[i] = pair_swap(br[i]);
    }
    uint32_t near = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int bit_l = i * 3 + j, bit_r = j * 3 + i;   // the mask is indexed (left body, right body)
            const bool feasible = (P.ll_mask >> (side == 0 ? bit_l : bit_r)) & 1u;
            const V3 d = bc[i] - oc[j];
            const float R = br[i] + orr[j];
            if (feasible && dot(d, d) < R * R) near |= 1u << (i * 3 + j);
        }
    // ---- narrow phase, body pair by body pair (wave-uniform skips)
    if (__any(near != 0u)) {
        constexpr int cnt[3] = {2, 2, 4}, off[3] = {8, 10, 12};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            uint32_t row = (near >> (i * 3)) & 7u;
            if (!__any(row != 0u)) continue;
            SphW mine[4];
#pragma unroll
            for (int a = 0; a < cnt[i]; ++a) mine[a] = sph_world(C.sph[off[i] + a], K[i]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // the partner lane must take part in the exchange: its (j, i) bit is this lane's (i, j) bit seen from the other side
                const bool pair_near = (near >> (i * 3 + j)) & 1u;
                if (!__any(pair_near)) continue;
                // other leg's body j shapes: computed by the partner as ITS body j, fetched by pair_swap
                SphW theirs[4];
#pragma unroll
                for (int b = 0; b < cnt[j]; ++b) theirs[b] = sph_swap(sph_world(C.sph[off[j] + b], K[j]));
                if (pair_near) {
#pragma unroll
                    for (int a = 0; a < cnt[i]; ++a)
#pragma unroll
                        for (int b = 0; b < cnt[j]; ++b) {
                            V3 F, pw;
                            if (sphere_pair(P, mine[a], theirs[b], mu, F, pw)) { o.fa[i] = o.fa[i] + cross(pw, F); o.fl[i] = o.fl[i] + F; }
                        }
                }
            }
        }
    }
    // ---- base-lump shapes x this lane's thigh shapes
    if (__any(C.nbc > 0)) {
        const SphW t0 = sph_world(C.sph[8], K[0]), t1 = sph_world(C.sph[9], K[0]);
        const ChainKin KB = {R0, zero, ang, vel};
        int prev_link = -1, slot = -1;
#pragma unroll
        for (int e = 0; e < GRX_MAX_BC; ++e) {
            if (!__any(e < C.nbc)) break;
            const BaseChainPair& q = C.bc[e];
            if (e < C.nbc) {
                if (q.link != prev_link) { prev_link = q.link; ++slot; }
                SphC sb; sb.x = q.x; sb.y = q.y; sb.z = q.z; sb.r = q.r; sb.dmax = q.dmax;
                const SphW b = sph_world(sb, KB);
                const SphW& t = q.tsel ? t1 : t0;
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
GRX_DEV void write_link_rows(const LinkForceOut& lfo, const SideConst& C, const V3 lf[8], V3 fl2, V3 fl3, V3 fl4, const SelfOut& so) {
    if (!lfo.last) return;
    // forces the self-collision puts on base-lump links: mine and the partner lane's (its thigh against the same or another link)
    int lk[4]; V3 fb[4];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        lk[s_] = self_base_link(C, s_); fb[s_] = so.fbase[s_];
        lk[2 + s_] = __builtin_bit_cast(int, pair_swap(__builtin_bit_cast(float, lk[s_])));
        fb[2 + s_] = v3(pair_swap(fb[s_].x), pair_swap(fb[s_].y), pair_swap(fb[s_].z));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (C.sph[i].link_last & 1) {
            V3 f = lf[i];
            const int link = sph_link(C.sph[i]);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) if (lk[s_] == link) f = f + fb[s_];
            put_link_force(lfo, C.sph[i], f);
        }
    put_link_force(lfo, C.sph[8], fl2 + so.fl[0]);
    put_link_force(lfo, C.sph[10], fl3 + so.fl[1]);
    put_link_force(lfo, C.sph[12], fl4 + so.fl[2]);
}
