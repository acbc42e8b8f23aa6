// grx_wavepipe.h -- the 4-waves-per-block physics pipeline of grx_step_kernel<HF, 4> (included by grx_kernels.hip
// inside its anonymous namespace, after the shared contact / kinematics helpers).
//
// At <= 8192 envs per GPU the step kernel is a handful of waves on a 1024-SIMD machine, each running alone on its
// SIMD at one instruction per 4 cycles: the instruction count of the longest wave IS the step time.  A block
// therefore spreads the sub-step of its 32 envs over the four SIMDs of a CU as a producer/consumer pipeline
// through LDS (lane i of every wave works on the same leg of the same env):
//
//   wave 0  "P" (owner of the state): motor torques; while the others start up, an outward walk that builds the
//            rigid-body inertias + joint axes for wave 1; then the BIAS half of the articulated-body recursion
//            (leaf -> root), floating-base solve, acceleration pass, integration -- and the rest of env.step()
//   wave 1  "I": the INERTIA half of the recursion (U, 1/d, rank-1 updates) as the rigid inertias arrive; streams
//            one record per joint (U, 1/d, updated 6x6) back to wave 0, which runs one joint behind; finally
//            the factorised base-level 6x6
//   wave 2  own outward walk with velocities: rigid-body bias forces + velocity-product accelerations of every
//            chain body (needed by wave 0 from its first joint on), then the 4 anchored foot spheres (it owns the
//            friction anchors) and the thigh spheres
//   wave 3  the base lump's bias force, the shank spheres (own walk to the knee), the base-lump contacts (torso,
//            head, arms ...)
//
// Synchronisation is by monotone sequence counters in LDS (release store by the producer after its data, acquire
// spin by the consumer), not block barriers, so each producer/consumer pair meets at its own time.  Every buffer is
// single: a producer only overwrites it after wave 0 has published the NEXT sub-step's state, which wave 0 does
// only after it has consumed all outputs of the current one.
#pragma once

#ifdef GRX_PROFILE_SECTIONS   // helper waves: cycles spent waiting for the next sub-step's state vs. in total
#define GRX_HELPER_PROF_BEGIN long long hp_idle = 0, hp_t0 = clock64(), hp_t = 0
#define GRX_HELPER_PROF_IDLE0 hp_t = clock64()
#define GRX_HELPER_PROF_IDLE1 hp_idle += clock64() - hp_t
#define GRX_HELPER_PROF_END(w) do { if (lane == 0) { P.prof[(size_t)blockIdx.x * 32 + 22 + 2 * (w)] = hp_idle; P.prof[(size_t)blockIdx.x * 32 + 23 + 2 * (w)] = clock64() - hp_t0; } } while (0)
#else
#define GRX_HELPER_PROF_BEGIN do {} while (0)
#define GRX_HELPER_PROF_IDLE0 do {} while (0)
#define GRX_HELPER_PROF_IDLE1 do {} while (0)
#define GRX_HELPER_PROF_END(w) do {} while (0)
#endif

#ifdef GRX_PROFILE_SECTIONS
#define GRX_WAIT(f, want, slot) do { long long w0_ = clock64(); flag_wait(f, want); tacc[slot] += clock64() - w0_; } while (0)
#else
#define GRX_WAIT(f, want, slot) flag_wait(f, want)
#endif

enum { FL_STATE = 0, FL_I = 1, FL_FOOT = 2, FL_LEGS = 3, FL_BASE = 4, FL_BIAS = 5, FL_REW = 6, FL_RI = 7, FL_BASEBIAS = 8, FL_SHANK = 9, FL_HZ = 10, FL_BHO1 = 11 /* ..13: waves 1..3 */, FL_RWB = 14, FL_COUNT = 16 };
constexpr int REC = 28;   // floats per joint record: ua 3, ul 3, 1/d, A' 6, B' 9, D' 6
constexpr int RIR = 15;   // floats per chain body from wave 0: rigid inertia about O (A 6, h 3), joint axis Sa 3, Ss 3
constexpr int PBR = 12;   // floats per chain body from wave 3: rigid-body bias force pa 3, pl 3, velocity-product acceleration ca 3, cl 3

struct PipeLds {
    float* base;   // [13][EPB]   base state at the start of the sub-step
    float* q;      // [2*LEG][64] q, qd of every lane's leg
    float* ri;     // [LEG][RIR][64] rigid inertias + joint axes (wave 0 -> I wave), leaf first
    float* rec;    // [LEG][REC][64] joint records of the I wave
    float* rec0;   // [21][64]    factorised base-level articulated inertia: inv(D) 6, inv(Schur) 6, B 9
    float* wc;     // [21][64]    contact wrenches about O on chain bodies 2, 3, 4; foot link velocity (3)
    float* pb;     // [LEG][PBR][64] + [6][64]: chain-body bias forces / accelerations (leaf first), base-lump bias force
    float* wr;     // [8][64]     base-lump wrench, termination flag, collision count
    int* flag;     // [FL_COUNT]
};

GRX_DEV void flag_set(int* f, int v, int lane) {
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
GRX_DEV void flag_wait(int* f, int want) {
    // pure spin (the waiter owns its SIMD; an s_sleep between polls only added detection latency: +1.3 % measured)
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) {}
}

// velocity-product (bias) force of a rigid body about O, from its centre-of-mass quantities (no 3x3 world inertia):
//   l = m (v + w x kap),  L_O = R Ic R^T w + kap x l,  p = (w x L_O + v x l ; w x l)
GRX_DEV void rigid_bias(const R3& R, V3 kap, float m, const S3& Ic, V3 w, V3 v, V3& pa, V3& pl) {
    const V3 l = (v + cross(w, kap)) * m;
    const V3 Lc = rot(R, mul(Ic, rotT(R, w)));
    const V3 ha = Lc + cross(kap, l);
    pa = cross(w, ha) + cross(v, l);
    pl = cross(w, l);
}

// rigid inertia of a body about O in world axes: A (rotational 3x3) and h = m kap (B = -skew(h), D = m 1)
GRX_DEV void rigid_inertia(const R3& R, V3 kap, float m, const S3& Ic, S3& A, V3& h) {
    A = rot_sym(R, Ic);
    const float kk = dot(kap, kap);
    A.xx += m * (kk - kap.x * kap.x); A.xy -= m * kap.x * kap.y; A.xz -= m * kap.x * kap.z;
    A.yy += m * (kk - kap.y * kap.y); A.yz -= m * kap.y * kap.z; A.zz += m * (kk - kap.z * kap.z);
    h = kap * m;
}
GRX_DEV void add_rigid(S3& A, M3& B, S3& D, const S3& Ak, V3 h, float m) {
    A = A + Ak;
    B.a01 -= h.z; B.a02 += h.y; B.a10 += h.z; B.a12 -= h.x; B.a20 -= h.y; B.a21 += h.x;
    D.xx += m; D.yy += m; D.zz += m;
}

// ---------------------------------------------------------------------------------------------------------------
// wave 0: one sub-step of the state owner
template <bool HF>
GRX_DEV void substep_p(KP P, const SideConst& C, const LaneConst& LC, LaneState& st, const float tau_m[LEG],
                       SubstepOut& out, FootKin& fk_before, const PipeLds& L, int lane, int seq, long long* tacc) {
    const float dt = P.sim_dt;
#ifdef GRX_PROFILE_SECTIONS
    long long tprev = clock64();
#endif
    V3 Sa[LEG], Ss[LEG], ca[LEG], cl[LEG], Ua[LEG], Ul[LEG];
    float dinv[LEG], uu[LEG];
    // ---- outward walk (positions only) + rigid inertias about O, leaf first, for the I wave: this wave has nothing
    // else to do until the first records come back
    {
        R3 RK[LEG];
        V3 rhoK[LEG];
        R3 R = quat_to_R(st.qx, st.qy, st.qz, st.qw);
        V3 rho = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            rho = rho + rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
            float sn, cs;
            grx_sincos(st.q[k], sn, cs);
            R = joint_rot_k(R, cs, sn, kAxis[k]);
            Sa[k] = axis_k(R, kAxis[k]);
            Ss[k] = cross(rho, Sa[k]);
            RK[k] = R; rhoK[k] = rho;
        }
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            const V3 kap = rhoK[k] + rot(RK[k], v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            const S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
            S3 Ak; V3 hk;
            rigid_inertia(RK[k], kap, C.body[k].mass, Ic, Ak, hk);
            float* o = L.ri + (size_t)(k * RIR) * 64 + lane;
            o[0 * 64] = Ak.xx; o[1 * 64] = Ak.xy; o[2 * 64] = Ak.xz; o[3 * 64] = Ak.yy; o[4 * 64] = Ak.yz; o[5 * 64] = Ak.zz;
            o[6 * 64] = hk.x; o[7 * 64] = hk.y; o[8 * 64] = hk.z;
            o[9 * 64] = Sa[k].x; o[10 * 64] = Sa[k].y; o[11 * 64] = Sa[k].z; o[12 * 64] = Ss[k].x; o[13 * 64] = Ss[k].y; o[14 * 64] = Ss[k].z;
            flag_set(L.flag + FL_RI, seq * 8 + (LEG - k), lane);
        }
    }
    // ---- pass 2, bias half (leaf -> root), one joint behind the I wave.  The chain-body CONTACT wrenches are not
    // waited for here: the recursion is linear in the bias forces, so they are propagated separately below.
    V3 pa = v3(0.f, 0.f, 0.f), pl = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        GRX_WAIT(L.flag + FL_I, seq * 8 + (LEG - k), 1);
        const float* r = L.rec + (size_t)(k * REC) * 64 + lane;
        const V3 ua = v3(r[0 * 64], r[1 * 64], r[2 * 64]), ul = v3(r[3 * 64], r[4 * 64], r[5 * 64]);
        const float di = r[6 * 64];
        const S3 A = {r[7 * 64], r[8 * 64], r[9 * 64], r[10 * 64], r[11 * 64], r[12 * 64]};
        const M3 B = {r[13 * 64], r[14 * 64], r[15 * 64], r[16 * 64], r[17 * 64], r[18 * 64], r[19 * 64], r[20 * 64], r[21 * 64]};
        const S3 D = {r[22 * 64], r[23 * 64], r[24 * 64], r[25 * 64], r[26 * 64], r[27 * 64]};
        {   // this body's rigid bias force joins the running articulated bias; its velocity-product acceleration
            GRX_WAIT(L.flag + FL_BIAS, seq * 8 + (LEG - k), 0);
            const float* b_ = L.pb + (size_t)(k * PBR) * 64 + lane;
            pa = pa + v3(b_[0 * 64], b_[1 * 64], b_[2 * 64]); pl = pl + v3(b_[3 * 64], b_[4 * 64], b_[5 * 64]);
            ca[k] = v3(b_[6 * 64], b_[7 * 64], b_[8 * 64]); cl[k] = v3(b_[9 * 64], b_[10 * 64], b_[11 * 64]);
        }
        const float qdk = st.qd[k];
        // joint-limit spring/damper (oracle substep()): added to the motor torque
        float t = tau_m[k];
        if (st.q[k] < C.body[k].qlo) t += C.body[k].Klim * (C.body[k].qlo - st.q[k]) - C.body[k].Clim * qdk;
        else if (st.q[k] > C.body[k].qhi) t += C.body[k].Klim * (C.body[k].qhi - st.q[k]) - C.body[k].Clim * qdk;
        const float u = t - (dot(Sa[k], pa) + dot(Ss[k], pl));
        const float ud = u * di;
        V3 npa = pa + mul(A, ca[k]) + mul(B, cl[k]) + ua * ud;
        V3 npl = pl + mulT(B, ca[k]) + mul(D, cl[k]) + ul * ud;
        Ua[k] = ua; Ul[k] = ul; dinv[k] = di; uu[k] = u;
        pa = npa; pl = npl;
    }
    // ---- contact wrenches on chain bodies 4 (foot), 3 (shank), 2 (thigh): delta recursion  dp -> dp + U (-S.dp)/d
    GRX_WAIT(L.flag + FL_FOOT, seq + 1, 2);
    GRX_WAIT(L.flag + FL_LEGS, seq + 1, 2);
    GRX_WAIT(L.flag + FL_SHANK, seq + 1, 2);
    {
        V3 da = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            if (k >= 2) {
                const float* c = L.wc + (size_t)((k - 2) * 6) * 64 + lane;
                const V3 fa = v3(c[0 * 64], c[1 * 64], c[2 * 64]), fl = v3(c[3 * 64], c[4 * 64], c[5 * 64]);
                da = da - fa; dl = dl - fl;
                if (k == LEG - 1) { out.foot_force = fl; fk_before.vel = v3(c[6 * 64], c[7 * 64], c[8 * 64]); }
            }
            const float du = -(dot(Sa[k], da) + dot(Ss[k], dl));
            uu[k] += du;
            const float dud = du * dinv[k];
            da = fma3(Ua[k], dud, da); dl = fma3(Ul[k], dud, dl);
        }
        pa = pa + da; pl = pl + dl;
    }
    // ---- base: both chains (DPP pair exchange) + base lump, 6x6 solve
    GRX_WAIT(L.flag + FL_BASE, seq + 1, 3);
    {
        const float* wr = L.wr + lane;
        const V3 f0a = v3(wr[0 * 64], wr[1 * 64], wr[2 * 64]), f0l = v3(wr[3 * 64], wr[4 * 64], wr[5 * 64]);
        out.term = wr[6 * 64] != 0.f;
        out.pen_count = wr[7 * 64];
        pa = pa - f0a; pl = pl - f0l;
    }
    pa = pair_sum(pa); pl = pair_sum(pl);
    {   // base-lump bias force (wave 3; both lanes of the pair add the same value after the pair sum)
        GRX_WAIT(L.flag + FL_BASEBIAS, seq + 1, 0);
        const float* b_ = L.pb + (size_t)(LEG * PBR) * 64 + lane;
        pa = pa + v3(b_[0 * 64], b_[1 * 64], b_[2 * 64]); pl = pl + v3(b_[3 * 64], b_[4 * 64], b_[5 * 64]);
    }
    // [A B; B^T D][alpha; acc] = -[pa; pl], factorised by the I wave: Di = inv(D), Sci = inv(A - B Di B^T)
    //   alpha = Sci (B Di pl - pa),  acc = -Di (pl + B^T alpha)
    GRX_WAIT(L.flag + FL_I, seq * 8 + LEG + 1, 4);
    const float* r0 = L.rec0 + lane;
    const S3 Di = {r0[0 * 64], r0[1 * 64], r0[2 * 64], r0[3 * 64], r0[4 * 64], r0[5 * 64]};
    const S3 Sci = {r0[6 * 64], r0[7 * 64], r0[8 * 64], r0[9 * 64], r0[10 * 64], r0[11 * 64]};
    const M3 B = {r0[12 * 64], r0[13 * 64], r0[14 * 64], r0[15 * 64], r0[16 * 64], r0[17 * 64], r0[18 * 64], r0[19 * 64], r0[20 * 64]};
    const V3 alpha = mul(Sci, mul(B, mul(Di, pl)) - pa);
    const V3 acc = neg(mul(Di, pl + mulT(B, alpha)));
    // ---- pass 3 (root -> leaf): accelerations
    float qdd[LEG];
    V3 aa = alpha, al = acc;
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        const V3 pa_ = aa + ca[k], pl_ = al + cl[k];
        const float qd2 = (uu[k] - (dot(Ua[k], pa_) + dot(Ul[k], pl_))) * dinv[k];
        qdd[k] = qd2;
        aa = fma3(Sa[k], qd2, pa_);
        al = fma3(Ss[k], qd2, pl_);
    }
    // ---- integrate (semi-implicit Euler)
    const V3 lin = acc + cross(st.ang, st.vel);  // classical acceleration of the base origin
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
    const float hx = 0.5f * dt * st.ang.x, hy = 0.5f * dt * st.ang.y, hz = 0.5f * dt * st.ang.z;
    const float x = st.qx, y = st.qy, z = st.qz, ww = st.qw;
    const float nx = x + hx * ww + hy * z - hz * y;
    const float ny = y - hx * z + hy * ww + hz * x;
    const float nz = z + hx * y - hy * x + hz * ww;
    const float nw = ww - hx * x - hy * y - hz * z;
    const float n = grx_rsq(nx * nx + ny * ny + nz * nz + nw * nw);
    st.qx = nx * n; st.qy = ny * n; st.qz = nz * n; st.qw = nw * n;
#ifdef GRX_PROFILE_SECTIONS
    tacc[5] += clock64() - tprev;   // whole sub-step
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// wave 1: inertia half of the articulated-body recursion for every sub-step of the policy step
GRX_DEV void iwave_loop(KP P, const SideConst& C, float base_m, V3 base_c, const S3& base_I, const PipeLds& L,
                        int lane, int el) {
    GRX_HELPER_PROF_BEGIN;
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        const float* b = L.base + el;
        const R3 R0 = quat_to_R(b[3 * EPB], b[4 * EPB], b[5 * EPB], b[6 * EPB]);
        // base lump first (needs only the base pose): ready long before wave 0's first rigid inertia arrives
        S3 A0; V3 h0;
        rigid_inertia(R0, rot(R0, base_c), base_m, base_I, A0, h0);
        // inward recursion, fed by wave 0
        S3 A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        M3 B = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            flag_wait(L.flag + FL_RI, seq * 8 + (LEG - k));
            const float* i_ = L.ri + (size_t)(k * RIR) * 64 + lane;
            {
                const S3 Ak = {i_[0 * 64], i_[1 * 64], i_[2 * 64], i_[3 * 64], i_[4 * 64], i_[5 * 64]};
                add_rigid(A, B, D, Ak, v3(i_[6 * 64], i_[7 * 64], i_[8 * 64]), C.body[k].mass);
            }
            const V3 a = v3(i_[9 * 64], i_[10 * 64], i_[11 * 64]), s = v3(i_[12 * 64], i_[13 * 64], i_[14 * 64]);
            const V3 ua = mul(A, a) + mul(B, s);
            const V3 ul = mulT(B, a) + mul(D, s);
            const float di = grx_rcp(dot(a, ua) + dot(s, ul));
            syr(A, ua, di); ger(B, ua, ul, di); syr(D, ul, di);
            float* r = L.rec + (size_t)(k * REC) * 64 + lane;
            r[0 * 64] = ua.x; r[1 * 64] = ua.y; r[2 * 64] = ua.z; r[3 * 64] = ul.x; r[4 * 64] = ul.y; r[5 * 64] = ul.z;
            r[6 * 64] = di;
            r[7 * 64] = A.xx; r[8 * 64] = A.xy; r[9 * 64] = A.xz; r[10 * 64] = A.yy; r[11 * 64] = A.yz; r[12 * 64] = A.zz;
            r[13 * 64] = B.a00; r[14 * 64] = B.a01; r[15 * 64] = B.a02; r[16 * 64] = B.a10; r[17 * 64] = B.a11;
            r[18 * 64] = B.a12; r[19 * 64] = B.a20; r[20 * 64] = B.a21; r[21 * 64] = B.a22;
            r[22 * 64] = D.xx; r[23 * 64] = D.xy; r[24 * 64] = D.xz; r[25 * 64] = D.yy; r[26 * 64] = D.yz; r[27 * 64] = D.zz;
            flag_set(L.flag + FL_I, seq * 8 + (LEG - k), lane);
        }
        // base level: both chains + the base lump
        A = pair_sum(A); B = pair_sum(B); D = pair_sum(D);
        add_rigid(A, B, D, A0, h0, base_m);
        {   // factorise for wave 0's solve: Di = inv(D), Schur complement Sc = A - B Di B^T, Sci = inv(Sc)
            const S3 Di = inv(D);
            const V3 b0 = v3(B.a00, B.a01, B.a02), b1 = v3(B.a10, B.a11, B.a12), b2 = v3(B.a20, B.a21, B.a22);
            const V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
            const S3 Sc = {A.xx - dot(b0, d0), A.xy - dot(b0, d1), A.xz - dot(b0, d2), A.yy - dot(b1, d1), A.yz - dot(b1, d2), A.zz - dot(b2, d2)};
            const S3 Sci = inv(Sc);
            float* r0 = L.rec0 + lane;
            r0[0 * 64] = Di.xx; r0[1 * 64] = Di.xy; r0[2 * 64] = Di.xz; r0[3 * 64] = Di.yy; r0[4 * 64] = Di.yz; r0[5 * 64] = Di.zz;
            r0[6 * 64] = Sci.xx; r0[7 * 64] = Sci.xy; r0[8 * 64] = Sci.xz; r0[9 * 64] = Sci.yy; r0[10 * 64] = Sci.yz; r0[11 * 64] = Sci.zz;
            r0[12 * 64] = B.a00; r0[13 * 64] = B.a01; r0[14 * 64] = B.a02; r0[15 * 64] = B.a10; r0[16 * 64] = B.a11;
            r0[17 * 64] = B.a12; r0[18 * 64] = B.a20; r0[19 * 64] = B.a21; r0[20 * 64] = B.a22;
        }
        flag_set(L.flag + FL_I, seq * 8 + LEG + 1, lane);
    }
    GRX_HELPER_PROF_END(1);
}

// ---------------------------------------------------------------------------------------------------------------
// wave 2: chain-body contacts (feet first: the bias recursion starts at the leaf)
template <bool HF>
GRX_DEV void chain_contact_loop(KP P, const SideConst& C, float mu, float hmax, LaneState& hs, const PipeLds& L,
                                int lane, int el) {
    GRX_HELPER_PROF_BEGIN;
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        const float* b = L.base + el;
        const V3 O = v3(b[0 * EPB], b[1 * EPB], b[2 * EPB]);
        const R3 R0 = quat_to_R(b[3 * EPB], b[4 * EPB], b[5 * EPB], b[6 * EPB]);
        const V3 vel = v3(b[7 * EPB], b[8 * EPB], b[9 * EPB]), ang = v3(b[10 * EPB], b[11 * EPB], b[12 * EPB]);
        const float* qs = L.q + lane;
        // outward walk with velocities; rigid-body bias forces + velocity-product accelerations of the chain bodies,
        // leaf first (wave 0's recursion starts at the foot and needs them before anything else this wave makes)
        ChainKin K = {R0, v3(0.f, 0.f, 0.f), ang, vel};
        ChainKin KK[LEG];
        V3 cak[LEG], clk[LEG];
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            const V3 wp = K.w, vp = K.v;   // parent velocity
            const float qdk = qs[(LEG + k) * 64];
            K.rho = K.rho + rot(K.R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
            float sn, cs;
            grx_sincos(qs[k * 64], sn, cs);
            K.R = joint_rot_k(K.R, cs, sn, kAxis[k]);
            const V3 a = axis_k(K.R, kAxis[k]);
            const V3 s = cross(K.rho, a);
            cak[k] = cross(wp, a) * qdk;
            clk[k] = (cross(vp, a) + cross(wp, s)) * qdk;
            K.w = fma3(a, qdk, wp); K.v = fma3(s, qdk, vp);
            KK[k] = K;
        }
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            const V3 kap = KK[k].rho + rot(KK[k].R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            const S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
            V3 pa, pl;
            rigid_bias(KK[k].R, kap, C.body[k].mass, Ic, KK[k].w, KK[k].v, pa, pl);
            float* o = L.pb + (size_t)(k * PBR) * 64 + lane;
            o[0 * 64] = pa.x; o[1 * 64] = pa.y; o[2 * 64] = pa.z; o[3 * 64] = pl.x; o[4 * 64] = pl.y; o[5 * 64] = pl.z;
            o[6 * 64] = cak[k].x; o[7 * 64] = cak[k].y; o[8 * 64] = cak[k].z; o[9 * 64] = clk[k].x; o[10 * 64] = clk[k].y; o[11 * 64] = clk[k].z;
            flag_set(L.flag + FL_BIAS, seq * 8 + (LEG - k), lane);
        }
        const ChainKin& K2 = KK[2];
        float* c_ = L.wc + lane;
        V3 fa, fl;
        foot_contacts<HF>(P, C, K, O, mu, hmax, hs, fa, fl);
        c_[12 * 64] = fa.x; c_[13 * 64] = fa.y; c_[14 * 64] = fa.z; c_[15 * 64] = fl.x; c_[16 * 64] = fl.y; c_[17 * 64] = fl.z;
        {   // foot link velocity BEFORE this sub-step's integration (sub-step averaged foot speed, fftai.py:79-81)
            const V3 fr = K.rho + rot(K.R, v3(C.foot_pos[0], C.foot_pos[1], C.foot_pos[2]));
            const V3 fv = K.v + cross(K.w, fr);
            c_[18 * 64] = fv.x; c_[19 * 64] = fv.y; c_[20 * 64] = fv.z;
        }
        flag_set(L.flag + FL_FOOT, seq + 1, lane);
        link_contacts<HF>(P, C, 2, K2, O, mu, hmax, fa, fl);      // thigh (the shank spheres are wave 3's)
        c_[0 * 64] = fa.x; c_[1 * 64] = fa.y; c_[2 * 64] = fa.z; c_[3 * 64] = fl.x; c_[4 * 64] = fl.y; c_[5 * 64] = fl.z;
        flag_set(L.flag + FL_LEGS, seq + 1, lane);
    }
    GRX_HELPER_PROF_END(2);
}

// wave 3: base-lump contacts
template <bool HF>
GRX_DEV void base_contact_loop(KP P, const SideConst& C, float mu, float hmax, float base_m, V3 base_c, const S3& base_I,
                               const PipeLds& L, int lane, int el) {
    GRX_HELPER_PROF_BEGIN;
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        const float* b = L.base + el;
        const V3 O = v3(b[0 * EPB], b[1 * EPB], b[2 * EPB]);
        const R3 R0 = quat_to_R(b[3 * EPB], b[4 * EPB], b[5 * EPB], b[6 * EPB]);
        const V3 vel = v3(b[7 * EPB], b[8 * EPB], b[9 * EPB]), ang = v3(b[10 * EPB], b[11 * EPB], b[12 * EPB]);
        {   // the base lump's rigid-body bias force (cheap; needed by wave 0 only at the base solve)
            V3 bpa, bpl;
            rigid_bias(R0, rot(R0, base_c), base_m, base_I, ang, vel, bpa, bpl);
            float* o = L.pb + (size_t)(LEG * PBR) * 64 + lane;
            o[0 * 64] = bpa.x; o[1 * 64] = bpa.y; o[2 * 64] = bpa.z; o[3 * 64] = bpl.x; o[4 * 64] = bpl.y; o[5 * 64] = bpl.z;
            flag_set(L.flag + FL_BASEBIAS, seq + 1, lane);
        }
        {   // shank spheres (own outward walk to the knee): shares the chain-contact load with wave 2
            const float* qs = L.q + lane;
            ChainKin K = {R0, v3(0.f, 0.f, 0.f), ang, vel};
#pragma unroll
            for (int k = 0; k <= 3; ++k) chain_step(C, k, qs[k * 64], qs[(LEG + k) * 64], K);
            V3 fa, fl;
            link_contacts<HF>(P, C, 3, K, O, mu, hmax, fa, fl);
            float* c_ = L.wc + lane;
            c_[6 * 64] = fa.x; c_[7 * 64] = fa.y; c_[8 * 64] = fa.z; c_[9 * 64] = fl.x; c_[10 * 64] = fl.y; c_[11 * 64] = fl.z;
            flag_set(L.flag + FL_SHANK, seq + 1, lane);
        }
        V3 f0a, f0l; bool term; float pen;
        base_lump_contacts<HF, false>(P, C, R0, O, ang, vel, mu, hmax, f0a, f0l, term, pen);
        float* w_ = L.wr + lane;
        w_[0 * 64] = f0a.x; w_[1 * 64] = f0a.y; w_[2 * 64] = f0a.z;
        w_[3 * 64] = f0l.x; w_[4 * 64] = f0l.y; w_[5 * 64] = f0l.z;
        w_[6 * 64] = term ? 1.f : 0.f; w_[7 * 64] = pen;
        flag_set(L.flag + FL_BASE, seq + 1, lane);
    }
    GRX_HELPER_PROF_END(3);
}
