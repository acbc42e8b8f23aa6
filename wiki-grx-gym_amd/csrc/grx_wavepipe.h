// grx_wavepipe.h -- the waves-per-block physics pipelines of grx_step_kernel<HF, 4> and <HF, 8> (included by grx_kernels.hip
// inside its anonymous namespace, after the shared contact / kinematics helpers).
//
// This header comment describes the FOUR-wave pipeline; the EIGHT-wave one (two waves per SIMD, the default since round 3) re-cuts
// the same work into eight roles and is described where its roles begin ("Eight waves per block", further down, and DESIGN.md 4.1).
//
// At <= 8192 envs per GPU the step kernel is a handful of waves on a 1024-SIMD machine, each running alone on its
// SIMD at one instruction per ~4.4 cycles: the instruction count of the longest dependent chain IS the step time.  A
// block therefore spreads the sub-step of its 32 envs over the four SIMDs of a CU (lane i of every wave works on the
// same leg of the same env), partitioned so that the four chains end together (round 2 balance, cycles per sub-step
// on rough terrain in brackets):
//
//   wave 0  owner of the state: motor torques; outward walk (positions); the whole inward recursion -- rigid inertias,
//            articulated inertias, bias forces -- in one pass, fed with the rigid-body bias forces of wave 2; contact
//            wrenches by delta recursion, floating-base solve, acceleration pass, integration -- and the rest of env.step()
//   wave 1  self-collision (grx_self.h) on the chain frames wave 2 publishes: leg x leg (the partner lane is one DPP
//            step away), thigh x base-lump shapes
//   wave 2  own walk with velocities; publishes the thigh / shank / foot frames; rigid-body bias forces and
//            velocity-product accelerations of the chain, leaf first, for wave 0; the 4 anchored foot spheres (it owns
//            the friction anchors)
//   wave 3  the base lump's bias force; the seldom-touching shapes (base lump, thigh, shank), lane-compacted
//            (grx_rare.h)
//
// (Rounds 1 and early 2 ran the inertia half of the recursion on wave 1, streaming joint records to wave 0: that hid
// ~1.7 k cycles of a ~9 k chain at the price of a wave; the self-collision of round 2 needs that wave more.)
// Round 1 had wave 2 evaluate the thigh shapes as well and wave 3 loop over all its base-lump and shank shapes
// whenever one env of the wave had any within reach (7 k cycles per sub-step): wave 0 waited 45 % of every sub-step.
//
// Synchronisation is by monotone sequence counters in LDS (release store by the producer after its data, acquire
// spin by the consumer), not block barriers, so each producer/consumer pair meets at its own time.  Every buffer is
// single: a producer only overwrites it after wave 0 has published the NEXT sub-step's state, which wave 0 does
// only after it has consumed all outputs of the current one.
#pragma once
#ifndef GRX_W8_WAITALL
#define GRX_W8_WAITALL 1
#endif
#ifndef GRX_W8_RARESPLIT
#define GRX_W8_RARESPLIT 1
#endif

#ifdef GRX_PROFILE_SECTIONS   // helper waves: cycles spent waiting for the next sub-step's state vs. in total
#define GRX_HELPER_PROF_BEGIN long long hp_idle = 0, hp_t0 = clock64(), hp_t = 0
#define GRX_HELPER_PROF_IDLE0 hp_t = clock64()
#define GRX_HELPER_PROF_IDLE1 hp_idle += clock64() - hp_t
#define GRX_HELPER_PROF_END(w) do { if (lane == 0) { P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 22 + 2 * (w)] = hp_idle; P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 23 + 2 * (w)] = clock64() - hp_t0; } } while (0)
#else
#define GRX_HELPER_PROF_BEGIN do {} while (0)
#define GRX_HELPER_PROF_IDLE0 do {} while (0)
#define GRX_HELPER_PROF_IDLE1 do {} while (0)
#define GRX_HELPER_PROF_END(w) do {} while (0)
#endif

#ifdef GRX_PROFILE_SECTIONS
// timeline of sub-step 5: event stamps in slots 48.. of the block's row (tools/gpu_sections.py prints them relative to EV 0)
#define GRX_EV(i) do { if (seq == 5 && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 48 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define GRX_EV(i) do {} while (0)
#endif
// keep a value's computation ABOVE this point (an empty asm the compiler must feed)
#define GRX_PIN(x) asm volatile("" : "+v"(x))
#ifdef GRX_PROFILE_SECTIONS
#define GRX_WAIT(f, want, slot) do { long long w0_ = clock64(); flag_wait(f, want); tacc[slot] += clock64() - w0_; } while (0)
#define GRX_WAIT_ALL(f, want_mine, lane, slot) do { long long w0_ = clock64(); flag_wait_all(f, want_mine, lane); tacc[slot] += clock64() - w0_; } while (0)
#else
#define GRX_WAIT(f, want, slot) flag_wait(f, want)
#define GRX_WAIT_ALL(f, want_mine, lane, slot) flag_wait_all(f, want_mine, lane)
#endif

enum { FL_STATE = 0, FL_FOOT = 2, FL_LEGS = 3, FL_FRAMES = 4, FL_BIAS = 5, FL_BIAS2 = 7 /* bias forces of bodies 2..0 when wave 1 computes them (heightfield) */, FL_SELF = 9, FL_REW = 6, FL_BASEBIAS = 8, FL_HZ = 10, FL_BHO1 = 11 /* ..13: waves 1..3 */, FL_RWB = 14, FL_RR = 15,
       FL_FACT = 16 /* W == 8: the base-level 6 x 6 is out (wave 0 -> wave 5) */, FL_FACTOUT = 17 /* ... and factorised */,
       FL_XK = 18 /* W == 8: rigid inertias of thigh, hip yaw, hip roll (wave 6 -> wave 0), seq * 4 + bodies out */,
       FL_CHAINW = 19 /* W == 8: thigh + shank terrain wrenches (wave 7) */, FL_BHO4 = 20 /* W == 8: fourth share of the observation height block */,
       FL_SB = 21 /* W == 8: the thigh x base-lump part of the self-collision is out (wave 3 with lane quads, wave 6 with lane pairs) */,
       FL_SCAN = 1 /* W == 8: COUNTER of the waves whose share of the height scan is in LDS (six: waves 1, 2, 4..7) */,
       FL_COUNT = 22 };
// every flag a slot of its own (FL_BHO1 stands for three: waves 1..3; the enumerators above are not in order, so a clash would go unnoticed)
constexpr bool flag_slots_distinct() {
    const int f[] = {FL_STATE, FL_SCAN, FL_FOOT, FL_LEGS, FL_FRAMES, FL_BIAS, FL_REW, FL_BIAS2, FL_BASEBIAS, FL_SELF, FL_HZ, FL_BHO1, FL_BHO1 + 1, FL_BHO1 + 2, FL_RWB, FL_RR,
                     FL_FACT, FL_FACTOUT, FL_XK, FL_CHAINW, FL_BHO4, FL_SB};
    constexpr int n = sizeof(f) / sizeof(f[0]);
    for (int i = 0; i < n; ++i) {
        if (f[i] < 0 || f[i] >= FL_COUNT) return false;
        for (int j = i + 1; j < n; ++j) if (f[i] == f[j]) return false;
    }
    return n == FL_COUNT;
}
static_assert(flag_slots_distinct(), "hand-over flags must occupy distinct slots of s_flag, all of them");
// Every record is laid out [quad][lane] in float4 units, so a lane moves it with ds_read_b128 / ds_write_b128: the
// kernel runs at one instruction issue per ~5 cycles whatever the instruction, and the records are ~350 dwords per
// lane and sub-step on wave 0 alone -- four dwords per LDS instruction instead of one is ~1.3k cycles per sub-step.
constexpr int PB4 = 2;    // quads per chain body from wave 2: rigid-body bias force pa 3, pl 3
constexpr int WC4 = 15;   // contact wrenches about O: thigh quads 0-1, shank 2-3, foot 4-6 (wrench 6 + foot link velocity 3);
                          // quads 7-14: self-collision (grx_self.h) on thigh, shank, foot, base lump + the forces on base-lump links
constexpr int Q4 = 3;     // q 5, qd 5 of the lane's leg

struct PipeLds {
    float4* bq;    // [4][EPB]    base state at the start of the sub-step: (pos, q.x) (q.yzw, vel.x) (vel.yz, ang.xy) (ang.z, -, -, -)
    float4* q;     // [Q4][64]    q, qd of every lane's leg
    float4* wc;    // [WC4][64]
    float4* pb;    // [LEG][PB4][64] + [2][64]: chain-body bias forces / accelerations (wave 2, leaf first), base-lump bias force (wave 3)
    float4* wr;    // [2][64]     base-lump wrench, termination flag, collision count
    int* flag;     // [FL_COUNT]
    float4* xk;    // [3][3][64]  W == 8: rigid inertia about O of chain body k = 2, 1, 0 (wave 6 -> wave 0): A 6, h = m kap 3
    float4* sb;    // [5][64]     W == 8: the thigh x base-lump part of the self-collision (wave 3 -> wave 0): thigh wrench 6, base-lump wrench 6,
                   //             forces on two base-lump links 6
    float4* fx;    // [8][64]     W == 8: quads 0-3 the base-level X, Y of both legs (wave 0 -> wave 5), 4-7 T = Y Xo^-1 and Sc^-1 (wave 5 -> wave 0)
};
GRX_DEV float4 f4(float a, float b, float c, float d) { float4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }
GRX_DEV void pipe_base_load(const float4* bq, int el, float b[13]) {
    const float4 a = bq[el], c = bq[EPB + el], d = bq[2 * EPB + el], e = bq[3 * EPB + el];
    b[0] = a.x; b[1] = a.y; b[2] = a.z; b[3] = a.w; b[4] = c.x; b[5] = c.y; b[6] = c.z; b[7] = c.w;
    b[8] = d.x; b[9] = d.y; b[10] = d.z; b[11] = d.w; b[12] = e.x;
}
GRX_DEV void pipe_base_store(float4* bq, int el, V3 pos, float qx, float qy, float qz, float qw, V3 vel, V3 ang) {
    bq[el] = f4(pos.x, pos.y, pos.z, qx); bq[EPB + el] = f4(qy, qz, qw, vel.x);
    bq[2 * EPB + el] = f4(vel.y, vel.z, ang.x, ang.y); bq[3 * EPB + el] = f4(ang.z, 0.f, 0.f, 0.f);
}

#include "grx_flags.h"   // flag_set / flag_wait / flag_wait_all / lds_barrier: shared with the litmus test tools/micro/lds_handover.hip

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
template <int HF, bool W8>   // W8: eight waves per block (see the second half of this file): no velocity-product terms, the rigid inertias of
                              // bodies 2, 1, 0 and all bias forces from other waves, one poll per group of hand-overs
GRX_DEV void substep_p(KP P, const SideConst& C, const LaneConst& LC, LaneState& st, const float tau_m[LEG],
                       SubstepOut& out, FootKin& fk_before, const PipeLds& L, const RareBuf& RB, int lane, int seq, long long* tacc,
                       const SideConst& Clds) {   // Clds: the LDS copy of C (tables read once per policy step)
    const float dt = P.sim_dt;
#ifdef GRX_PROFILE_SECTIONS
    long long tprev = clock64();
#endif
    V3 Sa[LEG], Ss[LEG], ca[LEG], cl[LEG], Ua[LEG], Ul[LEG];
    float dinv[LEG], uu[LEG];
    GRX_EV(0);
    // ---- outward walk with velocities: joint motion subspaces S = (a; rho x a), velocity-product accelerations and rigid
    // inertias about O of the chain bodies
    S3 AK[LEG];
    V3 hK[LEG];
    const R3 R0 = quat_to_R(st.qx, st.qy, st.qz, st.qw);
    {
        R3 R = R0;
        V3 rho = v3(0.f, 0.f, 0.f), w = st.ang, v = st.vel;
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            const float qdk = st.qd[k];
            rho = rho + rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
            float sn, cs;
            grx_sincos(st.q[k], sn, cs);
            R = joint_rot_k(R, cs, sn, kAxis[k]);
            const V3 a = axis_k(R, kAxis[k]);
            const V3 s_ = cross(rho, a);
            Sa[k] = a; Ss[k] = s_;
            if (!W8) {
                ca[k] = cross(w, a) * qdk;
                cl[k] = (cross(v, a) + cross(w, s_)) * qdk;
                w = fma3(a, qdk, w); v = fma3(s_, qdk, v);
            }
#ifndef GRX_P8_XK
#define GRX_P8_XK 1   // lane pairs, eight waves: 1 = the rigid inertias of bodies 2, 1, 0 come from wave 5 (as with lane quads)
#endif
            if (W8 && GRX_P8_XK && k < 2) continue;   // (rigid inertias of bodies 1, 0: wave 5 -- this wave would only wait for body 2's)
            const V3 kap = rho + rot(R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            const S3 Ic = {Clds.body[k].Ic[0], Clds.body[k].Ic[1], Clds.body[k].Ic[2], Clds.body[k].Ic[3], Clds.body[k].Ic[4], Clds.body[k].Ic[5]};   // (LDS: the register file is full)
            rigid_inertia(R, kap, C.body[k].mass, Ic, AK[k], hK[k]);
        }
    }
#ifdef GRX_PROFILE_SECTIONS   // (keep the walk above the stamp)
#pragma unroll
    for (int k = 0; k < LEG; ++k) { GRX_PIN(Ss[k].x); GRX_PIN(cl[k].x); GRX_PIN(AK[k].xx); GRX_PIN(AK[k].yz); GRX_PIN(hK[k].x); }
#endif
    GRX_EV(1);
    // ---- inward pass, inertia half (leaf -> root): articulated inertias, U = I^A S, 1/d, and the articulated inertia's
    // action on the velocity-product acceleration, I^a c -- everything the bias recursion below needs from this half, as
    // 6 + 6 + 1 numbers per joint.  No input from another wave: it runs while wave 2 is still producing the bias forces.
    V3 Ica[LEG], Icl[LEG];
    S3 A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    M3 B = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifndef GRX_PK_INERTIA
#define GRX_PK_INERTIA 0   // 1: hand-packed fp32 (v_pk_fma_f32) in the inertia half of the lane-pair eight-wave kernel.  MEASURED SLOWER (round 5, alternating runs,
                           // profiles/r05_experiments.md): 58.1 / 58.3 against 56.9 us per step at 8192 envs (52 spilled registers instead of 48; 36 updated entries
                           // instead of the symmetric form's 21 at 5.3 cycles per v_pk_fma against 4.5) -- kept as the record of the experiment, off
#endif
    constexpr bool kPk = W8 && GRX_PK_INERTIA;
    typedef float f2_t __attribute__((ext_vector_type(2)));
    struct C6 { f2_t r01, r23, r45; };
    C6 c6[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { c6[j].r01 = f2_t{0.f, 0.f}; c6[j].r23 = f2_t{0.f, 0.f}; c6[j].r45 = f2_t{0.f, 0.f}; }
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        if (W8 && GRX_P8_XK && k < 2) {
#ifndef GRX_P8_PINXK
#define GRX_P8_PINXK 1   // joints 4, 3, 2 of the inertia half stay IN FRONT of the wait for wave 5's rigid inertias (the compiler sinks register
                         // arithmetic across the spin: without the pins this wave idled at the flag and ran all five joints behind it)
#endif
            if (k == 1 && GRX_P8_PINXK && kPk) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { GRX_PIN(c6[j].r01); GRX_PIN(c6[j].r23); GRX_PIN(c6[j].r45); }
            }
            if (k == 1 && GRX_P8_PINXK && !kPk) {
                GRX_PIN(A.xx); GRX_PIN(A.xy); GRX_PIN(A.xz); GRX_PIN(A.yy); GRX_PIN(A.yz); GRX_PIN(A.zz);
                GRX_PIN(B.a00); GRX_PIN(B.a01); GRX_PIN(B.a02); GRX_PIN(B.a10); GRX_PIN(B.a11); GRX_PIN(B.a12); GRX_PIN(B.a20); GRX_PIN(B.a21); GRX_PIN(B.a22);
                GRX_PIN(D.xx); GRX_PIN(D.xy); GRX_PIN(D.xz); GRX_PIN(D.yy); GRX_PIN(D.yz); GRX_PIN(D.zz);
            }
#ifdef GRX_PROFILE_SECTIONS
            if (k == 1 && seq == 5 && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 91] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#endif
            if (k == 1) GRX_WAIT(L.flag + FL_XK, seq * 4 + 3, 4);
#ifdef GRX_PROFILE_SECTIONS
            if (k == 1 && seq == 5 && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 92] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#endif
            const float4* c = L.xk + (k * 3) * 64 + lane;
            const float4 a0 = c[0 * 64], a1 = c[1 * 64], a2 = c[2 * 64];
            AK[k].xx = a0.x; AK[k].xy = a0.y; AK[k].xz = a0.z; AK[k].yy = a0.w; AK[k].yz = a1.x; AK[k].zz = a1.y;
            hK[k] = v3(a1.z, a1.w, a2.x);
        }
        if (kPk) {   // the full 6 x 6 [A B; B^T D] as six columns of three float2 (rows 01 | 23 | 45): U = sum_j col_j S_j and col_j -= U (U_j / d) are
                     // v_pk_fma_f32 with a broadcast scalar (op_sel picks the half of a register pair: no moves) -- tools/micro/pk_recursion.hip
            const S3 K = AK[k];
            const V3 h = hK[k];
            const float m = C.body[k].mass;
            c6[0].r01 += f2_t{K.xx, K.xy}; c6[0].r23.x += K.xz;            c6[0].r45 += f2_t{-h.z, h.y};
            c6[1].r01 += f2_t{K.xy, K.yy}; c6[1].r23 += f2_t{K.yz, h.z};   c6[1].r45.y += -h.x;
            c6[2].r01 += f2_t{K.xz, K.yz}; c6[2].r23 += f2_t{K.zz, -h.y};  c6[2].r45.x += h.x;
            c6[3].r01.y += h.z;            c6[3].r23 += f2_t{-h.y, m};
            c6[4].r01.x += -h.z;           c6[4].r23.x += h.x;             c6[4].r45.x += m;
            c6[5].r01 += f2_t{h.y, -h.x};                                  c6[5].r45.y += m;
            const V3 a = Sa[k], s_ = Ss[k];
            const float sj[6] = {a.x, a.y, a.z, s_.x, s_.y, s_.z};
            f2_t u01 = c6[0].r01 * f2_t{sj[0], sj[0]}, u23 = c6[0].r23 * f2_t{sj[0], sj[0]}, u45 = c6[0].r45 * f2_t{sj[0], sj[0]};
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                u01 = __builtin_elementwise_fma(c6[j].r01, f2_t{sj[j], sj[j]}, u01);
                u23 = __builtin_elementwise_fma(c6[j].r23, f2_t{sj[j], sj[j]}, u23);
                u45 = __builtin_elementwise_fma(c6[j].r45, f2_t{sj[j], sj[j]}, u45);
            }
            f2_t t_ = f2_t{a.x, a.y} * u01;
            t_ = __builtin_elementwise_fma(f2_t{a.z, s_.x}, u23, t_);
            t_ = __builtin_elementwise_fma(f2_t{s_.y, s_.z}, u45, t_);
            const float di = grx_rcp(t_.x + t_.y);
            const f2_t w01 = u01 * f2_t{di, di}, w23 = u23 * f2_t{di, di}, w45 = u45 * f2_t{di, di};
            const float wj[6] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                c6[j].r01 = __builtin_elementwise_fma(-u01, f2_t{wj[j], wj[j]}, c6[j].r01);
                c6[j].r23 = __builtin_elementwise_fma(-u23, f2_t{wj[j], wj[j]}, c6[j].r23);
                c6[j].r45 = __builtin_elementwise_fma(-u45, f2_t{wj[j], wj[j]}, c6[j].r45);
            }
            Ua[k] = v3(u01.x, u01.y, u23.x); Ul[k] = v3(u23.y, u45.x, u45.y); dinv[k] = di;
            continue;
        }
        add_rigid(A, B, D, AK[k], hK[k], C.body[k].mass);
        const V3 a = Sa[k], s_ = Ss[k];
        const V3 ua = mul(A, a) + mul(B, s_);
        const V3 ul = mulT(B, a) + mul(D, s_);
        const float di = grx_rcp(dot(a, ua) + dot(s_, ul));
        syr(A, ua, di); ger(B, ua, ul, di); syr(D, ul, di);
        if (!W8) {
            Ica[k] = mul(A, ca[k]) + mul(B, cl[k]);
            Icl[k] = mulT(B, ca[k]) + mul(D, cl[k]);
        }
        Ua[k] = ua; Ul[k] = ul; dinv[k] = di;
    }
#ifdef GRX_PROFILE_SECTIONS
    GRX_PIN(A.xx); GRX_PIN(B.a00); GRX_PIN(D.xx);
#pragma unroll
    for (int k = 0; k < LEG; ++k) { GRX_PIN(Ica[k].x); GRX_PIN(Ica[k].y); GRX_PIN(Ica[k].z); GRX_PIN(Icl[k].x); GRX_PIN(Icl[k].y); GRX_PIN(Icl[k].z); }
    GRX_EV(3);
#endif
    if (kPk) {
        A.xx = c6[0].r01.x; A.xy = c6[0].r01.y; A.xz = c6[0].r23.x; A.yy = c6[1].r01.y; A.yz = c6[1].r23.x; A.zz = c6[2].r23.x;
        B.a00 = c6[3].r01.x; B.a10 = c6[3].r01.y; B.a20 = c6[3].r23.x; B.a01 = c6[4].r01.x; B.a11 = c6[4].r01.y; B.a21 = c6[4].r23.x;
        B.a02 = c6[5].r01.x; B.a12 = c6[5].r01.y; B.a22 = c6[5].r23.x;
        D.xx = c6[3].r23.y; D.xy = c6[4].r23.y; D.xz = c6[5].r23.y; D.yy = c6[4].r45.x; D.yz = c6[5].r45.x; D.zz = c6[5].r45.y;
    }
    // base level: both chains (DPP pair exchange) + the base lump, factorised for the solve below
    A = pair_sum(A); B = pair_sum(B); D = pair_sum(D);
    {
        S3 A0; V3 h0;
        rigid_inertia(R0, rot(R0, LC.base_c), LC.base_m, LC.base_I, A0, h0);
        add_rigid(A, B, D, A0, h0, LC.base_m);
    }
#ifdef GRX_PROFILE_SECTIONS
    GRX_PIN(A.xx); GRX_PIN(A.yz); GRX_PIN(B.a00); GRX_PIN(B.a22); GRX_PIN(D.xx); GRX_PIN(D.yz);
    GRX_EV(14);
#endif
    S3 Di = inv(D);
    S3 Sci;
    {
        const V3 b0 = v3(B.a00, B.a01, B.a02), b1 = v3(B.a10, B.a11, B.a12), b2 = v3(B.a20, B.a21, B.a22);
        const V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
        const S3 Sc = {A.xx - dot(b0, d0), A.xy - dot(b0, d1), A.xz - dot(b0, d2), A.yy - dot(b1, d1), A.yz - dot(b1, d2), A.zz - dot(b2, d2)};
        Sci = inv(Sc);
    }
#ifdef GRX_PROFILE_SECTIONS
    GRX_PIN(Sci.xx); GRX_PIN(Sci.yz); GRX_PIN(Di.xx); GRX_PIN(Di.yz);
    if (seq == 5 && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 79] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#endif
    // The spin-waits below are atomic loads: the compiler may sink pure register arithmetic across them, and did (the
    // whole inertia half ended up behind the last wait: 4 k cycles per sub-step, measured).  Pin the results here.
    GRX_PIN(Sci.xx); GRX_PIN(Sci.xy); GRX_PIN(Sci.xz); GRX_PIN(Sci.yy); GRX_PIN(Sci.yz); GRX_PIN(Sci.zz);
    GRX_PIN(Di.xx); GRX_PIN(Di.xy); GRX_PIN(Di.xz); GRX_PIN(Di.yy); GRX_PIN(Di.yz); GRX_PIN(Di.zz);
    if (!W8) {
#pragma unroll
        for (int k = 0; k < LEG; ++k) { GRX_PIN(Ica[k].x); GRX_PIN(Ica[k].y); GRX_PIN(Ica[k].z); GRX_PIN(Icl[k].x); GRX_PIN(Icl[k].y); GRX_PIN(Icl[k].z); }
    }
    GRX_EV(2);
    // ---- inward pass, bias half (leaf -> root): the rigid-body bias forces come from wave 2, leaf first.  The chain-body
    // CONTACT wrenches are not waited for here: the recursion is linear in the bias forces, they follow separately below.
    // (wave 2 is done with all five long before this point: one wait, and the ten LDS reads go out together -- one exposed
    //  LDS latency instead of one per joint)
    V3 pa = v3(0.f, 0.f, 0.f), pl = v3(0.f, 0.f, 0.f);
    if (W8) GRX_WAIT_ALL(L.flag, flag_want(lane, FL_BIAS, seq * 8 + 2, FL_BIAS2, seq + 1), lane, 0);
    else {
        GRX_WAIT(L.flag + FL_BIAS, seq * 8 + (HF ? 2 : LEG), 0);
        if (HF) GRX_WAIT(L.flag + FL_BIAS2, seq + 1, 0);   // bodies 2..0 come from wave 1 on a heightfield (see self_loop)
    }
    float4 bq0[LEG], bq1[LEG];
#pragma unroll
    for (int k = 0; k < LEG; ++k) { const float4* b_ = L.pb + (k * PB4) * 64 + lane; bq0[k] = b_[0 * 64]; bq1[k] = b_[1 * 64]; }
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        float tq_k;
        {
            const float4 b0_ = bq0[k], b1_ = bq1[k];
            pa = pa + v3(b0_.x, b0_.y, b0_.z); pl = pl + v3(b0_.w, b1_.x, b1_.y);
            tq_k = tau_m[k] + b1_.z;   // motor torque + joint-limit spring/damper (from wave 2, with the bias force)
        }
        const float u = tq_k - (dot(Sa[k], pa) + dot(Ss[k], pl));
        const float ud = u * dinv[k];
        uu[k] = u;
        if (!W8) { pa = pa + Ica[k]; pl = pl + Icl[k]; }
        pa = fma3(Ua[k], ud, pa);
        pl = fma3(Ul[k], ud, pl);
    }
    GRX_EV(7);
    // ---- contact wrenches on chain bodies 4 (foot), 3 (shank), 2 (thigh): delta recursion  dp -> dp + U (-S.dp)/d
    if (W8) {   // everything the rest of the sub-step consumes, in one poll
        GRX_WAIT_ALL(L.flag, flag_want(lane, FL_FOOT, seq + 1, FL_LEGS, seq + 1, FL_SELF, seq + 1, FL_BASEBIAS, seq + 1, FL_CHAINW, seq + 1, FL_SB, seq + 1), lane, 2);
        GRX_EV(5);
    } else {
        GRX_WAIT(L.flag + FL_FOOT, seq + 1, 2);
        GRX_EV(4);
        GRX_WAIT(L.flag + FL_LEGS, seq + 1, 2);
        GRX_EV(5);
        GRX_WAIT(L.flag + FL_SELF, seq + 1, 3);
    }
    SelfOut sc;   // self-collision wrenches (wave 1, after its recursion)
    {
        const float4* c = L.wc + 7 * 64 + lane;
        const float4 s0 = c[0 * 64], s1 = c[1 * 64], s2 = c[2 * 64], s3 = c[3 * 64], s4 = c[4 * 64], s5 = c[5 * 64];
        sc.fa[0] = v3(s0.x, s0.y, s0.z); sc.fl[0] = v3(s0.w, s1.x, s1.y);
        sc.fa[1] = v3(s1.z, s1.w, s2.x); sc.fl[1] = v3(s2.y, s2.z, s2.w);
        sc.fa[2] = v3(s3.x, s3.y, s3.z); sc.fl[2] = v3(s3.w, s4.x, s4.y);
        sc.f0a = v3(s4.z, s4.w, s5.x); sc.f0l = v3(s5.y, s5.z, s5.w);   // (quads 6, 7: forces on base-lump links, for wave 3's contact-force rows)
        if (W8) {   // the thigh x base-lump part (wave 3, with its FL_LEGS hand-over)
            const float4* d = L.sb + lane;
            const float4 d0 = d[0 * 64], d1 = d[1 * 64], d2 = d[2 * 64];
            sc.fa[0] = sc.fa[0] + v3(d0.x, d0.y, d0.z); sc.fl[0] = sc.fl[0] + v3(d0.w, d1.x, d1.y);
            sc.f0a = sc.f0a + v3(d1.z, d1.w, d2.x); sc.f0l = sc.f0l + v3(d2.y, d2.z, d2.w);
        }
    }
    {
        V3 da = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            if (k >= 2) {
                const float4* c = L.wc + ((k - 2) * 2) * 64 + lane;
                const float4 c0_ = c[0 * 64], c1_ = c[1 * 64];
                const V3 fa = v3(c0_.x, c0_.y, c0_.z), fl = v3(c0_.w, c1_.x, c1_.y);
                da = da - fa - sc.fa[k - 2]; dl = dl - fl - sc.fl[k - 2];
                if (k == LEG - 1) { const float4 c2_ = c[2 * 64]; out.foot_force = fl + sc.fl[2]; fk_before.vel = v3(c1_.z, c1_.w, c2_.x); }
            }
            const float du = -(dot(Sa[k], da) + dot(Ss[k], dl));
            uu[k] += du;
            const float dud = du * dinv[k];
            da = fma3(Ua[k], dud, da); dl = fma3(Ul[k], dud, dl);
        }
        pa = pa + da; pl = pl + dl;
    }
    // ---- base: both chains (DPP pair exchange) + base lump, 6x6 solve
    {   // (published together with the thigh / shank wrenches: FL_LEGS)
        const float4 w0_ = L.wr[lane], w1_ = L.wr[64 + lane];
        const V3 f0a = v3(w0_.x, w0_.y, w0_.z), f0l = v3(w0_.w, w1_.x, w1_.y);
        out.term = w1_.z != 0.f;
        out.pen_count = w1_.w;
        pa = pa - f0a - sc.f0a; pl = pl - f0l - sc.f0l;
    }
    // (GRX_T_CONTACT_FORCES rows of the last sub-step: written by wave 3, which has the base-lump links' forces at hand and is done
    //  before this wave -- base_contact_loop)
    pa = pair_sum(pa); pl = pair_sum(pl);
    {   // base-lump bias force (wave 3; both lanes of the pair add the same value after the pair sum)
        if (!W8) GRX_WAIT(L.flag + FL_BASEBIAS, seq + 1, 0);
        const float4* b_ = L.pb + (LEG * PB4) * 64 + lane;
        const float4 b0_ = b_[0 * 64], b1_ = b_[1 * 64];
        pa = pa + v3(b0_.x, b0_.y, b0_.z); pl = pl + v3(b0_.w, b1_.x, b1_.y);
    }
    // [A B; B^T D][alpha; acc] = -[pa; pl]:  alpha = Sci (B Di pl - pa),  acc = -Di (pl + B^T alpha)
    const V3 alpha = mul(Sci, mul(B, mul(Di, pl)) - pa);
    const V3 acc = neg(mul(Di, pl + mulT(B, alpha)));
    // ---- pass 3 (root -> leaf): accelerations
    float qdd[LEG];
    V3 aa = alpha, al = acc;
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        const V3 pa_ = W8 ? aa : aa + ca[k], pl_ = W8 ? al : al + cl[k];
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
        vq = fminf(fmaxf(vq, -Clds.body[k].vlim), Clds.body[k].vlim);
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
    GRX_EV(6);
#ifdef GRX_PROFILE_SECTIONS
    if (lane == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 64 + seq] = clock64() - tprev;   // duration of every sub-step
    tacc[5] += clock64() - tprev;   // whole sub-step
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// wave 0 with a lane PAIR per leg (GRX_LPE == 4): the sub-step of substep_p with every 6-vector and the 6 x 6 articulated
// inertia split by ROWS over the two lanes of the leg.  The "lo" half owns the angular rows -- [A B], U_a, p_a, I^a c (angular) --
// the "hi" half the linear ones -- [B^T D], U_l, p_l, I^a c (linear).  Written once in terms of a lane's OWN and the OTHER part:
//     u_own = X s_own + Y s_oth          X = A | D (symmetric),  Y = B | B^T,  (s_own, s_oth) = (a, s) | (s, a)
//     d     = s_own . u_own, summed over the two halves (one DPP step);  X -= u_own u_own^T / d,  Y -= u_own u_oth^T / d
// so both lanes run the same instruction stream on half the data: every dot product of the bias, delta and acceleration
// recursions is a 3-term product plus one quad_perm add instead of a 6-term one.  The floating-base 6 x 6 is solved in its two
// dual Schur forms at once: lo  alpha = (A - B D^-1 B^T)^-1 (B D^-1 p_l - p_a),  hi  acc = (D - B^T A^-1 B)^-1 (B^T A^-1 p_a - p_l)
// (substep_p eliminates acc only; the results agree to rounding).
GRX_DEV V3 sel3(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
GRX_DEV V3 row0(const M3& B) { return v3(B.a00, B.a01, B.a02); }
GRX_DEV V3 row1(const M3& B) { return v3(B.a10, B.a11, B.a12); }
GRX_DEV V3 row2(const M3& B) { return v3(B.a20, B.a21, B.a22); }
template <int HF, bool W8>
GRX_DEV void substep_q(KP P, const SideConst& C, const LaneConst& LC, LaneState& st, const float tau_m[LEG],
                       SubstepOut& out, FootKin& fk_before, const PipeLds& L, int lane, int seq, long long* tacc, const SideConst& Clds) {
    const float dt = P.sim_dt;
    const bool hi = lane_half(lane) != 0;
    const float sg = hi ? -1.f : 1.f;   // Y carries +skew(h) in the lo half (B), -skew(h) in the hi half (B^T)
#ifdef GRX_PROFILE_SECTIONS
    long long tprev = clock64();
#endif
    V3 So[LEG], St[LEG], co[LEG], ct[LEG], Uo[LEG], ic[LEG], hK[LEG];
    S3 XK[LEG];
    float dinv[LEG], uu[LEG];
    GRX_EV(0);
    const R3 R0 = quat_to_R(st.qx, st.qy, st.qz, st.qw);
    R3 R3s = R0, R4s = R0;
    V3 kap3 = v3(0.f, 0.f, 0.f), kap4 = kap3;
    {   // ---- outward walk (both halves: the chain is serial); own / other parts picked per lane
        R3 R = R0;
        V3 rho = v3(0.f, 0.f, 0.f), w = st.ang, v = st.vel;
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            const float qdk = st.qd[k];
            rho = rho + rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
            float sn, cs;
            grx_sincos(st.q[k], sn, cs);
            R = joint_rot_k(R, cs, sn, kAxis[k]);
            const V3 a = axis_k(R, kAxis[k]);
            const V3 s_ = cross(rho, a);
            So[k] = sel3(hi, s_, a); St[k] = sel3(hi, a, s_);

            if (!W8) {   // (eight waves: the c_k are folded into the bias forces, chain_bias_loop)
                const V3 ca = cross(w, a) * qdk;
                const V3 cl = (cross(v, a) + cross(w, s_)) * qdk;
                co[k] = sel3(hi, cl, ca); ct[k] = sel3(hi, ca, cl);
                w = fma3(a, qdk, w); v = fma3(s_, qdk, v);
            }
            if (W8 && k < LEG - 2) continue;   // eight waves: the rigid inertias of bodies 2, 1, 0 come from wave 6
            const V3 kap = rho + rot(R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            if (W8) {   // ... and those of shank and foot are computed in ONE pass below, a body per half
                hK[k] = kap * C.body[k].mass;
                if (k == LEG - 2) { R3s = R; kap3 = kap; }
                else { R4s = R; kap4 = kap; }
                continue;
            }
            const S3 Ic = {Clds.body[k].Ic[0], Clds.body[k].Ic[1], Clds.body[k].Ic[2], Clds.body[k].Ic[3], Clds.body[k].Ic[4], Clds.body[k].Ic[5]};
            S3 Ak;
            rigid_inertia(R, kap, C.body[k].mass, Ic, Ak, hK[k]);
            const float m = C.body[k].mass;
            XK[k].xx = hi ? m : Ak.xx; XK[k].xy = hi ? 0.f : Ak.xy; XK[k].xz = hi ? 0.f : Ak.xz;
            XK[k].yy = hi ? m : Ak.yy; XK[k].yz = hi ? 0.f : Ak.yz; XK[k].zz = hi ? m : Ak.zz;
        }
    }
    if (W8) {   // rotational inertia about O of the foot (lo half) and of the shank (hi half: the lo half needs it, one DPP step away)
        const int kb = hi ? LEG - 2 : LEG - 1;
        R3 Rs;
        Rs.cx = sel3(hi, R3s.cx, R4s.cx); Rs.cy = sel3(hi, R3s.cy, R4s.cy); Rs.cz = sel3(hi, R3s.cz, R4s.cz);
        const V3 kap = sel3(hi, kap3, kap4);
        const float ms = hi ? C.body[LEG - 2].mass : C.body[LEG - 1].mass;
        const S3 Ic = {Clds.body[kb].Ic[0], Clds.body[kb].Ic[1], Clds.body[kb].Ic[2], Clds.body[kb].Ic[3], Clds.body[kb].Ic[4], Clds.body[kb].Ic[5]};
        S3 Ak; V3 h_;
        rigid_inertia(Rs, kap, ms, Ic, Ak, h_);
        const S3 At = {half_swap(Ak.xx), half_swap(Ak.xy), half_swap(Ak.xz), half_swap(Ak.yy), half_swap(Ak.yz), half_swap(Ak.zz)};
        const float m4 = C.body[LEG - 1].mass, m3 = C.body[LEG - 2].mass;
        XK[LEG - 1].xx = hi ? m4 : Ak.xx; XK[LEG - 1].xy = hi ? 0.f : Ak.xy; XK[LEG - 1].xz = hi ? 0.f : Ak.xz;
        XK[LEG - 1].yy = hi ? m4 : Ak.yy; XK[LEG - 1].yz = hi ? 0.f : Ak.yz; XK[LEG - 1].zz = hi ? m4 : Ak.zz;
        XK[LEG - 2].xx = hi ? m3 : At.xx; XK[LEG - 2].xy = hi ? 0.f : At.xy; XK[LEG - 2].xz = hi ? 0.f : At.xz;
        XK[LEG - 2].yy = hi ? m3 : At.yy; XK[LEG - 2].yz = hi ? 0.f : At.yz; XK[LEG - 2].zz = hi ? m3 : At.zz;
    }
    GRX_EV(1);
    // ---- inward pass, inertia half (leaf -> root), rows split over the two halves
    S3 X = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    M3 Y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_skew = [&](V3 h) {   // Y += sg * skew(h)
        Y.a01 = fmaf(-sg, h.z, Y.a01); Y.a02 = fmaf(sg, h.y, Y.a02); Y.a10 = fmaf(sg, h.z, Y.a10);
        Y.a12 = fmaf(-sg, h.x, Y.a12); Y.a20 = fmaf(-sg, h.y, Y.a20); Y.a21 = fmaf(sg, h.x, Y.a21);
    };
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        if (W8 && k < LEG - 2) {
            if (k == LEG - 3) GRX_WAIT(L.flag + FL_XK, seq * 4 + 2, 4);   // bodies 2 and 1 come together,
            if (k == 0) GRX_WAIT(L.flag + FL_XK, seq * 4 + 3, 4);         // body 0 a little later
            // (wave 5 computes body 2 on the lo half of the leg, body 1 on the hi half, body 0 on both)
            const float4* c = L.xk + (k * 3) * 64 + (k == 0 ? lane : (k == 2 ? lane - lane_half(lane) : lane - lane_half(lane) + 1));
            const float4 a0 = c[0 * 64], a1 = c[1 * 64], a2 = c[2 * 64];
            const float m = C.body[k].mass;
            XK[k].xx = hi ? m : a0.x; XK[k].xy = hi ? 0.f : a0.y; XK[k].xz = hi ? 0.f : a0.z;
            XK[k].yy = hi ? m : a0.w; XK[k].yz = hi ? 0.f : a1.x; XK[k].zz = hi ? m : a1.y;
            hK[k] = v3(a1.z, a1.w, a2.x);
        }
        X = X + XK[k];
        add_skew(hK[k]);
        const V3 u = mul(X, So[k]) + mul(Y, St[k]);
        const float di = grx_rcp(half_sum(dot(So[k], u)));
        const V3 uo = half_swap(u);
        syr(X, u, di); ger(Y, u, uo, di);
        if (!W8) ic[k] = mul(X, co[k]) + mul(Y, ct[k]);
        Uo[k] = u; dinv[k] = di;
    }
    GRX_EV(3);
    // base level: both legs (quad_perm [2,3,0,1]) + the base lump, both Schur complements factorised
    X = pair_sum(X); Y = pair_sum(Y);
    S3 Xio;   // inverse of the OTHER half's diagonal block: lo holds D^-1, hi holds A^-1
    S3 Sci;
    if (W8) {   // eight waves: wave 5 adds the base lump and factorises (base_service_loop) while this wave runs the bias half
        float4* o = L.fx + lane;
        o[0 * 64] = f4(X.xx, X.xy, X.xz, X.yy);
        o[1 * 64] = f4(X.yz, X.zz, Y.a00, Y.a01);
        o[2 * 64] = f4(Y.a02, Y.a10, Y.a11, Y.a12);
        o[3 * 64] = f4(Y.a20, Y.a21, Y.a22, 0.f);
        flag_set(L.flag + FL_FACT, seq + 1, lane);
    } else {
    {
        S3 A0; V3 h0;
        rigid_inertia(R0, rot(R0, LC.base_c), LC.base_m, LC.base_I, A0, h0);
        const float m = LC.base_m;
        X.xx += hi ? m : A0.xx; X.xy += hi ? 0.f : A0.xy; X.xz += hi ? 0.f : A0.xz;
        X.yy += hi ? m : A0.yy; X.yz += hi ? 0.f : A0.yz; X.zz += hi ? m : A0.zz;
        add_skew(h0);
    }
    GRX_EV(14);
    {
        const S3 Xi = inv(X);
        Xio.xx = half_swap(Xi.xx); Xio.xy = half_swap(Xi.xy); Xio.xz = half_swap(Xi.xz);
        Xio.yy = half_swap(Xi.yy); Xio.yz = half_swap(Xi.yz); Xio.zz = half_swap(Xi.zz);
    }
    {
        const V3 y0 = row0(Y), y1 = row1(Y), y2 = row2(Y);
        const V3 t0 = mul(Xio, y0), t1 = mul(Xio, y1), t2 = mul(Xio, y2);
        const S3 Sc = {X.xx - dot(y0, t0), X.xy - dot(y0, t1), X.xz - dot(y0, t2), X.yy - dot(y1, t1), X.yz - dot(y1, t2), X.zz - dot(y2, t2)};
        Sci = inv(Sc);
    }
    // (keep all of the above in front of the spin-waits: see substep_p)
    GRX_PIN(Sci.xx); GRX_PIN(Sci.xy); GRX_PIN(Sci.xz); GRX_PIN(Sci.yy); GRX_PIN(Sci.yz); GRX_PIN(Sci.zz);
    GRX_PIN(Xio.xx); GRX_PIN(Xio.xy); GRX_PIN(Xio.xz); GRX_PIN(Xio.yy); GRX_PIN(Xio.yz); GRX_PIN(Xio.zz);
    }
#ifdef GRX_PROFILE_SECTIONS
    if (seq == 5 && lane == 0) { __builtin_amdgcn_sched_barrier(0); P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 79] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#endif
    if (!W8) {
#pragma unroll
        for (int k = 0; k < LEG; ++k) { GRX_PIN(ic[k].x); GRX_PIN(ic[k].y); GRX_PIN(ic[k].z); }
    }
    GRX_EV(2);
    // ---- inward pass, bias half (leaf -> root): rigid-body bias forces from waves 2 / 1, own rows only.  (Folding this chain of
    // dependent dot / DPP / fma steps into the loop above was tried: no gain -- a lone wave issues one VALU instruction per ~4.5
    // cycles whatever its dependencies, measured with tools/micro/operand_rate.hip -- and the wait for the bias forces moves up.)
    V3 po = v3(0.f, 0.f, 0.f);
    if (W8 && GRX_W8_WAITALL) GRX_WAIT_ALL(L.flag, flag_want(lane, FL_BIAS, seq * 8 + 2, FL_BIAS2, seq + 1), lane, 0);
    else if (W8) { GRX_WAIT(L.flag + FL_BIAS, seq * 8 + 2, 0); GRX_WAIT(L.flag + FL_BIAS2, seq + 1, 0); }
    else {
        GRX_WAIT(L.flag + FL_BIAS, seq * 8 + (HF ? 2 : LEG), 0);
        if (HF) GRX_WAIT(L.flag + FL_BIAS2, seq + 1, 0);
    }
    float4 bq0[LEG], bq1[LEG];
#pragma unroll
    for (int k = 0; k < LEG; ++k) {   // (eight waves: bodies 4, 2 were computed on the lo half of the leg, 3, 1 on the hi half, 0 on both)
        const int slot = !W8 || k == 0 ? lane : lane - lane_half(lane) + (k & 1);
        const float4* b_ = L.pb + (k * PB4) * 64 + slot; bq0[k] = b_[0 * 64]; bq1[k] = b_[1 * 64];
    }
#pragma unroll
    for (int k = LEG - 1; k >= 0; --k) {
        const float4 b0_ = bq0[k], b1_ = bq1[k];
        po = po + sel3(hi, v3(b0_.w, b1_.x, b1_.y), v3(b0_.x, b0_.y, b0_.z));
        const float u = (tau_m[k] + b1_.z) - half_sum(dot(So[k], po));   // motor torque + joint-limit spring/damper - S . p
        uu[k] = u;
        if (!W8) po = po + ic[k];
        po = fma3(Uo[k], u * dinv[k], po);
    }
    GRX_EV(7);
    // ---- contact wrenches on chain bodies 4, 3, 2: delta recursion on the own rows
    if (W8 && GRX_W8_WAITALL) {   // everything the rest of the sub-step consumes, in one poll
#ifndef GRX_W8_LATEFACT
#define GRX_W8_LATEFACT 1   // the factorisation (the last hand-over to arrive: 4.9 k cycles into the sub-step against 4.1 k for the contact wrenches) is waited for where it is used, behind the delta recursion
#endif
        if (GRX_W8_LATEFACT) GRX_WAIT_ALL(L.flag, flag_want(lane, FL_FOOT, seq + 1, FL_LEGS, seq + 1, FL_SELF, seq + 1, FL_BASEBIAS, seq + 1, FL_CHAINW, seq + 1, FL_SB, seq + 1), lane, 2);
        else GRX_WAIT_ALL(L.flag, flag_want(lane, FL_FOOT, seq + 1, FL_LEGS, seq + 1, FL_SELF, seq + 1, FL_BASEBIAS, seq + 1, FL_FACTOUT, seq + 1, FL_CHAINW, seq + 1, FL_SB, seq + 1), lane, 2);
        GRX_EV(5);
    } else {
        if (W8) { GRX_WAIT(L.flag + FL_BASEBIAS, seq + 1, 0); GRX_WAIT(L.flag + FL_FACTOUT, seq + 1, 1); GRX_WAIT(L.flag + FL_CHAINW, seq + 1, 2); }
        GRX_WAIT(L.flag + FL_FOOT, seq + 1, 2);
        GRX_EV(4);
        GRX_WAIT(L.flag + FL_LEGS, seq + 1, 2);
        GRX_EV(5);
        GRX_WAIT(L.flag + FL_SELF, seq + 1, 3);
    }
    V3 sco[3], sc0o;   // self-collision wrenches (wave 1), own rows
    V3 scfl2;          // ... and the force on the foot link
    {
        const float4* c = L.wc + 7 * 64 + lane;
        const float4 s0 = c[0 * 64], s1 = c[1 * 64], s2 = c[2 * 64], s3 = c[3 * 64], s4 = c[4 * 64], s5 = c[5 * 64];
        sco[0] = sel3(hi, v3(s0.w, s1.x, s1.y), v3(s0.x, s0.y, s0.z));
        sco[1] = sel3(hi, v3(s2.y, s2.z, s2.w), v3(s1.z, s1.w, s2.x));
        scfl2 = v3(s3.w, s4.x, s4.y);
        sco[2] = sel3(hi, scfl2, v3(s3.x, s3.y, s3.z));
        sc0o = sel3(hi, v3(s5.y, s5.z, s5.w), v3(s4.z, s4.w, s5.x));
        if (W8) {   // the thigh x base-lump part (wave 3, with its FL_LEGS hand-over)
            const float4* d = L.sb + lane;
            const float4 d0 = d[0 * 64], d1 = d[1 * 64], d2 = d[2 * 64];
            sco[0] = sco[0] + sel3(hi, v3(d0.w, d1.x, d1.y), v3(d0.x, d0.y, d0.z));
            sc0o = sc0o + sel3(hi, v3(d2.y, d2.z, d2.w), v3(d1.z, d1.w, d2.x));
        }
    }
    {
        V3 dlt = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            if (k >= 2) {
                const float4* c = L.wc + ((k - 2) * 2) * 64 + lane;
                const float4 c0_ = c[0 * 64], c1_ = c[1 * 64];
                const V3 fl = v3(c0_.w, c1_.x, c1_.y);
                dlt = dlt - sel3(hi, fl, v3(c0_.x, c0_.y, c0_.z)) - sco[k - 2];
                if (k == LEG - 1) { const float4 c2_ = c[2 * 64]; out.foot_force = fl + scfl2; fk_before.vel = v3(c1_.z, c1_.w, c2_.x); }
            }
            const float du = -half_sum(dot(So[k], dlt));
            uu[k] += du;
            dlt = fma3(Uo[k], du * dinv[k], dlt);
        }
        po = po + dlt;
    }
    {   // base-lump contact wrench + flags (published together with the thigh / shank wrenches: FL_LEGS)
        const float4 w0_ = L.wr[lane], w1_ = L.wr[64 + lane];
        out.term = w1_.z != 0.f;
        out.pen_count = w1_.w;
        po = po - sel3(hi, v3(w0_.w, w1_.x, w1_.y), v3(w0_.x, w0_.y, w0_.z)) - sc0o;
    }
    po = pair_sum(po);
    {   // base-lump bias force (wave 3; the lanes of both legs add the same value after the leg sum)
        if (!W8) GRX_WAIT(L.flag + FL_BASEBIAS, seq + 1, 0);
        const float4* b_ = L.pb + (LEG * PB4) * 64 + lane;
        const float4 b0_ = b_[0 * 64], b1_ = b_[1 * 64];
        po = po + sel3(hi, v3(b0_.w, b1_.x, b1_.y), v3(b0_.x, b0_.y, b0_.z));
    }
    // lo: alpha = Sa^-1 (B D^-1 p_l - p_a);  hi: acc = Sd^-1 (B^T A^-1 p_a - p_l)
    V3 xo;
    if (W8) {   // x = Sc^-1 (T p_other - p_own) with T = Y Xo^-1 and Sc^-1 from wave 5
        if (GRX_W8_WAITALL && GRX_W8_LATEFACT) {
            GRX_WAIT(L.flag + FL_FACTOUT, seq + 1, 1);
        }
        const float4* c = L.fx + 4 * 64 + lane;
        const float4 m0 = c[0 * 64], m1 = c[1 * 64], m2 = c[2 * 64], m3 = c[3 * 64];
        const V3 pt = half_swap(po);
        const S3 Si = {m2.y, m2.z, m2.w, m3.x, m3.y, m3.z};
        xo = mul(Si, v3(m0.x * pt.x + m0.y * pt.y + m0.z * pt.z, m0.w * pt.x + m1.x * pt.y + m1.y * pt.z, m1.z * pt.x + m1.w * pt.y + m2.x * pt.z) - po);
    } else {
        xo = mul(Sci, mul(Y, mul(Xio, half_swap(po))) - po);
    }
    // ---- pass 3 (root -> leaf): accelerations, own rows
    float qdd[LEG];
    {
        V3 ao = xo;
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            const V3 p_ = W8 ? ao : ao + co[k];
            const float qd2 = (uu[k] - half_sum(dot(Uo[k], p_))) * dinv[k];
            qdd[k] = qd2;
            ao = fma3(So[k], qd2, p_);
        }
    }
    // ---- integrate (semi-implicit Euler), both halves alike
    const V3 xt = half_swap(xo);
    const V3 alpha = sel3(hi, xt, xo), acc = sel3(hi, xo, xt);
    const V3 lin = acc + cross(st.ang, st.vel);  // classical acceleration of the base origin
    st.vel = v3(st.vel.x + (lin.x + P.gravity[0]) * dt, st.vel.y + (lin.y + P.gravity[1]) * dt, st.vel.z + (lin.z + P.gravity[2]) * dt);
    st.ang = fma3(alpha, dt, st.ang);
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        float vq = fmaf(qdd[k], dt, st.qd[k]);
        vq = fminf(fmaxf(vq, -Clds.body[k].vlim), Clds.body[k].vlim);
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
    GRX_EV(6);
#ifdef GRX_PROFILE_SECTIONS
    if (lane == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 64 + seq] = clock64() - tprev;   // duration of every sub-step
    tacc[5] += clock64() - tprev;   // whole sub-step
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// wave 1: self-collision (grx_self.h) -- leg against leg (the partner lane is one DPP step away), thigh against base-lump
// shapes -- on the chain frames wave 2 publishes right after its walk
// On a HEIGHTFIELD the foot wave (wave 2) is the late one (gathers, terrain normals), so this wave -- otherwise idle until
// wave 2's frames arrive -- walks the chain itself (same arithmetic, same frames) and takes over the bias forces of the
// bodies wave 0 reaches last: thigh, hip yaw, hip roll.
// `idle(seq)` runs after the hand-over of every sub-step, in the ~2 k cycles this wave then waits for wave 0's next state.
// OWNPOS (eight waves): positions-only walk of its own for the sphere centres and pair tests; the bodies' velocities (contact damping only)
// are picked up from wave 2's frames, which are out by the time the centres are staged
template <int HF, bool WALK, bool OWNPOS, class Idle>   // WALK: four waves on a heightfield (see above); else the frames come from wave 2
GRX_DEV void self_loop(KP P, const KTables& T, const SideConst& C, const RareBuf& RB, const float4* footfr, const SelfBuf& SB, float mu_self,
                       const PipeLds& L, int lane, int el, int side, Idle idle) {
    GRX_HELPER_PROF_BEGIN;
    SelfNear sn; sn.m = 0;   // self-collision broad phase of this policy step
#ifdef GRX_PROFILE_SECTIONS
    long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
    long long* sacc = nullptr;
#endif
    float lim_lo[3], lim_hi[3], lim_k[3], lim_c[3];   // joint-limit constants of joints 0..2 (HF)
#pragma unroll
    for (int k = 0; k < 3; ++k) { lim_lo[k] = C.body[k].qlo; lim_hi[k] = C.body[k].qhi; lim_k[k] = C.body[k].Klim; lim_c[k] = C.body[k].Clim; }
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        const V3 vel = v3(b[7], b[8], b[9]), ang = v3(b[10], b[11], b[12]);
        ChainKin KS[3];
        if (WALK) {
            float qs_q[LEG], qs_qd[LEG];
            {
                const float4 q0_ = L.q[lane], q1_ = L.q[64 + lane], q2_ = L.q[128 + lane];
                qs_q[0] = q0_.x; qs_q[1] = q0_.y; qs_q[2] = q0_.z; qs_q[3] = q0_.w; qs_q[4] = q1_.x;
                qs_qd[0] = q1_.y; qs_qd[1] = q1_.z; qs_qd[2] = q1_.w; qs_qd[3] = q2_.x; qs_qd[4] = q2_.y;
            }
            ChainKin K = {R0, v3(0.f, 0.f, 0.f), ang, vel};
            ChainKin KK[3];
#pragma unroll
            for (int k = 0; k < LEG; ++k) {
                chain_step(C, k, qs_q[k], qs_qd[k], K);
                if (k < 3) KK[k] = K;
                if (k >= 2) KS[k - 2] = K;
            }
#pragma unroll
            for (int k = 2; k >= 0; --k) {   // the hand-over wave 2 makes for the other bodies (chain_contact_loop: bias_out)
                const V3 kap = KK[k].rho + rot(KK[k].R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
                const S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
                V3 pa, pl;
                rigid_bias(KK[k].R, kap, C.body[k].mass, Ic, KK[k].w, KK[k].v, pa, pl);
                const float viol = qs_q[k] < lim_lo[k] ? lim_lo[k] - qs_q[k] : (qs_q[k] > lim_hi[k] ? lim_hi[k] - qs_q[k] : 0.f);
                const float tlim = lim_k[k] * viol - (viol != 0.f ? lim_c[k] * qs_qd[k] : 0.f);
                float4* o = L.pb + (k * PB4) * 64 + lane;
                o[0 * 64] = f4(pa.x, pa.y, pa.z, pl.x);
                o[1 * 64] = f4(pl.y, pl.z, tlim, 0.f);
            }
            flag_set(L.flag + FL_BIAS2, seq + 1, lane);
        } else if (OWNPOS) {
            const float4 q0_ = L.q[lane], q1_ = L.q[64 + lane];
            const float qs[LEG] = {q0_.x, q0_.y, q0_.z, q0_.w, q1_.x};
            ChainKin K = {R0, v3(0.f, 0.f, 0.f), ang, vel};
#pragma unroll
            for (int k = 0; k < LEG; ++k) {
                K.rho = K.rho + rot(K.R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
                float sn, cs;
                grx_sincos(qs[k], sn, cs);
                K.R = joint_rot_k(K.R, cs, sn, kAxis[k]);
                if (k >= 2) KS[k - 2] = K;
            }
        } else {
            flag_wait(L.flag + FL_FRAMES, seq + 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const RareFrame f = rare_load_frame(i < 2 ? RB.fchain + i * RC_FR4 * 64 + lane : footfr + lane, 64);
                KS[i].R = f.R; KS[i].rho = f.rho; KS[i].w = f.w; KS[i].v = f.v;
            }
        }
        const int fr_want = seq + 1;
        int* const fr_flag = L.flag + FL_FRAMES;
        auto velocities = [&]() {   // OWNPOS: (w, v) of thigh, shank, foot from wave 2's frames, once the geometry is done with
            if (!OWNPOS) return;
            flag_wait(fr_flag, fr_want);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4* f = i < 2 ? RB.fchain + i * RC_FR4 * 64 + lane : footfr + lane;
                const float4 d = f[3 * 64], e_ = f[4 * 64];
                KS[i].w = v3(d.x, d.y, d.z); KS[i].v = v3(d.w, e_.x, e_.y);
            }
        };
        if (seq == 0) sn = OWNPOS ? self_broad_phase<1>(P, C, side, R0, KS) : self_broad_phase(P, C, side, R0, KS);   // (eight waves: the other part on the wave that evaluates it)
        SelfOut sc;
        if (OWNPOS) self_collision<decltype(velocities), 1>(P, T, C, SB, lane, side, R0, ang, vel, KS, mu_self, sn, sc, sacc, velocities);   // (eight waves: thigh x base lump on wave 3)
        else self_collision(P, T, C, SB, lane, side, R0, ang, vel, KS, mu_self, sn, sc, sacc, velocities);
        float4* o = L.wc + 7 * 64 + lane;
        o[0 * 64] = f4(sc.fa[0].x, sc.fa[0].y, sc.fa[0].z, sc.fl[0].x);
        o[1 * 64] = f4(sc.fl[0].y, sc.fl[0].z, sc.fa[1].x, sc.fa[1].y);
        o[2 * 64] = f4(sc.fa[1].z, sc.fl[1].x, sc.fl[1].y, sc.fl[1].z);
        o[3 * 64] = f4(sc.fa[2].x, sc.fa[2].y, sc.fa[2].z, sc.fl[2].x);
        o[4 * 64] = f4(sc.fl[2].y, sc.fl[2].z, sc.f0a.x, sc.f0a.y);
        o[5 * 64] = f4(sc.f0a.z, sc.f0l.x, sc.f0l.y, sc.f0l.z);
        o[6 * 64] = f4(sc.fbase[0].x, sc.fbase[0].y, sc.fbase[0].z, sc.fbase[1].x);
        o[7 * 64] = f4(sc.fbase[1].y, sc.fbase[1].z, 0.f, 0.f);
        flag_set(L.flag + FL_SELF, seq + 1, lane);
        GRX_EV(13);
        idle(seq);
    }
    GRX_HELPER_PROF_END(1);
#ifdef GRX_PROFILE_SECTIONS
    if (lane == 0) for (int i = 0; i < 8; ++i) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 40 + i] = sacc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Eight waves per block (lane quads only; two waves per SIMD, each still issuing one instruction per ~4.5 cycles): the work the
// four-wave layout hangs on whichever wave has slack gets waves of its own, and wave 0 sheds what others can finish in time.
// The thigh x base-lump pairs of the self-collision (grx_self.h, PARTS = 2) on wave 2's thigh frame, with a broad phase of its own at
// the first sub-step: wave 3 with lane quads (after its base-lump shapes), wave 6 with lane pairs (after its bias forces: wave 3 tests
// eight shapes per lane there and is the last helper to finish as it is).  Record L.sb, hand-over FL_SB.
GRX_DEV void thigh_base_self_step(KP P, const KTables& T, const SideConst& C /* may be a register copy */, const RareBuf& RB, const float4* footfr, const PipeLds& L,
                                  int lane, int side, int seq, const R3& R0, V3 ang, V3 vel, float mu_self, SelfNear& sn3) {
    flag_wait(L.flag + FL_FRAMES, seq + 1);
    ChainKin KS[3];
    {
        const RareFrame f = rare_load_frame(RB.fchain + lane, 64);
        KS[0].R = f.R; KS[0].rho = f.rho; KS[0].w = f.w; KS[0].v = f.v;
        KS[1] = KS[0]; KS[2] = KS[0];
    }
    if (seq == 0) sn3 = self_broad_phase<2>(P, T.side[side], side, R0, KS);   // (this part looks at the thigh only; once per policy step: from the LDS table)
    SelfOut sb;
    const SelfBuf nosb = {nullptr, nullptr, nullptr};
    self_collision<SelfNoVel, 2>(P, T, C, nosb, lane, side, R0, ang, vel, KS, mu_self, sn3, sb);
    float4* o = L.sb + lane;
    o[0 * 64] = f4(sb.fa[0].x, sb.fa[0].y, sb.fa[0].z, sb.fl[0].x);
    o[1 * 64] = f4(sb.fl[0].y, sb.fl[0].z, sb.f0a.x, sb.f0a.y);
    o[2 * 64] = f4(sb.f0a.z, sb.f0l.x, sb.f0l.y, sb.f0l.z);
    o[3 * 64] = f4(sb.fbase[0].x, sb.fbase[0].y, sb.fbase[0].z, sb.fbase[1].x);
    o[4 * 64] = f4(sb.fbase[1].y, sb.fbase[1].z, 0.f, 0.f);
    flag_set(L.flag + FL_SB, seq + 1, lane);
}

// waves 4 and 6: own walk with velocities, then the bias forces of the chain bodies KHI .. KLO (wave 4: foot, shank; wave 6: thigh,
// hip yaw, hip roll) -- wave 2 keeps the foot contacts only.  The velocity-product accelerations c_k of the joints are FOLDED into
// these forces: with zeta_k = sum of c_j over the joints up to k (all about O in world axes, so a plain sum) and a_k = a^_k + zeta_k,
// body k obeys f_k = I_k a^_k + (p_k + I_k zeta_k) and the joints a^_k = a^_(k-1) + S_k qdd_k: the articulated-body recursion in a^
// has NO c terms -- wave 0 drops I^a c (18 instructions per joint), the c_k themselves and the chain velocities of its walk -- and
// the rigid-body bias force becomes the Newton-Euler force of body k moving with (w, v) and accelerating with zeta_k.
struct ChainKinZ { ChainKin K; V3 za, zl; };
GRX_DEV void chain_step_z(const SideConst& C, int k, float q, float qd, ChainKinZ& Z) {
    ChainKin& K = Z.K;
    K.rho = K.rho + rot(K.R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
    float sn, cs;
    grx_sincos(q, sn, cs);
    K.R = joint_rot_k(K.R, cs, sn, kAxis[k]);
    const V3 a = axis_k(K.R, kAxis[k]);
    const V3 s = cross(K.rho, a);
    Z.za = fma3(cross(K.w, a), qd, Z.za);
    Z.zl = fma3(cross(K.v, a) + cross(K.w, s), qd, Z.zl);
    K.w = fma3(a, qd, K.w); K.v = fma3(s, qd, K.v);
}
// F = m (zl + za x kap) + w x l, l = m (v + w x kap);  torque about the centre of mass n = R (Ic zb + wb x Ic wb);  about O: n + kap x F
GRX_DEV void rigid_bias_z(const R3& R, V3 kap, float m, const S3& Ic, V3 w, V3 v, V3 za, V3 zl, V3& pa, V3& pl) {
    const V3 wb = rotT(R, w), zb = rotT(R, za);
    const V3 nb = mul(Ic, zb) + cross(wb, mul(Ic, wb));
    const V3 l = (v + cross(w, kap)) * m;
    pl = cross(w, l) + (zl + cross(za, kap)) * m;
    pa = rot(R, nb) + cross(kap, pl);
}
// The two lanes of a leg compute DIFFERENT bodies in one pass (lo half: body KHI, hi half: body KHI - 1 -- same instructions, selected
// inputs; the table constants come from LDS with a lane-dependent index), a body left over (KLO, three bodies) on both halves: wave 0
// reads the record of body k from the slot of the lane that wrote it.
template <int KLO, int KHI, bool SELFB = false>   // SELFB: then the thigh x base-lump self-collision (lane pairs, wave 6)
GRX_DEV void chain_bias_loop(KP P, const SideConst& C, const SideConst& Clds, const PipeLds& L, int lane, int el,
                             const KTables* T = nullptr, const RareBuf* RB = nullptr, const float4* footfr = nullptr, int side = 0, float mu_self = 0.f) {
    constexpr int NB = KHI - KLO + 1;
    static_assert(NB == 2 || NB == 3, "a pair of bodies, or a pair and a single one");
    const bool hi = lane_half(lane) != 0;
    SelfNear sn3; sn3.m = 0;
    for (int seq = 0; seq < P.decimation; ++seq) {
        flag_wait(L.flag + FL_STATE, seq + 1);
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        const V3 vel = v3(b[7], b[8], b[9]), ang = v3(b[10], b[11], b[12]);
        float qs_q[LEG], qs_qd[LEG];
        {
            const float4 q0_ = L.q[lane], q1_ = L.q[64 + lane], q2_ = L.q[128 + lane];
            qs_q[0] = q0_.x; qs_q[1] = q0_.y; qs_q[2] = q0_.z; qs_q[3] = q0_.w; qs_q[4] = q1_.x;
            qs_qd[0] = q1_.y; qs_qd[1] = q1_.z; qs_qd[2] = q1_.w; qs_qd[3] = q2_.x; qs_qd[4] = q2_.y;
        }
        ChainKinZ Z = {{R0, v3(0.f, 0.f, 0.f), ang, vel}, v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
        ChainKinZ ZZ[NB];
#pragma unroll
        for (int k = 0; k <= KHI; ++k) { chain_step_z(C, k, qs_q[k], qs_qd[k], Z); if (k >= KLO) ZZ[k - KLO] = Z; }
        auto bias_out = [&](const ChainKinZ& B, const int kb, const float q, const float qd) {   // kb may differ between the halves
            const BodyC& T_ = Clds.body[kb];
            const V3 kap = B.K.rho + rot(B.K.R, v3(T_.com[0], T_.com[1], T_.com[2]));
            const S3 Ic = {T_.Ic[0], T_.Ic[1], T_.Ic[2], T_.Ic[3], T_.Ic[4], T_.Ic[5]};
            V3 pa, pl;
            rigid_bias_z(B.K.R, kap, T_.mass, Ic, B.K.w, B.K.v, B.za, B.zl, pa, pl);
            const float lo_ = T_.qlo, hi_ = T_.qhi;
            const float viol = q < lo_ ? lo_ - q : (q > hi_ ? hi_ - q : 0.f);
            const float tlim = T_.Klim * viol - (viol != 0.f ? T_.Clim * qd : 0.f);
            float4* o = L.pb + (kb * PB4) * 64 + lane;
            o[0 * 64] = f4(pa.x, pa.y, pa.z, pl.x);
            o[1 * 64] = f4(pl.y, pl.z, tlim, 0.f);
        };
        if (LPL == 1) {   // a lane per leg: the bodies one after the other
#pragma unroll
            for (int k = KHI; k > KLO; --k) bias_out(ZZ[k - KLO], k, qs_q[k], qs_qd[k]);
            if (NB == 2) bias_out(ZZ[0], KLO, qs_q[KLO], qs_qd[KLO]);
        } else {   // bodies KHI (lo half) and KHI - 1 (hi half)
            const ChainKinZ &A = ZZ[NB - 1], &Bq = ZZ[NB - 2];
            ChainKinZ S_;
            S_.K.R.cx = sel3(hi, Bq.K.R.cx, A.K.R.cx); S_.K.R.cy = sel3(hi, Bq.K.R.cy, A.K.R.cy); S_.K.R.cz = sel3(hi, Bq.K.R.cz, A.K.R.cz);
            S_.K.rho = sel3(hi, Bq.K.rho, A.K.rho); S_.K.w = sel3(hi, Bq.K.w, A.K.w); S_.K.v = sel3(hi, Bq.K.v, A.K.v);
            S_.za = sel3(hi, Bq.za, A.za); S_.zl = sel3(hi, Bq.zl, A.zl);
            bias_out(S_, hi ? KHI - 1 : KHI, hi ? qs_q[KHI - 1] : qs_q[KHI], hi ? qs_qd[KHI - 1] : qs_qd[KHI]);
        }
        if (NB == 3) bias_out(ZZ[0], KLO, qs_q[KLO], qs_qd[KLO]);
        if (KHI == LEG - 1) { flag_set(L.flag + FL_BIAS, seq * 8 + 2, lane); GRX_EV(26); }
        else { flag_set(L.flag + FL_BIAS2, seq + 1, lane); GRX_EV(28); }
        if (SELFB) thigh_base_self_step(P, *T, Clds, *RB, footfr, L, lane, side, seq, R0, ang, vel, mu_self, sn3);
    }
}

// wave 7: the thigh / shank shapes of the seldom-touching set (grx_rare.h), on the frames wave 2 publishes, with a compaction
// buffer of its own; wave 3 keeps the base-lump shapes, which need the base state only
template <int HF>
GRX_DEV void chain_rare_loop(KP P, const KTables& T, const SideConst& C, const RareBuf& RB, float mu, float hmax, const PipeLds& L, int lane, int el, int side) {
    for (int seq = 0; seq < P.decimation; ++seq) {
        flag_wait(L.flag + FL_STATE, seq + 1);
        if (!GRX_W8_RARESPLIT) { flag_set(L.flag + FL_CHAINW, seq + 1, lane); continue; }
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const V3 O = v3(b[0], b[1], b[2]);
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        const V3 zero = v3(0.f, 0.f, 0.f);
        // own positions-only walk to thigh and shank: the reach tests start ~0.6 k cycles before wave 2's frames (with velocities) are out
        // (taking R, rho from wave 0's walk instead was tried: the 15 record stores and two hand-overs cost wave 0 0.7 k cycles per sub-step)
        ChainKin K3 = {R0, zero, zero, zero}, K2 = K3;
        {
            const float4 q0_ = L.q[lane];
            const float qs[4] = {q0_.x, q0_.y, q0_.z, q0_.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                K3.rho = K3.rho + rot(K3.R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
                float sn, cs;
                grx_sincos(qs[k], sn, cs);
                K3.R = joint_rot_k(K3.R, cs, sn, kAxis[k]);
                if (k == 2) K2 = K3;
            }
        }
        RareOut ro;
        const int want = seq + 1;
        int* const fr_flag = L.flag + FL_FRAMES;
        rare_contacts<HF, 8, RC_NS, true, true>(P, T, C, RB, lane, el, side, R0, O, zero, zero, K2, K3, mu, hmax, ro, nullptr,
                                                [=]() { flag_wait(fr_flag, want); }, false);
        float4* c_ = L.wc + lane;
        c_[0 * 64] = f4(ro.fa2.x, ro.fa2.y, ro.fa2.z, ro.fl2.x);
        c_[1 * 64] = f4(ro.fl2.y, ro.fl2.z, 0.f, 0.f);
        c_[2 * 64] = f4(ro.fa3.x, ro.fa3.y, ro.fa3.z, ro.fl3.x);
        c_[3 * 64] = f4(ro.fl3.y, ro.fl3.z, 0.f, 0.f);
        flag_set(L.flag + FL_CHAINW, seq + 1, lane);
        GRX_EV(14);
    }
}

// wave 5: everything at the floating base that is not on wave 0's chain -- the base lump's bias force and rigid inertia, and the
// factorisation of the base-level 6 x 6 (substep_q's dual Schur forms) from the X, Y wave 0 hands over after its inertia half.
// Returns T = Y Xo^-1 and Sc^-1: wave 0's solve is then x = Sc^-1 (T p_other - p_own).
GRX_DEV void base_service_loop(KP P, const SideConst& C, const SideConst& Clds, float base_m, V3 base_c, const S3& base_I, const PipeLds& L, int lane, int el) {
    const bool hi = lane_half(lane) != 0;
    const float sg = hi ? -1.f : 1.f;
    for (int seq = 0; seq < P.decimation; ++seq) {
        flag_wait(L.flag + FL_STATE, seq + 1);
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        if (LPL == 2 || GRX_P8_XK) {   // first what wave 0 needs first: the rigid inertias about O of thigh, hip yaw, hip roll (its inertia half adds them at bodies 2,
            // 1, 0).  Positions-only walk of three bodies, then body 2 on the lo half of the leg and body 1 on the hi half in ONE pass.
            const float4 q0_ = L.q[lane];
            const float qs[3] = {q0_.x, q0_.y, q0_.z};
            R3 R = R0, Rk[3];
            V3 rho = v3(0.f, 0.f, 0.f), kapk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rho = rho + rot(R, v3(C.body[k].r[0], C.body[k].r[1], C.body[k].r[2]));
                float sn, cs;
                grx_sincos(qs[k], sn, cs);
                R = joint_rot_k(R, cs, sn, kAxis[k]);
                Rk[k] = R;
                kapk[k] = rho + rot(R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            }
            if (LPL == 1) {   // a lane per leg: bodies 1 and 0 one after the other (wave 0 computes body 2's itself), each into its own slot
#pragma unroll
                for (int kb = 1; kb >= 0; --kb) {
                    const S3 Ic = {C.body[kb].Ic[0], C.body[kb].Ic[1], C.body[kb].Ic[2], C.body[kb].Ic[3], C.body[kb].Ic[4], C.body[kb].Ic[5]};
                    S3 Ak; V3 h_;
                    rigid_inertia(Rk[kb], kapk[kb], C.body[kb].mass, Ic, Ak, h_);
                    float4* o = L.xk + (kb * 3) * 64 + lane;
                    o[0 * 64] = f4(Ak.xx, Ak.xy, Ak.xz, Ak.yy);
                    o[1 * 64] = f4(Ak.yz, Ak.zz, h_.x, h_.y);
                    o[2 * 64] = f4(h_.z, 0.f, 0.f, 0.f);
                }
                flag_set(L.flag + FL_XK, seq * 4 + 3, lane);
            } else {
                const int kb = hi ? 1 : 2;
                R3 Rs;
                Rs.cx = sel3(hi, Rk[1].cx, Rk[2].cx); Rs.cy = sel3(hi, Rk[1].cy, Rk[2].cy); Rs.cz = sel3(hi, Rk[1].cz, Rk[2].cz);
                const V3 kap = sel3(hi, kapk[1], kapk[2]);
                const float ms = hi ? C.body[1].mass : C.body[2].mass;
                const S3 Ic = {Clds.body[kb].Ic[0], Clds.body[kb].Ic[1], Clds.body[kb].Ic[2], Clds.body[kb].Ic[3], Clds.body[kb].Ic[4], Clds.body[kb].Ic[5]};
                S3 Ak; V3 h_;
                rigid_inertia(Rs, kap, ms, Ic, Ak, h_);
                float4* o = L.xk + (kb * 3) * 64 + lane;
                o[0 * 64] = f4(Ak.xx, Ak.xy, Ak.xz, Ak.yy);
                o[1 * 64] = f4(Ak.yz, Ak.zz, h_.x, h_.y);
                o[2 * 64] = f4(h_.z, 0.f, 0.f, 0.f);
                flag_set(L.flag + FL_XK, seq * 4 + 2, lane);
            }
            if (LPL == 2) {
                const S3 Ic = {C.body[0].Ic[0], C.body[0].Ic[1], C.body[0].Ic[2], C.body[0].Ic[3], C.body[0].Ic[4], C.body[0].Ic[5]};
                S3 Ak; V3 h_;
                rigid_inertia(Rk[0], kapk[0], C.body[0].mass, Ic, Ak, h_);
                float4* o = L.xk + lane;
                o[0 * 64] = f4(Ak.xx, Ak.xy, Ak.xz, Ak.yy);
                o[1 * 64] = f4(Ak.yz, Ak.zz, h_.x, h_.y);
                o[2 * 64] = f4(h_.z, 0.f, 0.f, 0.f);
                flag_set(L.flag + FL_XK, seq * 4 + 3, lane);
            }
            GRX_EV(27);
        }
        const V3 vel = v3(b[7], b[8], b[9]), ang = v3(b[10], b[11], b[12]);
        const V3 kap0 = rot(R0, base_c);
        {
            V3 bpa, bpl;
            rigid_bias(R0, kap0, base_m, base_I, ang, vel, bpa, bpl);
            float4* o = L.pb + (LEG * PB4) * 64 + lane;
            o[0 * 64] = f4(bpa.x, bpa.y, bpa.z, bpl.x);
            o[1 * 64] = f4(bpl.y, bpl.z, 0.f, 0.f);
            flag_set(L.flag + FL_BASEBIAS, seq + 1, lane);
        }
        if (LPL == 1) continue;   // (lane pairs: wave 0 factorises the base level itself)
        S3 A0; V3 h0;
        rigid_inertia(R0, kap0, base_m, base_I, A0, h0);
        flag_wait(L.flag + FL_FACT, seq + 1);
        GRX_EV(30);
        const float4* c = L.fx + lane;
        const float4 x0 = c[0 * 64], x1 = c[1 * 64], x2 = c[2 * 64], x3 = c[3 * 64];
        S3 X = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y};
        M3 Y = {x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z};
        X.xx += hi ? base_m : A0.xx; X.xy += hi ? 0.f : A0.xy; X.xz += hi ? 0.f : A0.xz;
        X.yy += hi ? base_m : A0.yy; X.yz += hi ? 0.f : A0.yz; X.zz += hi ? base_m : A0.zz;
        Y.a01 = fmaf(-sg, h0.z, Y.a01); Y.a02 = fmaf(sg, h0.y, Y.a02); Y.a10 = fmaf(sg, h0.z, Y.a10);
        Y.a12 = fmaf(-sg, h0.x, Y.a12); Y.a20 = fmaf(-sg, h0.y, Y.a20); Y.a21 = fmaf(sg, h0.x, Y.a21);
        S3 Xio;
        {
            const S3 Xi = inv(X);
            Xio.xx = half_swap(Xi.xx); Xio.xy = half_swap(Xi.xy); Xio.xz = half_swap(Xi.xz);
            Xio.yy = half_swap(Xi.yy); Xio.yz = half_swap(Xi.yz); Xio.zz = half_swap(Xi.zz);
        }
        const V3 y0 = row0(Y), y1 = row1(Y), y2 = row2(Y);
        const V3 t0 = mul(Xio, y0), t1 = mul(Xio, y1), t2 = mul(Xio, y2);   // rows of Y Xo^-1
        const S3 Sc = {X.xx - dot(y0, t0), X.xy - dot(y0, t1), X.xz - dot(y0, t2), X.yy - dot(y1, t1), X.yz - dot(y1, t2), X.zz - dot(y2, t2)};
        float4* o = L.fx + 4 * 64 + lane;   // (T = Y Xo^-1 goes out while Sc is being inverted: the product Sc^-1 T would sit on the chain)
        o[0 * 64] = f4(t0.x, t0.y, t0.z, t1.x);
        o[1 * 64] = f4(t1.y, t1.z, t2.x, t2.y);
        const S3 Si = inv(Sc);
        o[2 * 64] = f4(t2.z, Si.xx, Si.xy, Si.xz);
        o[3 * 64] = f4(Si.yy, Si.yz, Si.zz, 0.f);
        flag_set(L.flag + FL_FACTOUT, seq + 1, lane);
        GRX_EV(29);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wave 2: own walk with velocities; thigh / shank frames for wave 3; the anchored foot spheres
template <int HF, int NBIAS>   // NBIAS: this wave computes the bias forces of the last NBIAS chain bodies (5, 2, or 0: eight waves)
GRX_DEV void chain_contact_loop(KP P, const SideConst& C, const SideConst& Clds, const RareBuf& RB, float4* footfr, float mu, float hmax, float om_e, LaneState& hs, const PipeLds& L,
                                int lane, int el, int side, V3& rpy_acc) {   // rpy_acc: sum of the foot body's |angular velocity| at the START of sub-steps 1.. (avg_feet_speed_rpy)
    const int half = lane_half(lane);
    GRX_HELPER_PROF_BEGIN;
    float lim_lo[LEG], lim_hi[LEG], lim_k[LEG], lim_c[LEG];   // joint-limit constants: registers for the whole policy step
#pragma unroll
    for (int k = 0; k < LEG; ++k) { lim_lo[k] = C.body[k].qlo; lim_hi[k] = C.body[k].qhi; lim_k[k] = C.body[k].Klim; lim_c[k] = C.body[k].Clim; }
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const V3 O = v3(b[0], b[1], b[2]);
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        const V3 vel = v3(b[7], b[8], b[9]), ang = v3(b[10], b[11], b[12]);
        float qs_q[LEG], qs_qd[LEG];
        {
            const float4 q0_ = L.q[lane], q1_ = L.q[64 + lane], q2_ = L.q[128 + lane];
            qs_q[0] = q0_.x; qs_q[1] = q0_.y; qs_q[2] = q0_.z; qs_q[3] = q0_.w; qs_q[4] = q1_.x;
            qs_qd[0] = q1_.y; qs_qd[1] = q1_.z; qs_qd[2] = q1_.w; qs_qd[3] = q2_.x; qs_qd[4] = q2_.y;
        }
        // outward walk with velocities; thigh / shank frames for wave 3; then the rigid-body bias forces +
        // velocity-product accelerations of the chain bodies, leaf first (wave 0's bias recursion starts at the foot)
        ChainKin K = {R0, v3(0.f, 0.f, 0.f), ang, vel};
        ChainKin KK[LEG];
#pragma unroll
        for (int k = 0; k < LEG; ++k) {
            chain_step(C, k, qs_q[k], qs_qd[k], K);
            KK[k] = K;
            if (k == 2) rare_store_frame(RB.fchain + lane, 64, K.R, K.rho, K.w, K.v);                  // thigh
            if (k == 3) rare_store_frame(RB.fchain + RC_FR4 * 64 + lane, 64, K.R, K.rho, K.w, K.v);    // shank
            if (k == 4 && seq > 0) rpy_acc = v3(rpy_acc.x + fabsf(K.w.x), rpy_acc.y + fabsf(K.w.y), rpy_acc.z + fabsf(K.w.z));
            if (k == 4) { rare_store_frame(footfr + lane, 64, K.R, K.rho, K.w, K.v);   // foot (self-collision, wave 1)
                          flag_set(L.flag + FL_FRAMES, seq + 1, lane); GRX_EV(8); }
        }
        // the foot spheres' heightfield gathers go out now and land while the bias forces are computed
        FootProbe fp;
        FootProbeQ fpq;
        if (LPL == 2) foot_probe_q<HF>(P, C, Clds, half, K, O, hmax, fpq);
        else foot_probe<HF>(P, C, K, O, hmax, fp);
        // rigid-body bias force of chain body k -> wave 0.  The joint-limit spring/damper torque of joint k (oracle substep())
        // rides in the hand-over's spare slot: branch-free, constants in this wave's registers (on wave 0, fetched from LDS
        // behind data-dependent branches, it cost 1.8 k cycles per sub-step)
        auto bias_out = [&](const int k) {
            const V3 kap = KK[k].rho + rot(KK[k].R, v3(C.body[k].com[0], C.body[k].com[1], C.body[k].com[2]));
            const S3 Ic = {C.body[k].Ic[0], C.body[k].Ic[1], C.body[k].Ic[2], C.body[k].Ic[3], C.body[k].Ic[4], C.body[k].Ic[5]};
            V3 pa, pl;
            rigid_bias(KK[k].R, kap, C.body[k].mass, Ic, KK[k].w, KK[k].v, pa, pl);
            const float viol = qs_q[k] < lim_lo[k] ? lim_lo[k] - qs_q[k] : (qs_q[k] > lim_hi[k] ? lim_hi[k] - qs_q[k] : 0.f);
            const float tlim = lim_k[k] * viol - (viol != 0.f ? lim_c[k] * qs_qd[k] : 0.f);
            float4* o = L.pb + (k * PB4) * 64 + lane;
            o[0 * 64] = f4(pa.x, pa.y, pa.z, pl.x);
            o[1 * 64] = f4(pl.y, pl.z, tlim, 0.f);
            flag_set(L.flag + FL_BIAS, seq * 8 + (LEG - k), lane);
        };
        // (foot contacts between the bias forces of shank and thigh, so that the foot wrench is out early: tried, +1.2 us on
        //  rough terrain -- wave 0 then waits for the last bias forces instead)
        if (NBIAS >= 2) { bias_out(4); GRX_EV(9); bias_out(3); }
        if (NBIAS == 5) { bias_out(2); bias_out(1); bias_out(0); }   // else wave 1 (four waves, heightfield: self_loop) or waves 4, 6 (eight waves) compute them
        GRX_EV(10);
        float4* c_ = L.wc + lane;
        V3 fa, fl;
        if (LPL == 2) foot_contacts_q<HF>(P, Clds, half, K, O, mu, hmax, hs, fa, fl, om_e, fpq);
        else foot_contacts<HF>(P, C, K, O, mu, hmax, hs, fa, fl, om_e, fp);
        {   // foot link velocity BEFORE this sub-step's integration (sub-step averaged foot speed, fftai.py:79-81)
            const V3 fr = K.rho + rot(K.R, v3(C.foot_pos[0], C.foot_pos[1], C.foot_pos[2]));
            const V3 fv = K.v + cross(K.w, fr);
            c_[4 * 64] = f4(fa.x, fa.y, fa.z, fl.x);
            c_[5 * 64] = f4(fl.y, fl.z, fv.x, fv.y);
            c_[6 * 64] = f4(fv.z, 0.f, 0.f, 0.f);
        }
        flag_set(L.flag + FL_FOOT, seq + 1, lane);
        GRX_EV(11);
    }
    GRX_HELPER_PROF_END(2);
}

// wave 3: the seldom-touching shapes -- base lump (torso, head, arms), thigh, shank -- lane-compacted (grx_rare.h)
template <int HF, bool W8, bool SPLIT = W8 && GRX_W8_RARESPLIT>   // W8 (eight waves): base-lump shapes only -- thigh / shank shapes on wave 7 (chain_rare_loop), base bias force on wave 5
GRX_DEV void base_contact_loop(KP P, const KTables& T, const SideConst& C, const RareBuf& RB, float mu, float hmax, float base_m, V3 base_c,
                               const S3& base_I, const PipeLds& L, int lane, int el, int side, float* s_tp, LinkPrep& lp, V3 link_rows[11],
                               const float4* footfr, float mu_self) {
    GRX_HELPER_PROF_BEGIN;
    SelfNear sn3; sn3.m = 0;   // W8: this wave's copy of the self-collision broad phase (it evaluates the thigh x base-lump pairs)
    const SideConst Cself = T.side[side];   // (the pair table of that part in registers: read from LDS every sub-step it costs ~1.5 k cycles)
#ifdef GRX_PROFILE_SECTIONS
    long long racc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
    long long* racc = nullptr;
#endif
    for (int seq = 0; seq < P.decimation; ++seq) {
        GRX_HELPER_PROF_IDLE0;
        flag_wait(L.flag + FL_STATE, seq + 1);
        GRX_HELPER_PROF_IDLE1;
        float b[13]; pipe_base_load(L.bq, el, b);   // (base state: four quads per env)
        const V3 O = v3(b[0], b[1], b[2]);
        const R3 R0 = quat_to_R(b[3], b[4], b[5], b[6]);
        const V3 vel = v3(b[7], b[8], b[9]), ang = v3(b[10], b[11], b[12]);
#ifndef GRX_W8_LATE3
#define GRX_W8_LATE3 0   // (paid while this wave had the base-lump shapes only: +0.5 %; with the thigh x base-lump self-collision on it: -9 %)
#endif
        // eight waves: this role's output is needed last and takes 1.6 k cycles, and the first ~2 k cycles of a sub-step are the ones in
        // which all eight waves want to issue: it starts once the rigid inertias (the first hand-over on wave 0's chain) are out
        if (W8 && GRX_W8_LATE3 && LPL == 2) flag_wait(L.flag + FL_XK, seq * 4 + 2);   // (lane pairs: this wave tests all eight shapes per lane and is the last to finish as it is)
        if (!W8) {   // the base lump's rigid-body bias force (cheap; needed by wave 0 only at the base solve)
            V3 bpa, bpl;
            rigid_bias(R0, rot(R0, base_c), base_m, base_I, ang, vel, bpa, bpl);
            float4* o = L.pb + (LEG * PB4) * 64 + lane;
            o[0 * 64] = f4(bpa.x, bpa.y, bpa.z, bpl.x);
            o[1 * 64] = f4(bpl.y, bpl.z, 0.f, 0.f);
            flag_set(L.flag + FL_BASEBIAS, seq + 1, lane);
        }
        // base-lump shapes first (they need only the base state); the thigh / shank frames come from wave 2's walk
        RareOut ro;
        const int want = seq + 1;
        int* const fr_flag = L.flag + FL_FRAMES;
        rare_contacts<HF, 0, SPLIT ? 8 : RC_NS, true>(P, T, C, RB, lane, el, side, R0, O, ang, vel, ChainKin(), ChainKin(), mu, hmax, ro, racc,
                                                   [=]() { flag_wait(fr_flag, want); }, seq == P.decimation - 1);
        float4* c_ = L.wc + lane;
        if (!SPLIT) {
            c_[0 * 64] = f4(ro.fa2.x, ro.fa2.y, ro.fa2.z, ro.fl2.x);
            c_[1 * 64] = f4(ro.fl2.y, ro.fl2.z, 0.f, 0.f);
            c_[2 * 64] = f4(ro.fa3.x, ro.fa3.y, ro.fa3.z, ro.fl3.x);
            c_[3 * 64] = f4(ro.fl3.y, ro.fl3.z, 0.f, 0.f);
        }
        L.wr[lane] = f4(ro.f0a.x, ro.f0a.y, ro.f0a.z, ro.f0l.x);
        L.wr[64 + lane] = f4(ro.f0l.y, ro.f0l.z, ro.term ? 1.f : 0.f, ro.pen_count);
        flag_set(L.flag + FL_LEGS, seq + 1, lane);   // thigh + shank wrenches and the base-lump wrench, one hand-over
        // W8, lane quads: the thigh x base-lump self-collision (the leg x leg part runs on wave 1; lane pairs: wave 6 does this)
        if (W8 && LPL == 2) thigh_base_self_step(P, T, Cself, RB, footfr, L, lane, side, seq, R0, ang, vel, mu_self, sn3);
        GRX_EV(12);
        if (seq == P.decimation - 1) {
#ifdef GRX_PROFILE_SECTIONS
            if (lane == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 88] = clock64();
#endif
            // GRX_T_CONTACT_FORCES (net contact force per URDF link after the LAST sub-step, legged_robot.py:117, 266): this wave
            // has the base-lump links' forces; the foot's terrain force comes from wave 2, the self-collision forces from
            // wave 1 -- both handed over for wave 0 anyway.  Written here, off wave 0's path (it is the last to finish).
            lp = link_prep(C);   // (table reads: before the forces are in)
            if (SPLIT) flag_wait_all(L.flag, flag_want(lane, FL_FOOT, seq + 1, FL_SELF, seq + 1, FL_CHAINW, seq + 1, FL_SB, seq + 1), lane);
            else { flag_wait(L.flag + FL_FOOT, seq + 1); flag_wait(L.flag + FL_SELF, seq + 1); }
            if (SPLIT) {   // the terrain forces on thigh and shank come from wave 7
                const float4 a0 = c_[0 * 64], a1 = c_[1 * 64], a2 = c_[2 * 64], a3 = c_[3 * 64];
                ro.fl2 = v3(a0.w, a1.x, a1.y); ro.fl3 = v3(a2.w, a3.x, a3.y);
            }
            SelfOut sc;
            {
                const float4* c = L.wc + 7 * 64 + lane;
                const float4 s0 = c[0 * 64], s1 = c[1 * 64], s2 = c[2 * 64], s3 = c[3 * 64], s4 = c[4 * 64], s5 = c[5 * 64], s6 = c[6 * 64], s7 = c[7 * 64];
                sc.fa[0] = v3(s0.x, s0.y, s0.z); sc.fl[0] = v3(s0.w, s1.x, s1.y);
                sc.fa[1] = v3(s1.z, s1.w, s2.x); sc.fl[1] = v3(s2.y, s2.z, s2.w);
                sc.fa[2] = v3(s3.x, s3.y, s3.z); sc.fl[2] = v3(s3.w, s4.x, s4.y);
                sc.f0a = v3(s4.z, s4.w, s5.x); sc.f0l = v3(s5.y, s5.z, s5.w);
                sc.fbase[0] = v3(s6.x, s6.y, s6.z); sc.fbase[1] = v3(s6.w, s7.x, s7.y);
            }
            if (W8) {   // ... plus the thigh x base-lump part
                SelfOut sb;
                {
                    const float4* d = L.sb + lane;
                    const float4 d0 = d[0 * 64], d1 = d[1 * 64], d2 = d[2 * 64], d3 = d[3 * 64], d4 = d[4 * 64];
                    sb.fa[0] = v3(d0.x, d0.y, d0.z); sb.fl[0] = v3(d0.w, d1.x, d1.y);
                    sb.f0a = v3(d1.z, d1.w, d2.x); sb.f0l = v3(d2.y, d2.z, d2.w);
                    sb.fbase[0] = v3(d3.x, d3.y, d3.z); sb.fbase[1] = v3(d3.w, d4.x, d4.y);
                }
                sc.fa[0] = sc.fa[0] + sb.fa[0]; sc.fl[0] = sc.fl[0] + sb.fl[0];
                sc.f0a = sc.f0a + sb.f0a; sc.f0l = sc.f0l + sb.f0l;
                sc.fbase[0] = sc.fbase[0] + sb.fbase[0]; sc.fbase[1] = sc.fbase[1] + sb.fbase[1];
            }
            const float4 f0_ = c_[4 * 64], f1_ = c_[5 * 64];
            const V3 foot_terrain = v3(f0_.w, f1_.x, f1_.y);
#ifdef GRX_PROFILE_SECTIONS
            if (lane == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 89] = clock64();
#endif
            bool term; float pen;
            net_link_forces(P, lp, ro.lf, ro.fl2, ro.fl3, foot_terrain, sc, link_rows, term, pen);   // (the caller stores the rows once the block is past its barrier)
            s_tp[lane] = term ? 1.f : 0.f; s_tp[64 + lane] = pen;   // picked up by wave 0 behind the barrier that ends the sub-steps
#ifdef GRX_PROFILE_SECTIONS
            if (lane == 0) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 90] = clock64();
#endif
        }
    }
    GRX_HELPER_PROF_END(3);
#ifdef GRX_PROFILE_SECTIONS
    if (lane == 0) for (int i = 0; i < 8; ++i) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 32 + i] = racc[i];
#endif
}
