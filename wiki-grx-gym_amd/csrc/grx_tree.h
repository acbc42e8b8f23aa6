// grx_tree.h -- the tree step kernel: a LANE GROUP per env, a chain of the robot's tree per lane (included by
// grx_kernels.hip inside its anonymous namespace, after grx_generic.h whose helpers it shares).
//
// The 32-DOF full-body GR1T1 (BASELINE.json config 5) is five chains around the floating base: two 6-joint legs, the 3-joint
// waist that carries the torso and goes on into the head (3 joints), and -- hanging from the torso -- two 7-joint arms.  The
// generic-tree kernel (grx_generic.h) walks those 32 bodies one after the other on ONE lane per env: ~50 k instructions per
// sub-step on a wave that issues one instruction per ~4.5 cycles (tools/micro/operand_rate.hip).  Here an env is a group of
// GRX_TREE_G = 8 lanes and every lane owns one chain (a body's first child continues its chain, further children start new
// chains):
//
//   * at depth level g a lane works on its chain's body of depth g: the three passes of the articulated-body algorithm are
//     loops over the 10 depth levels of the tree instead of its 32 bodies;
//   * round 4: the LDS row of a body is 25 words instead of 53 -- its frame (R, rho, w, v: the child chains' start, the contact
//     pass, the self-collision, GRX_T_RIGID_BODY_STATES), its bias-force accumulator (contact and self-collision forces are added
//     by whichever lane finds them) and its motor torque; what pass 3 needs from pass 2 (U = I^A S, 1/d, u, the joint axis) is
//     parked in the slots of that row that are dead by then; the joint state lives in registers; the velocity-product
//     accelerations are folded into the bias forces (tree_outward) instead of being kept per body; rigid inertias are formed
//     where they are consumed.  Laid out [word][env of the wave] with an ODD body stride (the lanes of a group, which sit on
//     different bodies at the same word, fall on different banks); a chain parks its articulated inertia and bias force in its
//     slot for the parent's lane.  36 KB per 8-env wave of the 33-body robot instead of 71 KB: FOUR waves per CU, one on every
//     SIMD (round 3: two);
//   * terrain contacts run in a rolled pass of their own over the lane's bodies that carry shapes (a table: three rounds for
//     this robot instead of ten levels), on the frames in LDS -- the unrolled passes stay small (instruction cache);
//   * round 6: what does not depend on the parent's frame leaves the depth levels (tree_joint_phase: local rotations and motor torques
//     of ALL bodies round the group's lanes), a level of a pass is ONE batch of LDS reads requested one level ahead (tree_out_fetch /
//     tree_in_fetch / tree_acc_fetch), what a lane knows of its chain without the tables sits in registers (TreeChain), and the
//     self-collision's broad phase tests the model's sphere pairs exactly, on centres the contact pass leaves in LDS (TreeTab.sp) --
//     618 k -> 465 k cycles per policy step at 4096 envs (profiles/r06_experiments.md);
//   * all lanes of an env sit in one wave: program order is the only synchronisation;
//   * the same formulation as every kernel here -- spatial quantities in world axes about the base origin, so a child's
//     inertia simply ADDS into its parent --, the same contact, self-collision and env-pipeline arithmetic as grx_generic.h
//     (which stays as the fallback for trees with more than eight chains or ten levels, and as this kernel's cross-check:
//     GRX_TREE=0).
#pragma once

#ifndef GRX_TREE_GDEV
#define GRX_TREE_GDEV GRX_TREE_G   // (grx_tree16.hip compiles this header with 16)
#endif
constexpr int TG = GRX_TREE_GDEV, TEPW = 64 / TG;   // lanes per env, envs per wave
static_assert(TG == GRX_TREE_G || TG == GRX_TREE_GMAX, "an 8- or a 16-lane group per env");
constexpr int TWAVES_MAX = 4;                    // (the statistics rows at the kernel's end are added as four: keep in step) waves per block: 2 while the blocks fit the CUs in one round, else 4 (grx_capi.cpp)
// LDS workspace words per body (bodies 1 .. nb - 1; the base lives in registers)
enum { T_R = 0, T_RHO = 9, T_W = 12, T_V = 15, T_PA = 18, T_PL = 21, T_TAU = 24, T_NB = 25 };   // (T_TAU: the joint's motor torque of the current sub-step)
constexpr int T_UPW = 27;    // a chain's hand-over to its parent: A 6, B 9, D 6, pa 3, pl 3
constexpr int T_MISC = 14;   // foot link velocities before the sub-step (6), 2 spare, the terrain wrench on the base (6)
enum { TD_ACUR = 0, TD_ALAST = 1, TD_STR = 2, TD_N = 3 };   // per-dof rows every sub-step reads: clipped action, last action, motor strength
#define TBO(b) (((b) - 1) * T_NB)

struct TreeOff { int up, dof, lf, an, misc, total; };
// (the hand-over region doubles as the table of the spheres' world centres -- 3 words a sphere -- between the contact pass, which forms them,
//  and the inward pass: the self-collision's broad phase reads them, tree_self_collision)
__host__ __device__ inline TreeOff tree_offsets(int nb, int nlc, int nchain, int nsph) {
    TreeOff o;
    const int upw = nchain * T_UPW > nsph * 3 ? nchain * T_UPW : nsph * 3;
    o.up = (nb - 1) * T_NB; o.dof = o.up + upw; o.lf = o.dof + TD_N * GRX_MAX_DOFS; o.an = o.lf + nlc * 3; o.misc = o.an + 24; o.total = o.misc + T_MISC;
    return o;
}
// LDS addressing (round 5).  ds_read_b32 / ds_write_b32 bank by (word address mod 32) and serve a wave as its two 32-lane HALVES, one LDS cycle
// per half when its lanes fall on distinct banks (MI355X_MICROARCH.md, LDS).  Rounds 3-4 laid the rows out [word][env of the WAVE]: a half holds
// only half of the wave's envs, so half of the banks idled and the lanes of a group -- on different bodies at the same word -- shared
// 32 / TEPW = 4 (8) row classes: 26 % of the LDS-active cycles were conflicts (profiles/r04_pmc_sq_full_body_rough4096.json).  Now every HALF
// keeps its own block of rows, [word][env of the half]: bank = (row mod (32 / TEH)) * TEH + env -- 8 (16) row classes over all 32 banks, and
// with the odd body stride (25 = 1 mod 8, 9 mod 16) the bodies a group's lanes work on at one depth level (b, b + 6, b + 12, ...: the chains
// are numbered consecutively) fall in different classes.  `ei` below is the lane's offset (half * block + env of the half), not an env index.
constexpr int TEH = TEPW / 2;      // envs per half wave
__host__ __device__ inline int tree_half_words(int total) { return (total * TEH + 31) / 32 * 32; }   // a half's block, bank-aligned
#define TW(addr) wsw[(addr) * TEH + ei]

GRX_DEV float grp_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); if (TG == 16) v += __shfl_xor(v, 8); return v; }
GRX_DEV float grp_bcast(float v, int lane, int src) { return __shfl(v, (lane & ~(TG - 1)) | src); }
GRX_DEV V3 grp_bcast(V3 v, int lane, int src) { return v3(grp_bcast(v.x, lane, src), grp_bcast(v.y, lane, src), grp_bcast(v.z, lane, src)); }

GRX_DEV V3 tw_v3(const float* wsw, int ei, int a) { return v3(TW(a), TW(a + 1), TW(a + 2)); }
GRX_DEV void tw_put(float* wsw, int ei, int a, V3 x) { TW(a) = x.x; TW(a + 1) = x.y; TW(a + 2) = x.z; }
GRX_DEV R3 tw_R(const float* wsw, int ei, int a) { R3 R; R.cx = tw_v3(wsw, ei, a); R.cy = tw_v3(wsw, ei, a + 3); R.cz = tw_v3(wsw, ei, a + 6); return R; }
#ifdef GRX_PROFILE_SECTIONS
#define TLV(P_, slot) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 64) (P_).prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + (slot)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TLV(P_, slot) do {} while (0)
#endif
GRX_DEV void tree_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }   // (one wave: LDS operations complete in program order)

// Round 6 -- the JOINT-LOCAL phase.  A wave alone on its SIMD issues one instruction per ~4 cycles and every LDS round trip on the way costs
// ~100 more: rounds 4-5 rode the joint's rotation (sincos, Rodrigues, the rot0 product behind a branch on a table word), the motor torque (five
// table words, the action, the strength: three dependent batches) and the hanging chains' hand-over on EVERY ONE of the ten depth levels of the
// outward walk -- ~350 instructions and six dependent LDS round trips a level (from the ISA).  Nothing of that depends on the parent's frame:
// it is formed here for ALL bodies at once, the bodies going round the group's lanes (2 rounds of 16 lanes, 4 of 8), and parked in the body's
// row -- the local rotation rot0 * Rot(axis, q) in the rotation's nine slots (the walk reads it and writes the world rotation over it), the
// clipped motor torque in T_TAU.  q, qd come from two slots of the row (the rotation's, dead between the acceleration pass, which leaves the
// integrated state there, and this phase).
enum { T_Q = T_R + 2, T_QD = T_R + 3 };
// The loop takes TWO rounds of the group's lanes per iteration, their LDS operands requested together and without a branch in between (a lane
// past the last body works on the last body again and drops the result): one LDS round trip per pair of rounds instead of three per round.
#define TREE_BODY_ROUNDS2(b0_, ok0_, b1_, ok1_) \
    for (int r2_ = 1 + c; r2_ < T.nb; r2_ += 2 * TG) \
        if (const int b0_ = r2_, b1_ = min(r2_ + TG, T.nb - 1); true) \
            if (const bool ok0_ = true, ok1_ = r2_ + TG < T.nb; true)
struct TreeJointIn { float q, qd, ax, ay, az, r0[9], kp, kd, q0, effort, act, str; };
template <bool KIN>
GRX_DEV TreeJointIn tree_joint_fetch(const TreeTab& T, const float* wsw, int ei, const TreeOff& o, int b, bool use_last) {
    TreeJointIn x;
    const TreeBody& tb = T.body[b];
    const int j = b - 1, wb = TBO(b);
    x.q = TW(wb + T_Q); x.qd = TW(wb + T_QD);
    x.ax = tb.axis[0]; x.ay = tb.axis[1]; x.az = tb.axis[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) x.r0[k] = tb.rot0[k];
    x.kp = 0.f; x.kd = 0.f; x.q0 = 0.f; x.effort = 0.f; x.act = 0.f; x.str = 0.f;
    if (!KIN) {
        const TreeDof& td = T.dof[j];
        x.kp = td.kp; x.kd = td.kd; x.q0 = td.q0; x.effort = td.effort;
        x.act = TW(o.dof + (use_last ? TD_ALAST : TD_ACUR) * GRX_MAX_DOFS + j);
        x.str = TW(o.dof + TD_STR * GRX_MAX_DOFS + j);   // motor strength of this env (domain randomisation)
    }
    return x;
}
template <bool KIN>
GRX_DEV void tree_joint_one(KP P, float* wsw, int ei, int b, bool ok, const TreeJointIn& x, const float* qd_last_e) {
    const int wb = TBO(b);
    float sn, cs;
    grx_sincos(x.q, sn, cs);
    const float ax = x.ax, ay = x.ay, az = x.az, oc = 1.f - cs;
    const V3 qx = v3(cs + ax * ax * oc, az * sn + ax * ay * oc, -ay * sn + ax * az * oc);
    const V3 qy = v3(-az * sn + ax * ay * oc, cs + ay * ay * oc, ax * sn + ay * az * oc);
    const V3 qz = v3(ay * sn + ax * az * oc, -ax * sn + ay * az * oc, cs + az * az * oc);
    // rot0 * Rot(axis, q), unconditionally (only the shoulders of the GR1 carry a rotated joint frame; the product with the unit matrix is
    // exact, and the branch on a table word it replaces was a dependent LDS round trip)
    R3 J;
    J.cx = v3(x.r0[0], x.r0[3], x.r0[6]); J.cy = v3(x.r0[1], x.r0[4], x.r0[7]); J.cz = v3(x.r0[2], x.r0[5], x.r0[8]);
    const V3 lx = rot(J, qx), ly = rot(J, qy), lz = rot(J, qz);
    float tq = 0.f;
    if (!KIN) {   // _compute_torques (legged_robot.py:679-715): the 'P' law spelled out, the other control types (one with a global read) behind a uniform branch
        float t = x.kp * (x.act * P.action_scale + x.q0 - x.q) - x.kd * x.qd;
        if (P.control_type != GRX_CONTROL_P) t = control_torque(P, x.kp, x.kd, x.q0, x.act, x.q, x.qd, qd_last_e + (size_t)(b - 1) * (size_t)P.N);
        t *= x.str;
        tq = fminf(fmaxf(t, -x.effort), x.effort);
    }
    if (ok) {
        tw_put(wsw, ei, wb + T_R, lx); tw_put(wsw, ei, wb + T_R + 3, ly); tw_put(wsw, ei, wb + T_R + 6, lz);
        if (!KIN) TW(wb + T_TAU) = tq;
    }
}
template <bool KIN>
GRX_DEV void tree_joint_phase(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, bool use_last, const float* qd_last_e) {
    TREE_BODY_ROUNDS2(b0, ok0, b1, ok1) {
        const TreeJointIn x0 = tree_joint_fetch<KIN>(T, wsw, ei, o, b0, use_last), x1 = tree_joint_fetch<KIN>(T, wsw, ei, o, b1, use_last);
        tree_joint_one<KIN>(P, wsw, ei, b0, ok0, x0, qd_last_e);
        tree_joint_one<KIN>(P, wsw, ei, b1, ok1, x1, qd_last_e);
    }
    tree_fence();
}

// one sphere against the terrain (gen_sphere with the anchors and the link-force accumulators in the LDS workspace)
template <int HF, bool LF = true>   // LF: add the force to the link's accumulator here (false: the caller does, in its turn)
GRX_DEV V3 tree_sphere(KP P, const TreeSph& S, V3 w, V3 v, V3 O, float mu, float om_e, float hmax, float* wsw, int ei,
                       const TreeOff& o, V3 xr, const TerrainAt& th) {   // xr: the centre relative to O; th: the terrain under it (looked up by the caller, in batches)
    V3 F = v3(0.f, 0.f, 0.f);
    const float wz = O.z + xr.z, r = S.r;
    const int slot = S.slot;
    // the sphere's friction anchor (x, y, approach speed): requested up front, in one batch -- read where it is used, behind the nested
    // branches below, each of its three words was a dependent LDS round trip (round 6)
    const int sa = o.an + max(slot, 0) * 3;
    const float an_x = TW(sa), an_y = TW(sa + 1), an_v = TW(sa + 2);
    bool touching = false;
    float vimp = 0.f;
    if (wz - r <= hmax) {
        const float wx = O.x + xr.x, wy = O.y + xr.y;
        const float gx = th.gx, gy = th.gy;
        const float dv = th.h + r - wz;
        if (dv > 0.0f) {
            touching = true;
            const float nn = grx_rsq(1.0f + gx * gx + gy * gy);
            const V3 n = v3(-gx * nn, -gy * nn, nn);
            const float d = dv * nn;
            const V3 u = v + cross(w, xr);
            const float un = dot(u, n);
            float cd = fminf(P.kn * d * P.dn, S.dmax);
            if (slot >= 0) {   // restitution (sphere_contact's rule)
                vimp = an_v;
                if (vimp == 0.f) vimp = fmaxf(fmaxf(-un, 0.0f), 1e-6f);
                if (un > 0.0f && vimp > P.bounce_threshold) cd *= om_e;
            }
            const float fn = fmaxf(P.kn * d - cd * un, 0.0f);
            F = n * fn;
            const float fmax = mu * fn;
            if (slot >= 0) {
                float axx = an_x, ayy = an_y;
                if (an_v == 0.f) { axx = wx; ayy = wy; }
                float ftx = -P.kt * (wx - axx) - P.ct * u.x;
                float fty = -P.kt * (wy - ayy) - P.ct * u.y;
                const float ft = grx_sqrt(ftx * ftx + fty * fty);
                if (ft > fmax) {
                    const float sc = fmax * grx_rcp(ft);
                    ftx *= sc; fty *= sc;
                    axx = wx + ftx * P.inv_kt;
                    ayy = wy + fty * P.inv_kt;
                }
                TW(o.an + slot * 3) = axx; TW(o.an + slot * 3 + 1) = ayy;
                F.x += ftx; F.y += fty;
            } else {
                const float sp = grx_sqrt(u.x * u.x + u.y * u.y);
                const float ft = fminf(P.cv * sp, fmax);
                if (sp > 1e-9f) { const float k = -ft * grx_rcp(sp); F.x += k * u.x; F.y += k * u.y; }
            }
        }
    }
    if (slot >= 0) TW(o.an + slot * 3 + 2) = touching ? vimp : 0.f;
    if (LF) {
        const int L = S.link;
        TW(o.lf + L * 3) += F.x; TW(o.lf + L * 3 + 1) += F.y; TW(o.lf + L * 3 + 2) += F.z;
    }
    return F;
}

struct TreeEnv {   // what every lane of the env's group holds (redundantly)
    GenBase B;
    float base_m; V3 base_c; S3 base_I;
    float mu, om_e, hmax;
};

// what a lane keeps in REGISTERS for its own chain, indexed by the depth level: the level loops are UNROLLED (TNG levels, compile
// time).  Measured both ways on MI355X (round 4): the unrolled passes are 200 KB of code and still 12 % faster than the rolled
// ones (388 against 443 us per step at 4096 envs), whose lanes pick their joint state with chains of selects and keep the joint
// axis in LDS -- the kernel is bound by its instruction count (one wave per SIMD issues one instruction per ~4 cycles), not by
// instruction fetch.
constexpr int TNG = GRX_TREE_LEVELS;
struct TreeRegs {
    int sb[TNG];            // body of this lane at level g, -1: none
    float q[TNG], qd[TNG];  // joint state
    V3 Sa[TNG];             // joint axis in world axes (the motion subspace is S = (a; rho x a): rho comes back from the frame in LDS)
};

// what a chain's lane knows of its chain without asking the tables (registers, set once per launch): a level of a pass then needs ONE batch
// of LDS reads, all issued together, instead of a dependent round trip per table word it used to branch on
struct TreeChain {
    int first, last;   // the chain's run of levels (first > last: no chain)
    int hangp;         // the body the chain hangs from (0: the base)
    uint32_t hcmask;   // bit g: the chain's body of level g carries hanging chains
    uint32_t hlev, hlane;   // ... which ones: up to eight (level + 1, lane of the hanging chain) pairs, a nibble each, in the order the inward pass adds them up
                            // (level order, then the body's table order); nibble 0 of hlev: no more.  The inward pass used to ask the body table for them: two
                            // dependent LDS round trips per hanging chain at the torso's level
};

// pass 1 (root -> leaves) over the depth levels: frames and velocities into LDS; KIN: nothing else (the state after the last sub-step).
// The velocity-product accelerations c_k are FOLDED into the bodies' bias forces (as in the eight-wave lower-limb kernel,
// grx_wavepipe.h rigid_bias_z): with zeta_k = the sum of the c_j along the path from the base and a_k = a^_k + zeta_k, body k obeys
// f_k = I_k a^_k + (p_k + I_k zeta_k) and a^_k = a^_parent + S_k qdd_k -- the articulated-body recursion in a^ has no c terms at
// all: nothing to keep per level, no I^A c products in pass 2, no additions in pass 3.  A body's zeta is parked where its bias force goes
// (the force itself is formed for ALL bodies at once behind the walk); a chain that hangs from it PULLS it from there when it starts
// (rounds 4-5: the parent pushed it into every hanging chain's hand-over slot, a loop over a table count on every level).
// Round 6: a level is ONE batch of LDS reads (the joint-local rotation tree_joint_phase left in the row, the joint's origin and axis),
// ~80 multiply-adds and the stores -- see tree_joint_phase.
struct TreeOutIn { R3 L; V3 jp, ax; };
GRX_DEV TreeOutIn tree_out_fetch(const TreeTab& T, const float* wsw, int ei, int b) {
    TreeOutIn x;
    const TreeBody& tb = T.body[b];
    x.L = tw_R(wsw, ei, TBO(b) + T_R);
    x.jp = v3(tb.jpos[0], tb.jpos[1], tb.jpos[2]); x.ax = v3(tb.axis[0], tb.axis[1], tb.axis[2]);
    return x;
}
template <bool KIN>
GRX_DEV void tree_outward(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, const TreeEnv& E, const R3& R0, const TreeChain& CH, int nstep, TreeRegs& G) {
    R3 Rc = R0;
    V3 rho_c = v3(0.f, 0.f, 0.f), w_c = E.B.ang, v_c = E.B.vel;
    V3 za = v3(0.f, 0.f, 0.f), zl = v3(0.f, 0.f, 0.f);
    // final frames (KIN): only as deep as a foot, the torso, the forehead -- the arms hang four levels deeper -- unless every link frame is published
    const int kin_levels = KIN && !P.publish_rbs ? T.nstep_kin : TNG;
    // (a level's batch is requested one level ahead, unconditionally -- see tree_accel; the loop has no exit at the tree's depth either: with a
    //  loop-invariant bound a `break` makes the trip count a run-time value, the pass is no longer unrolled and its level-indexed registers go
    //  to scratch -- measured)
    TreeOutIn nx = tree_out_fetch(T, wsw, ei, max(G.sb[0], 1));
    if (!KIN) TLV(P, 44);
#pragma unroll
    for (int g = 0; g < TNG; ++g) {
        const TreeOutIn in = nx;
        if (g + 1 < TNG) nx = tree_out_fetch(T, wsw, ei, max(G.sb[g + 1 < TNG ? g + 1 : g], 1));
        if (G.sb[g] >= 0 && (!KIN || g < kin_levels)) {
            const int b = G.sb[g];
            const int wb = TBO(b);
            const R3 L = in.L;
            const V3 jp = in.jp, ax = in.ax;
            if (g == CH.first && CH.hangp != 0) {   // the chain hangs from another chain's body, processed one level earlier
                const int wp = TBO(CH.hangp);
                Rc = tw_R(wsw, ei, wp + T_R); rho_c = tw_v3(wsw, ei, wp + T_RHO);
                w_c = tw_v3(wsw, ei, wp + T_W); v_c = tw_v3(wsw, ei, wp + T_V);
                if (!KIN) { za = tw_v3(wsw, ei, wp + T_PA); zl = tw_v3(wsw, ei, wp + T_PL); }
            }
            const float qdj = G.qd[g];
            const V3 rho = rho_c + rot(Rc, jp);
            R3 R;
            R.cx = rot(Rc, L.cx); R.cy = rot(Rc, L.cy); R.cz = rot(Rc, L.cz);
            const V3 a = rot(R, ax);
            const V3 s = cross(rho, a);
            const V3 w = fma3(a, qdj, w_c), v = fma3(s, qdj, v_c);
            tw_put(wsw, ei, wb + T_R, R.cx); tw_put(wsw, ei, wb + T_R + 3, R.cy); tw_put(wsw, ei, wb + T_R + 6, R.cz);
            tw_put(wsw, ei, wb + T_RHO, rho); tw_put(wsw, ei, wb + T_W, w); tw_put(wsw, ei, wb + T_V, v);
            if (!KIN) {
                G.Sa[g] = a;
                za = fma3(cross(w_c, a), qdj, za);
                zl = fma3(cross(v_c, a) + cross(w_c, s), qdj, zl);
                tw_put(wsw, ei, wb + T_PA, za); tw_put(wsw, ei, wb + T_PL, zl);
            }
            Rc = R; rho_c = rho; w_c = w; v_c = v;
        }
        tree_fence();
        if (!KIN) TLV(P, 45 + g);
    }
}
// rigid-body bias forces p_k + I_k zeta_k of every body, from the frames and the zeta the walk left in LDS (contacts and self-collision add into these)
GRX_DEV void tree_bias_all(const TreeTab& T, float* wsw, int ei, int c) {
    for (int b = 1 + c; b < T.nb; b += TG) {   // (two rounds per iteration as in tree_joint_phase: measured, not faster -- this loop is bound by its arithmetic)
        const TreeBody& tb = T.body[b];
        const int wb = TBO(b);
        const R3 R = tw_R(wsw, ei, wb + T_R);
        const V3 rho = tw_v3(wsw, ei, wb + T_RHO), w = tw_v3(wsw, ei, wb + T_W), v = tw_v3(wsw, ei, wb + T_V);
        const V3 za_ = tw_v3(wsw, ei, wb + T_PA), zl_ = tw_v3(wsw, ei, wb + T_PL);
        const V3 kap = rho + rot(R, v3(tb.com[0], tb.com[1], tb.com[2]));
        const S3 Ic = {tb.Ic[0], tb.Ic[1], tb.Ic[2], tb.Ic[3], tb.Ic[4], tb.Ic[5]};
        V3 pa, pl;
        rigid_bias_z(R, kap, tb.mass, Ic, w, v, za_, zl_, pa, pl);
        tw_put(wsw, ei, wb + T_PA, pa); tw_put(wsw, ei, wb + T_PL, pl);
    }
    tree_fence();
}

// terrain contacts: a work list (TreeTab.cw) deals the bodies' shapes, two at a time, to ALL lanes of the group -- the frames are in
// LDS, so a lane need not own the body: the lanes without a chain work too and a foot's four spheres go to two lanes.  Lanes whose
// items share a body add their wrench (and the link forces) in turns.  The base's own shapes are items like any other (body 0: the
// frame is in registers, the wrench goes to the group through o.misc).  On the way: the foot link's velocity BEFORE this sub-step's
// integration (sub-step averaged foot speed, legged_robot_fftai.py:79-81).
// Round 6: a round is taken in two halves.  tree_contact_probe forms the item's sphere centres (left in LDS for the self-collision's broad
// phase) and ISSUES the terrain gathers -- ~1.2 us of memory latency on this part, which a wave alone on its SIMD cannot hide --;
// tree_contacts evaluates them.  The kernel puts the bias forces of all bodies (tree_bias_all) between the two halves of round 0.
struct TreeContactPre { V3 xr[2]; uint2 cc[2]; float tx[2], ty[2]; };   // cc: the raster cell's four corners as gathered (unpacked where they are used: a conversion next to the load would wait for it there)
template <int HF>
GRX_DEV TreeContactPre tree_contact_probe(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, const TreeEnv& E, const R3& R0, int r) {
    TreeContactPre pr;
    const int b = T.cw[r][c].body, s0 = T.cw[r][c].s0, s1 = T.cw[r][c].s1;
    R3 R = R0;
    V3 rho = v3(0.f, 0.f, 0.f);
    if (b > 0) { R = tw_R(wsw, ei, TBO(b) + T_R); rho = tw_v3(wsw, ei, TBO(b) + T_RHO); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // two shapes: their terrain lookups in flight together
        pr.xr[u] = v3(0.f, 0.f, 0.f);
        if (b >= 0 && s0 + u < s1) {
            const TreeSph& S = T.sph[s0 + u];
            pr.xr[u] = rho + rot(R, v3(S.x, S.y, S.z));
            tw_put(wsw, ei, o.up + (s0 + u) * 3, pr.xr[u]);   // (the self-collision's broad phase reads the centres from here)
        }
    }
    // the gathers are issued for EVERY lane, unconditionally (the cell index is clamped), back to back: both in flight together, nothing waits here
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        pr.cc[u] = make_uint2(0u, 0u); pr.tx[u] = 0.f; pr.ty[u] = 0.f;
        if (HF) {
            const int cell = terrain_locate(P, E.B.pos.x + pr.xr[u].x, E.B.pos.y + pr.xr[u].y, pr.tx[u], pr.ty[u]);
            pr.cc[u] = P.hf_cells[terrain_record<HF>(P, cell, pr.tx[u], pr.ty[u])];
        }
    }
    return pr;
}
template <int HF>
GRX_DEV void tree_contacts(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, const TreeEnv& E, const R3& R0, const TreeContactPre& pre0) {
    for (int r = 0; r < T.ncs; ++r) {
        const int b = T.cw[r][c].body, s0 = T.cw[r][c].s0, s1 = T.cw[r][c].s1, turn = T.cw[r][c].turn;
        V3 fa = v3(0.f, 0.f, 0.f), fl = v3(0.f, 0.f, 0.f), Fs[2] = {fa, fa};
        R3 R = R0;
        V3 rho = v3(0.f, 0.f, 0.f), w = E.B.ang, v = E.B.vel;
        TreeContactPre pr = pre0;
        if (r > 0) pr = tree_contact_probe<HF>(P, T, wsw, ei, c, o, E, R0, r);
        if (b >= 0) {
            if (b > 0) {
                const int wb = TBO(b);
                R = tw_R(wsw, ei, wb + T_R);
                rho = tw_v3(wsw, ei, wb + T_RHO); w = tw_v3(wsw, ei, wb + T_W); v = tw_v3(wsw, ei, wb + T_V);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (s0 + u < s1) {
                    TerrainAt th;
                    th.h = 0.f; th.gx = 0.f; th.gy = 0.f;
                    if (HF && E.B.pos.z + pr.xr[u].z - T.sph[s0 + u].r <= E.hmax) {
                        TerrainRaw raw;
                        terrain_unpack(pr.cc[u], raw);
                        raw.tx = pr.tx[u]; raw.ty = pr.ty[u];
                        th.h = terrain_eval<HF>(P, raw, th.gx, th.gy);
                    }
                    Fs[u] = tree_sphere<HF, false>(P, T.sph[s0 + u], w, v, E.B.pos, E.mu, E.om_e, E.hmax, wsw, ei, o, pr.xr[u], th);
                    if (HF == GRX_HF_TRIMESH && E.B.pos.z + pr.xr[u].z - T.sph[s0 + u].r <= E.hmax) {   // mesh_type 'trimesh': the vertical faces next to the shape
                        float wtx, wty;
                        const uint4 ww = wall_gather(P, E.B.pos.x + pr.xr[u].x, E.B.pos.y + pr.xr[u].y, wtx, wty);
                        Fs[u] = Fs[u] + wall_contact(P, ww, wtx, wty, E.B.pos.z + pr.xr[u].z, T.sph[s0 + u].r, T.sph[s0 + u].dmax, v + cross(w, pr.xr[u]), E.mu);
                    }
                    fa = fa + cross(pr.xr[u], Fs[u]); fl = fl + Fs[u];
                }
        }
        for (int t = 0; t < T.nturn; ++t) {
            if (b >= 0 && turn == t) {
                const int wa = b > 0 ? TBO(b) + T_PA : o.misc + 8;   // (the base: its terrain wrench, SUBTRACTED from the bias force like everyone's)
                TW(wa) -= fa.x; TW(wa + 1) -= fa.y; TW(wa + 2) -= fa.z; TW(wa + 3) -= fl.x; TW(wa + 4) -= fl.y; TW(wa + 5) -= fl.z;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (s0 + u < s1) { const int L = T.sph[s0 + u].link; TW(o.lf + L * 3) += Fs[u].x; TW(o.lf + L * 3 + 1) += Fs[u].y; TW(o.lf + L * 3 + 2) += Fs[u].z; }
                if (t == 0) {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        if (T.foot_body[f] == b) {
                            const V3 fr = rho + rot(R, v3(T.foot_pos[f][0], T.foot_pos[f][1], T.foot_pos[f][2]));
                            tw_put(wsw, ei, o.misc + f * 3, v + cross(w, fr));
                        }
                }
            }
            tree_fence();
        }
    }
}

// the hand-over of a chain: 27 words in the chain's slot
GRX_DEV void tree_put_up(float* wsw, int ei, int a, const S3& A, const M3& B, const S3& D, V3 pa, V3 pl) {
    TW(a) = A.xx; TW(a + 1) = A.xy; TW(a + 2) = A.xz; TW(a + 3) = A.yy; TW(a + 4) = A.yz; TW(a + 5) = A.zz;
    TW(a + 6) = B.a00; TW(a + 7) = B.a01; TW(a + 8) = B.a02; TW(a + 9) = B.a10; TW(a + 10) = B.a11; TW(a + 11) = B.a12; TW(a + 12) = B.a20; TW(a + 13) = B.a21; TW(a + 14) = B.a22;
    TW(a + 15) = D.xx; TW(a + 16) = D.xy; TW(a + 17) = D.xz; TW(a + 18) = D.yy; TW(a + 19) = D.yz; TW(a + 20) = D.zz;
    tw_put(wsw, ei, a + 21, pa); tw_put(wsw, ei, a + 24, pl);
}
GRX_DEV void tree_add_up(const float* wsw, int ei, int a, S3& A, M3& B, S3& D, V3& pa, V3& pl) {
    A.xx += TW(a); A.xy += TW(a + 1); A.xz += TW(a + 2); A.yy += TW(a + 3); A.yz += TW(a + 4); A.zz += TW(a + 5);
    B.a00 += TW(a + 6); B.a01 += TW(a + 7); B.a02 += TW(a + 8); B.a10 += TW(a + 9); B.a11 += TW(a + 10); B.a12 += TW(a + 11); B.a20 += TW(a + 12); B.a21 += TW(a + 13); B.a22 += TW(a + 14);
    D.xx += TW(a + 15); D.xy += TW(a + 16); D.xz += TW(a + 17); D.yy += TW(a + 18); D.yz += TW(a + 19); D.zz += TW(a + 20);
    pa = pa + tw_v3(wsw, ei, a + 21); pl = pl + tw_v3(wsw, ei, a + 24);
}

// pass 2 (leaves -> root): articulated inertias and bias forces; a chain's running [A B; B^T D], pa, pl stay in registers.  What
// pass 3 needs of a body is parked in slots of its LDS row that nobody reads any more in this sub-step: 1/d, u and the joint
// in the rotation's, U = I^A S in the bias force's.
enum { T_DI = T_R, T_U = T_R + 1, T_UA = T_PA, T_UL = T_PL };
// ... and what pass 2 itself needs of a body beyond its bias force -- the rigid inertia about O, A (6 words) and h = m kap (3) -- comes
// from a pass over ALL bodies at once (they go round the group's lanes: 2-4 rounds instead of the 10 depth levels), parked in the
// rotation's nine slots: contacts and self-collision are done with the frames by then, the next outward pass rewrites them
enum { T_AK = T_R, T_HK = T_R + 6 };
GRX_DEV void tree_rigid_inertias(const TreeTab& T, float* wsw, int ei, int c) {
    for (int b = 1 + c; b < T.nb; b += TG) {
        const TreeBody& tb = T.body[b];
        const int wb = TBO(b);
        const R3 R = tw_R(wsw, ei, wb + T_R);
        const V3 kap = tw_v3(wsw, ei, wb + T_RHO) + rot(R, v3(tb.com[0], tb.com[1], tb.com[2]));
        const S3 Ic = {tb.Ic[0], tb.Ic[1], tb.Ic[2], tb.Ic[3], tb.Ic[4], tb.Ic[5]};
        S3 Ar; V3 h;
        rigid_inertia(R, kap, tb.mass, Ic, Ar, h);
        TW(wb + T_AK) = Ar.xx; TW(wb + T_AK + 1) = Ar.xy; TW(wb + T_AK + 2) = Ar.xz; TW(wb + T_AK + 3) = Ar.yy; TW(wb + T_AK + 4) = Ar.yz; TW(wb + T_AK + 5) = Ar.zz;
        tw_put(wsw, ei, wb + T_HK, h);
    }
    tree_fence();
}
struct TreeInIn { V3 rho, pab, plb, hk; S3 Ar; float t, mass, arm, qlo, qhi, Klim, Clim; };
GRX_DEV TreeInIn tree_in_fetch(const TreeTab& T, const float* wsw, int ei, int b) {
    TreeInIn x;
    const TreeBody& tb = T.body[b];
    const TreeDof& td = T.dof[b - 1];
    const int wb = TBO(b);
    x.rho = tw_v3(wsw, ei, wb + T_RHO); x.pab = tw_v3(wsw, ei, wb + T_PA); x.plb = tw_v3(wsw, ei, wb + T_PL);
    x.t = TW(wb + T_TAU);
    x.Ar = S3{TW(wb + T_AK), TW(wb + T_AK + 1), TW(wb + T_AK + 2), TW(wb + T_AK + 3), TW(wb + T_AK + 4), TW(wb + T_AK + 5)};
    x.hk = tw_v3(wsw, ei, wb + T_HK);
    x.mass = tb.mass; x.arm = td.arm; x.qlo = td.qlo; x.qhi = td.qhi; x.Klim = td.Klim; x.Clim = td.Clim;
    return x;
}
GRX_DEV void tree_inward(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, const TreeChain& CH, int nstep, TreeRegs& G) {
    // The lane's chain is ONE run of levels (first .. last): its running articulated inertia [A B; B^T D] and bias force are the working
    // set itself -- zero before the chain's leaf, every level ADDS its rigid body and downdates in place, the chain's first body hands
    // them up.  (Round 4, from the ISA: separate per-level values copied into a carry cost 127 v_mov per level, a fifth of the pass.)
    // Round 6: a level's operands -- the body's row and the joint's limits and armature -- are ONE batch of LDS reads (the joint-limit
    // spring used to fetch its table words behind two branches: three dependent round trips a level), the limit torque is branch-free,
    // and whether a body carries hanging chains is a bit of a register, not a table word.  The batch -- contacts, self-collision and
    // tree_rigid_inertias are done with the rows -- is requested ONE LEVEL AHEAD (unconditionally: see tree_accel), so that its latency passes
    // behind the previous level's ~100 multiply-adds; only the hanging chains' hand-over, written one level earlier, is read in its level.
    S3 A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    M3 Bm = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    V3 pa = v3(0.f, 0.f, 0.f), pl = v3(0.f, 0.f, 0.f);
    TreeInIn nx = tree_in_fetch(T, wsw, ei, max(G.sb[TNG - 1], 1));
    TLV(P, 30);
#pragma unroll
    for (int g = TNG - 1; g >= 0; --g) {
        const TreeInIn in = nx;
        if (g > 0) nx = tree_in_fetch(T, wsw, ei, max(G.sb[g > 0 ? g - 1 : 0], 1));
        if (G.sb[g] >= 0) {
            const int b = G.sb[g];
            const int wb = TBO(b);
            const V3 rho = in.rho;
            float t = in.t;
            const float arm = in.arm, qlo = in.qlo, qhi = in.qhi, Klim = in.Klim, Clim = in.Clim;
            pa = pa + in.pab; pl = pl + in.plb;
            add_rigid(A, Bm, D, in.Ar, in.hk, in.mass);
            if (CH.hcmask & (1u << g)) {   // chains that hang from this body, fixed order
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((int)((CH.hlev >> (4 * k)) & 15u) == g + 1) tree_add_up(wsw, ei, o.up + (int)((CH.hlane >> (4 * k)) & 15u) * T_UPW, A, Bm, D, pa, pl);
            }
            const V3 a = G.Sa[g], s = cross(rho, a);
            const V3 ua = mul(A, a) + mul(Bm, s);
            const V3 ul = mulT(Bm, a) + mul(D, s);
            const float di = grx_rcp(dot(a, ua) + dot(s, ul) + arm);
            const float qj = G.q[g], qdj = G.qd[g];
            {   // joint-limit spring/damper on top of the motor torque
                const float viol = qj < qlo ? qlo - qj : (qj > qhi ? qhi - qj : 0.f);
                t += Klim * viol - (viol != 0.f ? Clim * qdj : 0.f);
            }
            const float u = t - (dot(a, pa) + dot(s, pl));
            syr(A, ua, di); ger(Bm, ua, ul, di); syr(D, ul, di);
            const float ud = u * di;
            pa = fma3(ua, ud, pa); pl = fma3(ul, ud, pl);
            TW(wb + T_DI) = di; TW(wb + T_U) = u;
            tw_put(wsw, ei, wb + T_UA, ua); tw_put(wsw, ei, wb + T_UL, ul);
            if (g == CH.first) tree_put_up(wsw, ei, o.up + c * T_UPW, A, Bm, D, pa, pl);
        }
        tree_fence();
        TLV(P, 31 + (TNG - 1 - g));
    }
}

// pass 3 (root -> leaves): accelerations a^ (see tree_outward), joint integration.  The pass is a chain of ten short dependent steps: what
// a level reads of its body's row (U, 1/d, u, rho: written by pass 2, nobody touches them any more) is requested ONE LEVEL AHEAD, so that the
// LDS latency passes behind the previous level's arithmetic (round 6; rounds 4-5 waited ~300 of a level's ~640 cycles for it).  The
// integrated joint state goes to the row's T_Q / T_QD for the next sub-step's joint-local phase.
struct TreeAccIn { V3 ua, ul, rho; float u, di, vlim; };
GRX_DEV TreeAccIn tree_acc_fetch(const TreeTab& T, const float* wsw, int ei, int b) {
    TreeAccIn x;
    const int wb = TBO(b);
    x.ua = tw_v3(wsw, ei, wb + T_UA); x.ul = tw_v3(wsw, ei, wb + T_UL); x.rho = tw_v3(wsw, ei, wb + T_RHO);
    x.u = TW(wb + T_U); x.di = TW(wb + T_DI); x.vlim = T.dof[b - 1].vlim;
    return x;
}
GRX_DEV void tree_accel(KP P, const TreeTab& T, float* wsw, int ei, int c, V3 alpha, V3 acc, const TreeChain& CH, int nstep, TreeRegs& G) {
    V3 aa_c = alpha, al_c = acc;
    const float dt = P.sim_dt;
    // (the request is UNCONDITIONAL -- a lane without a body on the next level reads body 1's row and drops it: a fetch under the lane's
    //  predicate makes the value a merge of two definitions, whose copies wait for the data at the top of the level; and the levels are not
    //  cut short at the tree's depth: a level nobody holds is one skipped branch)
    TreeAccIn nx = tree_acc_fetch(T, wsw, ei, max(G.sb[0], 1));
#pragma unroll
    for (int g = 0; g < TNG; ++g) {
        const TreeAccIn in = nx;
        if (g + 1 < TNG) nx = tree_acc_fetch(T, wsw, ei, max(G.sb[g + 1 < TNG ? g + 1 : g], 1));
        if (G.sb[g] >= 0) {
            const int b = G.sb[g], wb = TBO(b);
            if (g == CH.first && CH.hangp != 0) { aa_c = tw_v3(wsw, ei, TBO(CH.hangp) + T_PA); al_c = tw_v3(wsw, ei, TBO(CH.hangp) + T_PL); }   // (the parent parked its acceleration there)
            const V3 a = G.Sa[g];
            const float qdd = (in.u - (dot(in.ua, aa_c) + dot(in.ul, al_c))) * in.di;
            aa_c = fma3(a, qdd, aa_c); al_c = fma3(cross(in.rho, a), qdd, al_c);
            if (CH.hcmask & (1u << g)) { tw_put(wsw, ei, wb + T_PA, aa_c); tw_put(wsw, ei, wb + T_PL, al_c); }   // the hanging chains' parent acceleration (U has been read)
            float vq = fmaf(qdd, dt, G.qd[g]);
            vq = fminf(fmaxf(vq, -in.vlim), in.vlim);
            G.qd[g] = vq;
            G.q[g] = fmaf(vq, dt, G.q[g]);
            TW(wb + T_Q) = G.q[g]; TW(wb + T_QD) = vq;
        }
        tree_fence();
    }
}

// self-collision (the contact law of grx_self.h, the link-pair tables of grx_generic.h).  Round 6 -- an exact BROAD PHASE: the model's
// sphere pairs (TreeTab.sp: every sphere pair of every link pair that can meet) go round the group's lanes, each a centre-distance test on
// the world centres the contact pass left in LDS (o.up) against (ra + rb + margin)^2; a pair that passes raises the bit of its LINK pair.
// Links that do not touch stop here -- in a walking robot that is all of them, nearly always.  (Rounds 2-5 tested the links' bounding
// spheres, which for neighbours like upper arm x torso overlap in every pose: the ~800-instruction narrow phase ran in every round of
// every sub-step, 6.7 k of a sub-step's 55 k cycles.)  The raised link pairs are then evaluated as before: by the lanes that hold them,
// together; only the accumulation into the bodies' bias forces and the link forces goes one lane at a time, in table order -- no races,
// the same sums on every run, and bit for bit the sums of rounds 4-5 (the margin keeps the broad phase a superset of sphere_pair's own test).
GRX_DEV uint32_t grp_or(uint32_t v) {
    v |= (uint32_t)__shfl_xor((int)v, 1); v |= (uint32_t)__shfl_xor((int)v, 2); v |= (uint32_t)__shfl_xor((int)v, 4);
    if (TG == 16) v |= (uint32_t)__shfl_xor((int)v, 8);
    return v;
}
GRX_DEV void tree_self_collision(KP P, const TreeTab& T, float* wsw, int ei, int c, const TreeOff& o, const TreeEnv& E, const R3& R0,
                                 V3& pa0, V3& pl0) {
    uint32_t m0 = 0u, m1 = 0u;   // raised link pairs 0..31, 32..47
    for (int r0 = 0; r0 < T.nsp_batches; ++r0) {   // four rounds of the group's lanes per batch, branch-free: the table words and then the
        uint32_t ab[4]; float r2[4], d2[4];      // centres of all four are requested together (two LDS round trips a batch, not eight)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = (r0 * 4 + u) * TG + c; ab[u] = T.sp[k].ab; r2[u] = T.sp[k].r2; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ia = (int)(ab[u] & 255u), ib = (int)((ab[u] >> 8) & 255u);
            const V3 d = tw_v3(wsw, ei, o.up + ia * 3) - tw_v3(wsw, ei, o.up + ib * 3);
            d2[u] = dot(d, d);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t lp = ab[u] >> 16, bit = d2[u] < r2[u] ? 1u << (lp & 31u) : 0u;
            m0 |= lp < 32u ? bit : 0u; m1 |= lp < 32u ? 0u : bit;
        }
    }
    if (!__any((m0 | m1) != 0u)) return;
    m0 = grp_or(m0); m1 = grp_or(m1);
    const float mu_self = 2.0f * E.mu - P.terrain_friction;
    V3 dpa0 = v3(0.f, 0.f, 0.f), dpl0 = v3(0.f, 0.f, 0.f);
    for (int lp0 = 0; lp0 < T.nlp; lp0 += TG) {
        const int lp = lp0 + c;
        const bool hit = lp < T.nlp && (((lp < 32 ? m0 >> lp : m1 >> (lp - 32)) & 1u) != 0u);
        if (!__any(hit)) continue;
        // the lanes that hit evaluate their pairs TOGETHER (the nested sphere x sphere loops: ~800 instructions a pair)
        V3 Fa = v3(0.f, 0.f, 0.f), Ta = v3(0.f, 0.f, 0.f);
        int la = 0, lb = 0, ba = 0, bb = 0;
        if (hit) {
            ba = T.lp_ba[lp]; bb = T.lp_bb[lp];
            la = T.lp_a[lp]; lb = T.lp_b[lp];
            ChainKin Ka, Kb;
            if (ba == 0) Ka = ChainKin{R0, v3(0.f, 0.f, 0.f), E.B.ang, E.B.vel};
            else Ka = ChainKin{tw_R(wsw, ei, TBO(ba) + T_R), tw_v3(wsw, ei, TBO(ba) + T_RHO), tw_v3(wsw, ei, TBO(ba) + T_W), tw_v3(wsw, ei, TBO(ba) + T_V)};
            Kb = ChainKin{tw_R(wsw, ei, TBO(bb) + T_R), tw_v3(wsw, ei, TBO(bb) + T_RHO), tw_v3(wsw, ei, TBO(bb) + T_W), tw_v3(wsw, ei, TBO(bb) + T_V)};
            for (int i = T.lc_begin[la]; i < T.lc_begin[la + 1]; ++i) {
                SphC si; si.x = T.sph[i].x; si.y = T.sph[i].y; si.z = T.sph[i].z; si.r = T.sph[i].r; si.dmax = T.sph[i].dmax;
                const SphW a = sph_world(si, Ka);
                for (int jj = T.lc_begin[lb]; jj < T.lc_begin[lb + 1]; ++jj) {
                    SphC sj; sj.x = T.sph[jj].x; sj.y = T.sph[jj].y; sj.z = T.sph[jj].z; sj.r = T.sph[jj].r; sj.dmax = T.sph[jj].dmax;
                    const SphW b_ = sph_world(sj, Kb);
                    V3 F, pw;
                    if (sphere_pair(P, a, b_, mu_self, F, pw)) { Fa = Fa + F; Ta = Ta + cross(pw, F); }
                }
            }
        }
        for (int k = 0; k < TG; ++k) {   // table order within the round: lane k's pair
            if (!__any(hit && c == k)) continue;
            if (hit && c == k) {
                // F on link a (body ba), -F on link b (body bb)
                if (ba == 0) { dpa0 = dpa0 - Ta; dpl0 = dpl0 - Fa; }
                else { const int wa = TBO(ba); TW(wa + T_PA) -= Ta.x; TW(wa + T_PA + 1) -= Ta.y; TW(wa + T_PA + 2) -= Ta.z; TW(wa + T_PL) -= Fa.x; TW(wa + T_PL + 1) -= Fa.y; TW(wa + T_PL + 2) -= Fa.z; }
                const int wb_ = TBO(bb);
                TW(wb_ + T_PA) += Ta.x; TW(wb_ + T_PA + 1) += Ta.y; TW(wb_ + T_PA + 2) += Ta.z; TW(wb_ + T_PL) += Fa.x; TW(wb_ + T_PL + 1) += Fa.y; TW(wb_ + T_PL + 2) += Fa.z;
                TW(o.lf + la * 3) += Fa.x; TW(o.lf + la * 3 + 1) += Fa.y; TW(o.lf + la * 3 + 2) += Fa.z;
                TW(o.lf + lb * 3) -= Fa.x; TW(o.lf + lb * 3 + 1) -= Fa.y; TW(o.lf + lb * 3 + 2) -= Fa.z;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    // the base's share (only a few lanes ever hold one: skipped when nobody in the wave does): the group's lanes add up in the fixed
    // order of the butterfly
    if (__any(dot(dpa0, dpa0) + dot(dpl0, dpl0) != 0.f)) {
        pa0 = pa0 + v3(grp_sum(dpa0.x), grp_sum(dpa0.y), grp_sum(dpa0.z));
        pl0 = pl0 + v3(grp_sum(dpl0.x), grp_sum(dpl0.y), grp_sum(dpl0.z));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// blockDim = 64 * waves (2 or 4: grx_capi.cpp); dynamic LDS = the table + one workspace per wave.  One wave per SIMD.
// DBG (TEST-ONLY, grx_debug_post_physics; the lower-limb model forced through this kernel, i.e. nd = 10 like the debug rows): no
// sub-steps; foot forces / positions, sub-step averages, torques, termination contact and last_last_actions come from `dbg`
// (grx_kernels.hip DbgRow) -- the reference's golden fixtures reach the post-physics code config 5 runs.
#define GRX_STEP_TREE_ARGS const KParams* __restrict__ Pg, const TreeTab* __restrict__ Tt, const GenTables* __restrict__ Tg, const float* __restrict__ actions_in, float delay, long long common_step, \
                           const float* __restrict__ noise_in, float* __restrict__ obs_out, float* __restrict__ pri_out, const StepSeq sq, const float* __restrict__ dbg
template <bool HF, bool DBG = false>
__global__ __launch_bounds__(64 * TWAVES_MAX) __attribute__((amdgpu_waves_per_eu(1, 1))) void grx_step_tree(GRX_STEP_TREE_ARGS) {
#include "grx_step_tree_body.inc"
}
template <bool DBG = false>   // mesh_type 'trimesh' (grx_kernels.hip terrain_eval / wall_contact)
__global__ __launch_bounds__(64 * TWAVES_MAX) __attribute__((amdgpu_waves_per_eu(1, 1))) void grx_step_tree_trimesh(GRX_STEP_TREE_ARGS) {
    constexpr int HF = GRX_HF_TRIMESH;
#include "grx_step_tree_body.inc"
}
#undef TW
#undef TBO
