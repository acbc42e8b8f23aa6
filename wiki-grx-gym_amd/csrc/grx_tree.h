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
template <bool HF, bool LF = true>   // LF: add the force to the link's accumulator here (false: the caller does, in its turn)
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
template <bool HF>
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
            pr.cc[u] = P.hf_cells[terrain_record(P, cell, pr.tx[u], pr.ty[u])];
        }
    }
    return pr;
}
template <bool HF>
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
                    if (HF && P.vertical_faces && E.B.pos.z + pr.xr[u].z - T.sph[s0 + u].r <= E.hmax) {   // mesh_type 'trimesh': the vertical faces next to the shape
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
            const TreeBody& tb = T.body[b];
            const int wb = TBO(b);
            const V3 rho = in.rho;
            float t = in.t;
            const float arm = in.arm, qlo = in.qlo, qhi = in.qhi, Klim = in.Klim, Clim = in.Clim;
            pa = pa + in.pab; pl = pl + in.plb;
            add_rigid(A, Bm, D, in.Ar, in.hk, in.mass);
            if (CH.hcmask & (1u << g))   // chains that hang from this body, fixed order
                for (int k = 0; k < tb.nhc; ++k) tree_add_up(wsw, ei, o.up + tb.hc[k] * T_UPW, A, Bm, D, pa, pl);
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
template <bool HF, bool DBG = false>
__global__ __launch_bounds__(64 * TWAVES_MAX) __attribute__((amdgpu_waves_per_eu(1, 1))) void grx_step_tree(const KParams* __restrict__ Pg, const TreeTab* __restrict__ Tt, const GenTables* __restrict__ Tg,
                                                             const float* __restrict__ actions_in, float delay, long long common_step,
                                                             const float* __restrict__ noise_in, float* __restrict__ obs_out, float* __restrict__ pri_out,
                                                             const StepSeq sq, const float* __restrict__ dbg = nullptr) {
    KP P = GRX_PARAMS(Pg);
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    __shared__ float s_stat[TWAVES_MAX][NSTAT];   // a row per wave, added in wave order at the end: the same sums on every run (float atomics of four waves
                                                  // would add in arrival order)
    TreeTab& Tm = *reinterpret_cast<TreeTab*>(s_dyn);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(Tt);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_dyn);
        for (int i = threadIdx.x; i < (int)(sizeof(TreeTab) / 4); i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < TWAVES_MAX * NSTAT; i += blockDim.x) (&s_stat[0][0])[i] = 0.f;
    }
    __syncthreads();
    const TreeTab& T = Tm;
    const int nwaves = blockDim.x >> 6, tepb = TEPW * nwaves;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, ew = lane / TG, c = lane & (TG - 1);   // ew: the env of the wave
    if (wave == nwaves - 1) stats_fold_previous(P, sq, lane);   // the previous launch's episode statistics (and its ticket)
    const TreeOff o = tree_offsets(T.nb, T.nlc, T.nchain, T.nsph);
    const int thalf = tree_half_words(o.total);
    float* const wsw = s_dyn + sizeof(TreeTab) / 4 + (size_t)wave * 2 * thalf;
    const int ei = (ew / TEH) * thalf + (ew % TEH);   // (TW: the lane's offset into its half's block)
    const size_t N = (size_t)P.N;
    const int grp = step_group();   // (XCD-aware: grx_kernels.hip)
    const int e_raw = grp * tepb + wave * TEPW + ew;
    const bool act = e_raw < P.N;
    const int e = act ? e_raw : P.N - 1;
    const bool lead = c == 0, actl = act && lead;
    const int nd = T.nd;
    const uint32_t genv = (uint32_t)(P.env_offset + e), step = (uint32_t)common_step;
    const int nh = P.nh, nobs = 9 + 3 * nd, npri = P.num_pri_obs;
    const float dtp = P.sim_dt * (float)P.decimation;
    // ---- load the state: the base in every lane of the group, a chain's joints by its lane
    TreeEnv E;
    E.B.pos = v3(P.root[e], P.root[N + e], P.root[2 * N + e]);
    E.B.qx = P.root[3 * N + e]; E.B.qy = P.root[4 * N + e]; E.B.qz = P.root[5 * N + e]; E.B.qw = P.root[6 * N + e];
    E.B.vel = v3(P.root[7 * N + e], P.root[8 * N + e], P.root[9 * N + e]);
    E.B.ang = v3(P.root[10 * N + e], P.root[11 * N + e], P.root[12 * N + e]);
    E.base_m = P.base_m[e];
    E.base_c = v3(P.base_c[e], P.base_c[N + e], P.base_c[2 * N + e]);
    E.base_I = S3{P.base_I[e], P.base_I[N + e], P.base_I[2 * N + e], P.base_I[3 * N + e], P.base_I[4 * N + e], P.base_I[5 * N + e]};
    E.mu = 0.5f * (P.terrain_friction + P.friction[e]);
    E.om_e = 1.0f - 0.5f * (P.terrain_restitution + P.restitution[e]);
    E.hmax = 0.f;
    if (HF) {
        int ci = min(max((int)((E.B.pos.x + P.border_size) / (P.horizontal_scale * (float)GRX_COARSE)), 0), P.coarse_rows - 1);
        int cj = min(max((int)((E.B.pos.y + P.border_size) / (P.horizontal_scale * (float)GRX_COARSE)), 0), P.coarse_cols - 1);
        E.hmax = P.coarse_max[(size_t)ci * P.coarse_cols + cj];
    }
    TreeRegs G;
    const float *const qcol = P.q + e, *const qdcol = P.qd + e, *const lacol = P.last_actions + e, *const mscol = P.motor_strength + e;   // (per-lane column bases: the ten unrolled levels below each re-fetched the four pointers from the parameter block)
    TreeChain CH;
    CH.first = T.first[c]; CH.last = T.last[c]; CH.hangp = 0; CH.hcmask = 0u;
    const int nstep = __builtin_amdgcn_readfirstlane(T.nstep);
#pragma unroll
    for (int g = 0; g < TNG; ++g) {
        const int b = g < nstep ? (int)T.sched[c][g] : -1;
        G.sb[g] = b;
        G.q[g] = 0.f; G.qd[g] = 0.f; G.Sa[g] = v3(0.f, 0.f, 0.f);
        if (b >= 0) {
            const int j = b - 1;
            G.q[g] = qcol[(size_t)j * N]; G.qd[g] = qdcol[(size_t)j * N];
            TW(TBO(b) + T_Q) = G.q[g]; TW(TBO(b) + T_QD) = G.qd[g];   // (the joint-local phase reads them from the row)
            if (g == CH.first) CH.hangp = T.body[b].parent;
            if (T.body[b].nhc > 0) CH.hcmask |= 1u << g;
            const float a = actions_in ? actions_in[(size_t)e * nd + j] : 0.f;
            TW(o.dof + TD_ACUR * GRX_MAX_DOFS + j) = fminf(fmaxf(a, T.dof[j].amin), T.dof[j].amax);   // clip_actions (legged_robot_fftai.py:171-177)
            TW(o.dof + TD_ALAST * GRX_MAX_DOFS + j) = lacol[(size_t)j * N];
            TW(o.dof + TD_STR * GRX_MAX_DOFS + j) = mscol[(size_t)j * N];
        }
    }
    if (c < 8) {   // 8 anchor slots x (x, y, approach speed): one slot per lane (of the group's first eight)
#pragma unroll
        for (int k = 0; k < 3; ++k) TW(o.an + c * 3 + k) = P.anchors[(size_t)(c * 3 + k) * N + e];
    }
    tree_fence();
    // ---- during_physics_step (legged_robot_fftai.py:51-88)
    float avg_force[2] = {0.f, 0.f};
    V3 avg_speed[2] = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
    V3 avg_rpy[2] = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};   // avg_feet_speed_rpy (legged_robot_fftai.py:81, 88)
#ifdef GRX_PROFILE_SECTIONS
    long long tt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tt_prev = clock64();   // (8..: finer stamps inside the passes, slots 20.. of the block's row)
    const long long tt_begin = tt_prev;
#define TT(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); tt_acc[i] += t_ - tt_prev; tt_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TT(i) do {} while (0)
#endif
    if (DBG) {
        for (int i = c; i < T.nlc * 3; i += TG) TW(o.lf + i) = 0.f;
        tree_fence();
    }
    for (int deci = 0; deci < (DBG ? 0 : P.decimation); ++deci) {
        TT(7);
        asm volatile("" ::: "memory");   // (keeps the loop-invariant table reads of the unrolled passes in LDS: hoisted, they would spill)
        for (int i = c; i < T.nlc * 3; i += TG) TW(o.lf + i) = 0.f;
        if (c < 6) TW(o.misc + 8 + c) = 0.f;
        tree_fence();
        const R3 R0 = quat_to_R(E.B.qx, E.B.qy, E.B.qz, E.B.qw);
        tree_joint_phase<false>(P, T, wsw, ei, c, o, (float)deci < delay, P.last_dof_vel + e);
        TT(8);
        tree_outward<false>(P, T, wsw, ei, c, o, E, R0, CH, nstep, G);
        TT(9);
        {
            const TreeContactPre pre0 = tree_contact_probe<HF>(P, T, wsw, ei, c, o, E, R0, 0);   // (the terrain gathers fly behind the bias forces)
            TT(10);
            tree_bias_all(T, wsw, ei, c);
            TT(0);
            tree_contacts<HF>(P, T, wsw, ei, c, o, E, R0, pre0);
        }
        TT(1);
        // base: rigid lump (randomised per env); the terrain wrench on its own shapes was left in o.misc by the contact pass
        S3 Ab; V3 h0;
        rigid_inertia(R0, rot(R0, E.base_c), E.base_m, E.base_I, Ab, h0);
        V3 pa0, pl0;
        rigid_bias(R0, rot(R0, E.base_c), E.base_m, E.base_I, E.B.ang, E.B.vel, pa0, pl0);
        pa0 = pa0 + tw_v3(wsw, ei, o.misc + 8); pl0 = pl0 + tw_v3(wsw, ei, o.misc + 11);
        tree_fence();
        TT(2);
        if (P.self_collisions) tree_self_collision(P, T, wsw, ei, c, o, E, R0, pa0, pl0);
        TT(3);
        tree_rigid_inertias(T, wsw, ei, c);
        TT(11);
        tree_inward(P, T, wsw, ei, c, o, CH, nstep, G);
        TT(4);
        // ---- base: the chains that hang from it, in table order; [A B; B^T D][alpha; acc] = -[pa; pl]
        S3 Db = {E.base_m, 0.f, 0.f, E.base_m, 0.f, E.base_m};
        M3 Bb = {0.f, -h0.z, h0.y, h0.z, 0.f, -h0.x, -h0.y, h0.x, 0.f};
        for (int k = 0; k < T.nh0; ++k) tree_add_up(wsw, ei, o.up + T.heads0[k] * T_UPW, Ab, Bb, Db, pa0, pl0);
        const S3 Di = inv(Db);
        const V3 b0 = v3(Bb.a00, Bb.a01, Bb.a02), b1 = v3(Bb.a10, Bb.a11, Bb.a12), b2 = v3(Bb.a20, Bb.a21, Bb.a22);
        const V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
        const S3 Sc = {Ab.xx - dot(b0, d0), Ab.xy - dot(b0, d1), Ab.xz - dot(b0, d2), Ab.yy - dot(b1, d1), Ab.yz - dot(b1, d2), Ab.zz - dot(b2, d2)};
        const V3 alpha = mul(inv(Sc), mul(Bb, mul(Di, pl0)) - pa0);
        const V3 acc = neg(mul(Di, pl0 + mulT(Bb, alpha)));
        TT(5);
        tree_accel(P, T, wsw, ei, c, alpha, acc, CH, nstep, G);
        TT(6);
        {   // integrate the base (semi-implicit Euler), every lane of the group alike
            const float dt = P.sim_dt;
            GenBase& B = E.B;
            const V3 lin = acc + cross(B.ang, B.vel);
            B.vel = v3(B.vel.x + (lin.x + P.gravity[0]) * dt, B.vel.y + (lin.y + P.gravity[1]) * dt, B.vel.z + (lin.z + P.gravity[2]) * dt);
            B.ang = fma3(alpha, dt, B.ang);
            B.pos = fma3(B.vel, dt, B.pos);
            const float hx = 0.5f * dt * B.ang.x, hy = 0.5f * dt * B.ang.y, hz = 0.5f * dt * B.ang.z;
            const float x = B.qx, y = B.qy, z = B.qz, ww = B.qw;
            const float nx = x + hx * ww + hy * z - hz * y, ny = y - hx * z + hy * ww + hz * x;
            const float nz = z + hx * y - hy * x + hz * ww, nw = ww - hx * x - hy * y - hz * z;
            const float n = grx_rsq(nx * nx + ny * ny + nz * nz + nw * nw);
            B.qx = nx * n; B.qy = ny * n; B.qz = nz * n; B.qw = nw * n;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (deci > 0) {
                const V3 fv = tw_v3(wsw, ei, o.misc + f * 3);
                avg_speed[f] = v3(avg_speed[f].x + fabsf(fv.x), avg_speed[f].y + fabsf(fv.y), avg_speed[f].z + fabsf(fv.z));
                const V3 fw = tw_v3(wsw, ei, TBO(T.foot_body[f]) + T_W);   // (the walk's: BEFORE this sub-step's integration, like fv)
                avg_rpy[f] = v3(avg_rpy[f].x + fabsf(fw.x), avg_rpy[f].y + fabsf(fw.y), avg_rpy[f].z + fabsf(fw.z));
            }
            const V3 F = tw_v3(wsw, ei, o.lf + T.foot_link[f] * 3);
            avg_force[f] += grx_sqrt(dot(F, F));
        }
    }
#ifdef GRX_PROFILE_SECTIONS
    const long long tt_phys = clock64() - tt_begin;
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 8] = tt_phys;
#endif
    // ---- what only the env pipeline behind the sub-steps reads is requested HERE (round 6: loaded at the kernel's start, these ~17 values sat in
    // registers through all ten sub-steps of a kernel that needs every one of its 512); the final-frames walk below hides the latency
    EnvAux ea;
    ea.cmd[0] = P.commands[e]; ea.cmd[1] = P.commands[N + e]; ea.cmd[2] = P.commands[2 * N + e];
    ea.origin[0] = P.origins[e]; ea.origin[1] = P.origins[N + e]; ea.origin[2] = P.origins[2 * N + e];
    ea.level = P.levels[e]; ea.type = P.types[e];
    float air_time[2] = {P.air_time[e], P.air_time[N + e]}, land_time[2] = {P.land_time[e], P.land_time[N + e]};
    bool contact_last[2] = {P.feet_contact[e] != 0, P.feet_contact[N + e] != 0};
    const float bho_stale = P.base_heights_offset[e];
    long long ep_len = P.ep_len[e];
    // (the running episode sums too, AHEAD of the kernel's first global stores: vmcnt counts loads and stores in one order, so a load requested
    //  behind the contact_forces / height rows waits for those stores to be acknowledged by memory -- 12 k cycles of this section, measured)
    // The reward scales, ONE per lane (lane t holds term t's), and the set of active terms as one scalar mask: the three loops over the terms
    // below asked the parameter block for scale[t] -- and for the episode_sums pointer again -- behind a branch per term: ~100 dependent
    // scalar-load round trips, 12 k cycles of this kernel's tail (from the ISA: s_load_dword / s_waitcnt lgkmcnt(0) / branch / s_load_dwordx2 ...).
    static_assert(NT <= 64, "a reward term per lane");
    const float scale_v = P.reward_scale_dt[lane < NT ? lane : 0];
    const unsigned long long term_on = __ballot(lane < NT && scale_v != 0.f);
    auto scale_of = [&](int t) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scale_v), t)); };
    float* const es_col = P.episode_sums + e;
    float es_old[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) es_old[t] = ((term_on >> t) & 1ull) ? es_col[(size_t)t * N] : 0.f;
    // ---- refresh_rigid_body_state_tensor after the last sub-step: frames of the final state
    {
        const R3 R0 = quat_to_R(E.B.qx, E.B.qy, E.B.qz, E.B.qw);
        tree_joint_phase<true>(P, T, wsw, ei, c, o, false, nullptr);
        tree_outward<true>(P, T, wsw, ei, c, o, E, R0, CH, nstep, G);
    }
    if (P.publish_rbs) {   // GRX_T_RIGID_BODY_STATES (legged_robot.py:113,134): every URDF link frame of that state, the links go round the lanes
        const LinkTab& LT = *P.link_tab;
        const R3 Rb0 = quat_to_R(E.B.qx, E.B.qy, E.B.qz, E.B.qw);
        for (int l = c; l < LT.n; l += TG) {
            const int b = LT.body[l];
            const R3 Rb = b == 0 ? Rb0 : tw_R(wsw, ei, TBO(b) + T_R);
            const V3 rho_b = b == 0 ? v3(0.f, 0.f, 0.f) : tw_v3(wsw, ei, TBO(b) + T_RHO);
            const V3 w_b = b == 0 ? E.B.ang : tw_v3(wsw, ei, TBO(b) + T_W), v_b = b == 0 ? E.B.vel : tw_v3(wsw, ei, TBO(b) + T_V);
            const V3 r_ = rho_b + rot(Rb, v3(LT.pos[l][0], LT.pos[l][1], LT.pos[l][2]));
            const V3 vl = v_b + cross(w_b, r_);
            // R_link = R_body * (link -> body), row-major entries m[i][k]
            float m[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const V3 col = rot(Rb, v3(LT.rot[l][k], LT.rot[l][3 + k], LT.rot[l][6 + k]));   // column k of the link rotation, in world axes
                m[k] = col.x; m[3 + k] = col.y; m[6 + k] = col.z;
            }
            float qx, qy, qz, qw;   // largest-component form (the oracle's m3_to_quat)
            const float t0 = 1 + m[0] - m[4] - m[8], t1 = 1 - m[0] + m[4] - m[8], t2 = 1 - m[0] - m[4] + m[8], t3 = 1 + m[0] + m[4] + m[8];
            if (t3 >= t0 && t3 >= t1 && t3 >= t2) { qx = m[7] - m[5]; qy = m[2] - m[6]; qz = m[3] - m[1]; qw = t3; }
            else if (t0 >= t1 && t0 >= t2) { qx = t0; qy = m[1] + m[3]; qz = m[2] + m[6]; qw = m[7] - m[5]; }
            else if (t1 >= t2) { qx = m[1] + m[3]; qy = t1; qz = m[5] + m[7]; qw = m[2] - m[6]; }
            else { qx = m[2] + m[6]; qy = m[5] + m[7]; qz = t2; qw = m[3] - m[1]; }
            const float qn = grx_rsq(qx * qx + qy * qy + qz * qz + qw * qw);
            if (act) {
                float* o_ = P.rbs + (size_t)(l * 13) * N + e;
                o_[0] = E.B.pos.x + r_.x; o_[N] = E.B.pos.y + r_.y; o_[2 * N] = E.B.pos.z + r_.z;
                o_[3 * N] = qx * qn; o_[4 * N] = qy * qn; o_[5 * N] = qz * qn; o_[6 * N] = qw * qn;
                o_[7 * N] = vl.x; o_[8 * N] = vl.y; o_[9 * N] = vl.z;
                o_[10 * N] = w_b.x; o_[11 * N] = w_b.y; o_[12 * N] = w_b.z;
            }
        }
    }
    V3 fpos[2], fvel[2], foot_force[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int b = T.foot_body[f];
        const R3 R = tw_R(wsw, ei, TBO(b) + T_R);
        const V3 fr = tw_v3(wsw, ei, TBO(b) + T_RHO) + rot(R, v3(T.foot_pos[f][0], T.foot_pos[f][1], T.foot_pos[f][2]));
        fpos[f] = E.B.pos + fr;
        fvel[f] = tw_v3(wsw, ei, TBO(b) + T_V) + cross(tw_v3(wsw, ei, TBO(b) + T_W), fr);
        avg_speed[f] = v3((avg_speed[f].x + fabsf(fvel[f].x)) / (float)P.decimation, (avg_speed[f].y + fabsf(fvel[f].y)) / (float)P.decimation,
                          (avg_speed[f].z + fabsf(fvel[f].z)) / (float)P.decimation);
        avg_force[f] /= (float)P.decimation;
        {
            const V3 fw = tw_v3(wsw, ei, TBO(b) + T_W);
            avg_rpy[f] = v3((avg_rpy[f].x + fabsf(fw.x)) / (float)P.decimation, (avg_rpy[f].y + fabsf(fw.y)) / (float)P.decimation, (avg_rpy[f].z + fabsf(fw.z)) / (float)P.decimation);
        }
        foot_force[f] = tw_v3(wsw, ei, o.lf + T.foot_link[f] * 3);
    }
    bool dbg_apply_reset = true;
    if (DBG) {   // injected "physics results" (rows of DbgRow, [row][N])
        const float* d = dbg + e;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            foot_force[f] = v3(d[(size_t)(DBG_FEET_FORCE + f * 3) * N], d[(size_t)(DBG_FEET_FORCE + f * 3 + 1) * N], d[(size_t)(DBG_FEET_FORCE + f * 3 + 2) * N]);
            fpos[f] = v3(d[(size_t)(DBG_FEET_POS + f * 3) * N], d[(size_t)(DBG_FEET_POS + f * 3 + 1) * N], d[(size_t)(DBG_FEET_POS + f * 3 + 2) * N]);
            avg_force[f] = d[(size_t)(DBG_AVG_FORCE + f) * N];
            avg_speed[f] = v3(d[(size_t)(DBG_AVG_SPEED + f * 3) * N], d[(size_t)(DBG_AVG_SPEED + f * 3 + 1) * N], d[(size_t)(DBG_AVG_SPEED + f * 3 + 2) * N]);
        }
#pragma unroll
        for (int g = 0; g < TNG; ++g)
            if (G.sb[g] >= 0) TW(TBO(G.sb[g]) + T_TAU) = d[(size_t)(DBG_TORQUES + G.sb[g] - 1) * N];
        dbg_apply_reset = d[(size_t)DBG_APPLY_RESET * N] != 0.f;
        tree_fence();
    }
    // termination / collision from the per-link net forces of the LAST sub-step (legged_robot.py:336-353); contact_forces rows
    bool term_contact = false;
    float pen_count = 0.f;
    for (int L = 0; L < T.nlc; ++L) {
        const V3 F = tw_v3(wsw, ei, o.lf + L * 3);
        const float n2 = dot(F, F);
        if ((T.link_flags[L] & GRX_SPH_TERMINATE) && n2 > P.termination_force * P.termination_force) term_contact = true;
        if ((T.link_flags[L] & GRX_SPH_PENALISE) && n2 > 0.01f) pen_count += 1.f;
        if (act && (L & (TG - 1)) == c) {
            float* cf = P.contact_forces + (size_t)(T.link_urdf[L] * 3) * N + e;
            cf[0] = F.x; cf[N] = F.y; cf[2 * N] = F.z;
        }
    }
    if (DBG) { term_contact = dbg[(size_t)DBG_TERM_CONTACT * N + e] != 0.f; pen_count = 0.f; }
    float torso_g[2] = {0.f, 0.f}, fore_g[2] = {0.f, 0.f};
    if (T.torso_body >= 0) {
        const R3 R = T.torso_body == 0 ? quat_to_R(E.B.qx, E.B.qy, E.B.qz, E.B.qw) : tw_R(wsw, ei, TBO(T.torso_body) + T_R);
        torso_g[0] = -(R.cx.z * T.torso_rot[0] + R.cy.z * T.torso_rot[3] + R.cz.z * T.torso_rot[6]);
        torso_g[1] = -(R.cx.z * T.torso_rot[1] + R.cy.z * T.torso_rot[4] + R.cz.z * T.torso_rot[7]);
    }
    if (T.forehead_body >= 0) {
        const R3 R = T.forehead_body == 0 ? quat_to_R(E.B.qx, E.B.qy, E.B.qz, E.B.qw) : tw_R(wsw, ei, TBO(T.forehead_body) + T_R);
        fore_g[0] = -(R.cx.z * T.forehead_rot[0] + R.cy.z * T.forehead_rot[3] + R.cz.z * T.forehead_rot[6]);
        fore_g[1] = -(R.cx.z * T.forehead_rot[1] + R.cy.z * T.forehead_rot[4] + R.cz.z * T.forehead_rot[7]);
    }
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 10] = clock64() - tt_begin;
#endif
    // ---- post_physics_step (legged_robot.py:269-334)
    GenBase& B = E.B;
    ep_len += 1;
    const V3 qv = v3(B.qx, B.qy, B.qz);
    const V3 blv = quat_rotate_inverse(qv, B.qw, B.vel), bav = quat_rotate_inverse(qv, B.qw, B.ang);
    const V3 pg = quat_rotate_inverse(qv, B.qw, v3(0.f, 0.f, -1.f));
    if (P.resample_command_interval > 0 && ((uint32_t)ep_len % (uint32_t)P.resample_command_interval) == 0)
        resample_commands(P, genv, step, GRX_RNG_CMD_TIME, ea.cmd);
    if (P.heading_command) ea.cmd[2] = heading_yaw_command(P, qv, B.qw);   // legged_robot.py:320-326
    float* heights = P.heights + e;   // raw measured heights: the scan's points go round the group's lanes
    float hsum = 0.f;
    // the lane's points of the scan (k = c, c + TG, ...): unrolled over the most a lane can hold, so that the points' table reads and raster
    // gathers are in flight together (a rolled loop made every point two exposed memory round trips), and kept in registers for the
    // observation block below (it used to read them back from memory)
    constexpr int HPL = (GRX_MAX_HEIGHT_POINTS + TG - 1) / TG;
    float hraw[HPL];
#pragma unroll
    for (int i = 0; i < HPL; ++i) hraw[i] = 0.f;
    if (HF && P.measure_heights) {
        const float yaw_n = fmaxf(sqrtf(B.qz * B.qz + B.qw * B.qw), 1e-9f);
        const float yz = B.qz / yaw_n, yw = B.qw / yaw_n;
        // (a lane's points are sampled WITHOUT a branch, the index clamped -- as the fused kernels' height_scan_share does: under `if (k < nh)`
        //  every point re-fetched the raster's parameters from the parameter block behind its own branch, eight dependent scalar round trips)
        const bool pub_h = act && P.publish_heights;
#pragma unroll
        for (int i = 0; i < HPL; ++i) hraw[i] = height_sample(P, *P.tables, yz, yw, B.pos, min(c + i * TG, nh - 1));
#pragma unroll
        for (int i = 0; i < HPL; ++i) {   // (summed in the order of the rolled loop)
            const int k = c + i * TG;
            if (k < nh) { if (pub_h) heights[(size_t)k * N] = hraw[i]; hsum += hraw[i]; } else hraw[i] = 0.f;
        }
        hsum = grp_sum(hsum);
    } else {
        const bool pub_h = act && P.publish_heights;
#pragma unroll
        for (int i = 0; i < HPL; ++i) { const int k = c + i * TG; if (k < nh && pub_h) heights[(size_t)k * N] = 0.f; }
    }
    if (P.push_robots && P.push_interval > 0 && (step % (uint32_t)P.push_interval) == 0) {
        if (P.stash_pre_reset && actl) { P.pre_push_vel[e] = B.vel.x; P.pre_push_vel[(size_t)N + e] = B.vel.y; }   // (grx_refresh: link frames of the state BEFORE the push)
        B.vel.x = urand(P, genv, step, GRX_RNG_PUSH, 0, -P.max_push_vel_xy, P.max_push_vel_xy);
        B.vel.y = urand(P, genv, step, GRX_RNG_PUSH, 1, -P.max_push_vel_xy, P.max_push_vel_xy);
    }
    // feet timers (legged_robot_fftai.py:108-133)
    bool contact[2], contact_filt[2], first_contact[2];
    float feet_height[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        contact[f] = foot_force[f].z > 1.0f;
        contact_filt[f] = contact[f] || contact_last[f];
        contact_last[f] = contact[f];
        first_contact[f] = (air_time[f] > 0.f) && contact_filt[f];
        air_time[f] += dtp;
        feet_height[f] = nh > 0 ? (fpos[f].z * (float)nh - hsum) / (float)nh : fpos[f].z;
        land_time[f] = (land_time[f] + dtp) * (contact[f] ? 1.f : 0.f);
    }
    bool reset = term_contact || (fabsf(pg.z) < P.termination_gravity_z);
    const bool time_out = (float)ep_len > P.max_episode_length;
    reset = reset || time_out;
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 11] = clock64() - tt_begin;
#endif
    // ---- compute_reward (legged_robot.py:355-375; terms legged_robot_fftai.py:180-352, gr1t1.py:338-589): a lane sums over
    // its chain's joints, the group adds up
    float r[NT];
    // (the running episode sums were requested in one batch behind the sub-steps: read inside the loop that folds them, behind its wave-uniform
    //  branches, each of the 36 loads was an exposed round trip: 27 k cycles, gpu_tree_sections.py)
    // The joint terms go round the group's lanes BY JOINT (two rounds of 16 lanes, four of 8) instead of riding on the ten depth levels of
    // the lane's chain (a wave executes all ten whatever its lanes hold): the chains' owners publish q, qd in two slots of the body's row
    // that are dead behind the sub-steps.
    enum { T_QPUB = T_PL + 1, T_QDPUB = T_PL + 2 };
#pragma unroll
    for (int g = 0; g < TNG; ++g)
        if (G.sb[g] >= 0) { TW(TBO(G.sb[g]) + T_QPUB) = G.q[g]; TW(TBO(G.sb[g]) + T_QDPUB) = G.qd[g]; }
    tree_fence();
    {
        const float as = P.action_scale, H = P.swing_feet_height_target, Tt_ = P.feet_air_time_target;
        const GRX_AS4 float* sg = P.reward_sigma;
        float s2 = 0.f;   // DBG: the injected last_last_actions (otherwise last_last_actions == last_actions, legged_robot_fftai.py:94)
        float s1 = 0.f, s3 = 0.f, sacc = 0.f, stor = 0.f, svel = 0.f, spose = 0.f, sla = 0.f, slp = 0.f, slt = 0.f, slv = 0.f, shy = 0.f;
        float tor_hr = 0.f, vel_kn = 0.f, ank[2] = {0.f, 0.f};
        for (int j = c; j < nd; j += TG) {
            const TreeDof& td = T.dof[j];
            const float ac = TW(o.dof + TD_ACUR * GRX_MAX_DOFS + j), al = TW(o.dof + TD_ALAST * GRX_MAX_DOFS + j), tj = TW(TBO(j + 1) + T_TAU);
            const float qj = TW(TBO(j + 1) + T_QPUB), qdj = TW(TBO(j + 1) + T_QDPUB);
            const uint32_t bit = 1u << j;
            s1 += fabsf((al - ac) * as);
            if (DBG) s2 += fabsf((al - ac) * as - (dbg[(size_t)(DBG_LAST_LAST_ACTIONS + j) * N + e] - al) * as);
            if (P.knee_mask & bit) { s3 += fabsf((ac - al) * as); vel_kn += fabsf(qdj); }
            sacc += fabsf((qdj - P.last_dof_vel[(size_t)j * N + e]) / dtp);
            stor += fabsf(tj);
            svel += fabsf(qdj);
            const float po = fabsf(qj - td.q0);
            spose += po;
            if (P.hip_yaw_mask & bit) shy += po;
            if (P.hip_roll_mask & bit) tor_hr += fabsf(tj);
            if (P.ankle_left_mask & bit) ank[0] += fabsf(tj);
            if (P.ankle_right_mask & bit) ank[1] += fabsf(tj);
            const float a = ac * as;
            float oa = 0.f, op = 0.f;
            if (a - td.slo < 0.f) oa += -(a - td.slo);
            if (a - td.shi > 0.f) oa += (a - td.shi);
            sla += oa * oa;
            if (qj - td.slo < 0.f) op += -(qj - td.slo);
            if (qj - td.shi > 0.f) op += (qj - td.shi);
            slp += fabsf(op);
            slv += fminf(fmaxf(fabsf(qdj) - td.vlim * P.soft_dof_vel_limit, 0.f), 1.f);
            slt += fmaxf(fabsf(tj) - td.effort * P.soft_torque_limit, 0.f);
        }
#ifdef GRX_PROFILE_SECTIONS
        if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 14] = clock64() - tt_begin;
#endif
        s2 = DBG ? grp_sum(s2) : 0.f;
        s1 = grp_sum(s1); s3 = grp_sum(s3); sacc = grp_sum(sacc); stor = grp_sum(stor); svel = grp_sum(svel); spose = grp_sum(spose);
        sla = grp_sum(sla); slp = grp_sum(slp); slt = grp_sum(slt); slv = grp_sum(slv); shy = grp_sum(shy);
        tor_hr = grp_sum(tor_hr); vel_kn = grp_sum(vel_kn); ank[0] = grp_sum(ank[0]); ank[1] = grp_sum(ank[1]);
#ifdef GRX_PROFILE_SECTIONS
        if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 15] = clock64() - tt_begin;
#endif
        const float hmin = fminf(feet_height[0], feet_height[1]);
        float lift = 0.f, af = 0.f, ah = 0.f, at = 0.f, lt = 0.f, exy = 0.f, ez = 0.f, stum = 0.f, ncontact = 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float h = feet_height[f];
            lift += ank[f] * fabsf(h) * (h > H * 0.5f ? 1.f : 0.f);
            const float mid = fabsf(air_time[f] - Tt_ * 0.5f);
            af += mid * avg_force[f];
            ah += mid * fabsf(h - hmin - H);
            at += expf(sg[GRX_REW_FEET_AIR_TIME] * fabsf(air_time[f] - Tt_)) * (first_contact[f] ? 1.f : 0.f);
            const float le = (land_time[f] - P.feet_land_time_max) * (land_time[f] > P.feet_land_time_max ? 1.f : 0.f);
            lt += 1.f - expf(sg[GRX_REW_FEET_LAND_TIME] * le);
            const float close = fabsf(h - H * 0.25f) * (h < H * 0.25f ? 1.f : 0.f) / (H * 0.25f);
            exy += sqrtf(avg_speed[f].x * avg_speed[f].x + avg_speed[f].y * avg_speed[f].y) * close;
            const float far = fabsf(h - H * 3.f / 4.f) * (h > H * 3.f / 4.f ? 1.f : 0.f) / (H * 1.f / 4.f);
            ez += fabsf(avg_speed[f].z) * far;
            const V3 F = foot_force[f];
            float serr = sqrtf(F.x * F.x + F.y * F.y) - P.feet_stumble_ratio * fabsf(F.z);
            serr = serr * (serr > 0.f ? 1.f : 0.f);
            stum += 1.f - expf(sg[GRX_REW_FEET_STUMBLE] * serr);
            ncontact += contact[f] ? 1.f : 0.f;
        }
        const float cmd_n = sqrtf(ea.cmd[0] * ea.cmd[0] + ea.cmd[1] * ea.cmd[1]);
        const float moving = cmd_n > 0.1f ? 1.f : 0.f;
        r[GRX_REW_ACTION_DIFF] = 1.f - expf(sg[GRX_REW_ACTION_DIFF] * s1);
        r[GRX_REW_ACTION_DIFF_DIFF] = 1.f - expf(sg[GRX_REW_ACTION_DIFF_DIFF] * (DBG ? s2 : s1));   // last_last_actions == last_actions
        r[GRX_REW_ACTION_DIFF_KNEE] = 1.f - expf(sg[GRX_REW_ACTION_DIFF_KNEE] * s3);
        r[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] * fabsf(0.f - bav.y));
        r[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] * fabsf(0.f - bav.x));
        r[GRX_REW_CMD_DIFF_ANG_VEL_YAW] = expf(sg[GRX_REW_CMD_DIFF_ANG_VEL_YAW] * fabsf(ea.cmd[2] - bav.z));
        r[GRX_REW_CMD_DIFF_BASE_HEIGHT] = expf(sg[GRX_REW_CMD_DIFF_BASE_HEIGHT] * (fabsf(bho_stale) * (bho_stale < 0.f ? 1.f : 0.f)));
        r[GRX_REW_CMD_DIFF_BASE_ORIENT] = expf(sg[GRX_REW_CMD_DIFF_BASE_ORIENT] * (fabsf(pg.x) + fabsf(pg.y)));
        r[GRX_REW_CMD_DIFF_TORSO_ORIENT] = T.torso_body >= 0 ? expf(sg[GRX_REW_CMD_DIFF_TORSO_ORIENT] * (fabsf(torso_g[0]) + fabsf(torso_g[1]))) : 0.f;
        r[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] = T.forehead_body >= 0 ? expf(sg[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] * (fabsf(fore_g[0]) + fabsf(fore_g[1]))) : 0.f;
        r[GRX_REW_CMD_DIFF_LIN_VEL_X] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_X] * fabsf(ea.cmd[0] - blv.x));
        r[GRX_REW_CMD_DIFF_LIN_VEL_Y] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_Y] * fabsf(ea.cmd[1] - blv.y));
        r[GRX_REW_CMD_DIFF_LIN_VEL_Z] = expf(sg[GRX_REW_CMD_DIFF_LIN_VEL_Z] * fabsf(0.f - blv.z));
        r[GRX_REW_COLLISION] = 1.f - expf(sg[GRX_REW_COLLISION] * pen_count);
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
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 16] = clock64() - tt_begin;
#endif
    float rew = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float sc_t = scale_of(t);
        float rt = 0.f;
        if (t != GRX_REW_TERMINATION && ((term_on >> t) & 1ull)) { rt = r[t] * sc_t; rew += rt; }
        r[t] = rt;
    }
    if (P.only_positive_rewards) rew = fmaxf(rew, 0.f);
    if ((term_on >> GRX_REW_TERMINATION) & 1ull) {
        const float rt = r[GRX_REW_TERMINATION] = ((reset && !time_out) ? 1.f : 0.f) * scale_of(GRX_REW_TERMINATION);
        rew += rt;
    }
    // episode sums (the group's first lane); finished episodes -> the block's statistics row (deterministic lane order)
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 17] = clock64() - tt_begin;
#endif
    const unsigned long long reset_mask = __ballot(reset && actl);
    const bool publish_debug = P.publish_debug != 0;
    float* const rt_col = P.reward_terms + e;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float es = es_old[t] + r[t];
        if (reset_mask) {   // the finished episodes' sums, added in lane order: the groups' first lanes through v_readlane (an LDS shuffle per
            // finished env and term -- 36 dependent round trips per env -- was most of this section)
            const int esm = __float_as_int((reset && actl) ? es : 0.f);
            float acc_ = 0.f;
#pragma unroll
            for (int k = 0; k < TEPW; ++k) acc_ += __int_as_float(__builtin_amdgcn_readlane(esm, k * TG));
            if (lane == 0) s_stat[wave][t] = acc_;   // (the wave's row was zeroed at the kernel's start and this is its only writer: no read-modify-write)
        }
        if (actl && ((term_on >> t) & 1ull)) {
            es_col[(size_t)t * N] = (reset && dbg_apply_reset) ? 0.f : es;
            if (publish_debug) rt_col[(size_t)t * N] = r[t];
        }
    }
    if (lane == 0) s_stat[wave][NT] = (float)__popcll(reset_mask);
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 12] = clock64() - tt_begin;
#endif
    // ---- reset_idx (masked, in-kernel): a chain's joints by its lane, the base in every lane (same counters, same values)
    if (P.stash_pre_reset && reset && act) {   // on-demand tensors (grx_refresh) show the state BEFORE reset_idx: a chain's joints by its lane, the base by lane 0 of the group
        const size_t n_ = (size_t)N;
#pragma unroll
        for (int g = 0; g < TNG; ++g)
            if (G.sb[g] >= 0) { const int j = G.sb[g] - 1; P.pre_q[(size_t)j * n_ + e] = G.q[g]; P.pre_qd[(size_t)j * n_ + e] = G.qd[g]; }
        if (c == 0) {
            float* r_ = P.pre_root + e;
            r_[0] = B.pos.x; r_[n_] = B.pos.y; r_[2 * n_] = B.pos.z; r_[3 * n_] = B.qx; r_[4 * n_] = B.qy; r_[5 * n_] = B.qz; r_[6 * n_] = B.qw;
            r_[7 * n_] = B.vel.x; r_[8 * n_] = B.vel.y; r_[9 * n_] = B.vel.z; r_[10 * n_] = B.ang.x; r_[11 * n_] = B.ang.y; r_[12 * n_] = B.ang.z;
        }
    }
    const bool reported_reset = reset;   // (the debug entry may report a reset without applying it)
    if (DBG && !dbg_apply_reset) reset = false;
    if (reset) {
        if (P.curriculum && P.terrain_type != GRX_TERRAIN_PLANE) {
            const float dx = B.pos.x - ea.origin[0], dy = B.pos.y - ea.origin[1];
            const float dist = sqrtf(dx * dx + dy * dy);
            const int up = dist > P.terrain_length * 0.5f;
            const float cn = sqrtf(ea.cmd[0] * ea.cmd[0] + ea.cmd[1] * ea.cmd[1]);
            const int down = (dist < cn * P.max_episode_length_s * 0.5f) && !up;
            ea.level += up - down;
            if (ea.level >= P.num_terrain_rows) {
                const float u = grx_rand(P.seed, genv, step, GRX_RNG_CURRICULUM, 0);
                ea.level = min((int)(u * (float)P.num_terrain_rows), P.num_terrain_rows - 1);
            } else if (ea.level < 0)
                ea.level = 0;
            const float* og = P.terrain_origins + ((size_t)ea.level * P.num_terrain_cols + ea.type) * 3;
            ea.origin[0] = og[0]; ea.origin[1] = og[1]; ea.origin[2] = og[2];
        }
#pragma unroll
        for (int g = 0; g < TNG; ++g) {
            if (G.sb[g] < 0) continue;
            const int j = G.sb[g] - 1;
            const float f_ = P.randomize_init_dof_pos ? urand(P, genv, step, GRX_RNG_RESET_DOF, (uint32_t)j, 0.5f, 1.5f) : 1.0f;
            G.q[g] = f_ * T.dof[j].q0;
            G.qd[g] = 0.f;
        }
        B.pos = v3(P.init_pos[0] + ea.origin[0], P.init_pos[1] + ea.origin[1], P.init_pos[2] + ea.origin[2]);
        if (P.terrain_type != GRX_TERRAIN_PLANE) {
            B.pos.x += urand(P, genv, step, GRX_RNG_RESET_ROOT, 0, -1.0f, 1.0f);
            B.pos.y += urand(P, genv, step, GRX_RNG_RESET_ROOT, 1, -1.0f, 1.0f);
        }
        const float yaw = urand(P, genv, step, GRX_RNG_RESET_ROOT, 2, -6.283185307179586f, 6.283185307179586f);
        float sy, cy;
        sincosf(yaw * 0.5f, &sy, &cy);
        B.qx = 0.f; B.qy = 0.f; B.qz = sy; B.qw = cy;
        if (P.randomize_init_base_velocity) {
            B.vel = v3(urand(P, genv, step, GRX_RNG_RESET_ROOT, 3, -0.5f, 0.5f), urand(P, genv, step, GRX_RNG_RESET_ROOT, 4, -0.5f, 0.5f),
                       urand(P, genv, step, GRX_RNG_RESET_ROOT, 5, -0.5f, 0.5f));
            B.ang = v3(urand(P, genv, step, GRX_RNG_RESET_ROOT, 6, -0.5f, 0.5f), urand(P, genv, step, GRX_RNG_RESET_ROOT, 7, -0.5f, 0.5f),
                       urand(P, genv, step, GRX_RNG_RESET_ROOT, 8, -0.5f, 0.5f));
        } else {
            B.vel = v3(0.f, 0.f, 0.f);
            B.ang = v3(0.f, 0.f, 0.f);
        }
        resample_commands(P, genv, step, GRX_RNG_CMD_RESET, ea.cmd);
        if (c < 8) TW(o.an + c * 3 + 2) = 0.f;   // the lane's anchor slot
        for (int f = 0; f < 2; ++f) { air_time[f] = 0.f; land_time[f] = 0.f; contact_last[f] = false; }
        ep_len = 0;
    }
    {   // statistics row NT + 1: terrain levels after this step's curriculum moves (legged_robot.py:427-428)
        const float ls = level_sum(ea.level, actl);
        if (lane == 0) s_stat[wave][NT + 1] = ls;
    }
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) P.prof[(size_t)blockIdx.x * GRX_PROF_SLOTS + 13] = clock64() - tt_begin;
#endif
    // ---- compute_observations (legged_robot_fftai.py:148-167, gr1t1.py:281-336)
    float* obs = (obs_out ? obs_out : P.obs) + (size_t)e * nobs;
    float* pri = (pri_out ? pri_out : P.pri_obs) + (size_t)e * npri;
    const float clipo = P.clip_observations;
    float bho = 0.f;
    {
        float sum = 0.f;
        const float bht = P.base_height_target, osh = P.obs_scale_height;   // (fetched once, not behind every point's branch)
        float* const prih = pri + nobs + 8;
#pragma unroll
        for (int i = 0; i < HPL; ++i) {
            const int k = c + i * TG;
            if (k < nh) {
                float d = B.pos.z - bht - hraw[i];
                d = fminf(fmaxf(d, -1.f), 1.f) * osh;
                if (act) prih[k] = fminf(fmaxf(d * osh, -clipo), clipo);
                sum += d;
            }
        }
        sum = grp_sum(sum);
        bho = nh > 0 ? sum / (float)nh : 0.f;
    }
    // observation noise: the env's Philox blocks -- 2 of the base stream, ceil(3 (nd / 2) / 4) per half of the dof range (the oracle's scheme:
    // item i of a stream is word i & 3 of block i >> 2) -- computed ONCE, the blocks going round the group's lanes, and parked in the
    // bias-force slots of bodies 1.. (dead behind the sub-steps).  (Round 4, from the section profile: one whole Philox block per
    // observation ELEMENT, 30 per lane with the level loop unrolled, and an integer division to find the element's stream made this
    // section 40 k cycles of a 650 k step.)
#pragma unroll
    for (int g = 0; g < TNG; ++g)   // q, qd after reset_idx, for the joint-parallel loop below
        if (G.sb[g] >= 0) { TW(TBO(G.sb[g]) + T_QPUB) = G.q[g]; TW(TBO(G.sb[g]) + T_QDPUB) = G.qd[g]; }
    tree_fence();
    const int half_ = nd / 2, nblk_dof = (3 * half_ + 3) / 4, nblk = 2 + 2 * nblk_dof;
    const bool own_noise = P.add_noise && !noise_in, noise_lds = own_noise && nblk <= T.nb - 1;
    if (noise_lds) {
        for (int k = c; k < nblk; k += TG) {
            const uint32_t stream = k < 2 ? (uint32_t)GRX_RNG_NOISE : (k < 2 + nblk_dof ? (uint32_t)GRX_RNG_NOISE_DOF_L : (uint32_t)GRX_RNG_NOISE_DOF_R);
            const uint32_t blk = k < 2 ? (uint32_t)k : (uint32_t)(k < 2 + nblk_dof ? k - 2 : k - 2 - nblk_dof);
            const U4 o4 = grx_philox4x32_10(genv, step, stream, blk, (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
            const int wa = TBO(k + 1) + T_PA;
            TW(wa) = __uint_as_float(o4.x); TW(wa + 1) = __uint_as_float(o4.y); TW(wa + 2) = __uint_as_float(o4.z); TW(wa + 3) = __uint_as_float(o4.w);
        }
        tree_fence();
    }
    if (act) {
        // slot: block of the env's noise table (0, 1: base stream; 2..: left dofs; 2 + nblk_dof..: right dofs), item: index within the stream
        auto noise_u = [&](uint32_t stream, int slot0, int item) -> float {
            if (noise_lds) return grx_u01(__float_as_uint(TW(TBO(slot0 + (item >> 2) + 1) + T_PA + (item & 3))));
            return grx_rand(P.seed, genv, step, stream, (uint32_t)item);
        };
        // idx: column of the observation; (stream, slot0, item): where its noise uniform comes from
        auto put = [&](int idx, float val, float nscale, uint32_t stream, int slot0, int item) {
            pri[idx] = fminf(fmaxf(val, -clipo), clipo);   // pri_obs copies obs BEFORE noise
            float ov = val;
            if (P.add_noise && nscale != 0.f) {
                const float u = noise_in ? noise_in[(size_t)e * nobs + idx] : noise_u(stream, slot0, item);
                ov += (2.f * u - 1.f) * nscale;
            }
            obs[idx] = fminf(fmaxf(ov, -clipo), clipo);
        };
        const float np_ = P.noise_dof_pos * P.noise_level * P.obs_scale_dof_pos, nv = P.noise_dof_vel * P.noise_level * P.obs_scale_dof_vel;
        const float nac = P.noise_action * P.noise_level * P.obs_scale_action;
        for (int j = c; j < nd; j += TG) {   // the joints go round the group's lanes (as in the reward terms): observations, history, state
            const size_t oj = (size_t)j * N + e;
            const float qj = TW(TBO(j + 1) + T_QPUB), qdj = TW(TBO(j + 1) + T_QDPUB), ac = TW(o.dof + TD_ACUR * GRX_MAX_DOFS + j);
            // dof terms: one stream per half of the dof range, item = group * (nd / 2) + joint within the half (group 0 pos, 1 vel, 2 action)
            const bool right = j >= half_;
            const uint32_t ds_ = right ? (uint32_t)GRX_RNG_NOISE_DOF_R : (uint32_t)GRX_RNG_NOISE_DOF_L;
            const int s0_ = right ? 2 + nblk_dof : 2, jj = right ? j - half_ : j;
            put(9 + j, (qj - T.dof[j].q0) * P.obs_scale_dof_pos, np_, ds_, s0_, jj);
            put(9 + nd + j, qdj * P.obs_scale_dof_vel, nv, ds_, s0_, half_ + jj);
            put(9 + 2 * nd + j, ac * P.obs_scale_action, nac, ds_, s0_, 2 * half_ + jj);
            P.q[oj] = qj; P.qd[oj] = qdj; P.actions[oj] = ac; P.torques[oj] = TW(TBO(j + 1) + T_TAU);
            P.last_actions[oj] = ac; P.last_dof_vel[oj] = qdj;   // history (legged_robot.py:299-300, after reset_idx)
        }
        if (c < 8) {
#pragma unroll
            for (int k = 0; k < 3; ++k) P.anchors[(size_t)(c * 3 + k) * N + e] = TW(o.an + c * 3 + k);
        }
        if (lead) {
            put(0, ea.cmd[0], 0.f, 0u, 0, 0); put(1, ea.cmd[1], 0.f, 0u, 0, 0); put(2, ea.cmd[2], 0.f, 0u, 0, 0);
            const float na = P.noise_ang_vel * P.noise_level * P.obs_scale_ang_vel, ng = P.noise_gravity * P.noise_level * P.obs_scale_gravity;
            const float bo[6] = {bav.x * P.obs_scale_ang_vel, bav.y * P.obs_scale_ang_vel, bav.z * P.obs_scale_ang_vel,
                                 pg.x * P.obs_scale_gravity, pg.y * P.obs_scale_gravity, pg.z * P.obs_scale_gravity};
#pragma unroll
            for (int i = 0; i < 6; ++i) put(3 + i, bo[i], i < 3 ? na : ng, (uint32_t)GRX_RNG_NOISE, 0, i);   // base stream: item = column - 3
            pri[nobs + 0] = fminf(fmaxf(blv.x * P.obs_scale_lin_vel, -clipo), clipo);
            pri[nobs + 1] = fminf(fmaxf(blv.y * P.obs_scale_lin_vel, -clipo), clipo);
            pri[nobs + 2] = fminf(fmaxf(blv.z * P.obs_scale_lin_vel, -clipo), clipo);
            pri[nobs + 3] = fminf(fmaxf(bho * P.obs_scale_height, -clipo), clipo);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                pri[nobs + 4 + f] = (reset ? false : contact[f]) ? 1.f : 0.f;   // feet_contact[env_ids] = 0 (legged_robot_fftai.py:141)
                pri[nobs + 6 + f] = fminf(fmaxf(feet_height[f] * P.obs_scale_height, -clipo), clipo);
            }
            // ---- store the env's state
            const float rs[13] = {B.pos.x, B.pos.y, B.pos.z, B.qx, B.qy, B.qz, B.qw, B.vel.x, B.vel.y, B.vel.z, B.ang.x, B.ang.y, B.ang.z};
#pragma unroll
            for (int i = 0; i < 13; ++i) P.root[(size_t)i * N + e] = rs[i];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                P.air_time[(size_t)f * N + e] = air_time[f] * (contact_filt[f] ? 0.f : 1.f);   // legged_robot_fftai.py:97
                P.land_time[(size_t)f * N + e] = land_time[f];
                P.feet_contact[(size_t)f * N + e] = (reset ? false : contact[f]) ? 1 : 0;
                P.feet_height[(size_t)f * N + e] = feet_height[f];
                P.avg_force[(size_t)f * N + e] = avg_force[f];
                const float ff[3] = {foot_force[f].x, foot_force[f].y, foot_force[f].z}, fp[3] = {fpos[f].x, fpos[f].y, fpos[f].z};
                const float as_[3] = {avg_speed[f].x, avg_speed[f].y, avg_speed[f].z}, ar_[3] = {avg_rpy[f].x, avg_rpy[f].y, avg_rpy[f].z};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    P.avg_speed_rpy[(size_t)(f * 3 + i) * N + e] = ar_[i];
                    P.feet_force[(size_t)(f * 3 + i) * N + e] = ff[i];
                    P.feet_pos[(size_t)(f * 3 + i) * N + e] = fp[i];
                    P.avg_speed[(size_t)(f * 3 + i) * N + e] = as_[i];
                }
            }
            P.commands[e] = ea.cmd[0]; P.commands[N + e] = ea.cmd[1]; P.commands[2 * N + e] = ea.cmd[2];
            P.base_lin_vel[e] = blv.x; P.base_lin_vel[N + e] = blv.y; P.base_lin_vel[2 * N + e] = blv.z;
            P.base_ang_vel[e] = bav.x; P.base_ang_vel[N + e] = bav.y; P.base_ang_vel[2 * N + e] = bav.z;
            P.proj_grav[e] = pg.x; P.proj_grav[N + e] = pg.y; P.proj_grav[2 * N + e] = pg.z;
            P.origins[e] = ea.origin[0]; P.origins[N + e] = ea.origin[1]; P.origins[2 * N + e] = ea.origin[2];
            P.levels[e] = ea.level;
            P.base_heights_offset[e] = bho;
            P.ep_len[e] = ep_len;
            P.rew[e] = rew;
            P.reset[e] = reported_reset ? 1 : 0;
            P.time_out[e] = time_out ? 1 : 0;
            P.term_contact[e] = term_contact ? 1 : 0;
        }
    }
#ifdef GRX_PROFILE_SECTIONS
    if (threadIdx.x == 0 && blockIdx.x < 64) {   // sections summed over the sub-steps: outward, contacts, base, self-collision, inward, base solve, accel, rest; [8] physics, [9] whole kernel
        long long* pr = P.prof + (size_t)blockIdx.x * GRX_PROF_SLOTS;
        for (int i = 0; i < 8; ++i) { pr[i] = tt_acc[i]; pr[20 + i] = tt_acc[8 + i]; }
        pr[9] = clock64() - tt_begin;
    }
#endif
    __syncthreads();
    if (threadIdx.x < NSTAT) stat_row(P, sq.seq, threadIdx.x)[grp] = (s_stat[0][threadIdx.x] + s_stat[1][threadIdx.x]) + (s_stat[2][threadIdx.x] + s_stat[3][threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 0) P.stat_nblocks[sq.seq & 1] = (int)gridDim.x;
}
#undef TW
#undef TBO
