// grx_generic.h -- the generic-tree step kernel (included by grx_kernels.hip inside its anonymous namespace).
//
// The fast kernel (grx_step_kernel) is specialised for the GR1 lower-limb tree: two 5-joint chains, one lane pair
// per env.  Every other robot model that include/grx.h can describe -- in particular the 32-DOF full-body GR1T1
// (config 5 of BASELINE.json: legs with ankle roll, 3-joint waist, head, two 7-joint arms hanging from the torso)
// -- runs through this kernel instead:
//
//   * one env per lane, 64 envs per single-wave block; loops over bodies / dofs with run-time trip counts;
//   * the same formulation as the fast kernel: spatial quantities in WORLD axes about the base origin O, so a
//     child's articulated inertia and bias force simply ADD into the parent at every junction of the tree (no 6x6
//     frame transforms), revolute joints about arbitrary axes (Rodrigues), Schur-complement base solve;
//   * per-body intermediates (frames, joint axes, articulated inertias, ...) live in a global-memory workspace laid
//     out [slot][body][env], so every access of a wave is one coalesced row -- 72 floats per body per env; the
//     workspace of 4096 full-body envs is 39 MB and stays cache-resident;
//   * the env pipeline (torques, timers, termination, all reward terms, reset, observations + noise) is the fast
//     kernel's, written as plain loops over the dofs; same Philox streams (the dof-noise streams split the dof range
//     in halves exactly as the oracle does).
//
// This is the correctness-first path for models the fast kernel does not cover; it is latency-bound on the
// workspace round trips (DESIGN.md section 4.3 has the measured rate).
#pragma once

constexpr int GEN_MAXB = GRX_MAX_BODIES, GEN_MAXD = GRX_MAX_DOFS, GEN_MAXS = GRX_MAX_SPHERES, GEN_MAXLC = 24, GEN_MAXLP = 48;
constexpr int WSB = 72;   // workspace floats per body
enum { W_R = 0, W_RHO = 9, W_W = 12, W_V = 15, W_A = 18, W_S = 21, W_CA = 24, W_CL = 27, W_IA = 30, W_IB = 36, W_ID = 45,
       W_PA = 51, W_PL = 54, W_UA = 57, W_UL = 60, W_DI = 63, W_U = 64, W_AA = 65, W_AL = 68 };

struct GenTables {
    int32_t nb, nd, nsph, nlc;
    int32_t parent[GEN_MAXB];
    float axis[GEN_MAXB][3], rot0[GEN_MAXB][9], jpos[GEN_MAXB][3], mass[GEN_MAXB], com[GEN_MAXB][3], Ic[GEN_MAXB][6];
    float kp[GEN_MAXD], kd[GEN_MAXD], q0[GEN_MAXD], effort[GEN_MAXD], vlim[GEN_MAXD], qlo[GEN_MAXD], qhi[GEN_MAXD];
    float slo[GEN_MAXD], shi[GEN_MAXD], amin[GEN_MAXD], amax[GEN_MAXD], Klim[GEN_MAXD], Clim[GEN_MAXD];
    float arm[GEN_MAXD];               // joint-space armature (grx_model.dof_armature)
    int32_t sph_begin[GEN_MAXB + 1];   // spheres are sorted by carrying body
    float sx[GEN_MAXS], sy[GEN_MAXS], sz[GEN_MAXS], sr[GEN_MAXS], sdmax[GEN_MAXS];
    int32_t sslot[GEN_MAXS];           // friction-anchor slot 0..7 of an anchored foot sphere, -1 otherwise
    int32_t slink[GEN_MAXS];           // compact id of the URDF link the shape belongs to (force netting)
    uint32_t link_flags[GEN_MAXLC];    // GRX_SPH_TERMINATE / GRX_SPH_PENALISE of the compact links
    int32_t link_urdf[GEN_MAXLC];      // URDF link index of the compact links (row of GRX_T_CONTACT_FORCES)
    int32_t foot_body[2], foot_link[2];
    float foot_pos[2][3];
    int32_t torso_body, forehead_body;
    float torso_rot[9], forehead_rot[9];
    // self-collision (grx_model.pair_a / pair_b grouped by link pair): compact links a, b on bodies ba, bb, bounding
    // spheres of their shapes (body frame: xyz, radius); a compact link's shapes are sx[lc_begin[l] .. lc_begin[l + 1])
    int32_t nlp;
    int32_t lp_a[GEN_MAXLP], lp_b[GEN_MAXLP], lp_ba[GEN_MAXLP], lp_bb[GEN_MAXLP];
    float lp_ca[GEN_MAXLP][4], lp_cb[GEN_MAXLP][4];
    int32_t lc_begin[GEN_MAXLC + 1];
};
typedef const GRX_AS4 GenTables& GT;

#define WSX(b, slot) ws[((size_t)((b) * WSB + (slot))) * WN]   // WN: workspace stride (envs per row)
GRX_DEV V3 ws_v3(const float* ws, size_t WN, int b, int slot) { return v3(WSX(b, slot), WSX(b, slot + 1), WSX(b, slot + 2)); }
GRX_DEV void ws_put(float* ws, size_t WN, int b, int slot, V3 x) { WSX(b, slot) = x.x; WSX(b, slot + 1) = x.y; WSX(b, slot + 2) = x.z; }
GRX_DEV R3 ws_R(const float* ws, size_t WN, int b) {
    R3 R;
    R.cx = ws_v3(ws, WN, b, W_R); R.cy = ws_v3(ws, WN, b, W_R + 3); R.cz = ws_v3(ws, WN, b, W_R + 6);
    return R;
}
GRX_DEV void ws_putR(float* ws, size_t WN, int b, const R3& R) { ws_put(ws, WN, b, W_R, R.cx); ws_put(ws, WN, b, W_R + 3, R.cy); ws_put(ws, WN, b, W_R + 6, R.cz); }

// child rotation: R_parent * rot0 * Rot(axis, q) for a unit axis in the child frame (columns = body axes in the world)
GRX_DEV R3 gen_joint_rot(const R3& Rp, GT T, int b, float q) {
    float sn, cs;
    grx_sincos(q, sn, cs);
    const float ax = T.axis[b][0], ay = T.axis[b][1], az = T.axis[b][2], oc = 1.f - cs;
    // Rodrigues, columns of Rq
    const V3 qx = v3(cs + ax * ax * oc, az * sn + ax * ay * oc, -ay * sn + ax * az * oc);
    const V3 qy = v3(-az * sn + ax * ay * oc, cs + ay * ay * oc, ax * sn + ay * az * oc);
    const V3 qz = v3(ay * sn + ax * az * oc, -ax * sn + ay * az * oc, cs + az * az * oc);
    // rot0 is row-major child(q=0) -> parent: J = Rp * rot0
    R3 J;
    J.cx = rot(Rp, v3(T.rot0[b][0], T.rot0[b][3], T.rot0[b][6]));
    J.cy = rot(Rp, v3(T.rot0[b][1], T.rot0[b][4], T.rot0[b][7]));
    J.cz = rot(Rp, v3(T.rot0[b][2], T.rot0[b][5], T.rot0[b][8]));
    R3 R;
    R.cx = rot(J, qx); R.cy = rot(J, qy); R.cz = rot(J, qz);
    return R;
}

// one sphere of body `b` against the terrain with a run-time anchor slot; adds its force into the link accumulator
template <int HF>
GRX_DEV V3 gen_sphere(KP P, GT T, int i, const R3& R, V3 rho, V3 w, V3 v, V3 O, float mu, float om_e, float hmax, float* ws, size_t WN, size_t N, int e,
                      int lfbase, V3& xr) {
    xr = rho + rot(R, v3(T.sx[i], T.sy[i], T.sz[i]));
    V3 F = v3(0.f, 0.f, 0.f);
    const float wz = O.z + xr.z, r = T.sr[i];
    const int slot = T.sslot[i];
    bool touching = false;
    float vimp = 0.f;
    if (wz - r <= hmax) {
        const float wx = O.x + xr.x, wy = O.y + xr.y;
        float gx, gy;
        const float dv = terrain_height<HF>(P, wx, wy, gx, gy) + r - wz;
        if (dv > 0.0f) {
            touching = true;
            const float nn = grx_rsq(1.0f + gx * gx + gy * gy);   // surface normal from the gradient (fast kernel: sphere_contact)
            const V3 n = v3(-gx * nn, -gy * nn, nn);
            const float d = dv * nn;
            const V3 u = v + cross(w, xr);
            const float un = dot(u, n);
            float cd = fminf(P.kn * d * P.dn, T.sdmax[i]);
            if (slot >= 0) {   // restitution (same rule as the fast kernel's sphere_contact)
                vimp = P.anchors[(size_t)(slot * 3 + 2) * N + e];
                if (vimp == 0.f) vimp = fmaxf(fmaxf(-un, 0.0f), 1e-6f);
                if (un > 0.0f && vimp > P.bounce_threshold) cd *= om_e;
            }
            const float fn = fmaxf(P.kn * d - cd * un, 0.0f);
            F = n * fn;
            const float fmax = mu * fn;
            if (slot >= 0) {
                float* an = P.anchors + (size_t)(slot * 3) * N + e;
                float axx = an[0], ayy = an[N];
                if (an[2 * N] == 0.f) { axx = wx; ayy = wy; }
                float ftx = -P.kt * (wx - axx) - P.ct * u.x;
                float fty = -P.kt * (wy - ayy) - P.ct * u.y;
                const float ft = grx_sqrt(ftx * ftx + fty * fty);
                if (ft > fmax) {
                    const float sc = fmax * grx_rcp(ft);
                    ftx *= sc; fty *= sc;
                    axx = wx + ftx * P.inv_kt;
                    ayy = wy + fty * P.inv_kt;
                }
                an[0] = axx; an[N] = ayy;
                F.x += ftx; F.y += fty;
            } else {
                const float sp = grx_sqrt(u.x * u.x + u.y * u.y);
                const float ft = fminf(P.cv * sp, fmax);
                if (sp > 1e-9f) { const float k = -ft * grx_rcp(sp); F.x += k * u.x; F.y += k * u.y; }
            }
        }
        if (HF == GRX_HF_TRIMESH) {   // mesh_type 'trimesh': the vertical faces next to the shape (grx_kernels.hip wall_contact)
            float wtx, wty;
            const uint4 ww = wall_gather(P, wx, wy, wtx, wty);
            F = F + wall_contact(P, ww, wtx, wty, wz, r, T.sdmax[i], v + cross(w, xr), mu);
        }
    }
    if (slot >= 0) P.anchors[(size_t)(slot * 3 + 2) * N + e] = touching ? vimp : 0.f;
    const int L = T.slink[i];
    ws[(size_t)(lfbase + L * 3 + 0) * WN] += F.x; ws[(size_t)(lfbase + L * 3 + 1) * WN] += F.y; ws[(size_t)(lfbase + L * 3 + 2) * WN] += F.z;
    return F;
}

struct GenBase { V3 pos, vel, ang; float qx, qy, qz, qw; };

// foot link frames of the CURRENT state (positions walk only, up to the two foot bodies)
GRX_DEV void gen_foot_frames(KP P, GT T, const GenBase& B, const float* q, const float* qd, float* ws, size_t WN, size_t N, int e, V3 fpos[2], V3 fvel[2]) {
    ws_putR(ws, WN, 0, quat_to_R(B.qx, B.qy, B.qz, B.qw));
    ws_put(ws, WN, 0, W_RHO, v3(0.f, 0.f, 0.f)); ws_put(ws, WN, 0, W_W, B.ang); ws_put(ws, WN, 0, W_V, B.vel);
    for (int b = 1; b < T.nb; ++b) {
        const int p = T.parent[b];
        const R3 Rp = ws_R(ws, WN, p);
        const V3 rho = ws_v3(ws, WN, p, W_RHO) + rot(Rp, v3(T.jpos[b][0], T.jpos[b][1], T.jpos[b][2]));
        const R3 R = gen_joint_rot(Rp, T, b, q[(size_t)(b - 1) * N]);
        const V3 a = rot(R, v3(T.axis[b][0], T.axis[b][1], T.axis[b][2]));
        const float qdk = qd[(size_t)(b - 1) * N];
        const V3 wp = ws_v3(ws, WN, p, W_W), vp = ws_v3(ws, WN, p, W_V);
        ws_putR(ws, WN, b, R); ws_put(ws, WN, b, W_RHO, rho);
        ws_put(ws, WN, b, W_W, fma3(a, qdk, wp)); ws_put(ws, WN, b, W_V, fma3(cross(rho, a), qdk, vp));
    }
    for (int f = 0; f < 2; ++f) {
        const int b = T.foot_body[f];
        const R3 R = ws_R(ws, WN, b);
        const V3 fr = ws_v3(ws, WN, b, W_RHO) + rot(R, v3(T.foot_pos[f][0], T.foot_pos[f][1], T.foot_pos[f][2]));
        fpos[f] = B.pos + fr;
        fvel[f] = ws_v3(ws, WN, b, W_V) + cross(ws_v3(ws, WN, b, W_W), fr);
    }
}

// One physics sub-step.  q / qd / torques: this env's columns of the SoA state arrays (stride N).
template <int HF>
GRX_DEV void gen_substep(KP P, GT T, GenBase& B, float* q, float* qd, const float* tau, float* ws, size_t WN, size_t N, int e,
                         float base_m, V3 base_c, const S3& base_I, float mu, float om_e, float hmax, V3 foot_vel_before[2]) {
    const int nb = T.nb, lfbase = nb * WSB;
    const float dt = P.sim_dt;
    const R3 R0 = quat_to_R(B.qx, B.qy, B.qz, B.qw);
    const V3 O = B.pos;
    for (int i = 0; i < T.nlc * 3; ++i) ws[(size_t)(lfbase + i) * WN] = 0.f;
    // ---- pass 1 (root -> leaves): frames, joint axes, velocity-product accelerations, rigid inertias, bias, contacts
    ws_putR(ws, WN, 0, R0);
    ws_put(ws, WN, 0, W_RHO, v3(0.f, 0.f, 0.f)); ws_put(ws, WN, 0, W_W, B.ang); ws_put(ws, WN, 0, W_V, B.vel);
    R3 Rprev = R0;                       // frame of body b-1: most bodies hang from their predecessor (chains), which
    V3 rho_prev = v3(0.f, 0.f, 0.f), w_prev = B.ang, v_prev = B.vel;   // then never has to be read back from the workspace
    for (int b = 1; b < nb; ++b) {
        const int p = T.parent[b];
        const bool chain = p == b - 1;
        const R3 Rp = chain ? Rprev : ws_R(ws, WN, p);
        const V3 wp = chain ? w_prev : ws_v3(ws, WN, p, W_W), vp = chain ? v_prev : ws_v3(ws, WN, p, W_V);
        const V3 rho = (chain ? rho_prev : ws_v3(ws, WN, p, W_RHO)) + rot(Rp, v3(T.jpos[b][0], T.jpos[b][1], T.jpos[b][2]));
        const R3 R = gen_joint_rot(Rp, T, b, q[(size_t)(b - 1) * N]);
        const V3 a = rot(R, v3(T.axis[b][0], T.axis[b][1], T.axis[b][2]));
        const V3 s = cross(rho, a);
        const float qdk = qd[(size_t)(b - 1) * N];
        const V3 ca = cross(wp, a) * qdk;
        const V3 cl = (cross(vp, a) + cross(wp, s)) * qdk;
        const V3 w = fma3(a, qdk, wp), v = fma3(s, qdk, vp);
        const V3 kap = rho + rot(R, v3(T.com[b][0], T.com[b][1], T.com[b][2]));
        const S3 Ic = {T.Ic[b][0], T.Ic[b][1], T.Ic[b][2], T.Ic[b][3], T.Ic[b][4], T.Ic[b][5]};
        const float m = T.mass[b];
        S3 Ak; V3 h;
        rigid_inertia(R, kap, m, Ic, Ak, h);
        V3 pa, pl;
        rigid_bias(R, kap, m, Ic, w, v, pa, pl);
        for (int i = T.sph_begin[b]; i < T.sph_begin[b + 1]; ++i) {
            V3 xr;
            const V3 F = gen_sphere<HF>(P, T, i, R, rho, w, v, O, mu, om_e, hmax, ws, WN, N, e, lfbase, xr);
            pa = pa - cross(xr, F); pl = pl - F;
        }
        for (int f = 0; f < 2; ++f)
            if (T.foot_body[f] == b) {   // foot link velocity BEFORE this sub-step's integration
                const V3 fr = rho + rot(R, v3(T.foot_pos[f][0], T.foot_pos[f][1], T.foot_pos[f][2]));
                foot_vel_before[f] = v + cross(w, fr);
            }
        ws_putR(ws, WN, b, R); ws_put(ws, WN, b, W_RHO, rho); ws_put(ws, WN, b, W_W, w); ws_put(ws, WN, b, W_V, v);
        Rprev = R; rho_prev = rho; w_prev = w; v_prev = v;
        ws_put(ws, WN, b, W_A, a); ws_put(ws, WN, b, W_S, s); ws_put(ws, WN, b, W_CA, ca); ws_put(ws, WN, b, W_CL, cl);
        WSX(b, W_IA + 0) = Ak.xx; WSX(b, W_IA + 1) = Ak.xy; WSX(b, W_IA + 2) = Ak.xz; WSX(b, W_IA + 3) = Ak.yy; WSX(b, W_IA + 4) = Ak.yz; WSX(b, W_IA + 5) = Ak.zz;
        WSX(b, W_IB + 0) = 0.f; WSX(b, W_IB + 1) = -h.z; WSX(b, W_IB + 2) = h.y; WSX(b, W_IB + 3) = h.z; WSX(b, W_IB + 4) = 0.f;
        WSX(b, W_IB + 5) = -h.x; WSX(b, W_IB + 6) = -h.y; WSX(b, W_IB + 7) = h.x; WSX(b, W_IB + 8) = 0.f;
        WSX(b, W_ID + 0) = m; WSX(b, W_ID + 1) = 0.f; WSX(b, W_ID + 2) = 0.f; WSX(b, W_ID + 3) = m; WSX(b, W_ID + 4) = 0.f; WSX(b, W_ID + 5) = m;
        ws_put(ws, WN, b, W_PA, pa); ws_put(ws, WN, b, W_PL, pl);
    }
    // base: rigid lump (randomised per env) + its contacts
    S3 A0; V3 h0;
    rigid_inertia(R0, rot(R0, base_c), base_m, base_I, A0, h0);
    V3 pa0, pl0;
    rigid_bias(R0, rot(R0, base_c), base_m, base_I, B.ang, B.vel, pa0, pl0);
    for (int i = T.sph_begin[0]; i < T.sph_begin[1]; ++i) {
        V3 xr;
        const V3 F = gen_sphere<HF>(P, T, i, R0, v3(0.f, 0.f, 0.f), B.ang, B.vel, O, mu, om_e, hmax, ws, WN, N, e, lfbase, xr);
        pa0 = pa0 - cross(xr, F); pl0 = pl0 - F;
    }
    // self-collision (grx_self.h has the contact law): link pairs that can touch, bounding spheres first
    if (P.self_collisions) {
        const float mu_self = 2.0f * mu - P.terrain_friction;
        for (int lp = 0; lp < T.nlp; ++lp) {
            const int ba = T.lp_ba[lp], bb = T.lp_bb[lp];
            ChainKin Ka, Kb;
            if (ba == 0) Ka = ChainKin{R0, v3(0.f, 0.f, 0.f), B.ang, B.vel};
            else Ka = ChainKin{ws_R(ws, WN, ba), ws_v3(ws, WN, ba, W_RHO), ws_v3(ws, WN, ba, W_W), ws_v3(ws, WN, ba, W_V)};
            Kb = ChainKin{ws_R(ws, WN, bb), ws_v3(ws, WN, bb, W_RHO), ws_v3(ws, WN, bb, W_W), ws_v3(ws, WN, bb, W_V)};
            const V3 ca = Ka.rho + rot(Ka.R, v3(T.lp_ca[lp][0], T.lp_ca[lp][1], T.lp_ca[lp][2]));
            const V3 cb = Kb.rho + rot(Kb.R, v3(T.lp_cb[lp][0], T.lp_cb[lp][1], T.lp_cb[lp][2]));
            const V3 d = ca - cb;
            const float R = T.lp_ca[lp][3] + T.lp_cb[lp][3];
            if (!(dot(d, d) < R * R)) continue;
            const int la = T.lp_a[lp], lb = T.lp_b[lp];
            V3 Fa = v3(0.f, 0.f, 0.f), Ta = v3(0.f, 0.f, 0.f);
            for (int i = T.lc_begin[la]; i < T.lc_begin[la + 1]; ++i) {
                SphC si; si.x = T.sx[i]; si.y = T.sy[i]; si.z = T.sz[i]; si.r = T.sr[i]; si.dmax = T.sdmax[i];
                const SphW a = sph_world(si, Ka);
                for (int j = T.lc_begin[lb]; j < T.lc_begin[lb + 1]; ++j) {
                    SphC sj; sj.x = T.sx[j]; sj.y = T.sy[j]; sj.z = T.sz[j]; sj.r = T.sr[j]; sj.dmax = T.sdmax[j];
                    const SphW b = sph_world(sj, Kb);
                    V3 F, pw;
                    if (sphere_pair(P, a, b, mu_self, F, pw)) { Fa = Fa + F; Ta = Ta + cross(pw, F); }
                }
            }
            // F on link a (body ba), -F on link b (body bb)
            if (ba == 0) { pa0 = pa0 - Ta; pl0 = pl0 - Fa; }
            else { WSX(ba, W_PA) -= Ta.x; WSX(ba, W_PA + 1) -= Ta.y; WSX(ba, W_PA + 2) -= Ta.z; WSX(ba, W_PL) -= Fa.x; WSX(ba, W_PL + 1) -= Fa.y; WSX(ba, W_PL + 2) -= Fa.z; }
            WSX(bb, W_PA) += Ta.x; WSX(bb, W_PA + 1) += Ta.y; WSX(bb, W_PA + 2) += Ta.z; WSX(bb, W_PL) += Fa.x; WSX(bb, W_PL + 1) += Fa.y; WSX(bb, W_PL + 2) += Fa.z;
            ws[(size_t)(lfbase + la * 3 + 0) * WN] += Fa.x; ws[(size_t)(lfbase + la * 3 + 1) * WN] += Fa.y; ws[(size_t)(lfbase + la * 3 + 2) * WN] += Fa.z;
            ws[(size_t)(lfbase + lb * 3 + 0) * WN] -= Fa.x; ws[(size_t)(lfbase + lb * 3 + 1) * WN] -= Fa.y; ws[(size_t)(lfbase + lb * 3 + 2) * WN] -= Fa.z;
        }
    }
    S3 Ab = A0, Db = {base_m, 0.f, 0.f, base_m, 0.f, base_m};
    M3 Bb = {0.f, -h0.z, h0.y, h0.z, 0.f, -h0.x, -h0.y, h0.x, 0.f};
    // ---- pass 2 (leaves -> root): articulated inertias and bias forces; a child adds into its parent
    int carry_to = -1;                   // a chain child (parent == b-1, processed next) hands its contribution on in registers
    S3 cA = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cD = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    M3 cB = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    V3 cpa = v3(0.f, 0.f, 0.f), cpl = v3(0.f, 0.f, 0.f);
    for (int b = nb - 1; b >= 1; --b) {
        const int p = T.parent[b], j = b - 1;
        S3 A = {WSX(b, W_IA), WSX(b, W_IA + 1), WSX(b, W_IA + 2), WSX(b, W_IA + 3), WSX(b, W_IA + 4), WSX(b, W_IA + 5)};
        M3 Bm = {WSX(b, W_IB), WSX(b, W_IB + 1), WSX(b, W_IB + 2), WSX(b, W_IB + 3), WSX(b, W_IB + 4), WSX(b, W_IB + 5), WSX(b, W_IB + 6), WSX(b, W_IB + 7), WSX(b, W_IB + 8)};
        S3 D = {WSX(b, W_ID), WSX(b, W_ID + 1), WSX(b, W_ID + 2), WSX(b, W_ID + 3), WSX(b, W_ID + 4), WSX(b, W_ID + 5)};
        const V3 a = ws_v3(ws, WN, b, W_A), s = ws_v3(ws, WN, b, W_S), ca = ws_v3(ws, WN, b, W_CA), cl = ws_v3(ws, WN, b, W_CL);
        V3 pa = ws_v3(ws, WN, b, W_PA), pl = ws_v3(ws, WN, b, W_PL);
        if (carry_to == b) { A = A + cA; Bm = Bm + cB; D = D + cD; pa = pa + cpa; pl = pl + cpl; }
        carry_to = -1;
        const V3 ua = mul(A, a) + mul(Bm, s);
        const V3 ul = mulT(Bm, a) + mul(D, s);
        const float di = grx_rcp(dot(a, ua) + dot(s, ul) + T.arm[j]);
        const float qj = q[(size_t)j * N], qdj = qd[(size_t)j * N];
        float t = tau[(size_t)j * N];   // joint-limit spring/damper on top of the motor torque
        if (qj < T.qlo[j]) t += T.Klim[j] * (T.qlo[j] - qj) - T.Clim[j] * qdj;
        else if (qj > T.qhi[j]) t += T.Klim[j] * (T.qhi[j] - qj) - T.Clim[j] * qdj;
        const float u = t - (dot(a, pa) + dot(s, pl));
        syr(A, ua, di); ger(Bm, ua, ul, di); syr(D, ul, di);
        const float ud = u * di;
        const V3 npa = pa + mul(A, ca) + mul(Bm, cl) + ua * ud;
        const V3 npl = pl + mulT(Bm, ca) + mul(D, cl) + ul * ud;
        ws_put(ws, WN, b, W_UA, ua); ws_put(ws, WN, b, W_UL, ul); WSX(b, W_DI) = di; WSX(b, W_U) = u;
        if (p == 0) {
            Ab = Ab + A; Bb = Bb + Bm; Db = Db + D; pa0 = pa0 + npa; pl0 = pl0 + npl;
        } else if (p == b - 1) {
            carry_to = p; cA = A; cB = Bm; cD = D; cpa = npa; cpl = npl;
        } else {
            WSX(p, W_IA) += A.xx; WSX(p, W_IA + 1) += A.xy; WSX(p, W_IA + 2) += A.xz; WSX(p, W_IA + 3) += A.yy; WSX(p, W_IA + 4) += A.yz; WSX(p, W_IA + 5) += A.zz;
            WSX(p, W_IB) += Bm.a00; WSX(p, W_IB + 1) += Bm.a01; WSX(p, W_IB + 2) += Bm.a02; WSX(p, W_IB + 3) += Bm.a10; WSX(p, W_IB + 4) += Bm.a11;
            WSX(p, W_IB + 5) += Bm.a12; WSX(p, W_IB + 6) += Bm.a20; WSX(p, W_IB + 7) += Bm.a21; WSX(p, W_IB + 8) += Bm.a22;
            WSX(p, W_ID) += D.xx; WSX(p, W_ID + 1) += D.xy; WSX(p, W_ID + 2) += D.xz; WSX(p, W_ID + 3) += D.yy; WSX(p, W_ID + 4) += D.yz; WSX(p, W_ID + 5) += D.zz;
            WSX(p, W_PA) += npa.x; WSX(p, W_PA + 1) += npa.y; WSX(p, W_PA + 2) += npa.z;
            WSX(p, W_PL) += npl.x; WSX(p, W_PL + 1) += npl.y; WSX(p, W_PL + 2) += npl.z;
        }
    }
    // ---- base: [A B; B^T D][alpha; acc] = -[pa; pl]
    const S3 Di = inv(Db);
    const V3 b0 = v3(Bb.a00, Bb.a01, Bb.a02), b1 = v3(Bb.a10, Bb.a11, Bb.a12), b2 = v3(Bb.a20, Bb.a21, Bb.a22);
    const V3 d0 = mul(Di, b0), d1 = mul(Di, b1), d2 = mul(Di, b2);
    const S3 Sc = {Ab.xx - dot(b0, d0), Ab.xy - dot(b0, d1), Ab.xz - dot(b0, d2), Ab.yy - dot(b1, d1), Ab.yz - dot(b1, d2), Ab.zz - dot(b2, d2)};
    const V3 alpha = mul(inv(Sc), mul(Bb, mul(Di, pl0)) - pa0);
    const V3 acc = neg(mul(Di, pl0 + mulT(Bb, alpha)));
    // ---- pass 3 (root -> leaves): accelerations, joint integration
    ws_put(ws, WN, 0, W_AA, alpha); ws_put(ws, WN, 0, W_AL, acc);
    V3 aa_prev = alpha, al_prev = acc;
    for (int b = 1; b < nb; ++b) {
        const int p = T.parent[b], j = b - 1;
        const bool chain = p == b - 1;
        const V3 a = ws_v3(ws, WN, b, W_A), s = ws_v3(ws, WN, b, W_S);
        const V3 pa_ = (chain ? aa_prev : ws_v3(ws, WN, p, W_AA)) + ws_v3(ws, WN, b, W_CA), pl_ = (chain ? al_prev : ws_v3(ws, WN, p, W_AL)) + ws_v3(ws, WN, b, W_CL);
        const float qdd = (WSX(b, W_U) - (dot(ws_v3(ws, WN, b, W_UA), pa_) + dot(ws_v3(ws, WN, b, W_UL), pl_))) * WSX(b, W_DI);
        aa_prev = fma3(a, qdd, pa_); al_prev = fma3(s, qdd, pl_);
        ws_put(ws, WN, b, W_AA, aa_prev); ws_put(ws, WN, b, W_AL, al_prev);
        float vq = fmaf(qdd, dt, qd[(size_t)j * N]);
        vq = fminf(fmaxf(vq, -T.vlim[j]), T.vlim[j]);
        qd[(size_t)j * N] = vq;
        q[(size_t)j * N] = fmaf(vq, dt, q[(size_t)j * N]);
    }
    // ---- integrate the base (semi-implicit Euler)
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

// reset_idx for one env (legged_robot.py:377-440, 717-826; legged_robot_fftai.py:137-146); state written to memory
GRX_DEV void gen_reset_env(KP P, GT T, uint32_t genv, uint32_t step, bool init_done, GenBase& B, EnvAux& ea, float* q, float* qd, size_t N, int e) {
    if (P.curriculum && P.terrain_type != GRX_TERRAIN_PLANE && init_done) {
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
        const float* o = P.terrain_origins + ((size_t)ea.level * P.num_terrain_cols + ea.type) * 3;
        ea.origin[0] = o[0]; ea.origin[1] = o[1]; ea.origin[2] = o[2];
    }
    for (int j = 0; j < T.nd; ++j) {
        const float f = P.randomize_init_dof_pos ? urand(P, genv, step, GRX_RNG_RESET_DOF, (uint32_t)j, 0.5f, 1.5f) : 1.0f;
        q[(size_t)j * N] = f * T.q0[j];
        qd[(size_t)j * N] = 0.f;
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
    for (int i = 0; i < 8; ++i) P.anchors[(size_t)(i * 3 + 2) * N + e] = 0.f;
}

GRX_DEV float gen_masked_abs_sum(const float* a, size_t N, int nd, uint32_t mask) {
    float s = 0.f;
    for (int j = 0; j < nd; ++j) if (mask & (1u << j)) s += fabsf(a[(size_t)j * N]);
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
#define GRX_STEP_GENERIC_ARGS const KParams* __restrict__ Pg, const GenTables* __restrict__ Tg, float* __restrict__ wsg, const float* __restrict__ actions_in, float delay, long long common_step, \
                              const float* __restrict__ noise_in, float* __restrict__ obs_out, float* __restrict__ pri_out, long long seq
template <bool HF>
__global__ __launch_bounds__(64) void grx_step_generic(GRX_STEP_GENERIC_ARGS) {
#include "grx_step_generic_body.inc"
}
#ifndef GRX_TREE16_TU   // (not a template: it would be emitted in grx_tree16.hip's object too, which launches none of the generic kernels)
__global__ __launch_bounds__(64) void grx_step_generic_trimesh(GRX_STEP_GENERIC_ARGS) {   // mesh_type 'trimesh'
    constexpr int HF = GRX_HF_TRIMESH;
#include "grx_step_generic_body.inc"
}
#endif

// BaseTask.reset() first half for the generic path
// mask: as in grx_reset_all_kernel (nullptr = every env, else reset_idx of the flagged ones)
__global__ __launch_bounds__(64) void grx_reset_all_generic(const KParams* __restrict__ Pg, const GenTables* __restrict__ Tg, uint32_t step, long long seq, uint8_t* __restrict__ mask) {
    KP P = GRX_PARAMS(Pg);
    GT T = *reinterpret_cast<const GRX_AS4 GenTables*>(reinterpret_cast<uintptr_t>(Tg));
    const size_t N = (size_t)P.N;
    const int lane = threadIdx.x, epb = blockDim.x, e_raw = blockIdx.x * epb + lane;
    const bool act = e_raw < P.N;
    const int e = act ? e_raw : P.N - 1;
    const uint32_t genv = (uint32_t)(P.env_offset + e);
    const bool sel = act && (!mask || mask[e] != 0);   // this env resets
    for (int t = 0; t < NT; ++t) {   // extras["episode"]: every resetting env is "finished"
        float contrib = sel ? P.episode_sums[(size_t)t * N + e] : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) contrib += __shfl_xor(contrib, off);
        if (lane == 0) stat_row(P, seq, t)[blockIdx.x] = contrib;
    }
    GenBase B;
    B.pos = v3(P.root[e], P.root[N + e], P.root[2 * N + e]);
    EnvAux ea;
    ea.cmd[0] = P.commands[e]; ea.cmd[1] = P.commands[N + e]; ea.cmd[2] = P.commands[2 * N + e];
    ea.origin[0] = P.origins[e]; ea.origin[1] = P.origins[N + e]; ea.origin[2] = P.origins[2 * N + e];
    ea.level = P.levels[e]; ea.type = P.types[e];
    const int level_before = ea.level;
    if (sel && P.stash_pre_reset) {   // on-demand tensors (grx_refresh) show the state before the reset
        for (int j = 0; j < T.nd; ++j) { P.pre_q[(size_t)j * N + e] = P.q[(size_t)j * N + e]; P.pre_qd[(size_t)j * N + e] = P.qd[(size_t)j * N + e]; }
        for (int i = 0; i < 13; ++i) P.pre_root[(size_t)i * N + e] = P.root[(size_t)i * N + e];
    }
    if (sel) gen_reset_env(P, T, genv, step, mask != nullptr, B, ea, P.q + e, P.qd + e, N, e);
    {
        const unsigned long long wm = __ballot(sel);
        const float ls = level_sum(sel ? ea.level : level_before, act);
        if (lane == 0) {
            stat_row(P, seq, NT)[blockIdx.x] = (float)__popcll(wm);
            stat_row(P, seq, NT + 1)[blockIdx.x] = ls;
            if (blockIdx.x == 0) P.stat_nblocks[seq & 1] = (int)gridDim.x;
        }
    }
    if (!sel) return;
    if (mask) {
        mask[e] = 0;
        P.origins[e] = ea.origin[0]; P.origins[N + e] = ea.origin[1]; P.origins[2 * N + e] = ea.origin[2];
        P.levels[e] = ea.level;
    }
    for (int j = 0; j < T.nd; ++j) { P.last_actions[(size_t)j * N + e] = 0.f; P.last_dof_vel[(size_t)j * N + e] = 0.f; }
    for (int f = 0; f < 2; ++f) { P.air_time[(size_t)f * N + e] = 0.f; P.land_time[(size_t)f * N + e] = 0.f; P.feet_contact[(size_t)f * N + e] = 0; }
    const float rs[13] = {B.pos.x, B.pos.y, B.pos.z, B.qx, B.qy, B.qz, B.qw, B.vel.x, B.vel.y, B.vel.z, B.ang.x, B.ang.y, B.ang.z};
    for (int i = 0; i < 13; ++i) P.root[(size_t)i * N + e] = rs[i];
    P.commands[e] = ea.cmd[0]; P.commands[N + e] = ea.cmd[1]; P.commands[2 * N + e] = ea.cmd[2];
    P.ep_len[e] = 0;
    P.reset[e] = 1;
    for (int t = 0; t < NT; ++t) P.episode_sums[(size_t)t * N + e] = 0.f;
}
