/*
 * grx_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement ("oracle") of the GRx environment step.
 *
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  It exists to check the HIP kernels (wiki-grx-gym_amd/csrc) and is
 * deliberately written the slow, readable way: generic kinematic tree, dense 6x6 spatial
 * algebra (Featherstone, "Rigid Body Dynamics Algorithms", 2008, ch. 7 ABA / ch. 9 floating
 * base), one env at a time, AoS state.  The HIP kernels are an independent, topology-specialised
 * implementation of the same equations.
 *
 * What it restates, with the reference file:line each part follows
 * (paths relative to /root/reference/legged_gym/legged_gym/envs/):
 *   step()                      base/legged_robot.py:222-246
 *   clip_actions                fftai/legged_robot_fftai.py:171-177
 *   during_physics_step         fftai/legged_robot_fftai.py:51-88   (action latency, sub-step averages)
 *   _compute_torques            base/legged_robot.py:679-715
 *   post_physics_step           base/legged_robot.py:269-305, fftai/legged_robot_fftai.py:90-99
 *   post_physics_step_update_state  base/legged_robot.py:307-334, fftai/...:101-133
 *   check_termination           base/legged_robot.py:336-353
 *   compute_reward + 36 terms   base/legged_robot.py:355-375, fftai/...:180-352, gr1t1/gr1t1.py:338-589
 *   reset_idx & friends         base/legged_robot.py:377-440, 650-677, 717-826, fftai/...:137-146
 *   compute_observations        base/legged_robot.py:442-481, fftai/...:148-167, gr1t1/gr1t1.py:281-336
 *   _get_heights                base/legged_robot.py:1219-1274, utils/math.py:38-42
 *   quaternion helpers          isaacgym/torch_utils.py:43-81,176-190
 *
 * PARITY STATUS
 *   - env pipeline (everything above except gym.simulate): PINNED against golden vectors produced
 *     by importing the reference Python in the build container (tools/gen_golden.py ->
 *     tests/golden/ (npz fixtures), tests/test_oracle_golden.py).
 *   - physics (gym.simulate, legged_robot_fftai.py:68): **parity unpinned**.  The reference's
 *     physics is the closed NVIDIA Isaac Gym 1.0.preview4 / PhysX 5 binary (absent from the
 *     checkout, .MISSING_LARGE_BLOBS:7-21); there is no source, no golden rollout and no test
 *     in the reference that pins its output.  This file implements a textbook articulated-body
 *     forward dynamics + compliant contact model pinned only by physics
 *     invariants (tests/test_oracle_physics.py): ABA vs CRBA, energy, free fall, static stance;
 *     round 2: self-collision (legs kept apart, internal force pair conserves momentum; the reference
 *     enables it: legged_robot_config.py:121, legged_robot.py:1022-1028), restitution with the bounce
 *     threshold (legged_robot_config.py:46-49, legged_robot.py:565-575), contact along the terrain
 *     normal, 'trimesh' risers (legged_robot.py:903-921) -- each this build's own model of what PhysX
 *     does there, equally unpinned.
 *
 * Build: see oracle/Makefile (REAL=float -> libgrx_oracle_f32.so, REAL=double -> ..._f64.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/grx.h"
#include "philox.h"

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define NB_MAX GRX_MAX_BODIES
#define ND_MAX GRX_MAX_DOFS
#define NS_MAX GRX_MAX_SPHERES
#define NL_MAX GRX_MAX_LINKS /* URDF links */
#define NT GRX_NUM_REWARD_TERMS
#define NFS 8     /* anchored foot spheres (4 per foot) */

static __thread char g_err[512];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

/* ------------------------------------------------------------------ small linear algebra */
static void v3_cross(const real a[3], const real b[3], real o[3]) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static real v3_dot(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3_mulv(const real M[9], const real v[3], real o[3]) {
    real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    real y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    real z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_tmulv(const real M[9], const real v[3], real o[3]) { /* M^T v */
    real x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    real y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    real z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_mul(const real A[9], const real B[9], real O[9]) {
    real T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(O, T, sizeof T);
}
static void m3_transpose(const real A[9], real O[9]) {
    real T[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
    memcpy(O, T, sizeof T);
}
static void m3_skew(const real v[3], real O[9]) {
    O[0] = 0; O[1] = -v[2]; O[2] = v[1];
    O[3] = v[2]; O[4] = 0; O[5] = -v[0];
    O[6] = -v[1]; O[7] = v[0]; O[8] = 0;
}
/* Rodrigues: rotation by angle about unit axis (maps child coords -> parent coords) */
static void m3_axis_angle(const real a[3], real ang, real R[9]) {
    real c = cos(ang), s = sin(ang), t = 1 - c;
    R[0] = c + a[0] * a[0] * t;        R[1] = a[0] * a[1] * t - a[2] * s; R[2] = a[0] * a[2] * t + a[1] * s;
    R[3] = a[1] * a[0] * t + a[2] * s; R[4] = c + a[1] * a[1] * t;        R[5] = a[1] * a[2] * t - a[0] * s;
    R[6] = a[2] * a[0] * t - a[1] * s; R[7] = a[2] * a[1] * t + a[0] * s; R[8] = c + a[2] * a[2] * t;
}
/* quaternion (x,y,z,w) -> rotation matrix body->world */
static void quat_to_m3(const real q[4], real R[9]) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
/* isaacgym/torch_utils.py:71-81  quat_rotate_inverse: a - b + c */
static void quat_rotate_inverse(const real q[4], const real v[3], real o[3]) {
    real w = q[3];
    real s = (real)2.0 * w * w - (real)1.0;
    real cr[3];
    v3_cross(q, v, cr);
    real d = v3_dot(q, v);
    for (int i = 0; i < 3; ++i) o[i] = v[i] * s - cr[i] * w * (real)2.0 + q[i] * d * (real)2.0;
}
/* isaacgym/torch_utils.py:48-55 quat_apply */
static void quat_apply(const real q[4], const real b[3], real o[3]) {
    real t[3], u[3];
    v3_cross(q, b, t);
    for (int i = 0; i < 3; ++i) t[i] *= 2;
    v3_cross(q, t, u);
    for (int i = 0; i < 3; ++i) o[i] = b[i] + q[3] * t[i] + u[i];
}
/* utils/math.py:38-42 quat_apply_yaw: zero x,y, normalise (eps 1e-9, torch_utils.py:43-45), apply */
static void quat_apply_yaw(const real q[4], const real b[3], real o[3]) {
    real qy[4] = {0, 0, q[2], q[3]};
    real n = sqrt(qy[2] * qy[2] + qy[3] * qy[3]);
    if (n < (real)1e-9) n = (real)1e-9;
    qy[2] /= n; qy[3] /= n;
    quat_apply(qy, b, o);
}

/* ------------------------------------------------------------------ 6x6 spatial algebra */
typedef struct { real m[6][6]; } sm6;
typedef struct { real v[6]; } sv6;

/* motion transform parent->child: X = [E 0; -E rx E], E = parent->child rotation, r = child origin in parent */
static void sx_motion(const real E[9], const real r[3], sm6* X) {
    real rx[9], Erx[9];
    m3_skew(r, rx);
    m3_mul(E, rx, Erx);
    memset(X, 0, sizeof *X);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            X->m[i][j] = E[3 * i + j];
            X->m[3 + i][3 + j] = E[3 * i + j];
            X->m[3 + i][j] = -Erx[3 * i + j];
        }
}
static void sm_mulv(const sm6* A, const sv6* x, sv6* o) {
    sv6 t;
    for (int i = 0; i < 6; ++i) {
        real s = 0;
        for (int j = 0; j < 6; ++j) s += A->m[i][j] * x->v[j];
        t.v[i] = s;
    }
    *o = t;
}
static void sm_tmulv(const sm6* A, const sv6* x, sv6* o) { /* A^T x */
    sv6 t;
    for (int i = 0; i < 6; ++i) {
        real s = 0;
        for (int j = 0; j < 6; ++j) s += A->m[j][i] * x->v[j];
        t.v[i] = s;
    }
    *o = t;
}
/* O += X^T A X */
static void sm_add_congruence(const sm6* X, const sm6* A, sm6* O) {
    sm6 T;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            real s = 0;
            for (int k = 0; k < 6; ++k) s += A->m[i][k] * X->m[k][j];
            T.m[i][j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            real s = 0;
            for (int k = 0; k < 6; ++k) s += X->m[k][i] * T.m[k][j];
            O->m[i][j] += s;
        }
}
/* v x (motion), v x* (force) */
static void s_crm(const sv6* v, const sv6* m, sv6* o) {
    real a[3], b[3], c[3];
    v3_cross(v->v, m->v, a);
    v3_cross(v->v + 3, m->v, b);
    v3_cross(v->v, m->v + 3, c);
    for (int i = 0; i < 3; ++i) { o->v[i] = a[i]; o->v[3 + i] = b[i] + c[i]; }
}
static void s_crf(const sv6* v, const sv6* f, sv6* o) {
    real a[3], b[3], c[3];
    v3_cross(v->v, f->v, a);
    v3_cross(v->v + 3, f->v + 3, b);
    v3_cross(v->v, f->v + 3, c);
    for (int i = 0; i < 3; ++i) { o->v[i] = a[i] + b[i]; o->v[3 + i] = c[i]; }
}
/* rigid-body spatial inertia at the body origin from (m, com, Ic[xx xy xz yy yz zz]) */
static void s_rigid_inertia(real m, const real c[3], const real Ic6[6], sm6* I) {
    real cx[9], cxcx[9];
    m3_skew(c, cx);
    m3_mul(cx, cx, cxcx);
    real Ic[9] = {Ic6[0], Ic6[1], Ic6[2], Ic6[1], Ic6[3], Ic6[4], Ic6[2], Ic6[4], Ic6[5]};
    memset(I, 0, sizeof *I);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            I->m[i][j] = Ic[3 * i + j] - m * cxcx[3 * i + j];
            I->m[i][3 + j] = m * cx[3 * i + j];
            I->m[3 + i][j] = -m * cx[3 * i + j];
        }
    for (int i = 0; i < 3; ++i) I->m[3 + i][3 + i] = m;
}
/* solve A x = b for SPD 6x6 by Cholesky */
static int s_solve_spd(const sm6* A, const sv6* b, sv6* x) {
    real L[6][6];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            real s = A->m[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s <= 0) return -1;
                L[i][i] = sqrt(s);
            } else
                L[i][j] = s / L[j][j];
        }
    real y[6];
    for (int i = 0; i < 6; ++i) {
        real s = b->v[i];
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        real s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x->v[k];
        x->v[i] = s / L[i][i];
    }
    return 0;
}

/* ------------------------------------------------------------------ per-env state */
typedef struct {
    real q[ND_MAX], qd[ND_MAX];
    real pos[3], quat[4], vel[3], ang[3]; /* root state, world frame (gymapi root tensor layout) */
    real anchor[NFS][2];
    int anchor_on[NFS];
    real anchor_vimp[NFS]; /* normal approach speed at the first touch of the current contact (restitution) */
    /* env pipeline state */
    real actions[ND_MAX], last_actions[ND_MAX], last_last_actions[ND_MAX], last_dof_vel[ND_MAX];
    real torques[ND_MAX];
    real commands[3];
    real base_lin_vel[3], base_ang_vel[3], proj_grav[3];
    real air_time[2], land_time[2];
    int contact[2], contact_last[2], contact_filt[2], first_contact[2];
    real feet_height[2];
    real feet_pos[2][3], feet_vel[2][3], feet_ang[2][3];
    real torso_quat_R[9], forehead_R[9];
    real avg_force[2], avg_speed[2][3], avg_rpy[2][3];
    real link_force[NL_MAX][3]; /* net contact force per URDF link, last sub-step */
    real feet_force[2][3];      /* = link_force of the two *_foot_roll_link */
    real heights[GRX_MAX_HEIGHT_POINTS];
    real base_heights_offset;
    real episode_sums[NT], reward_terms[NT];
    int64_t episode_length;
    real rew;
    int reset, time_out;
    /* per-env constants (domain randomisation at creation) */
    real motor_strength[ND_MAX];
    real friction;      /* shape friction of this env */
    real restitution;
    real base_link_mass, base_link_com[3];
    real base_m, base_c[3], base_I[6]; /* randomised base lump */
    real origin[3];
    int level, type;
} env_t;

struct grx_sim {
    grx_config cfg;
    int N, nd, nb;
    env_t* env;
    int16_t* hf;
    int16_t* tm_cells;   /* mesh_type 'trimesh': [cell][6] ground corners of the reference's slope-corrected mesh under the cell's two triangle halves: (e00, e01, e11) where ty >= tx, (e00, e10, e11) where tx > ty; raster units */
    int16_t* tm_walls;   /* [cell][8] tops of the vertical faces on the cell's sides (x-, x+, y-, y+) and of the posts at its corners (00, 10, 01, 11); TM_NONE = none */
    float* torigins;
    /* published float32 views (row-major) */
    float *t_obs, *t_pri, *t_rew;
    uint8_t *t_reset, *t_timeout;
    int64_t* t_eplen;
    float* scratch[GRX_NUM_TENSORS];
    float* t_rbs;   /* GRX_T_RIGID_BODY_STATES (N, GRX_MAX_LINKS, 13), written by the step */
    uint8_t* scratch_u8[GRX_NUM_TENSORS];
    int32_t* scratch_i32[GRX_NUM_TENSORS];
    float stats[NT + 2];   /* [NT] episodes that ended, [NT + 1] mean terrain level after that step's curriculum moves */
    float hist[GRX_STATS_HISTORY][NT + 2];   /* GRX_T_EPISODE_STATS_HISTORY */
    int64_t seq;           /* launches that may finish episodes (steps, resets): history row = seq % GRX_STATS_HISTORY */
    int64_t nsteps;
    int num_links;
    uint32_t reset_count;
};

/* ------------------------------------------------------------------ terrain */
/* mesh_type 'trimesh' (vertical_faces): the surface IS the reference's slope-corrected triangle mesh.
 * convert_heightfield_to_trimesh (isaacgym terrain_utils.py:286-350, called by legged_robot.py:903-921) keeps the raster's heights
 * and MOVES vertices by whole cells: a vertex whose +x / -x / +y / -y / diagonal neighbour stands more than slope_threshold above
 * it goes under that neighbour (:313-325), so the low ground runs up to the high vertex's grid line and the face there is vertical.
 * Every vertex of the corrected mesh is therefore still a grid point, and the mesh over one raster cell is described by two tables
 * built once (trimesh_build):
 *   tm_cells: the height of the top surface at the cell's four corners, approached from inside the cell -- the two planes found by
 *     a vertical ray cast at the centroids of the cell's two triangle halves (split along the (0,0)-(1,1) diagonal like :335-347),
 *     evaluated at the corners.  Exact wherever one plane covers each half: every undeformed cell, and every cell a flat tread was
 *     stretched over (stairs, obstacles, stones, gaps, pits: tests/golden/trimesh_tiles.npz); where a SLOPED neighbour was stretched
 *     over the cell its two-cell interpolation is kept at the corners and rounded to the raster unit.
 *   tm_walls: the vertical faces (triangles whose projection is a segment of a grid line) that stand on the cell's four sides and,
 *     as posts, the ends of faces that run away from its four corners; kept where they rise above the cell's own ground.
 * terrain_query interpolates tm_cells per triangle half; wall_contact (below) is the sphere against the faces and posts. */
#define TM_NONE INT16_MIN
typedef struct { double x, y, z; } tmv_t;
static void tm_vertex(const int16_t* H, const int8_t* mv, int C, int a, int b, tmv_t* v) {
    size_t k = (size_t)a * C + b;
    v->x = a + mv[2 * k]; v->y = b + mv[2 * k + 1]; v->z = H[k];
}
/* the two triangles of raster cell (a, b) in the reference's order (terrain_utils.py:339-347): (ind0, ind3, ind1), (ind0, ind2, ind3) */
static void tm_triangle(const int16_t* H, const int8_t* mv, int C, int a, int b, int second, tmv_t t[3]) {
    tm_vertex(H, mv, C, a, b, &t[0]);
    if (!second) { tm_vertex(H, mv, C, a + 1, b + 1, &t[1]); tm_vertex(H, mv, C, a, b + 1, &t[2]); }
    else { tm_vertex(H, mv, C, a + 1, b, &t[1]); tm_vertex(H, mv, C, a + 1, b + 1, &t[2]); }
}
/* vertical ray at the raster point (px, py): plane (z there, dz/dx, dz/dy) of the highest triangle hit; 0 if none */
static int tm_plane_at(const int16_t* H, const int8_t* mv, int R, int C, double px, double py, double pl[3]) {
    int ci = (int)floor(px), cj = (int)floor(py), found = 0;
    for (int a = ci - 1; a <= ci + 1; ++a) {
        if (a < 0 || a > R - 2) continue;
        for (int b = cj - 1; b <= cj + 1; ++b) {
            if (b < 0 || b > C - 2) continue;
            for (int k = 0; k < 2; ++k) {
                tmv_t t[3];
                tm_triangle(H, mv, C, a, b, k, t);
                double ux = t[1].x - t[0].x, uy = t[1].y - t[0].y, vx = t[2].x - t[0].x, vy = t[2].y - t[0].y;
                double den = ux * vy - vx * uy;
                if (fabs(den) < 1e-9) continue;   /* a vertical face */
                double qx = px - t[0].x, qy = py - t[0].y;
                double w1 = (qx * vy - vx * qy) / den, w2 = (ux * qy - qx * uy) / den;
                if (w1 < -1e-9 || w2 < -1e-9 || 1 - w1 - w2 < -1e-9) continue;
                double uz = t[1].z - t[0].z, vz = t[2].z - t[0].z;
                double z = t[0].z + w1 * uz + w2 * vz;
                if (!found || z > pl[0]) { pl[0] = z; pl[1] = (uz * vy - vz * uy) / den; pl[2] = (ux * vz - vx * uz) / den; found = 1; }
            }
        }
    }
    return found;
}
static int16_t tm_round(double z) { return (int16_t)lrint(z < -32767 ? -32767 : z > 32767 ? 32767 : z); }
static void trimesh_build(struct grx_sim* s) {
    const grx_config* c = &s->cfg;
    const int R = c->hf_rows, C = c->hf_cols;
    const int16_t* H = s->hf;
    const size_t n = (size_t)R * C;
    /* the vertex moves (terrain_utils.py:310-325): in double like numpy, threshold in raster units */
    const double thr = (double)c->slope_threshold * ((double)c->horizontal_scale / (double)c->vertical_scale);
    int8_t* mv = (int8_t*)calloc(2 * n, 1);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int h = H[(size_t)i * C + j];
#define TM_UP(di, dj) ((i + (di) >= 0 && i + (di) < R && j + (dj) >= 0 && j + (dj) < C && H[(size_t)(i + (di)) * C + j + (dj)] - h > thr) ? 1 : 0)
            const int mx = TM_UP(1, 0) - TM_UP(-1, 0), my = TM_UP(0, 1) - TM_UP(0, -1), mc = TM_UP(1, 1) - TM_UP(-1, -1);
#undef TM_UP
            mv[2 * ((size_t)i * C + j)] = (int8_t)(mx + (mx == 0 ? mc : 0));
            mv[2 * ((size_t)i * C + j) + 1] = (int8_t)(my + (my == 0 ? mc : 0));
        }
    /* ground corners, per triangle half (the surface may break along the cell's diagonal: a concave corner of a raised block leaves one
     * half on the upper level and the other on the lower one) */
    s->tm_cells = (int16_t*)malloc(6 * n * sizeof(int16_t));
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int i1 = i + 1 < R ? i + 1 : R - 1, j1 = j + 1 < C ? j + 1 : C - 1;
            int16_t* e = s->tm_cells + 6 * ((size_t)i * C + j);
            e[0] = e[3] = H[(size_t)i * C + j]; e[1] = H[(size_t)i * C + j1]; e[4] = H[(size_t)i1 * C + j]; e[2] = e[5] = H[(size_t)i1 * C + j1];
            if (i > R - 2 || j > C - 2) continue;
            int touched = 0;   /* a vertex of the 3 x 3 cells around this one was moved: else the cell is its own two triangles */
            for (int a = (i > 0 ? i - 1 : 0); a <= (i + 2 < R ? i + 2 : R - 1) && !touched; ++a)
                for (int b = (j > 0 ? j - 1 : 0); b <= (j + 2 < C ? j + 2 : C - 1); ++b)
                    if (mv[2 * ((size_t)a * C + b)] | mv[2 * ((size_t)a * C + b) + 1]) { touched = 1; break; }
            if (!touched) continue;
            double p[3];
            const double x1 = i + 1.0 / 3, y1 = j + 2.0 / 3, x2 = i + 2.0 / 3, y2 = j + 1.0 / 3;   /* centroids of the halves ty >= tx, tx > ty */
#define TM_AT(px, py, cx, cy) tm_round(p[0] + p[1] * ((cx) - (px)) + p[2] * ((cy) - (py)))
            if (tm_plane_at(H, mv, R, C, x1, y1, p)) { e[0] = TM_AT(x1, y1, i, j); e[1] = TM_AT(x1, y1, i, j + 1); e[2] = TM_AT(x1, y1, i + 1, j + 1); }
            if (tm_plane_at(H, mv, R, C, x2, y2, p)) { e[3] = TM_AT(x2, y2, i, j); e[4] = TM_AT(x2, y2, i + 1, j); e[5] = TM_AT(x2, y2, i + 1, j + 1); }
#undef TM_AT
        }
    /* vertical faces: per unit segment of the grid lines x = X (segx[X][k]: y in [k, k + 1]) and y = Y (segy[Y][k]: x in [k, k + 1]) the
     * top of the faces on it AT ITS TWO ENDS ([0]: at k, [1]: at k + 1) -- where three levels meet, the vertices the reference slides along a
     * face leave it triangular (a top running down to the lower level within one cell) */
    int16_t* segx = (int16_t*)malloc(2 * n * sizeof(int16_t));
    int16_t* segy = (int16_t*)malloc(2 * n * sizeof(int16_t));
    for (size_t k = 0; k < 2 * n; ++k) segx[k] = segy[k] = TM_NONE;
    for (int a = 0; a < R - 1; ++a)
        for (int b = 0; b < C - 1; ++b)
            for (int k = 0; k < 2; ++k) {
                tmv_t t[3];
                tm_triangle(H, mv, C, a, b, k, t);
                const double den = (t[1].x - t[0].x) * (t[2].y - t[0].y) - (t[2].x - t[0].x) * (t[1].y - t[0].y);
                if (fabs(den) > 1e-9) continue;   /* (an undeformed triangle: den = -1) */
                if (fmax(t[0].z, fmax(t[1].z, t[2].z)) <= fmin(t[0].z, fmin(t[1].z, t[2].z))) continue;
                const int along_y = t[0].x == t[1].x && t[0].x == t[2].x, along_x = t[0].y == t[1].y && t[0].y == t[2].y;
                if (along_y == along_x) continue;   /* a needle, or a face across the grid (not produced by axis-aligned steps) */
                const int L = (int)(along_y ? t[0].x : t[0].y);
                double sp[3];   /* position along the line */
                for (int q = 0; q < 3; ++q) sp[q] = along_y ? t[q].y : t[q].x;
                const int lo = (int)fmin(sp[0], fmin(sp[1], sp[2])), hi = (int)fmax(sp[0], fmax(sp[1], sp[2]));
                if (hi <= lo) continue;   /* a needle */
                int16_t* seg = along_y ? segx : segy;
                const int nl = along_y ? R : C, ns = along_y ? C - 1 : R - 1, stride = along_y ? C : R;
                if (L < 0 || L >= nl) continue;
                for (int q = lo < 0 ? 0 : lo; q < hi && q < ns; ++q)
                    for (int end = 0; end < 2; ++end) {
                        const double at = q + end;
                        double top = -1e30;   /* the highest point of the triangle over `at` */
                        for (int m_ = 0; m_ < 3; ++m_) {
                            const int m2 = (m_ + 1) % 3;
                            if (at < fmin(sp[m_], sp[m2]) || at > fmax(sp[m_], sp[m2])) continue;
                            const double z = sp[m_] == sp[m2] ? fmax(t[m_].z, t[m2].z) : t[m_].z + (t[m2].z - t[m_].z) * (at - sp[m_]) / (sp[m2] - sp[m_]);
                            if (z > top) top = z;
                        }
                        int16_t* o = seg + 2 * ((size_t)L * stride + q) + end;
                        if (top > -1e29 && tm_round(top) > *o) *o = tm_round(top);
                    }
            }
    s->tm_walls = (int16_t*)malloc(8 * n * sizeof(int16_t));
#define SEGX(X, k, end) (((X) >= 0 && (X) < R && (k) >= 0 && (k) < C - 1) ? segx[2 * ((size_t)(X) * C + (k)) + (end)] : TM_NONE)
#define SEGY(Y, k, end) (((Y) >= 0 && (Y) < C && (k) >= 0 && (k) < R - 1) ? segy[2 * ((size_t)(Y) * R + (k)) + (end)] : TM_NONE)
#define MIN2(a_, b_) ((a_) < (b_) ? (a_) : (b_))
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            const int16_t* e = s->tm_cells + 6 * ((size_t)i * C + j);
            int16_t* w = s->tm_walls + 8 * ((size_t)i * C + j);
#define MAX2(a_, b_) ((a_) > (b_) ? (a_) : (b_))
            /* a side's face: the rectangle up to the lower of its two end tops (TM_NONE when an end has none) */
            const int16_t side[4] = {MIN2(SEGX(i, j, 0), SEGX(i, j, 1)), MIN2(SEGX(i + 1, j, 0), SEGX(i + 1, j, 1)),
                                     MIN2(SEGY(j, i, 0), SEGY(j, i, 1)), MIN2(SEGY(j + 1, i, 0), SEGY(j + 1, i, 1))};
            /* the cell's own ground along the side: x- and y+ bound the half ty >= tx, x+ and y- the half tx > ty */
            const int16_t ground[4] = {MAX2(e[0], e[1]), MAX2(e[4], e[5]), MAX2(e[3], e[4]), MAX2(e[1], e[2])};
            for (int q = 0; q < 4; ++q) w[q] = side[q] > ground[q] ? side[q] : TM_NONE;
            /* posts: faces that run AWAY from a corner along either grid line through it */
            const int16_t pa[4] = {SEGX(i, j - 1, 1), SEGX(i + 1, j - 1, 1), SEGX(i, j + 1, 0), SEGX(i + 1, j + 1, 0)};
            const int16_t pb[4] = {SEGY(j, i - 1, 1), SEGY(j, i + 1, 0), SEGY(j + 1, i - 1, 1), SEGY(j + 1, i + 1, 0)};
            const int16_t pg[4] = {MAX2(e[0], e[3]), e[4], e[1], MAX2(e[2], e[5])};   /* corners 00, 10, 01, 11 */
            for (int q = 0; q < 4; ++q) { const int16_t t_ = MAX2(pa[q], pb[q]); w[4 + q] = t_ > pg[q] ? t_ : TM_NONE; }
#undef MAX2
        }
#undef SEGX
#undef SEGY
#undef MIN2
    free(segx); free(segy); free(mv);
}

/* Physics terrain query: height under (x, y) and the gradient of the surface there (g[2] = dh/dx, dh/dy).
 *  - heightfield: bilinear patch of the int16 raster.
 *  - trimesh: the triangle half of the cell under the point, on the corrected mesh's corner heights (tm_cells). */
static int terrain_locate(const grx_config* c, real x, real y, real* tx, real* ty) {
    real fx = (x + c->border_size) / c->horizontal_scale;
    real fy = (y + c->border_size) / c->horizontal_scale;
    if (fx < 0) fx = 0;
    if (fy < 0) fy = 0;
    if (fx > c->hf_rows - 1) fx = (real)(c->hf_rows - 1);
    if (fy > c->hf_cols - 1) fy = (real)(c->hf_cols - 1);
    int ix = (int)fx, iy = (int)fy;
    if (ix > c->hf_rows - 2) ix = c->hf_rows - 2;
    if (iy > c->hf_cols - 2) iy = c->hf_cols - 2;
    *tx = fx - ix; *ty = fy - iy;
    return ix * c->hf_cols + iy;
}
static real terrain_query(const struct grx_sim* s, real x, real y, real g[2]) {
    const grx_config* c = &s->cfg;
    g[0] = 0; g[1] = 0;
    if (c->terrain_type == GRX_TERRAIN_PLANE) return 0;
    real tx, ty;
    const int cell = terrain_locate(c, x, y, &tx, &ty);
    const int C = c->hf_cols;
    const real sc = c->vertical_scale / c->horizontal_scale;
    if (c->vertical_faces) {
        const int16_t* e = s->tm_cells + 6 * (size_t)cell + (ty >= tx ? 0 : 3);
        const real e00 = e[0], e11 = e[2];
        real h;
        if (ty >= tx) { g[0] = e11 - e[1]; g[1] = e[1] - e00; }
        else { g[0] = e[1] - e00; g[1] = e11 - e[1]; }
        h = e00 + g[0] * tx + g[1] * ty;
        g[0] *= sc; g[1] *= sc;
        return h * c->vertical_scale;
    }
    const int16_t* H = s->hf;
    real h00 = H[cell], h10 = H[cell + C], h01 = H[cell + 1], h11 = H[cell + C + 1];
    real h = (h00 * (1 - tx) + h10 * tx) * (1 - ty) + (h01 * (1 - tx) + h11 * tx) * ty;
    g[0] = ((h10 - h00) * (1 - ty) + (h11 - h01) * ty) * sc;
    g[1] = ((h01 - h00) * (1 - tx) + (h11 - h10) * tx) * sc;
    return h * c->vertical_scale;
}
/* mesh_type 'trimesh': the deepest overlap of a sphere (centre x, radius r) with the vertical faces on the sides of the raster cell
 * under its centre and with the posts at that cell's corners (tm_walls).  A face spans its whole side, from the ground up to its top;
 * the closest point on it is at the centre's own position along the side and at min(centre height, top) -- above the top the contact
 * is with the face's upper edge.  Returns the overlap (<= 0: none) and the unit direction n from the closest point to the centre. */
static real wall_overlap(const struct grx_sim* s, const real x[3], real r, real n[3]) {
    const grx_config* c = &s->cfg;
    real tx, ty;
    const int cell = terrain_locate(c, x[0], x[1], &tx, &ty);
    const int16_t* w = s->tm_walls + 8 * (size_t)cell;
    const real hs = c->horizontal_scale;
    const real dx[2] = {tx * hs, (tx - 1) * hs}, dy[2] = {ty * hs, (ty - 1) * hs};   /* from the low / high grid line to the centre */
    real best = 0;
    for (int q = 0; q < 8; ++q) {
        if (w[q] == TM_NONE) continue;
        real d[3];
        if (q < 2) { d[0] = dx[q]; d[1] = 0; }
        else if (q < 4) { d[0] = 0; d[1] = dy[q - 2]; }
        else { d[0] = dx[(q - 4) & 1]; d[1] = dy[(q - 4) >> 1]; }
        d[2] = x[2] - w[q] * c->vertical_scale;
        if (d[2] < 0) d[2] = 0;
        const real dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (r - dist > best && dist > (real)1e-9) { best = r - dist; for (int j = 0; j < 3; ++j) n[j] = d[j] / dist; }
    }
    return best;
}

/* legged_robot.py:1235-1274 _get_heights */
static void measure_heights(const struct grx_sim* s, env_t* e) {
    const grx_config* c = &s->cfg;
    int nh = c->num_height_points;
    if (c->terrain_type == GRX_TERRAIN_PLANE) { /* legged_robot.py:1248-1249 */
        for (int k = 0; k < nh; ++k) e->heights[k] = 0;
        return;
    }
    for (int k = 0; k < nh; ++k) {
        real b[3] = {c->height_points[k][0], c->height_points[k][1], 0}, p[3];
        quat_apply_yaw(e->quat, b, p);
        real px = (p[0] + e->pos[0] + c->border_size) / c->horizontal_scale;
        real py = (p[1] + e->pos[1] + c->border_size) / c->horizontal_scale;
        long ix = (long)px, iy = (long)py; /* .long(): truncation toward zero (legged_robot.py:1259) */
        if (ix < 0) ix = 0;
        if (iy < 0) iy = 0;
        if (ix > c->hf_rows - 2) ix = c->hf_rows - 2;
        if (iy > c->hf_cols - 2) iy = c->hf_cols - 2;
        int C = c->hf_cols;
        int16_t h1 = s->hf[ix * C + iy], h2 = s->hf[(ix + 1) * C + iy], h3 = s->hf[ix * C + iy + 1];
        int16_t h = h1 < h2 ? h1 : h2;
        h = h < h3 ? h : h3;
        e->heights[k] = (real)h * c->vertical_scale;
    }
}

/* ------------------------------------------------------------------ dynamics */
typedef struct {
    real R[NB_MAX][9]; /* body -> world */
    real p[NB_MAX][3]; /* origin, world */
    sm6 X[NB_MAX];     /* motion transform parent -> body */
    sv6 v[NB_MAX];     /* spatial velocity, body coords */
} kin_t;

static void base_lump(const grx_model* m, env_t* e) {
    /* base lump = rest (all fixed descendants) + base_link with randomised mass / COM
     * (legged_robot.py:618-648: props[0] is base_link only; inertia tensor scaled with the mass) */
    real m1 = m->base_rest_mass, m2 = e->base_link_mass;
    real scale = m->base_link_mass > 0 ? m2 / m->base_link_mass : 1;
    real M = m1 + m2;
    real c[3];
    for (int i = 0; i < 3; ++i) c[i] = (m1 * m->base_rest_com[i] + m2 * e->base_link_com[i]) / M;
    real I[6];
    for (int i = 0; i < 6; ++i) I[i] = m->base_rest_inertia[i] + scale * m->base_link_inertia[i];
    const real* cs[2] = {0, 0};
    real c1[3] = {m->base_rest_com[0], m->base_rest_com[1], m->base_rest_com[2]};
    real ms[2] = {m1, m2};
    cs[0] = c1; cs[1] = e->base_link_com;
    for (int k = 0; k < 2; ++k) {
        real d[3] = {cs[k][0] - c[0], cs[k][1] - c[1], cs[k][2] - c[2]};
        real dd = v3_dot(d, d);
        I[0] += ms[k] * (dd - d[0] * d[0]); I[1] -= ms[k] * d[0] * d[1]; I[2] -= ms[k] * d[0] * d[2];
        I[3] += ms[k] * (dd - d[1] * d[1]); I[4] -= ms[k] * d[1] * d[2];
        I[5] += ms[k] * (dd - d[2] * d[2]);
    }
    e->base_m = M;
    for (int i = 0; i < 3; ++i) e->base_c[i] = c[i];
    for (int i = 0; i < 6; ++i) e->base_I[i] = I[i];
}

static void forward_kinematics(const struct grx_sim* s, const env_t* e, kin_t* k) {
    const grx_model* m = &s->cfg.model;
    quat_to_m3(e->quat, k->R[0]);
    for (int i = 0; i < 3; ++i) k->p[0][i] = e->pos[i];
    real wb[3], vb[3];
    m3_tmulv(k->R[0], e->ang, wb);
    m3_tmulv(k->R[0], e->vel, vb);
    for (int i = 0; i < 3; ++i) { k->v[0].v[i] = wb[i]; k->v[0].v[3 + i] = vb[i]; }
    for (int b = 1; b < s->nb; ++b) {
        int par = m->parent[b];
        real ax[3] = {m->joint_axis[b][0], m->joint_axis[b][1], m->joint_axis[b][2]};
        real R0[9], Rq[9], Rcp[9], E[9], r[3];
        for (int i = 0; i < 9; ++i) R0[i] = m->joint_rot0[b][i];
        for (int i = 0; i < 3; ++i) r[i] = m->joint_pos[b][i];
        m3_axis_angle(ax, e->q[b - 1], Rq);
        m3_mul(R0, Rq, Rcp); /* child -> parent */
        m3_transpose(Rcp, E);
        sx_motion(E, r, &k->X[b]);
        m3_mul(k->R[par], Rcp, k->R[b]);
        real rw[3];
        m3_mulv(k->R[par], r, rw);
        for (int i = 0; i < 3; ++i) k->p[b][i] = k->p[par][i] + rw[i];
        sm_mulv(&k->X[b], &k->v[par], &k->v[b]);
        for (int i = 0; i < 3; ++i) k->v[b].v[i] += ax[i] * e->qd[b - 1];
    }
}

/* contact: penalty normal force (Hunt-Crossley) + anchored stick/slip friction on the 8 foot
 * spheres, viscous-capped friction on the others.  Accumulates body-frame spatial forces. */
static void contact_forces(const struct grx_sim* s, env_t* e, const kin_t* k, sv6 fext[NB_MAX]) {
    const grx_config* c = &s->cfg;
    const grx_model* m = &c->model;
    const grx_contact_params* cp = &c->contact;
    for (int b = 0; b < s->nb; ++b) memset(&fext[b], 0, sizeof(sv6));
    memset(e->link_force, 0, sizeof e->link_force);
    memset(e->feet_force, 0, sizeof e->feet_force);
    real mu = (real)0.5 * (cp->terrain_friction + e->friction); /* PhysX friction combine: average */
    real e_rest = (real)0.5 * (c->terrain_restitution + e->restitution);
    int foot_slot[2] = {0, 0};
    for (int i = 0; i < m->num_spheres; ++i) {
        int b = m->sph_body[i];
        uint32_t fl = m->sph_flags[i];
        int slot = -1;
        if (fl & GRX_SPH_FOOT_LEFT) slot = foot_slot[0]++;
        else if (fl & GRX_SPH_FOOT_RIGHT) slot = 4 + foot_slot[1]++;
        real sb[3] = {m->sph_pos[i][0], m->sph_pos[i][1], m->sph_pos[i][2]}, sw[3], x[3];
        m3_mulv(k->R[b], sb, sw);
        for (int j = 0; j < 3; ++j) x[j] = k->p[b][j] + sw[j];
        real g[2];
        real h = terrain_query(s, x[0], x[1], g);
        real dv = h + m->sph_radius[i] - x[2];   /* vertical overlap */
        real F[3] = {0, 0, 0};
        /* sphere-centre velocity, world */
        real wxs[3], ub[3], u[3];
        v3_cross(k->v[b].v, sb, wxs);
        for (int j = 0; j < 3; ++j) ub[j] = k->v[b].v[3 + j] + wxs[j];
        m3_mulv(k->R[b], ub, u);
        if (dv <= 0) {
            if (slot >= 0) e->anchor_on[slot] = 0;
        } else {
        /* surface normal from the gradient of the patch; overlap along it (locally planar terrain) */
        real nn = 1 / sqrt(1 + g[0] * g[0] + g[1] * g[1]);
        real n[3] = {-g[0] * nn, -g[1] * nn, nn};
        real d = dv * nn;
        real un = u[0] * n[0] + u[1] * n[1] + u[2] * n[2];   /* > 0: separating */
        /* Hunt-Crossley damping kn*d*dn, capped by the mass-aware bound that keeps explicit
         * integration of the (light) foot stable; normal force never pulls */
        real cd = cp->kn * d * cp->dn;
        if (cd > m->sph_damp_max[i]) cd = m->sph_damp_max[i];
        if (slot >= 0 && slot < NFS) {
            if (!e->anchor_on[slot]) {
                e->anchor_on[slot] = 1;
                e->anchor[slot][0] = x[0];
                e->anchor[slot][1] = x[1];
                e->anchor_vimp[slot] = un < 0 ? -un : 0;
            }
            /* restitution (legged_robot.py:565-575; PhysX combines by averaging, bounce threshold
             * legged_robot_config.py:48): a contact that began faster than the threshold keeps only the
             * fraction (1 - e) of its damping while the sphere separates again */
            if (un > 0 && e->anchor_vimp[slot] > c->bounce_threshold_velocity) cd *= 1 - e_rest;
        }
        real fn = cp->kn * d - cd * un;
        if (fn < 0) fn = 0;
        for (int j = 0; j < 3; ++j) F[j] = fn * n[j];
        /* friction: in the horizontal plane (anchored stick/slip on the foot spheres, viscous-capped elsewhere) */
        if (slot >= 0 && slot < NFS) {
            real ftx = -cp->kt * (x[0] - e->anchor[slot][0]) - cp->ct * u[0];
            real fty = -cp->kt * (x[1] - e->anchor[slot][1]) - cp->ct * u[1];
            real ft = sqrt(ftx * ftx + fty * fty), fmax = mu * fn;
            if (ft > fmax) { /* slip: clamp to the cone, drag the anchor along */
                real sc = fmax / ft;
                ftx *= sc; fty *= sc;
                e->anchor[slot][0] = x[0] + ftx / cp->kt;
                e->anchor[slot][1] = x[1] + fty / cp->kt;
            }
            F[0] += ftx; F[1] += fty;
        } else {
            real sp = sqrt(u[0] * u[0] + u[1] * u[1]);
            real ft = cp->cv * sp, fmax = mu * fn;
            if (ft > fmax) ft = fmax;
            if (sp > (real)1e-9) { F[0] -= ft * u[0] / sp; F[1] -= ft * u[1] / sp; }
        }
        }
        /* mesh_type 'trimesh': the vertical faces of the corrected mesh next to the sphere (wall_overlap) -- the same normal law,
         * friction viscous and capped by the cone in the face's tangent plane; one contact, the deepest */
        if (c->terrain_type == GRX_TERRAIN_HEIGHTFIELD && c->vertical_faces) {
            real n[3];
            const real d = wall_overlap(s, x, m->sph_radius[i], n);
            if (d > 0) {
                const real un = u[0] * n[0] + u[1] * n[1] + u[2] * n[2];
                real cd = cp->kn * d * cp->dn;
                if (cd > m->sph_damp_max[i]) cd = m->sph_damp_max[i];
                real fn = cp->kn * d - cd * un;
                if (fn < 0) fn = 0;
                real ut[3] = {u[0] - un * n[0], u[1] - un * n[1], u[2] - un * n[2]};
                const real sp = sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
                real ft = cp->cv * sp;
                if (ft > mu * fn) ft = mu * fn;
                for (int j = 0; j < 3; ++j) F[j] += fn * n[j] - (sp > (real)1e-9 ? ft * ut[j] / sp : 0);
            }
        }
        if (F[0] == 0 && F[1] == 0 && F[2] == 0) continue;
        int L = m->sph_link[i];
        for (int j = 0; j < 3; ++j) e->link_force[L][j] += F[j];
        if (fl & GRX_SPH_FOOT_LEFT) for (int j = 0; j < 3; ++j) e->feet_force[0][j] += F[j];
        if (fl & GRX_SPH_FOOT_RIGHT) for (int j = 0; j < 3; ++j) e->feet_force[1][j] += F[j];
        real fb[3], nb_[3];
        m3_tmulv(k->R[b], F, fb);
        v3_cross(sb, fb, nb_);
        for (int j = 0; j < 3; ++j) { fext[b].v[j] += nb_[j]; fext[b].v[3 + j] += fb[j]; }
    }
    /* self-collision (self_collisions = 0 = enabled, legged_robot_config.py:121): sphere pairs of links that can touch.
     * Penalty contact along the line of centres (same Hunt-Crossley law as the terrain), viscous-capped friction with
     * the shapes' own (per-env) friction; equal and opposite forces applied at the middle of the overlap. */
    if (c->self_collisions) {
        for (int pi = 0; pi < m->num_pairs; ++pi) {
            int ia = m->pair_a[pi], ib = m->pair_b[pi];
            int ba = m->sph_body[ia], bb = m->sph_body[ib];
            real xa[3], xb[3], ua[3], ub_[3];
            for (int w = 0; w < 2; ++w) {
                int i = w ? ib : ia, b = w ? bb : ba;
                real sb[3] = {m->sph_pos[i][0], m->sph_pos[i][1], m->sph_pos[i][2]}, sw[3], wxs[3], ul[3];
                m3_mulv(k->R[b], sb, sw);
                v3_cross(k->v[b].v, sb, wxs);
                for (int j = 0; j < 3; ++j) ul[j] = k->v[b].v[3 + j] + wxs[j];
                real* x = w ? xb : xa; real* u = w ? ub_ : ua;
                for (int j = 0; j < 3; ++j) x[j] = k->p[b][j] + sw[j];
                m3_mulv(k->R[b], ul, u);
            }
            real dv[3] = {xa[0] - xb[0], xa[1] - xb[1], xa[2] - xb[2]};
            real d2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
            real Rs = m->sph_radius[ia] + m->sph_radius[ib];
            if (d2 >= Rs * Rs || d2 < (real)1e-12) continue;
            real dist = sqrt(d2), pen = Rs - dist;
            real n[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist}; /* from b to a */
            real ur[3] = {ua[0] - ub_[0], ua[1] - ub_[1], ua[2] - ub_[2]};
            real un = ur[0] * n[0] + ur[1] * n[1] + ur[2] * n[2];
            real cd = cp->kn * pen * cp->dn;
            real dm = m->sph_damp_max[ia] < m->sph_damp_max[ib] ? m->sph_damp_max[ia] : m->sph_damp_max[ib];
            if (cd > dm) cd = dm;
            real fn = cp->kn * pen - cd * un;
            if (fn < 0) fn = 0;
            real ut[3] = {ur[0] - un * n[0], ur[1] - un * n[1], ur[2] - un * n[2]};
            real sp = sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
            real ft = cp->cv * sp, fmax = e->friction * fn;
            if (ft > fmax) ft = fmax;
            real F[3] = {fn * n[0], fn * n[1], fn * n[2]};
            if (sp > (real)1e-9) for (int j = 0; j < 3; ++j) F[j] -= ft * ut[j] / sp;
            real pw[3];   /* point of application: middle of the overlap, world */
            for (int j = 0; j < 3; ++j) pw[j] = xb[j] + n[j] * (m->sph_radius[ib] - (real)0.5 * pen);
            for (int w = 0; w < 2; ++w) {
                int i = w ? ib : ia, b = w ? bb : ba;
                real Fw[3] = {w ? -F[0] : F[0], w ? -F[1] : F[1], w ? -F[2] : F[2]};
                int L = m->sph_link[i];
                uint32_t fl = m->sph_flags[i];
                for (int j = 0; j < 3; ++j) e->link_force[L][j] += Fw[j];
                if (fl & GRX_SPH_FOOT_LEFT) for (int j = 0; j < 3; ++j) e->feet_force[0][j] += Fw[j];
                if (fl & GRX_SPH_FOOT_RIGHT) for (int j = 0; j < 3; ++j) e->feet_force[1][j] += Fw[j];
                real rel[3] = {pw[0] - k->p[b][0], pw[1] - k->p[b][1], pw[2] - k->p[b][2]}, rb[3], fb[3], nb_[3];
                m3_tmulv(k->R[b], rel, rb);
                m3_tmulv(k->R[b], Fw, fb);
                v3_cross(rb, fb, nb_);
                for (int j = 0; j < 3; ++j) { fext[b].v[j] += nb_[j]; fext[b].v[3 + j] += fb[j]; }
            }
        }
    }
}

/* Articulated-Body Algorithm, floating base, gravity-free (gravity is added to the base
 * acceleration afterwards: uniform-field equivalence).  RBDA Table 7.1 + sec. 9.4. */
static int aba(const struct grx_sim* s, const env_t* e, const kin_t* k, const real tau[ND_MAX],
               const sv6 fext[NB_MAX], real qdd[ND_MAX], sv6* a0_out) {
    const grx_model* m = &s->cfg.model;
    int nb = s->nb;
    static __thread sm6 IA[NB_MAX];
    sv6 pA[NB_MAX], c[NB_MAX], U[NB_MAX], a[NB_MAX];
    real dinv[NB_MAX], u[NB_MAX];
    for (int b = 0; b < nb; ++b) {
        if (b == 0) s_rigid_inertia(e->base_m, e->base_c, e->base_I, &IA[0]);
        else {
            real cc[3] = {m->com[b][0], m->com[b][1], m->com[b][2]};
            real I6[6];
            for (int i = 0; i < 6; ++i) I6[i] = m->inertia[b][i];
            s_rigid_inertia(m->mass[b], cc, I6, &IA[b]);
        }
        sv6 Iv, t;
        sm_mulv(&IA[b], &k->v[b], &Iv);
        s_crf(&k->v[b], &Iv, &t);
        for (int i = 0; i < 6; ++i) pA[b].v[i] = t.v[i] - fext[b].v[i];
        memset(&c[b], 0, sizeof(sv6));
        if (b > 0) {
            sv6 Sq;
            memset(&Sq, 0, sizeof Sq);
            for (int i = 0; i < 3; ++i) Sq.v[i] = m->joint_axis[b][i] * e->qd[b - 1];
            s_crm(&k->v[b], &Sq, &c[b]);
        }
    }
    for (int b = nb - 1; b >= 1; --b) {
        sv6 S;
        memset(&S, 0, sizeof S);
        for (int i = 0; i < 3; ++i) S.v[i] = m->joint_axis[b][i];
        sm_mulv(&IA[b], &S, &U[b]);
        real d = m->dof_armature[b - 1], sp = 0;   /* joint-space armature (asset_options.armature, legged_robot_config.py:125) */
        for (int i = 0; i < 6; ++i) { d += S.v[i] * U[b].v[i]; sp += S.v[i] * pA[b].v[i]; }
        dinv[b] = 1 / d;
        u[b] = tau[b - 1] - sp;
        sm6 Ia = IA[b];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) Ia.m[i][j] -= U[b].v[i] * U[b].v[j] * dinv[b];
        sv6 pa, Iac;
        sm_mulv(&Ia, &c[b], &Iac);
        for (int i = 0; i < 6; ++i) pa.v[i] = pA[b].v[i] + Iac.v[i] + U[b].v[i] * u[b] * dinv[b];
        int par = m->parent[b];
        sm_add_congruence(&k->X[b], &Ia, &IA[par]);
        sv6 pp;
        sm_tmulv(&k->X[b], &pa, &pp);
        for (int i = 0; i < 6; ++i) pA[par].v[i] += pp.v[i];
    }
    sv6 rhs;
    for (int i = 0; i < 6; ++i) rhs.v[i] = -pA[0].v[i];
    if (s_solve_spd(&IA[0], &rhs, &a[0])) return -1;
    for (int b = 1; b < nb; ++b) {
        int par = m->parent[b];
        sv6 ap;
        sm_mulv(&k->X[b], &a[par], &ap);
        real Ua = 0;
        for (int i = 0; i < 6; ++i) { ap.v[i] += c[b].v[i]; Ua += U[b].v[i] * ap.v[i]; }
        qdd[b - 1] = (u[b] - Ua) * dinv[b];
        a[b] = ap;
        for (int i = 0; i < 3; ++i) a[b].v[i] += m->joint_axis[b][i] * qdd[b - 1];
    }
    *a0_out = a[0];
    return 0;
}

/* one physics sub-step: gym.simulate(dt) (legged_robot_fftai.py:68) */
static int substep(const struct grx_sim* s, env_t* e, const real tau_motor[ND_MAX], kin_t* k) {
    const grx_config* c = &s->cfg;
    const grx_model* m = &c->model;
    real dt = c->sim_dt;
    sv6 fext[NB_MAX];
    contact_forces(s, e, k, fext);
    real tau[ND_MAX];
    for (int j = 0; j < s->nd; ++j) {
        real K = c->contact.k_limit * m->dof_effort[j], C = c->contact.c_limit * K, t = tau_motor[j];
        if (e->q[j] < m->dof_lower[j]) t += K * (m->dof_lower[j] - e->q[j]) - C * e->qd[j];
        else if (e->q[j] > m->dof_upper[j]) t += K * (m->dof_upper[j] - e->q[j]) - C * e->qd[j];
        tau[j] = t;
    }
    real qdd[ND_MAX];
    sv6 a0;
    if (aba(s, e, k, tau, fext, qdd, &a0)) return -1;
    /* base: spatial -> classical acceleration, to world, add gravity */
    real wxv[3], al[3], aw[3], alw[3];
    v3_cross(k->v[0].v, k->v[0].v + 3, wxv);
    for (int i = 0; i < 3; ++i) al[i] = a0.v[3 + i] + wxv[i];
    m3_mulv(k->R[0], al, alw);
    m3_mulv(k->R[0], a0.v, aw);
    for (int i = 0; i < 3; ++i) {
        e->vel[i] += (alw[i] + c->gravity[i]) * dt;
        e->ang[i] += aw[i] * dt;
    }
    /* semi-implicit Euler: positions advance with the NEW velocities */
    for (int j = 0; j < s->nd; ++j) {
        real v = e->qd[j] + qdd[j] * dt, vl = m->dof_vel_limit[j];
        if (v > vl) v = vl;
        if (v < -vl) v = -vl;
        e->qd[j] = v;
        e->q[j] += v * dt;
    }
    for (int i = 0; i < 3; ++i) e->pos[i] += e->vel[i] * dt;
    /* quaternion: q <- normalise(dq * q), dq = (w dt / 2, 1) */
    real hx = (real)0.5 * dt * e->ang[0], hy = (real)0.5 * dt * e->ang[1], hz = (real)0.5 * dt * e->ang[2];
    real x = e->quat[0], y = e->quat[1], z = e->quat[2], w = e->quat[3];
    real nx = x + hx * w + hy * z - hz * y;
    real ny = y - hx * z + hy * w + hz * x;
    real nz = z + hx * y - hy * x + hz * w;
    real nw = w - hx * x - hy * y - hz * z;
    real n = 1 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    e->quat[0] = nx * n; e->quat[1] = ny * n; e->quat[2] = nz * n; e->quat[3] = nw * n;
    return 0;
}

/* foot / torso frames from a kinematics pass */
static void named_frames(const struct grx_sim* s, env_t* e, const kin_t* k) {
    const grx_model* m = &s->cfg.model;
    for (int f = 0; f < 2; ++f) {
        int b = m->foot_body[f];
        real sb[3] = {m->foot_pos[f][0], m->foot_pos[f][1], m->foot_pos[f][2]}, sw[3], wxs[3], ub[3];
        m3_mulv(k->R[b], sb, sw);
        for (int i = 0; i < 3; ++i) e->feet_pos[f][i] = k->p[b][i] + sw[i];
        v3_cross(k->v[b].v, sb, wxs);
        for (int i = 0; i < 3; ++i) ub[i] = k->v[b].v[3 + i] + wxs[i];
        m3_mulv(k->R[b], ub, e->feet_vel[f]);
        m3_mulv(k->R[b], k->v[b].v, e->feet_ang[f]);
    }
    if (m->torso_body >= 0) {
        real T[9];
        for (int i = 0; i < 9; ++i) T[i] = m->torso_rot[i];
        m3_mul(k->R[m->torso_body], T, e->torso_quat_R);
    }
    if (m->forehead_body >= 0) {
        real T[9];
        for (int i = 0; i < 9; ++i) T[i] = m->forehead_rot[i];
        m3_mul(k->R[m->forehead_body], T, e->forehead_R);
    }
}

/* GRX_T_RIGID_BODY_STATES row of env le (gym.acquire_rigid_body_state_tensor, legged_robot.py:113,134): every URDF link frame,
 * p3 q4(xyzw) v3 w3 in world axes, from the body frames after the last sub-step */
static void m3_to_quat(const real R[9], real q[4]) { /* largest-component form */
    real t0 = 1 + R[0] - R[4] - R[8], t1 = 1 - R[0] + R[4] - R[8], t2 = 1 - R[0] - R[4] + R[8], t3 = 1 + R[0] + R[4] + R[8];
    if (t3 >= t0 && t3 >= t1 && t3 >= t2) { q[0] = R[7] - R[5]; q[1] = R[2] - R[6]; q[2] = R[3] - R[1]; q[3] = t3; }
    else if (t0 >= t1 && t0 >= t2) { q[0] = t0; q[1] = R[1] + R[3]; q[2] = R[2] + R[6]; q[3] = R[7] - R[5]; }
    else if (t1 >= t2) { q[0] = R[1] + R[3]; q[1] = t1; q[2] = R[5] + R[7]; q[3] = R[2] - R[6]; }
    else { q[0] = R[2] + R[6]; q[1] = R[5] + R[7]; q[2] = t2; q[3] = R[3] - R[1]; }
    real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
}
static void link_frames(struct grx_sim* s, int le, const kin_t* k) {
    const grx_model* m = &s->cfg.model;
    float* out = s->t_rbs + (size_t)le * GRX_MAX_LINKS * 13;
    for (int l = 0; l < m->num_links; ++l) {
        int b = m->link_body[l];
        real lp[3] = {m->link_pos[l][0], m->link_pos[l][1], m->link_pos[l][2]}, off[3], wxs[3], ub[3], vw[3], ww[3], Rl[9], LR[9], q[4];
        m3_mulv(k->R[b], lp, off);
        v3_cross(k->v[b].v, lp, wxs);
        for (int i = 0; i < 3; ++i) ub[i] = k->v[b].v[3 + i] + wxs[i];
        m3_mulv(k->R[b], ub, vw);
        m3_mulv(k->R[b], k->v[b].v, ww);
        for (int i = 0; i < 9; ++i) LR[i] = m->link_rot[l][i];
        m3_mul(k->R[b], LR, Rl);
        m3_to_quat(Rl, q);
        float* o = out + l * 13;
        for (int i = 0; i < 3; ++i) { o[i] = (float)(k->p[b][i] + off[i]); o[7 + i] = (float)vw[i]; o[10 + i] = (float)ww[i]; }
        for (int i = 0; i < 4; ++i) o[3 + i] = (float)q[i];
    }
}

/* ------------------------------------------------------------------ env pipeline */
static real urand(const struct grx_sim* s, int le, uint32_t step, uint32_t stream, uint32_t i, real lo, real hi) {
    float u = gro_rand(s->cfg.seed, (uint32_t)(s->cfg.env_offset + le), step, stream, i);
    return (hi - lo) * (real)u + lo; /* torch_rand_float: (upper-lower)*rand + lower (torch_utils.py:193-196) */
}

/* _compute_torques before the motor-strength ratio and the clip (legged_robot.py:693-707): a = the clipped action */
static real control_torque(const grx_config* c, const env_t* e, int j, real a) {
    const real as = a * c->action_scale;
    if (c->control_type == GRX_CONTROL_V)
        return c->kp[j] * (as - e->qd[j]) - c->kd[j] * (e->qd[j] - e->last_dof_vel[j]) / c->sim_dt;
    if (c->control_type == GRX_CONTROL_T) return as;
    return c->kp[j] * (as + c->default_dof_pos[j] - e->q[j]) - c->kd[j] * e->qd[j];
}

/* isaacgym torch_utils / legged_gym math.py:38-41 wrap_to_pi: angles %= 2 pi (Python: result in [0, 2 pi)); angles -= 2 pi (angles > pi) */
static real wrap_to_pi(real x) {
    const real tp = (real)(2.0 * 3.14159265358979323846);
    real r = fmod(x, tp);
    if (r < 0) r += tp;
    if (r > (real)3.14159265358979323846) r -= tp;
    return r;
}

/* legged_robot.py:320-326: the yaw command from the heading error, commands_heading = 0 (gr1t1.py:124: never written) */
static void heading_rule(const grx_config* c, env_t* e) {
    if (!c->heading_command) return;
    const real fwd0[3] = {1, 0, 0};
    real fwd[3];
    quat_apply(e->quat, fwd0, fwd);
    real y = (real)0.5 * wrap_to_pi((real)0 - atan2(fwd[1], fwd[0]));
    if (y < c->cmd_ang_vel_yaw[0]) y = c->cmd_ang_vel_yaw[0];
    if (y > c->cmd_ang_vel_yaw[1]) y = c->cmd_ang_vel_yaw[1];
    e->commands[2] = y;
}

/* legged_robot.py:650-677 */
static void resample_commands(const struct grx_sim* s, env_t* e, int le, uint32_t step, uint32_t stream) {
    const grx_config* c = &s->cfg;
    e->commands[0] = urand(s, le, step, stream, 0, c->cmd_lin_vel_x[0], c->cmd_lin_vel_x[1]);
    e->commands[1] = urand(s, le, step, stream, 1, c->cmd_lin_vel_y[0], c->cmd_lin_vel_y[1]);
    real n = sqrt(e->commands[0] * e->commands[0] + e->commands[1] * e->commands[1]);
    real keep = n > (real)0.1 ? 1 : 0; /* set small commands to zero (legged_robot.py:666) */
    e->commands[0] *= keep;
    e->commands[1] *= keep;
    /* heading mode: the draw goes to commands[:, 3], which nothing reads (legged_robot.py:668-671; commands_heading stays 0) */
    if (!c->heading_command) e->commands[2] = urand(s, le, step, stream, 2, c->cmd_ang_vel_yaw[0], c->cmd_ang_vel_yaw[1]);
}

static void env_origin_from_terrain(const struct grx_sim* s, env_t* e) {
    const grx_config* c = &s->cfg;
    const float* o = s->torigins + ((size_t)e->level * c->num_terrain_cols + e->type) * 3;
    for (int i = 0; i < 3; ++i) e->origin[i] = o[i];
}

/* legged_robot.py:799-826 */
static void update_terrain_curriculum(const struct grx_sim* s, env_t* e, int le, uint32_t step) {
    const grx_config* c = &s->cfg;
    real dx = e->pos[0] - e->origin[0], dy = e->pos[1] - e->origin[1];
    real dist = sqrt(dx * dx + dy * dy);
    int up = dist > c->terrain_length / 2;
    real cn = sqrt(e->commands[0] * e->commands[0] + e->commands[1] * e->commands[1]);
    int down = (dist < cn * c->max_episode_length_s * (real)0.5) && !up;
    e->level += up - down;
    if (e->level >= c->num_terrain_rows) {
        float u = gro_rand(c->seed, (uint32_t)(c->env_offset + le), step, GRO_RNG_CURRICULUM, 0);
        int r = (int)(u * c->num_terrain_rows);
        if (r >= c->num_terrain_rows) r = c->num_terrain_rows - 1;
        e->level = r;
    } else if (e->level < 0)
        e->level = 0;
    env_origin_from_terrain(s, e);
}

/* legged_robot.py:377-440 + legged_robot_fftai.py:137-146 (the per-env part) */
static void reset_env(struct grx_sim* s, env_t* e, int le, uint32_t step, int init_done) {
    const grx_config* c = &s->cfg;
    if (c->curriculum && c->terrain_type != GRX_TERRAIN_PLANE && init_done) update_terrain_curriculum(s, e, le, step);
    /* _reset_dofs legged_robot.py:717-740 */
    for (int j = 0; j < s->nd; ++j) {
        real f = c->randomize_init_dof_pos ? urand(s, le, step, GRO_RNG_RESET_DOF, j, (real)0.5, (real)1.5) : 1;
        e->q[j] = f * c->default_dof_pos[j];
        e->qd[j] = 0;
    }
    /* _reset_root_states legged_robot.py:742-784 */
    for (int i = 0; i < 3; ++i) e->pos[i] = c->init_pos[i] + e->origin[i];
    if (c->terrain_type != GRX_TERRAIN_PLANE) { /* custom_origins */
        e->pos[0] += urand(s, le, step, GRO_RNG_RESET_ROOT, 0, -1, 1);
        e->pos[1] += urand(s, le, step, GRO_RNG_RESET_ROOT, 1, -1, 1);
    }
    real yaw = urand(s, le, step, GRO_RNG_RESET_ROOT, 2, (real)(-2 * M_PI), (real)(2 * M_PI));
    /* quat_from_euler_xyz(0, 0, yaw) torch_utils.py:176-190 */
    e->quat[0] = 0; e->quat[1] = 0; e->quat[2] = sin(yaw * (real)0.5); e->quat[3] = cos(yaw * (real)0.5);
    for (int i = 0; i < 3; ++i) {
        if (c->randomize_init_base_velocity) {
            e->vel[i] = urand(s, le, step, GRO_RNG_RESET_ROOT, 3 + i, (real)-0.5, (real)0.5);
            e->ang[i] = urand(s, le, step, GRO_RNG_RESET_ROOT, 6 + i, (real)-0.5, (real)0.5);
        } else { e->vel[i] = 0; e->ang[i] = 0; }
    }
    resample_commands(s, e, le, step, GRO_RNG_CMD_RESET);
    for (int j = 0; j < s->nd; ++j) { e->last_actions[j] = 0; e->last_dof_vel[j] = 0; e->last_last_actions[j] = 0; }
    for (int f = 0; f < 2; ++f) {
        e->air_time[f] = 0; e->land_time[f] = 0;
        e->contact[f] = 0; e->contact_last[f] = 0; /* feet_contact and feet_contact_last alias (fftai:113,141) */
    }
    e->episode_length = 0;
    for (int i = 0; i < NFS; ++i) e->anchor_on[i] = 0;
    /* link_force / feet_force are NOT cleared: reset_idx does not touch the contact-force tensor, which keeps the values of
       the last gym.simulate until the next refresh (legged_robot.py:377-440 vs :266) */
}

static real sum_abs_masked(const real* a, int n, uint32_t mask) {
    real s = 0;
    for (int j = 0; j < n; ++j) if (mask & (1u << j)) s += fabs(a[j]);
    return s;
}

/* the reward terms: returns unscaled r_i for every term (legged_robot_fftai.py:180-352, gr1t1.py:338-589) */
static void reward_terms(const struct grx_sim* s, const env_t* e, real r[NT]) {
    const grx_config* c = &s->cfg;
    const grx_model* m = &c->model;
    const float* sg = c->reward_sigma;
    int nd = s->nd;
    real as = c->action_scale;
    memset(r, 0, sizeof(real) * NT);
    real cmd_n = sqrt(e->commands[0] * e->commands[0] + e->commands[1] * e->commands[1]);
    real moving = cmd_n > (real)0.1 ? 1 : 0;
    real H = c->swing_feet_height_target, T = c->feet_air_time_target;
    real s1 = 0, s2 = 0, s3 = 0;
    for (int j = 0; j < nd; ++j) {
        real d1 = (e->last_actions[j] - e->actions[j]) * as;
        real d2 = (e->last_last_actions[j] - e->last_actions[j]) * as;
        s1 += fabs(d1);
        s2 += fabs(d1 - d2);
        if (c->knee_mask & (1u << j)) s3 += fabs((e->actions[j] - e->last_actions[j]) * as);
    }
    r[GRX_REW_ACTION_DIFF] = 1 - exp(sg[GRX_REW_ACTION_DIFF] * s1);
    r[GRX_REW_ACTION_DIFF_DIFF] = 1 - exp(sg[GRX_REW_ACTION_DIFF_DIFF] * s2);
    r[GRX_REW_ACTION_DIFF_KNEE] = 1 - exp(sg[GRX_REW_ACTION_DIFF_KNEE] * s3);
    r[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] = exp(sg[GRX_REW_CMD_DIFF_ANG_VEL_PITCH] * fabs(0 - e->base_ang_vel[1]));
    r[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] = exp(sg[GRX_REW_CMD_DIFF_ANG_VEL_ROLL] * fabs(0 - e->base_ang_vel[0]));
    r[GRX_REW_CMD_DIFF_ANG_VEL_YAW] = exp(sg[GRX_REW_CMD_DIFF_ANG_VEL_YAW] * fabs(e->commands[2] - e->base_ang_vel[2]));
    {   /* uses the PREVIOUS step's base_heights_offset */
        real h = e->base_heights_offset;
        real err = fabs(h) * (h < 0 ? 1 : 0);
        r[GRX_REW_CMD_DIFF_BASE_HEIGHT] = exp(sg[GRX_REW_CMD_DIFF_BASE_HEIGHT] * err);
    }
    r[GRX_REW_CMD_DIFF_BASE_ORIENT] = exp(sg[GRX_REW_CMD_DIFF_BASE_ORIENT] * (fabs(e->proj_grav[0]) + fabs(e->proj_grav[1])));
    if (m->forehead_body >= 0) {
        /* R^T (0,0,-1) = -third row of R (body->world) */
        real gx = -e->forehead_R[6], gy = -e->forehead_R[7];
        r[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] = exp(sg[GRX_REW_CMD_DIFF_FOREHEAD_ORIENT] * (fabs(gx) + fabs(gy)));
    }
    r[GRX_REW_CMD_DIFF_LIN_VEL_X] = exp(sg[GRX_REW_CMD_DIFF_LIN_VEL_X] * fabs(e->commands[0] - e->base_lin_vel[0]));
    r[GRX_REW_CMD_DIFF_LIN_VEL_Y] = exp(sg[GRX_REW_CMD_DIFF_LIN_VEL_Y] * fabs(e->commands[1] - e->base_lin_vel[1]));
    r[GRX_REW_CMD_DIFF_LIN_VEL_Z] = exp(sg[GRX_REW_CMD_DIFF_LIN_VEL_Z] * fabs(0 - e->base_lin_vel[2]));
    if (m->torso_body >= 0) {
        real gx = -e->torso_quat_R[6], gy = -e->torso_quat_R[7];
        r[GRX_REW_CMD_DIFF_TORSO_ORIENT] = exp(sg[GRX_REW_CMD_DIFF_TORSO_ORIENT] * (fabs(gx) + fabs(gy)));
    }
    {   /* collision: count penalised links with |F| > 0.1 (legged_robot_fftai.py:185-194) */
        real cnt = 0;
        int seen[NL_MAX] = {0};
        for (int i = 0; i < m->num_spheres; ++i) {
            int L = m->sph_link[i];
            if (!(m->sph_flags[i] & GRX_SPH_PENALISE) || seen[L]) continue;
            seen[L] = 1;
            const real* F = e->link_force[L];
            if (sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) > (real)0.1) cnt += 1;
        }
        r[GRX_REW_COLLISION] = 1 - exp(sg[GRX_REW_COLLISION] * cnt);
    }
    real sacc = 0, stor = 0, svel = 0, spose = 0, slim_a = 0, slim_p = 0, slim_t = 0, slim_v = 0;
    for (int j = 0; j < nd; ++j) {
        sacc += fabs((e->qd[j] - e->last_dof_vel[j]) / (c->sim_dt * c->decimation));
        stor += fabs(e->torques[j]);
        svel += fabs(e->qd[j]);
        spose += fabs(e->q[j] - c->default_dof_pos[j]);
        real mid = (m->dof_lower[j] + m->dof_upper[j]) / 2, rng = m->dof_upper[j] - m->dof_lower[j];
        real lo = mid - (real)0.5 * rng * c->soft_dof_pos_limit, hi = mid + (real)0.5 * rng * c->soft_dof_pos_limit;
        real a = e->actions[j] * as, oa = 0, op = 0;
        if (a - lo < 0) oa += -(a - lo);
        if (a - hi > 0) oa += (a - hi);
        slim_a += oa * oa;
        if (e->q[j] - lo < 0) op += -(e->q[j] - lo);
        if (e->q[j] - hi > 0) op += (e->q[j] - hi);
        slim_p += fabs(op);
        real ov = fabs(e->qd[j]) - m->dof_vel_limit[j] * c->soft_dof_vel_limit;
        if (ov < 0) ov = 0;
        if (ov > 1) ov = 1;
        slim_v += ov;
        real ot = fabs(e->torques[j]) - m->dof_effort[j] * c->soft_torque_limit;
        if (ot < 0) ot = 0;
        slim_t += ot;
    }
    r[GRX_REW_DOF_ACC_NEW] = 1 - exp(sg[GRX_REW_DOF_ACC_NEW] * sacc);
    r[GRX_REW_DOF_TOR_NEW] = 1 - exp(sg[GRX_REW_DOF_TOR_NEW] * stor);
    r[GRX_REW_DOF_TOR_NEW_HIP_ROLL] = 1 - exp(sg[GRX_REW_DOF_TOR_NEW_HIP_ROLL] * sum_abs_masked(e->torques, nd, c->hip_roll_mask));
    r[GRX_REW_DOF_VEL_NEW] = 1 - exp(sg[GRX_REW_DOF_VEL_NEW] * svel);
    r[GRX_REW_DOF_VEL_NEW_KNEE] = 1 - exp(sg[GRX_REW_DOF_VEL_NEW_KNEE] * sum_abs_masked(e->qd, nd, c->knee_mask));
    {   /* gr1t1.py:398-421 */
        real err = 0;
        uint32_t masks[2] = {c->ankle_left_mask, c->ankle_right_mask};
        for (int f = 0; f < 2; ++f) {
            real h = e->feet_height[f];
            err += sum_abs_masked(e->torques, nd, masks[f]) * fabs(h) * (h > H / 2 ? 1 : 0);
        }
        r[GRX_REW_DOF_TOR_ANKLE_FEET_LIFT_UP] = 1 - exp(sg[GRX_REW_DOF_TOR_ANKLE_FEET_LIFT_UP] * err);
    }
    {   /* gr1t1.py:534-549 / 502-532 / 490-500 / 551-560 */
        real af = 0, ah = 0, at = 0, lt = 0;
        real hmin = e->feet_height[0] < e->feet_height[1] ? e->feet_height[0] : e->feet_height[1];
        for (int f = 0; f < 2; ++f) {
            real mid = fabs(e->air_time[f] - T / 2);
            af += mid * e->avg_force[f];
            ah += mid * fabs(e->feet_height[f] - hmin - H);
            at += exp(sg[GRX_REW_FEET_AIR_TIME] * fabs(e->air_time[f] - T)) * (e->first_contact[f] ? 1 : 0);
            real le = (e->land_time[f] - c->feet_land_time_max) * (e->land_time[f] > c->feet_land_time_max ? 1 : 0);
            lt += 1 - exp(sg[GRX_REW_FEET_LAND_TIME] * le);
        }
        r[GRX_REW_FEET_AIR_FORCE] = exp(sg[GRX_REW_FEET_AIR_FORCE] * af) * moving;
        r[GRX_REW_FEET_AIR_HEIGHT] = exp(sg[GRX_REW_FEET_AIR_HEIGHT] * ah) * moving;
        r[GRX_REW_FEET_AIR_TIME] = at * moving;
        r[GRX_REW_FEET_LAND_TIME] = lt * moving;
    }
    {   /* gr1t1.py:425-452, 454-486 */
        real exy = 0, ez = 0;
        for (int f = 0; f < 2; ++f) {
            real h = e->feet_height[f];
            real close = fabs(h - H / 4) * (h < H / 4 ? 1 : 0) / (H / 4);
            real sxy = sqrt(e->avg_speed[f][0] * e->avg_speed[f][0] + e->avg_speed[f][1] * e->avg_speed[f][1]);
            exy += sxy * close;
            real far = fabs(h - H * 3 / 4) * (h > H * 3 / 4 ? 1 : 0) / (H * 1 / 4);
            ez += fabs(e->avg_speed[f][2]) * far;
        }
        r[GRX_REW_FEET_SPEED_XY_CLOSE_TO_GROUND] = exp(sg[GRX_REW_FEET_SPEED_XY_CLOSE_TO_GROUND] * exy);
        r[GRX_REW_FEET_SPEED_Z_CLOSE_TO_HEIGHT_TARGET] = exp(sg[GRX_REW_FEET_SPEED_Z_CLOSE_TO_HEIGHT_TARGET] * ez);
    }
    {   /* gr1t1.py:571-589 */
        real st = 0;
        for (int f = 0; f < 2; ++f) {
            const real* F = e->feet_force[f];
            real err = sqrt(F[0] * F[0] + F[1] * F[1]) - c->feet_stumble_ratio * fabs(F[2]);
            err = err * (err > 0 ? 1 : 0);
            st += 1 - exp(sg[GRX_REW_FEET_STUMBLE] * err);
        }
        r[GRX_REW_FEET_STUMBLE] = st;
    }
    r[GRX_REW_LIMITS_ACTIONS] = 1 - exp(sg[GRX_REW_LIMITS_ACTIONS] * slim_a);
    r[GRX_REW_LIMITS_DOF_POS] = 1 - exp(sg[GRX_REW_LIMITS_DOF_POS] * slim_p);
    r[GRX_REW_LIMITS_DOF_TOR] = 1 - exp(sg[GRX_REW_LIMITS_DOF_TOR] * slim_t);
    r[GRX_REW_LIMITS_DOF_VEL] = 1 - exp(sg[GRX_REW_LIMITS_DOF_VEL] * slim_v);
    r[GRX_REW_ON_THE_AIR] = (e->contact[0] + e->contact[1]) == 0 ? 1 : 0;
    r[GRX_REW_POSE_OFFSET] = exp(sg[GRX_REW_POSE_OFFSET] * spose);
    {
        real sh = 0;
        for (int j = 0; j < nd; ++j) if (c->hip_yaw_mask & (1u << j)) sh += fabs(e->q[j] - c->default_dof_pos[j]);
        r[GRX_REW_POSE_OFFSET_HIP_YAW] = 1 - exp(sg[GRX_REW_POSE_OFFSET_HIP_YAW] * sh);
    }
    r[GRX_REW_STAND_STILL] = exp(sg[GRX_REW_STAND_STILL] * spose) * (cmd_n < (real)0.1 ? 1 : 0);
    r[GRX_REW_TERMINATION] = (e->reset && !e->time_out) ? 1 : 0;
}

/* ------------------------------------------------------------------ the step */
static void publish(struct grx_sim* s);

static void build_observations(struct grx_sim* s, env_t* e, int le, const grx_step_args* args, uint32_t step) {
    const grx_config* c = &s->cfg;
    int nd = s->nd, nh = c->measure_heights ? c->num_height_points : 0;
    float* obs = ((args && args->obs_out) ? args->obs_out : s->t_obs) + (size_t)le * c->num_obs;
    float* pri = ((args && args->pri_obs_out) ? args->pri_obs_out : s->t_pri) + (size_t)le * c->num_pri_obs;
    /* compute_observation_variables legged_robot_fftai.py:148-167 (uses post-reset root z, pre-reset heights) */
    real sum = 0, sur[GRX_MAX_HEIGHT_POINTS];
    for (int k = 0; k < nh; ++k) {
        real d = e->pos[2] - c->base_height_target - e->heights[k];
        if (d < -1) d = -1;
        if (d > 1) d = 1;
        sur[k] = d * c->obs_scale_height;
        sum += sur[k];
    }
    e->base_heights_offset = nh > 0 ? sum / nh : 0;
    /* compute_observation_profile gr1t1.py:281-313 */
    real o[3 + 3 + 3 + 3 * ND_MAX];
    int n = 0;
    for (int i = 0; i < 3; ++i) o[n++] = e->commands[i] * 1; /* commands_scale = ones (gr1t1.py:124) */
    for (int i = 0; i < 3; ++i) o[n++] = e->base_ang_vel[i] * c->obs_scale_ang_vel;
    for (int i = 0; i < 3; ++i) o[n++] = e->proj_grav[i] * c->obs_scale_gravity;
    for (int j = 0; j < nd; ++j) o[n++] = (e->q[j] - c->default_dof_pos[j]) * c->obs_scale_dof_pos;
    for (int j = 0; j < nd; ++j) o[n++] = e->qd[j] * c->obs_scale_dof_vel;
    for (int j = 0; j < nd; ++j) o[n++] = e->actions[j] * c->obs_scale_action;
    /* pri_obs copies obs BEFORE noise */
    int p = 0;
    for (int i = 0; i < n; ++i) pri[p++] = (float)o[i];
    for (int i = 0; i < 3; ++i) pri[p++] = (float)(e->base_lin_vel[i] * c->obs_scale_lin_vel);
    pri[p++] = (float)(e->base_heights_offset * c->obs_scale_height);
    for (int f = 0; f < 2; ++f) pri[p++] = (float)e->contact[f];
    for (int f = 0; f < 2; ++f) pri[p++] = (float)(e->feet_height[f] * c->obs_scale_height);
    for (int k = 0; k < nh; ++k) pri[p++] = (float)(sur[k] * c->obs_scale_height);
    /* compute_observation_noise legged_robot.py:478-481, noise vec gr1t1.py:315-336 */
    if (c->add_noise) {
        for (int i = 0; i < n; ++i) {
            real sc = 0;
            if (i >= 3 && i < 6) sc = c->noise_ang_vel * c->noise_level * c->obs_scale_ang_vel;
            else if (i >= 6 && i < 9) sc = c->noise_gravity * c->noise_level * c->obs_scale_gravity;
            else if (i >= 9 && i < 9 + nd) sc = c->noise_dof_pos * c->noise_level * c->obs_scale_dof_pos;
            else if (i >= 9 + nd && i < 9 + 2 * nd) sc = c->noise_dof_vel * c->noise_level * c->obs_scale_dof_vel;
            else if (i >= 9 + 2 * nd) sc = c->noise_action * c->noise_level * c->obs_scale_action;
            if (sc == 0) continue;
            float u;
            if (args && args->noise_uniform) u = args->noise_uniform[(size_t)le * c->num_obs + i];
            else if (i < 9) u = gro_rand(c->seed, (uint32_t)(c->env_offset + le), step, GRO_RNG_NOISE, (uint32_t)(i - 3));
            else { /* dof terms: one stream per leg half so that both GPU lanes of an env index their blocks statically */
                int g = (i - 9) / nd, j = (i - 9) % nd, half = nd / 2;
                int right = j >= half, k = right ? j - half : j;
                u = gro_rand(c->seed, (uint32_t)(c->env_offset + le), step, right ? GRO_RNG_NOISE_DOF_R : GRO_RNG_NOISE_DOF_L, (uint32_t)(g * half + k));
            }
            o[i] += (2 * (real)u - 1) * sc;
        }
    }
    real clip = c->clip_observations;
    for (int i = 0; i < n; ++i) {
        real v = o[i];
        if (v > clip) v = clip;
        if (v < -clip) v = -clip;
        obs[i] = (float)v;
    }
    for (int i = 0; i < p; ++i) {
        if (pri[i] > (float)clip) pri[i] = (float)clip;
        if (pri[i] < -(float)clip) pri[i] = -(float)clip;
    }
}

static int step_env(struct grx_sim* s, int le, const grx_step_args* args, real stats_sum[NT], int* stats_cnt) {
    const grx_config* c = &s->cfg;
    const grx_model* m = &c->model;
    env_t* e = &s->env[le];
    int nd = s->nd;
    uint32_t step = (uint32_t)args->common_step_counter;
    real dt = c->sim_dt * c->decimation;
    /* clip_actions legged_robot_fftai.py:171-177 */
    for (int j = 0; j < nd; ++j) {
        real a = args->actions ? (real)args->actions[(size_t)le * nd + j] : 0;
        if (a < c->clip_actions_min[j]) a = c->clip_actions_min[j];
        if (a > c->clip_actions_max[j]) a = c->clip_actions_max[j];
        e->actions[j] = a;
    }
    /* before/during_physics_step legged_robot_fftai.py:46-88 */
    for (int f = 0; f < 2; ++f) {
        e->avg_force[f] = 0;
        for (int i = 0; i < 3; ++i) { e->avg_speed[f][i] = 0; e->avg_rpy[f][i] = 0; }
    }
    static __thread kin_t k;
    for (int deci = 0; deci < c->decimation; ++deci) {
        const real* act = ((real)deci < (real)args->delay_substeps) ? e->last_actions : e->actions;
        for (int j = 0; j < nd; ++j) { /* _compute_torques legged_robot.py:679-715 */
            real t = control_torque(c, e, j, act[j]);
            t *= e->motor_strength[j];
            real lim = m->dof_effort[j];
            if (t > lim) t = lim;
            if (t < -lim) t = -lim;
            e->torques[j] = t;
        }
        forward_kinematics(s, e, &k);
        if (substep(s, e, e->torques, &k)) return -1;
        forward_kinematics(s, e, &k); /* refresh_*_tensor: body states AFTER the sub-step */
        named_frames(s, e, &k);
        if (deci == c->decimation - 1 && c->publish_rigid_body_states) link_frames(s, le, &k);   /* before reset_idx, like the reference's tensor */
        for (int f = 0; f < 2; ++f) {
            const real* F = e->feet_force[f];
            e->avg_force[f] += sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]);
            for (int i = 0; i < 3; ++i) {
                e->avg_speed[f][i] += fabs(e->feet_vel[f][i]);
                e->avg_rpy[f][i] += fabs(e->feet_ang[f][i]);
            }
        }
    }
    for (int f = 0; f < 2; ++f) {
        e->avg_force[f] /= c->decimation;
        for (int i = 0; i < 3; ++i) { e->avg_speed[f][i] /= c->decimation; e->avg_rpy[f][i] /= c->decimation; }
    }
    /* post_physics_step legged_robot.py:269-305 */
    e->episode_length += 1;
    /* post_physics_step_update_state legged_robot.py:307-334 */
    real g[3] = {0, 0, -1};
    quat_rotate_inverse(e->quat, e->vel, e->base_lin_vel);
    quat_rotate_inverse(e->quat, e->ang, e->base_ang_vel);
    quat_rotate_inverse(e->quat, g, e->proj_grav);
    if (c->resample_command_interval > 0 && e->episode_length % c->resample_command_interval == 0)
        resample_commands(s, e, le, step, GRO_RNG_CMD_TIME);
    heading_rule(c, e);
    if (c->measure_heights) measure_heights(s, e);
    if (c->push_robots && c->push_interval > 0 && args->common_step_counter % c->push_interval == 0) {
        /* _push_robots legged_robot.py:786-797 */
        e->vel[0] = urand(s, le, step, GRO_RNG_PUSH, 0, -c->max_push_vel_xy, c->max_push_vel_xy);
        e->vel[1] = urand(s, le, step, GRO_RNG_PUSH, 1, -c->max_push_vel_xy, c->max_push_vel_xy);
    }
    /* _calculate_air_time / _feet_height / _land_time legged_robot_fftai.py:108-133 */
    int nh = c->measure_heights ? c->num_height_points : 0;
    for (int f = 0; f < 2; ++f) {
        e->contact[f] = e->feet_force[f][2] > (real)1.0;
        e->contact_filt[f] = e->contact[f] || e->contact_last[f];
        e->contact_last[f] = e->contact[f];
        e->first_contact[f] = (e->air_time[f] > 0) && e->contact_filt[f];
        e->air_time[f] += dt;
        real hs = 0;
        for (int kk = 0; kk < nh; ++kk) hs += e->feet_pos[f][2] - e->heights[kk];
        e->feet_height[f] = nh > 0 ? hs / nh : e->feet_pos[f][2];
        e->land_time[f] = (e->land_time[f] + dt) * (e->contact[f] ? 1 : 0);
    }
    /* check_termination legged_robot.py:336-353 */
    int term = 0;
    {
        int seen[NL_MAX] = {0};
        for (int i = 0; i < m->num_spheres; ++i) {
            int L = m->sph_link[i];
            if (!(m->sph_flags[i] & GRX_SPH_TERMINATE) || seen[L]) continue;
            seen[L] = 1;
            const real* F = e->link_force[L];
            if (sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) > c->termination_force) term = 1;
        }
    }
    e->reset = term || (fabs(e->proj_grav[2]) < c->termination_gravity_z);
    e->time_out = (real)e->episode_length > c->max_episode_length;
    e->reset = e->reset || e->time_out;
    /* compute_reward legged_robot.py:355-375 */
    real r[NT];
    reward_terms(s, e, r);
    e->rew = 0;
    for (int t = 0; t < NT; ++t) {
        e->reward_terms[t] = 0;
        if (t == GRX_REW_TERMINATION || c->reward_scale[t] == 0) continue;
        real rew = r[t] * (c->reward_scale[t] * dt);
        e->reward_terms[t] = rew;
        e->rew += rew;
        e->episode_sums[t] += rew;
    }
    if (c->only_positive_rewards && e->rew < 0) e->rew = 0;
    if (c->reward_scale[GRX_REW_TERMINATION] != 0) {
        real rew = r[GRX_REW_TERMINATION] * (c->reward_scale[GRX_REW_TERMINATION] * dt);
        e->reward_terms[GRX_REW_TERMINATION] = rew;
        e->rew += rew;
        e->episode_sums[GRX_REW_TERMINATION] += rew;
    }
    /* reset_idx legged_robot.py:377-440 */
    if (e->reset) {
        for (int t = 0; t < NT; ++t) { stats_sum[t] += e->episode_sums[t]; e->episode_sums[t] = 0; }
        *stats_cnt += 1;
        reset_env(s, e, le, step, 1);
    }
    /* compute_observations legged_robot.py:442-452 */
    build_observations(s, e, le, args, step);
    /* history legged_robot.py:299-300, legged_robot_fftai.py:94-97 */
    for (int j = 0; j < nd; ++j) {
        e->last_actions[j] = e->actions[j];
        e->last_dof_vel[j] = e->qd[j];
        e->last_last_actions[j] = e->last_actions[j]; /* copied AFTER last_actions was overwritten */
    }
    for (int f = 0; f < 2; ++f) e->air_time[f] = e->air_time[f] * (e->contact_filt[f] ? 0 : 1);
    return 0;
}

static void stats_file(struct grx_sim* s, int cnt) {   /* after a step / reset: terrain-level mean (legged_robot.py:427-428), history row */
    if (cnt > 0) {
        double lv = 0;
        for (int i = 0; i < s->N; ++i) lv += s->env[i].level;
        s->stats[NT + 1] = (float)(lv / s->N);
    }
    ++s->seq;
    memcpy(s->hist[s->seq & (GRX_STATS_HISTORY - 1)], s->stats, sizeof s->stats);
}

int gro_step(grx_handle s, grx_step_args* args, void* stream) {
    (void)stream;
    if (!s || !args) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_step: null argument");
    /* caller may have written episode_length_buf (on_policy_runner.py:126) */
    for (int i = 0; i < s->N; ++i) s->env[i].episode_length = s->t_eplen[i];
    real sum[NT];
    memset(sum, 0, sizeof sum);
    int cnt = 0, bad = 0;
#ifdef _OPENMP
#pragma omp parallel
    {
        real lsum[NT];
        memset(lsum, 0, sizeof lsum);
        int lcnt = 0, lbad = 0;
#pragma omp for schedule(static)
        for (int i = 0; i < s->N; ++i) if (step_env(s, i, args, lsum, &lcnt)) lbad = 1;
#pragma omp critical
        {
            for (int t = 0; t < NT; ++t) sum[t] += lsum[t];
            cnt += lcnt;
            bad |= lbad;
        }
    }
#else
    for (int i = 0; i < s->N; ++i) if (step_env(s, i, args, sum, &cnt)) bad = 1;
#endif
    if (cnt > 0) { /* extras["episode"] legged_robot.py:420-424 */
        for (int t = 0; t < NT; ++t) s->stats[t] = (float)(sum[t] / cnt / s->cfg.max_episode_length_s);
        s->stats[NT] = (float)cnt;
    }
    stats_file(s, cnt);
    args->stats_slot = s->seq & (GRX_STATS_HISTORY - 1);
    args->stats_seq = s->seq;
    s->nsteps++;
    publish(s);
    return bad ? fail(GRX_ERR_INVALID_ARGUMENT, "gro_step: articulated inertia not SPD (diverged state)") : GRX_OK;
}

int gro_reset_all(grx_handle s, void* stream) {
    (void)stream;
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_reset_all: null handle");
    uint32_t step = 0x80000000u + (s->reset_count++);
    real sum[NT];
    memset(sum, 0, sizeof sum);
    for (int i = 0; i < s->N; ++i) {
        env_t* e = &s->env[i];
        for (int t = 0; t < NT; ++t) { sum[t] += e->episode_sums[t]; e->episode_sums[t] = 0; }
        reset_env(s, e, i, step, 0);
        e->reset = 1;
    }
    for (int t = 0; t < NT; ++t) s->stats[t] = (float)(sum[t] / s->N / s->cfg.max_episode_length_s);
    s->stats[NT] = (float)s->N;
    stats_file(s, s->N);
    publish(s);
    return GRX_OK;
}

/* LeggedRobot.reset_idx(env_ids) outside a step (legged_robot.py:377-440); env_ids: host int32[n] here */
int gro_reset_idx(grx_handle s, const int32_t* env_ids, int32_t n, void* stream) {
    (void)stream;
    if (!s || (n > 0 && !env_ids)) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_reset_idx: null argument");
    if (n <= 0) return GRX_OK;
    uint32_t step = 0x80000000u + (s->reset_count++);
    real sum[NT];
    memset(sum, 0, sizeof sum);
    char* flag = (char*)calloc((size_t)s->N, 1);
    int cnt = 0;
    for (int k = 0; k < n; ++k) if (env_ids[k] >= 0 && env_ids[k] < s->N) flag[env_ids[k]] = 1;
    for (int i = 0; i < s->N; ++i) {
        if (!flag[i]) continue;
        env_t* e = &s->env[i];
        for (int t = 0; t < NT; ++t) { sum[t] += e->episode_sums[t]; e->episode_sums[t] = 0; }
        reset_env(s, e, i, step, 1);
        e->reset = 1;
        ++cnt;
    }
    free(flag);
    if (cnt > 0) {
        for (int t = 0; t < NT; ++t) s->stats[t] = (float)(sum[t] / cnt / s->cfg.max_episode_length_s);
        s->stats[NT] = (float)cnt;
    }
    stats_file(s, cnt);
    publish(s);
    return GRX_OK;
}

/* every tensor of the oracle is written by its step (grx_publish_mode has no ON_REFRESH leg in the checker): nothing to do */
int gro_refresh(grx_handle s, int id, void* stream) { (void)stream; (void)id; return s ? GRX_OK : fail(GRX_ERR_INVALID_ARGUMENT, "gro_refresh: null handle"); }
int gro_flush_stats(grx_handle s, void* stream) { (void)stream; return s ? GRX_OK : fail(GRX_ERR_INVALID_ARGUMENT, "gro_flush_stats: null handle"); }

/* c10::div_floor_floating (what torch.div(..., rounding_mode='floor') evaluates in float32) */
static float torch_div_floor(float a, float b) {
    float mod = fmodf(a, b);
    float div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div -= 1.0f;
    if (div == 0.0f) return copysignf(0.0f, a / b);
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.0f;
    return fl;
}

/* ------------------------------------------------------------------ create / tensors */
static void* zalloc(size_t n) { return calloc(n ? n : 1, 1); }

int gro_create(const grx_config* cfg, int device_id, grx_handle* out) {
    (void)device_id;
    if (!cfg || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: null argument");
    if (cfg->abi_version != GRX_ABI_VERSION || cfg->struct_size != (int)sizeof(grx_config))
        return fail(GRX_ERR_ABI_MISMATCH, "gro_create: grx_config ABI mismatch");
    const grx_model* m = &cfg->model;
    if (m->num_bodies < 1 || m->num_bodies > NB_MAX || cfg->num_envs < 1)
        return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: bad sizes");
    for (int b = 1; b < m->num_bodies; ++b)
        if (m->parent[b] < 0 || m->parent[b] >= b) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: parent[b] must be < b");
    struct grx_sim* s = (struct grx_sim*)zalloc(sizeof *s);
    s->cfg = *cfg;
    s->N = cfg->num_envs;
    s->nb = m->num_bodies;
    s->nd = m->num_bodies - 1;
    int N = s->N, nd = s->nd;
    if (cfg->num_obs != 9 + 3 * nd) { free(s); return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: num_obs != 9 + 3*num_dofs"); }
    int nh = cfg->measure_heights ? cfg->num_height_points : 0;
    if (cfg->num_pri_obs != cfg->num_obs + 3 + 1 + 2 + 2 + nh) { free(s); return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: num_pri_obs mismatch"); }
    s->env = (env_t*)zalloc(sizeof(env_t) * N);
    if (cfg->terrain_type == GRX_TERRAIN_HEIGHTFIELD) {
        if (!cfg->height_samples || !cfg->terrain_origins) { free(s->env); free(s); return fail(GRX_ERR_INVALID_ARGUMENT, "gro_create: heightfield terrain needs height_samples and terrain_origins"); }
        size_t n = (size_t)cfg->hf_rows * cfg->hf_cols;
        s->hf = (int16_t*)malloc(n * sizeof(int16_t));
        memcpy(s->hf, cfg->height_samples, n * sizeof(int16_t));
        if (cfg->vertical_faces) trimesh_build(s);
        size_t no = (size_t)cfg->num_terrain_rows * cfg->num_terrain_cols * 3;
        s->torigins = (float*)malloc(no * sizeof(float));
        memcpy(s->torigins, cfg->terrain_origins, no * sizeof(float));
    }
    s->cfg.height_samples = NULL;
    s->cfg.terrain_origins = NULL;
    s->t_obs = (float*)zalloc(sizeof(float) * N * cfg->num_obs);
    s->t_pri = (float*)zalloc(sizeof(float) * N * cfg->num_pri_obs);
    s->t_rew = (float*)zalloc(sizeof(float) * N);
    s->t_reset = (uint8_t*)zalloc(N);
    s->t_timeout = (uint8_t*)zalloc(N);
    s->t_eplen = (int64_t*)zalloc(sizeof(int64_t) * N);
    s->t_rbs = (float*)zalloc(sizeof(float) * (size_t)N * GRX_MAX_LINKS * 13);
    memset(s->t_reset, 1, N); /* reset_buf starts as ones (base_task.py:71) */
    for (int id = 0; id < GRX_NUM_TENSORS; ++id) {
        s->scratch[id] = (float*)zalloc(sizeof(float) * (size_t)N * 256);
        s->scratch_u8[id] = NULL; s->scratch_i32[id] = NULL;
    }
    s->scratch_u8[GRX_T_FEET_CONTACT] = (uint8_t*)zalloc((size_t)N * 2);
    s->scratch_u8[GRX_T_TERM_CONTACT] = (uint8_t*)zalloc((size_t)N);
    s->scratch_i32[GRX_T_TERRAIN_LEVELS] = (int32_t*)zalloc(sizeof(int32_t) * N);
    s->scratch_i32[GRX_T_TERRAIN_TYPES] = (int32_t*)zalloc(sizeof(int32_t) * N);
    int maxlink = 0;
    for (int i = 0; i < m->num_spheres; ++i) if (m->sph_link[i] > maxlink) maxlink = m->sph_link[i];
    s->num_links = maxlink + 1;
    /* per-env constants: origins, domain randomisation (legged_robot.py:538-648, 1060-1064, 1163-1195) */
    for (int i = 0; i < N; ++i) {
        env_t* e = &s->env[i];
        uint32_t ge = (uint32_t)(cfg->env_offset + i);
        e->quat[3] = 1;
        if (cfg->terrain_type == GRX_TERRAIN_HEIGHTFIELD) {
            int max_init = cfg->curriculum ? cfg->max_init_terrain_level : cfg->num_terrain_rows - 1;
            float u = gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_LEVEL, 0);
            e->level = (int)(u * (max_init + 1));
            if (e->level > max_init) e->level = max_init;
            /* torch.div(arange(N), N / num_cols, rounding_mode='floor') evaluates in float32 (legged_robot.py:1177-1180) */
            float per = (float)((double)cfg->total_envs / cfg->num_terrain_cols);
            e->type = (int)torch_div_floor((float)ge, per);
            if (e->type > cfg->num_terrain_cols - 1) e->type = cfg->num_terrain_cols - 1;
            env_origin_from_terrain(s, e);
        } else {
            int ncols = (int)floor(sqrt((double)cfg->total_envs));
            if (ncols < 1) ncols = 1;
            e->origin[0] = cfg->env_spacing * (real)(ge / ncols);
            e->origin[1] = cfg->env_spacing * (real)(ge % ncols);
            e->origin[2] = 0;
        }
        e->friction = 1; /* URDF default shape friction */
        if (cfg->randomize_friction) {
            uint32_t b = (uint32_t)(gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_DR, 0) * 64);
            if (b > 63) b = 63;
            e->friction = cfg->friction_range[0] + (cfg->friction_range[1] - cfg->friction_range[0]) * gro_rand(cfg->seed, b, 1, GRO_RNG_INIT_DR, 0);
        }
        if (cfg->randomize_restitution) {
            uint32_t b = (uint32_t)(gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_DR, 1) * 64);
            if (b > 63) b = 63;
            e->restitution = cfg->restitution_range[0] + (cfg->restitution_range[1] - cfg->restitution_range[0]) * gro_rand(cfg->seed, b, 1, GRO_RNG_INIT_DR, 1);
        }
        e->base_link_mass = m->base_link_mass;
        for (int k2 = 0; k2 < 3; ++k2) e->base_link_com[k2] = m->base_link_com[k2];
        if (cfg->randomize_base_mass)
            e->base_link_mass *= cfg->base_mass_range[0] + (cfg->base_mass_range[1] - cfg->base_mass_range[0]) * gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_DR, 2);
        if (cfg->randomize_base_com)
            for (int k2 = 0; k2 < 3; ++k2)
                e->base_link_com[k2] += cfg->base_com_range[k2][0] + (cfg->base_com_range[k2][1] - cfg->base_com_range[k2][0]) * gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_DR, 3 + k2);
        base_lump(m, e);
        for (int j = 0; j < nd; ++j) {
            e->motor_strength[j] = 1;
            if (cfg->randomize_motor_strength)
                e->motor_strength[j] = cfg->motor_strength_range[0] + (cfg->motor_strength_range[1] - cfg->motor_strength_range[0]) * gro_rand(cfg->seed, ge, 0, GRO_RNG_INIT_DR, 8 + j);
        }
        for (int j = 0; j < nd; ++j) e->q[j] = cfg->default_dof_pos[j];
        for (int k2 = 0; k2 < 3; ++k2) e->pos[k2] = cfg->init_pos[k2] + e->origin[k2];
        e->reset = 1;
    }
    publish(s);
    *out = s;
    return GRX_OK;
}

int gro_destroy(grx_handle s) {
    if (!s) return GRX_OK;
    for (int id = 0; id < GRX_NUM_TENSORS; ++id) { free(s->scratch[id]); free(s->scratch_u8[id]); free(s->scratch_i32[id]); }
    free(s->t_obs); free(s->t_pri); free(s->t_rew); free(s->t_reset); free(s->t_timeout); free(s->t_eplen); free(s->t_rbs);
    free(s->hf); free(s->tm_cells); free(s->tm_walls); free(s->torigins); free(s->env); free(s);
    return GRX_OK;
}

#define PUB(id, k, expr) do { float* d_ = s->scratch[id]; for (int i = 0; i < N; ++i) { const env_t* e = &s->env[i]; for (int j = 0; j < (k); ++j) d_[(size_t)i * (k) + j] = (float)(expr); } } while (0)

static void publish(struct grx_sim* s) {
    int N = s->N, nd = s->nd, nh = s->cfg.measure_heights ? s->cfg.num_height_points : 0;
    for (int i = 0; i < N; ++i) {
        const env_t* e = &s->env[i];
        s->t_rew[i] = (float)e->rew;
        s->t_reset[i] = (uint8_t)e->reset;
        s->t_timeout[i] = (uint8_t)e->time_out;
        s->t_eplen[i] = e->episode_length;
        s->scratch_u8[GRX_T_FEET_CONTACT][2 * i] = (uint8_t)e->contact[0];
        s->scratch_u8[GRX_T_FEET_CONTACT][2 * i + 1] = (uint8_t)e->contact[1];
        s->scratch_i32[GRX_T_TERRAIN_LEVELS][i] = e->level;
        s->scratch_i32[GRX_T_TERRAIN_TYPES][i] = e->type;
        float* root = s->scratch[GRX_T_ROOT_STATES] + (size_t)i * 13;
        for (int k = 0; k < 3; ++k) { root[k] = (float)e->pos[k]; root[7 + k] = (float)e->vel[k]; root[10 + k] = (float)e->ang[k]; }
        for (int k = 0; k < 4; ++k) root[3 + k] = (float)e->quat[k];
        float* an = s->scratch[GRX_T_ANCHORS] + (size_t)i * NFS * 3;
        for (int k = 0; k < NFS; ++k) { an[3 * k] = (float)e->anchor[k][0]; an[3 * k + 1] = (float)e->anchor[k][1]; an[3 * k + 2] = e->anchor_on[k] ? (float)(e->anchor_vimp[k] > (real)1e-6 ? e->anchor_vimp[k] : (real)1e-6) : 0.f; }
        for (int t = 0; t < NT; ++t) {
            s->scratch[GRX_T_EPISODE_SUMS][(size_t)t * N + i] = (float)e->episode_sums[t];
            s->scratch[GRX_T_REWARD_TERMS][(size_t)t * N + i] = (float)e->reward_terms[t];
        }
        float* bm = s->scratch[GRX_T_BASE_MASS_COM] + (size_t)i * 4;
        bm[0] = (float)e->base_link_mass;
        for (int k = 0; k < 3; ++k) bm[1 + k] = (float)e->base_link_com[k];
        s->scratch[GRX_T_FRICTION][i] = (float)e->friction;
        s->scratch[GRX_T_BASE_HEIGHTS_OFFSET][i] = (float)e->base_heights_offset;
        int tc = 0;
        const grx_model* m = &s->cfg.model;
        for (int k = 0; k < m->num_spheres; ++k) if (m->sph_flags[k] & GRX_SPH_TERMINATE) {
            const real* F = e->link_force[m->sph_link[k]];
            if (sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) > s->cfg.termination_force) tc = 1;
        }
        s->scratch_u8[GRX_T_TERM_CONTACT][i] = (uint8_t)tc;
    }
    PUB(GRX_T_DOF_POS, nd, e->q[j]);
    PUB(GRX_T_DOF_VEL, nd, e->qd[j]);
    PUB(GRX_T_TORQUES, nd, e->torques[j]);
    PUB(GRX_T_ACTIONS, nd, e->actions[j]);
    PUB(GRX_T_LAST_ACTIONS, nd, e->last_actions[j]);
    PUB(GRX_T_LAST_DOF_VEL, nd, e->last_dof_vel[j]);
    PUB(GRX_T_COMMANDS, 3, e->commands[j]);
    PUB(GRX_T_BASE_LIN_VEL, 3, e->base_lin_vel[j]);
    PUB(GRX_T_BASE_ANG_VEL, 3, e->base_ang_vel[j]);
    PUB(GRX_T_PROJECTED_GRAVITY, 3, e->proj_grav[j]);
    PUB(GRX_T_FEET_CONTACT_FORCE, 6, e->feet_force[j / 3][j % 3]);
    PUB(GRX_T_FEET_POS, 6, e->feet_pos[j / 3][j % 3]);
    PUB(GRX_T_CONTACT_FORCES, 3 * GRX_MAX_LINKS, e->link_force[j / 3][j % 3]);   /* legged_robot.py:117 contact_forces */
    PUB(GRX_T_FEET_HEIGHT, 2, e->feet_height[j]);
    PUB(GRX_T_FEET_AIR_TIME, 2, e->air_time[j]);
    PUB(GRX_T_FEET_LAND_TIME, 2, e->land_time[j]);
    PUB(GRX_T_AVG_FEET_FORCE, 2, e->avg_force[j]);
    PUB(GRX_T_AVG_FEET_SPEED, 6, e->avg_speed[j / 3][j % 3]);
    PUB(GRX_T_AVG_FEET_SPEED_RPY, 6, e->avg_rpy[j / 3][j % 3]);   /* legged_robot_fftai.py:81, 88 */
    PUB(GRX_T_MEASURED_HEIGHTS, nh, e->heights[j]);
    PUB(GRX_T_ENV_ORIGINS, 3, e->origin[j]);
    PUB(GRX_T_MOTOR_STRENGTH, nd, e->motor_strength[j]);
    memcpy(s->scratch[GRX_T_EPISODE_STATS], s->stats, sizeof s->stats);
}

static void desc_set(grx_tensor_desc* d, void* p, int dtype, int ndim, int64_t a, int64_t b, int64_t c3) {
    d->data = p; d->dtype = dtype; d->ndim = ndim;
    d->shape[0] = a; d->shape[1] = b; d->shape[2] = c3; d->shape[3] = 1;
    int64_t dims[4] = {a, b, c3, 1};
    int64_t st = 1;
    for (int i = ndim - 1; i >= 0; --i) { d->stride[i] = st; st *= dims[i]; }
    for (int i = ndim; i < 4; ++i) { d->stride[i] = 1; d->shape[i] = 1; }
}

int gro_tensor(grx_handle s, int id, grx_tensor_desc* d) {
    if (!s || !d) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_tensor: null argument");
    int N = s->N, nd = s->nd, nh = s->cfg.measure_heights ? s->cfg.num_height_points : 0;
    switch (id) {
    case GRX_T_OBS: desc_set(d, s->t_obs, GRX_F32, 2, N, s->cfg.num_obs, 1); break;
    case GRX_T_PRI_OBS: desc_set(d, s->t_pri, GRX_F32, 2, N, s->cfg.num_pri_obs, 1); break;
    case GRX_T_REW: desc_set(d, s->t_rew, GRX_F32, 1, N, 1, 1); break;
    case GRX_T_RESET: desc_set(d, s->t_reset, GRX_U8, 1, N, 1, 1); break;
    case GRX_T_TIME_OUT: desc_set(d, s->t_timeout, GRX_U8, 1, N, 1, 1); break;
    case GRX_T_EPISODE_LENGTH: desc_set(d, s->t_eplen, GRX_I64, 1, N, 1, 1); break;
    case GRX_T_DOF_POS: case GRX_T_DOF_VEL: case GRX_T_TORQUES: case GRX_T_ACTIONS:
    case GRX_T_LAST_ACTIONS: case GRX_T_LAST_DOF_VEL: case GRX_T_MOTOR_STRENGTH:
        desc_set(d, s->scratch[id], GRX_F32, 2, N, nd, 1); break;
    case GRX_T_COMMANDS: case GRX_T_BASE_LIN_VEL: case GRX_T_BASE_ANG_VEL: case GRX_T_PROJECTED_GRAVITY:
    case GRX_T_ENV_ORIGINS:
        desc_set(d, s->scratch[id], GRX_F32, 2, N, 3, 1); break;
    case GRX_T_ROOT_STATES: desc_set(d, s->scratch[id], GRX_F32, 2, N, 13, 1); break;
    case GRX_T_FEET_CONTACT_FORCE: case GRX_T_FEET_POS: case GRX_T_AVG_FEET_SPEED: case GRX_T_AVG_FEET_SPEED_RPY:
        desc_set(d, s->scratch[id], GRX_F32, 3, N, 2, 3); break;
    case GRX_T_FEET_HEIGHT: case GRX_T_FEET_AIR_TIME: case GRX_T_FEET_LAND_TIME: case GRX_T_AVG_FEET_FORCE:
        desc_set(d, s->scratch[id], GRX_F32, 2, N, 2, 1); break;
    case GRX_T_FEET_CONTACT: desc_set(d, s->scratch_u8[id], GRX_U8, 2, N, 2, 1); break;
    case GRX_T_MEASURED_HEIGHTS: desc_set(d, s->scratch[id], GRX_F32, 2, N, nh, 1); break;
    case GRX_T_BASE_HEIGHTS_OFFSET: case GRX_T_FRICTION: desc_set(d, s->scratch[id], GRX_F32, 1, N, 1, 1); break;
    case GRX_T_EPISODE_SUMS: case GRX_T_REWARD_TERMS: desc_set(d, s->scratch[id], GRX_F32, 2, NT, N, 1); break;
    case GRX_T_TERRAIN_LEVELS: case GRX_T_TERRAIN_TYPES: desc_set(d, s->scratch_i32[id], GRX_I32, 1, N, 1, 1); break;
    case GRX_T_BASE_MASS_COM: desc_set(d, s->scratch[id], GRX_F32, 2, N, 4, 1); break;
    case GRX_T_TERM_CONTACT: desc_set(d, s->scratch_u8[id], GRX_U8, 1, N, 1, 1); break;
    case GRX_T_EPISODE_STATS: desc_set(d, s->scratch[id], GRX_F32, 1, NT + 2, 1, 1); break;
    case GRX_T_EPISODE_STATS_HISTORY: desc_set(d, &s->hist[0][0], GRX_F32, 2, GRX_STATS_HISTORY, NT + 2, 1); break;
    case GRX_T_ANCHORS: desc_set(d, s->scratch[id], GRX_F32, 3, N, NFS, 3); break;
    case GRX_T_CONTACT_FORCES: desc_set(d, s->scratch[id], GRX_F32, 3, N, GRX_MAX_LINKS, 3); break;
    case GRX_T_RIGID_BODY_STATES: desc_set(d, s->t_rbs, GRX_F32, 3, N, GRX_MAX_LINKS, 13); break;
    default: return fail(GRX_ERR_INVALID_ARGUMENT, "gro_tensor: unknown tensor id");
    }
    return GRX_OK;
}

static int set_state_rows(grx_handle s, const int32_t* env_ids, int n, const float* root, const float* q, const float* qd);
int gro_set_state(grx_handle s, const float* root, const float* q, const float* qd, void* stream) {
    (void)stream;
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_set_state: null handle");
    return set_state_rows(s, NULL, s->N, root, q, qd);
}
int gro_set_state_indexed(grx_handle s, const int32_t* env_ids, int32_t n, const float* root, const float* q, const float* qd, void* stream) {
    (void)stream;
    if (!s || (n > 0 && !env_ids)) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_set_state_indexed: null argument");
    return n > 0 ? set_state_rows(s, env_ids, n, root, q, qd) : GRX_OK;
}
static int set_state_rows(grx_handle s, const int32_t* env_ids, int n, const float* root, const float* q, const float* qd) {
    for (int k = 0; k < n; ++k) {
        const int i = env_ids ? env_ids[k] : k;
        if (i < 0 || i >= s->N) continue;
        env_t* e = &s->env[i];
        if (root) {
            const float* r = root + (size_t)i * 13;
            for (int k = 0; k < 3; ++k) { e->pos[k] = r[k]; e->vel[k] = r[7 + k]; e->ang[k] = r[10 + k]; }
            for (int k = 0; k < 4; ++k) e->quat[k] = r[3 + k];
            real n = sqrt(e->quat[0] * e->quat[0] + e->quat[1] * e->quat[1] + e->quat[2] * e->quat[2] + e->quat[3] * e->quat[3]);
            for (int k = 0; k < 4; ++k) e->quat[k] /= n; /* callers pass unit quaternions up to fp32 rounding */
        }
        if (q) for (int j = 0; j < s->nd; ++j) e->q[j] = q[(size_t)i * s->nd + j];
        if (qd) for (int j = 0; j < s->nd; ++j) e->qd[j] = qd[(size_t)i * s->nd + j];
        for (int k = 0; k < NFS; ++k) e->anchor_on[k] = 0;
    }
    publish(s);
    return GRX_OK;
}

int gro_episode_stats(grx_handle s, float* host_out, void* stream) {
    (void)stream;
    if (!s || !host_out) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_episode_stats: null argument");
    memcpy(host_out, s->stats, sizeof s->stats);
    return GRX_OK;
}

/* TEST SUPPORT: the inverse of publish() for the simulation STATE -- what tests/helpers.STATE_TENSORS lists, i.e. exactly what the HIP
 * library keeps between two steps.  The float32 views handed out by gro_tensor are OUTPUTS of the oracle (its state lives in `real`
 * env_t records); a test that wants the oracle to continue from a state it wrote into those views (the perturbed twins of
 * tests/test_hip_parity.py: the oracle's own sensitivity to a 1e-6 nudge, per env) calls this first. */
int gro_debug_import_state(grx_handle s) {
    if (!s) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_debug_import_state: null handle");
    const int N = s->N, nd = s->nd;
    for (int i = 0; i < N; ++i) {
        env_t* e = &s->env[i];
        for (int j = 0; j < nd; ++j) {
            e->q[j] = s->scratch[GRX_T_DOF_POS][(size_t)i * nd + j];
            e->qd[j] = s->scratch[GRX_T_DOF_VEL][(size_t)i * nd + j];
            e->last_actions[j] = s->scratch[GRX_T_LAST_ACTIONS][(size_t)i * nd + j];
            e->last_last_actions[j] = e->last_actions[j];   /* legged_robot_fftai.py:94: equal after every step */
            e->last_dof_vel[j] = s->scratch[GRX_T_LAST_DOF_VEL][(size_t)i * nd + j];
        }
        const float* root = s->scratch[GRX_T_ROOT_STATES] + (size_t)i * 13;
        for (int k = 0; k < 3; ++k) { e->pos[k] = root[k]; e->vel[k] = root[7 + k]; e->ang[k] = root[10 + k]; }
        for (int k = 0; k < 4; ++k) e->quat[k] = root[3 + k];
        const float* an = s->scratch[GRX_T_ANCHORS] + (size_t)i * NFS * 3;
        for (int k = 0; k < NFS; ++k) {
            e->anchor[k][0] = an[3 * k]; e->anchor[k][1] = an[3 * k + 1];
            e->anchor_on[k] = an[3 * k + 2] > 0.f; e->anchor_vimp[k] = an[3 * k + 2] > 0.f ? an[3 * k + 2] : 0;
        }
        for (int k = 0; k < 3; ++k) { e->commands[k] = s->scratch[GRX_T_COMMANDS][(size_t)i * 3 + k]; e->origin[k] = s->scratch[GRX_T_ENV_ORIGINS][(size_t)i * 3 + k]; }
        for (int f = 0; f < 2; ++f) {
            e->air_time[f] = s->scratch[GRX_T_FEET_AIR_TIME][(size_t)i * 2 + f];
            e->land_time[f] = s->scratch[GRX_T_FEET_LAND_TIME][(size_t)i * 2 + f];
            e->contact[f] = e->contact_last[f] = s->scratch_u8[GRX_T_FEET_CONTACT][2 * i + f];
        }
        e->base_heights_offset = s->scratch[GRX_T_BASE_HEIGHTS_OFFSET][i];
        e->episode_length = s->t_eplen[i];
        for (int t = 0; t < NT; ++t) e->episode_sums[t] = s->scratch[GRX_T_EPISODE_SUMS][(size_t)t * N + i];
        e->level = s->scratch_i32[GRX_T_TERRAIN_LEVELS][i];
    }
    return GRX_OK;
}

const char* gro_last_error(void) { return g_err; }
int gro_abi_version(void) { return GRX_ABI_VERSION; }
int gro_stats_seq(grx_handle s, int64_t* out) { if (!s || !out) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_stats_seq: null argument"); *out = s->seq; return GRX_OK; }
int gro_real_size(void) { return (int)sizeof(real); }

static const char* k_term_names[NT] = {
    "action_diff", "action_diff_diff", "action_diff_knee", "cmd_diff_ang_vel_pitch", "cmd_diff_ang_vel_roll",
    "cmd_diff_ang_vel_yaw", "cmd_diff_base_height", "cmd_diff_base_orient", "cmd_diff_forehead_orient",
    "cmd_diff_lin_vel_x", "cmd_diff_lin_vel_y", "cmd_diff_lin_vel_z", "cmd_diff_torso_orient", "collision",
    "dof_acc_new", "dof_tor_ankle_feet_lift_up", "dof_tor_new", "dof_tor_new_hip_roll", "dof_vel_new",
    "dof_vel_new_knee", "feet_air_force", "feet_air_height", "feet_air_time", "feet_land_time",
    "feet_speed_xy_close_to_ground", "feet_speed_z_close_to_height_target", "feet_stumble", "limits_actions",
    "limits_dof_pos", "limits_dof_tor", "limits_dof_vel", "on_the_air", "pose_offset", "pose_offset_hip_yaw",
    "stand_still", "termination"};
const char* gro_reward_term_name(int t) { return (t >= 0 && t < NT) ? k_term_names[t] : ""; }

/* ------------------------------------------------------------------ debug hooks (physics pins) */
/* Overwrite the env-pipeline state of env `le` (golden-vector tests drive the pipeline stages
 * from arbitrary synthetic state exactly as tools/gen_golden.py drives the reference). */
typedef struct gro_pipeline_state {
    float q[ND_MAX], qd[ND_MAX], root[13];
    float actions[ND_MAX], last_actions[ND_MAX], last_last_actions[ND_MAX], last_dof_vel[ND_MAX], torques[ND_MAX];
    float commands[3];
    float air_time[2], land_time[2];
    int32_t contact_last[2];
    float feet_force[2][3], feet_pos[2][3];
    float avg_force[2], avg_speed[2][3];
    float torso_R[9];
    float heights[GRX_MAX_HEIGHT_POINTS];
    float base_heights_offset;
    int64_t episode_length;
    int32_t term_contact; /* pretend a terminating link carries |F| > threshold */
} gro_pipeline_state;

/* Runs post_physics_step (everything after the sub-step loop) on injected state; no physics, no reset
 * (reset is reported in RESET but not applied when apply_reset == 0). */
int gro_debug_post_physics(grx_handle s, int le, const gro_pipeline_state* ps, int apply_reset,
                           const grx_step_args* args) {
    if (!s || !ps || le < 0 || le >= s->N) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_debug_post_physics: bad argument");
    const grx_config* c = &s->cfg;
    const grx_model* m = &c->model;
    env_t* e = &s->env[le];
    int nd = s->nd;
    real dt = c->sim_dt * c->decimation;
    for (int j = 0; j < nd; ++j) {
        e->q[j] = ps->q[j]; e->qd[j] = ps->qd[j]; e->actions[j] = ps->actions[j];
        e->last_actions[j] = ps->last_actions[j]; e->last_last_actions[j] = ps->last_last_actions[j];
        e->last_dof_vel[j] = ps->last_dof_vel[j]; e->torques[j] = ps->torques[j];
    }
    for (int k = 0; k < 3; ++k) { e->pos[k] = ps->root[k]; e->vel[k] = ps->root[7 + k]; e->ang[k] = ps->root[10 + k]; e->commands[k] = ps->commands[k]; }
    for (int k = 0; k < 4; ++k) e->quat[k] = ps->root[3 + k];
    for (int f = 0; f < 2; ++f) {
        e->air_time[f] = ps->air_time[f]; e->land_time[f] = ps->land_time[f]; e->contact_last[f] = ps->contact_last[f];
        e->avg_force[f] = ps->avg_force[f];
        for (int k = 0; k < 3; ++k) { e->feet_force[f][k] = ps->feet_force[f][k]; e->feet_pos[f][k] = ps->feet_pos[f][k]; e->avg_speed[f][k] = ps->avg_speed[f][k]; }
    }
    for (int k = 0; k < 9; ++k) e->torso_quat_R[k] = ps->torso_R[k];
    e->base_heights_offset = ps->base_heights_offset;
    e->episode_length = ps->episode_length;
    memset(e->link_force, 0, sizeof e->link_force);
    if (ps->term_contact)
        for (int i = 0; i < m->num_spheres; ++i) if (m->sph_flags[i] & GRX_SPH_TERMINATE) { e->link_force[m->sph_link[i]][2] = 10; break; }
    uint32_t step = (uint32_t)args->common_step_counter;
    /* --- same sequence as step_env after the sub-step loop --- */
    e->episode_length += 1;
    real g[3] = {0, 0, -1};
    quat_rotate_inverse(e->quat, e->vel, e->base_lin_vel);
    quat_rotate_inverse(e->quat, e->ang, e->base_ang_vel);
    quat_rotate_inverse(e->quat, g, e->proj_grav);
    heading_rule(c, e);   /* (the time-based resample itself is left out here: its draws are not the reference's) */
    int nh = c->measure_heights ? c->num_height_points : 0;
    if (c->measure_heights) {
        if (c->terrain_type == GRX_TERRAIN_PLANE) for (int k = 0; k < nh; ++k) e->heights[k] = ps->heights[k];
        else measure_heights(s, e);
    }
    for (int f = 0; f < 2; ++f) {
        e->contact[f] = e->feet_force[f][2] > (real)1.0;
        e->contact_filt[f] = e->contact[f] || e->contact_last[f];
        e->contact_last[f] = e->contact[f];
        e->first_contact[f] = (e->air_time[f] > 0) && e->contact_filt[f];
        e->air_time[f] += dt;
        real hs = 0;
        for (int kk = 0; kk < nh; ++kk) hs += e->feet_pos[f][2] - e->heights[kk];
        e->feet_height[f] = nh > 0 ? hs / nh : e->feet_pos[f][2];
        e->land_time[f] = (e->land_time[f] + dt) * (e->contact[f] ? 1 : 0);
    }
    int term = ps->term_contact ? 1 : 0;
    e->reset = term || (fabs(e->proj_grav[2]) < c->termination_gravity_z);
    e->time_out = (real)e->episode_length > c->max_episode_length;
    e->reset = e->reset || e->time_out;
    real r[NT];
    reward_terms(s, e, r);
    e->rew = 0;
    for (int t = 0; t < NT; ++t) {
        e->reward_terms[t] = 0;
        if (t == GRX_REW_TERMINATION || c->reward_scale[t] == 0) continue;
        real rew = r[t] * (c->reward_scale[t] * dt);
        e->reward_terms[t] = rew; e->rew += rew; e->episode_sums[t] += rew;
    }
    if (c->only_positive_rewards && e->rew < 0) e->rew = 0;
    if (c->reward_scale[GRX_REW_TERMINATION] != 0) {
        real rew = r[GRX_REW_TERMINATION] * (c->reward_scale[GRX_REW_TERMINATION] * dt);
        e->reward_terms[GRX_REW_TERMINATION] = rew; e->rew += rew; e->episode_sums[GRX_REW_TERMINATION] += rew;
    }
    if (e->reset && apply_reset) reset_env(s, e, le, step, 1);
    build_observations(s, e, le, args, step);
    for (int j = 0; j < nd; ++j) { e->last_actions[j] = e->actions[j]; e->last_dof_vel[j] = e->qd[j]; e->last_last_actions[j] = e->last_actions[j]; }
    for (int f = 0; f < 2; ++f) e->air_time[f] = e->air_time[f] * (e->contact_filt[f] ? 0 : 1);
    publish(s);
    return GRX_OK;
}

/* physics terrain query at world (x, y): out = {height, dh/dx, dh/dy} */
int gro_debug_terrain(grx_handle s, double x, double y, double* out) {
    real g[2];
    out[0] = terrain_query(s, (real)x, (real)y, g);
    out[1] = g[0]; out[2] = g[1];
    return GRX_OK;
}

/* mesh_type 'trimesh': the per-cell tables of trimesh_build -- ground int16[rows * cols][6], walls int16[rows * cols][8] */
int gro_debug_trimesh_tables(grx_handle s, int16_t* ground, int16_t* walls) {
    if (!s || !s->tm_cells || !s->tm_walls) return fail(GRX_ERR_INVALID_ARGUMENT, "gro_debug_trimesh_tables: not a trimesh handle");
    const size_t n = (size_t)s->cfg.hf_rows * s->cfg.hf_cols;
    memcpy(ground, s->tm_cells, 6 * n * sizeof(int16_t));
    memcpy(walls, s->tm_walls, 8 * n * sizeof(int16_t));
    return GRX_OK;
}

/* mesh_type 'trimesh': overlap of a sphere (centre x y z, radius r) with the vertical faces next to it: out = {overlap, nx, ny, nz} (0 0 0 0: none) */
int gro_debug_wall(grx_handle s, double x, double y, double z, double r, double* out) {
    real n[3] = {0, 0, 0}, c[3] = {(real)x, (real)y, (real)z};
    real d = 0;
    if (s->cfg.terrain_type == GRX_TERRAIN_HEIGHTFIELD && s->cfg.vertical_faces) d = wall_overlap(s, c, (real)r, n);
    out[0] = d > 0 ? d : 0;
    for (int j = 0; j < 3; ++j) out[1 + j] = d > 0 ? n[j] : 0;
    return GRX_OK;
}

/* unscaled reward terms of env le as last evaluated by reward_terms() on its current state */
int gro_debug_reward_terms(grx_handle s, int le, float* out) {
    real r[NT];
    reward_terms(s, &s->env[le], r);
    for (int t = 0; t < NT; ++t) out[t] = (float)r[t];
    return GRX_OK;
}

/* torques for (actions) from the current state, no physics: _compute_torques + clip_actions */
int gro_debug_torques(grx_handle s, const float* actions, float* clipped, float* torques) {
    const grx_config* c = &s->cfg;
    for (int i = 0; i < s->N; ++i) {
        const env_t* e = &s->env[i];
        for (int j = 0; j < s->nd; ++j) {
            real a = actions[(size_t)i * s->nd + j];
            if (a < c->clip_actions_min[j]) a = c->clip_actions_min[j];
            if (a > c->clip_actions_max[j]) a = c->clip_actions_max[j];
            clipped[(size_t)i * s->nd + j] = (float)a;
            real t = control_torque(c, e, j, a);
            t *= e->motor_strength[j];
            real lim = c->model.dof_effort[j];
            if (t > lim) t = lim;
            if (t < -lim) t = -lim;
            torques[(size_t)i * s->nd + j] = (float)t;
        }
    }
    return GRX_OK;
}

/* Forward dynamics of env le for given joint torques, no contact: returns qdd (nd) and the base
 * classical acceleration in world frame (lin 3, ang 3). */
int gro_debug_forward_dynamics(grx_handle s, int le, const double* tau_in, int with_contact, double* qdd_out, double* base_acc_out) {
    env_t* e = &s->env[le];
    static __thread kin_t k;
    forward_kinematics(s, e, &k);
    sv6 fext[NB_MAX];
    if (with_contact) contact_forces(s, e, &k, fext);
    else for (int b = 0; b < s->nb; ++b) memset(&fext[b], 0, sizeof(sv6));
    real tau[ND_MAX], qdd[ND_MAX];
    for (int j = 0; j < s->nd; ++j) tau[j] = (real)tau_in[j];
    sv6 a0;
    if (aba(s, e, &k, tau, fext, qdd, &a0)) return fail(GRX_ERR_INVALID_ARGUMENT, "aba failed");
    for (int j = 0; j < s->nd; ++j) qdd_out[j] = qdd[j];
    real wxv[3], al[3], alw[3], aw[3];
    v3_cross(k.v[0].v, k.v[0].v + 3, wxv);
    for (int i = 0; i < 3; ++i) al[i] = a0.v[3 + i] + wxv[i];
    m3_mulv(k.R[0], al, alw);
    m3_mulv(k.R[0], a0.v, aw);
    for (int i = 0; i < 3; ++i) { base_acc_out[i] = alw[i] + s->cfg.gravity[i]; base_acc_out[3 + i] = aw[i]; }
    return GRX_OK;
}

/* Recursive Newton-Euler inverse dynamics (RBDA Table 5.1, floating base as a 6-DOF joint):
 * given qdd and the base acceleration (world, classical), returns joint torques (nd) and the
 * residual wrench on the base (6, body coords) which must vanish for a free-floating base.
 * Independent of aba(): tests assert ID(FD(tau)) == tau. */
int gro_debug_inverse_dynamics(grx_handle s, int le, const double* qdd_in, const double* base_acc_in, double* tau_out, double* base_wrench_out) {
    const grx_model* m = &s->cfg.model;
    env_t* e = &s->env[le];
    static __thread kin_t k;
    forward_kinematics(s, e, &k);
    sv6 a[NB_MAX], f[NB_MAX];
    /* base spatial acceleration in body coords, gravity folded in as a fictitious acceleration */
    real lw[3], awv[3], lb[3], ab[3], wxv[3];
    for (int i = 0; i < 3; ++i) { lw[i] = (real)base_acc_in[i] - s->cfg.gravity[i]; awv[i] = (real)base_acc_in[3 + i]; }
    m3_tmulv(k.R[0], lw, lb);
    m3_tmulv(k.R[0], awv, ab);
    v3_cross(k.v[0].v, k.v[0].v + 3, wxv);
    for (int i = 0; i < 3; ++i) { a[0].v[i] = ab[i]; a[0].v[3 + i] = lb[i] - wxv[i]; }
    for (int b = 0; b < s->nb; ++b) {
        sm6 I;
        if (b == 0) s_rigid_inertia(e->base_m, e->base_c, e->base_I, &I);
        else {
            real cc[3] = {m->com[b][0], m->com[b][1], m->com[b][2]}, I6[6];
            for (int i = 0; i < 6; ++i) I6[i] = m->inertia[b][i];
            s_rigid_inertia(m->mass[b], cc, I6, &I);
            sv6 Sq, cq;
            memset(&Sq, 0, sizeof Sq);
            for (int i = 0; i < 3; ++i) Sq.v[i] = m->joint_axis[b][i] * e->qd[b - 1];
            s_crm(&k.v[b], &Sq, &cq);
            sm_mulv(&k.X[b], &a[m->parent[b]], &a[b]);
            for (int i = 0; i < 6; ++i) a[b].v[i] += cq.v[i];
            for (int i = 0; i < 3; ++i) a[b].v[i] += m->joint_axis[b][i] * (real)qdd_in[b - 1];
        }
        sv6 Ia, Iv, t;
        sm_mulv(&I, &a[b], &Ia);
        sm_mulv(&I, &k.v[b], &Iv);
        s_crf(&k.v[b], &Iv, &t);
        for (int i = 0; i < 6; ++i) f[b].v[i] = Ia.v[i] + t.v[i];
    }
    for (int b = s->nb - 1; b >= 1; --b) {
        real tq = m->dof_armature[b - 1] * (real)qdd_in[b - 1];
        for (int i = 0; i < 3; ++i) tq += m->joint_axis[b][i] * f[b].v[i];
        tau_out[b - 1] = tq;
        sv6 fp;
        sm_tmulv(&k.X[b], &f[b], &fp);
        for (int i = 0; i < 6; ++i) f[m->parent[b]].v[i] += fp.v[i];
    }
    for (int i = 0; i < 6; ++i) base_wrench_out[i] = f[0].v[i];
    return GRX_OK;
}

/* total mechanical energy (kinetic + gravitational potential) and linear momentum (world) of env le */
int gro_debug_energy(grx_handle s, int le, double* out /* [KE, PE, px, py, pz, mass] */) {
    const grx_model* m = &s->cfg.model;
    env_t* e = &s->env[le];
    static __thread kin_t k;
    forward_kinematics(s, e, &k);
    double KE = 0, PE = 0, P[3] = {0, 0, 0}, M = 0;
    for (int b = 0; b < s->nb; ++b) {
        sm6 I;
        real mass, cc[3];
        if (b == 0) { s_rigid_inertia(e->base_m, e->base_c, e->base_I, &I); mass = e->base_m; for (int i = 0; i < 3; ++i) cc[i] = e->base_c[i]; }
        else {
            real I6[6];
            for (int i = 0; i < 3; ++i) cc[i] = m->com[b][i];
            for (int i = 0; i < 6; ++i) I6[i] = m->inertia[b][i];
            mass = m->mass[b];
            s_rigid_inertia(mass, cc, I6, &I);
        }
        sv6 Iv;
        sm_mulv(&I, &k.v[b], &Iv);
        double ke = 0;
        for (int i = 0; i < 6; ++i) ke += 0.5 * (double)k.v[b].v[i] * (double)Iv.v[i];
        KE += ke;
        real cw[3], pw[3];
        m3_mulv(k.R[b], cc, cw);
        double hz = (double)k.p[b][2] + cw[2];
        PE += -(double)mass * s->cfg.gravity[2] * hz;
        m3_mulv(k.R[b], Iv.v + 3, pw); /* linear momentum = linear part of I v */
        for (int i = 0; i < 3; ++i) P[i] += pw[i];
        M += mass;
    }
    out[0] = KE; out[1] = PE; out[2] = P[0]; out[3] = P[1]; out[4] = P[2]; out[5] = M;
    return GRX_OK;
}

/* n raw sub-steps with constant joint torques (no env pipeline): integrator-level tests */
int gro_debug_substeps(grx_handle s, int le, const double* tau_in, int n, int contact_on) {
    env_t* e = &s->env[le];
    static __thread kin_t k;
    real tau[ND_MAX];
    for (int j = 0; j < s->nd; ++j) tau[j] = (real)tau_in[j];
    float saved_kn = s->cfg.contact.kn;
    if (!contact_on) s->cfg.contact.kn = 0;
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        forward_kinematics(s, e, &k);
        rc = substep(s, e, tau, &k);
    }
    s->cfg.contact.kn = saved_kn;
    forward_kinematics(s, e, &k);
    named_frames(s, e, &k);
    publish(s);
    return rc ? fail(GRX_ERR_INVALID_ARGUMENT, "substep failed") : GRX_OK;
}

/* world pose of body b of env le: R (9, body->world) and p (3) */
int gro_debug_body_pose(grx_handle s, int le, int b, double* R, double* p) {
    static __thread kin_t k;
    forward_kinematics(s, &s->env[le], &k);
    for (int i = 0; i < 9; ++i) R[i] = k.R[b][i];
    for (int i = 0; i < 3; ++i) p[i] = k.p[b][i];
    return GRX_OK;
}

/* per-link net contact force of env le (num_links x 3), last contact evaluation */
int gro_debug_link_forces(grx_handle s, int le, double* out, int max_links) {
    for (int L = 0; L < max_links && L < NL_MAX; ++L) for (int i = 0; i < 3; ++i) out[3 * L + i] = s->env[le].link_force[L][i];
    return GRX_OK;
}
