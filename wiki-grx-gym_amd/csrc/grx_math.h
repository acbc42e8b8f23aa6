// grx_math.h -- device-side small-vector algebra for the fused GRx step kernel (gfx950).
// Everything is by-value structs of floats so the compiler keeps them in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#define GRX_DEV __device__ __forceinline__

// hardware reciprocal / sqrt (1 ulp): no denormal pre-scaling sequences in the issue-bound inner loop
GRX_DEV float grx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
GRX_DEV float grx_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
GRX_DEV float grx_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

struct V3 { float x, y, z; };
struct S3 { float xx, xy, xz, yy, yz, zz; };          // symmetric 3x3
struct M3 { float a00, a01, a02, a10, a11, a12, a20, a21, a22; };  // full 3x3, row-major
struct R3 { V3 cx, cy, cz; };                          // rotation, stored as columns (body -> world)

GRX_DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
GRX_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
GRX_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
GRX_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
GRX_DEV V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
GRX_DEV V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
GRX_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
GRX_DEV V3 cross(V3 a, V3 b) {
    return v3(fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x));
}
GRX_DEV V3 fma3(V3 a, float s, V3 c) { return v3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }

GRX_DEV V3 rot(const R3& R, V3 v) { return fma3(R.cx, v.x, fma3(R.cy, v.y, R.cz * v.z)); }
GRX_DEV V3 rotT(const R3& R, V3 v) { return v3(dot(R.cx, v), dot(R.cy, v), dot(R.cz, v)); }

GRX_DEV V3 mul(const S3& A, V3 v) {
    return v3(fmaf(A.xx, v.x, fmaf(A.xy, v.y, A.xz * v.z)), fmaf(A.xy, v.x, fmaf(A.yy, v.y, A.yz * v.z)),
              fmaf(A.xz, v.x, fmaf(A.yz, v.y, A.zz * v.z)));
}
GRX_DEV V3 mul(const M3& B, V3 v) {
    return v3(fmaf(B.a00, v.x, fmaf(B.a01, v.y, B.a02 * v.z)), fmaf(B.a10, v.x, fmaf(B.a11, v.y, B.a12 * v.z)),
              fmaf(B.a20, v.x, fmaf(B.a21, v.y, B.a22 * v.z)));
}
GRX_DEV V3 mulT(const M3& B, V3 v) {
    return v3(fmaf(B.a00, v.x, fmaf(B.a10, v.y, B.a20 * v.z)), fmaf(B.a01, v.x, fmaf(B.a11, v.y, B.a21 * v.z)),
              fmaf(B.a02, v.x, fmaf(B.a12, v.y, B.a22 * v.z)));
}
GRX_DEV S3 operator+(const S3& a, const S3& b) {
    S3 r = {a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz};
    return r;
}
GRX_DEV M3 operator+(const M3& a, const M3& b) {
    M3 r = {a.a00 + b.a00, a.a01 + b.a01, a.a02 + b.a02, a.a10 + b.a10, a.a11 + b.a11,
            a.a12 + b.a12, a.a20 + b.a20, a.a21 + b.a21, a.a22 + b.a22};
    return r;
}
// A -= u u^T * s   (symmetric rank-1)
GRX_DEV void syr(S3& A, V3 u, float s) {
    V3 us = u * s;
    A.xx = fmaf(-us.x, u.x, A.xx); A.xy = fmaf(-us.x, u.y, A.xy); A.xz = fmaf(-us.x, u.z, A.xz);
    A.yy = fmaf(-us.y, u.y, A.yy); A.yz = fmaf(-us.y, u.z, A.yz); A.zz = fmaf(-us.z, u.z, A.zz);
}
// B -= u w^T * s
GRX_DEV void ger(M3& B, V3 u, V3 w, float s) {
    V3 us = u * s;
    B.a00 = fmaf(-us.x, w.x, B.a00); B.a01 = fmaf(-us.x, w.y, B.a01); B.a02 = fmaf(-us.x, w.z, B.a02);
    B.a10 = fmaf(-us.y, w.x, B.a10); B.a11 = fmaf(-us.y, w.y, B.a11); B.a12 = fmaf(-us.y, w.z, B.a12);
    B.a20 = fmaf(-us.z, w.x, B.a20); B.a21 = fmaf(-us.z, w.y, B.a21); B.a22 = fmaf(-us.z, w.z, B.a22);
}
// inverse of an SPD symmetric 3x3 (cofactors)
GRX_DEV S3 inv(const S3& A) {
    float c00 = fmaf(A.yy, A.zz, -A.yz * A.yz);
    float c01 = fmaf(A.xz, A.yz, -A.xy * A.zz);
    float c02 = fmaf(A.xy, A.yz, -A.xz * A.yy);
    float det = fmaf(A.xx, c00, fmaf(A.xy, c01, A.xz * c02));
    float id = grx_rcp(det);
    S3 r;
    r.xx = c00 * id; r.xy = c01 * id; r.xz = c02 * id;
    r.yy = fmaf(A.xx, A.zz, -A.xz * A.xz) * id;
    r.yz = fmaf(A.xy, A.xz, -A.xx * A.yz) * id;
    r.zz = fmaf(A.xx, A.yy, -A.xy * A.xy) * id;
    return r;
}
// R Ic R^T for symmetric Ic (result symmetric)
GRX_DEV S3 rot_sym(const R3& R, const S3& I) {
    // M = R * I  (columns of M^T ...): rows of R are (cx.x,cy.x,cz.x) etc.
    V3 r0 = v3(R.cx.x, R.cy.x, R.cz.x), r1 = v3(R.cx.y, R.cy.y, R.cz.y), r2 = v3(R.cx.z, R.cy.z, R.cz.z);
    V3 m0 = mul(I, r0), m1 = mul(I, r1), m2 = mul(I, r2);  // I symmetric: (R I)_row_i = I r_i
    S3 o;
    o.xx = dot(m0, r0); o.xy = dot(m0, r1); o.xz = dot(m0, r2);
    o.yy = dot(m1, r1); o.yz = dot(m1, r2); o.zz = dot(m2, r2);
    return o;
}
// quaternion (x,y,z,w) -> rotation columns
GRX_DEV R3 quat_to_R(float x, float y, float z, float w) {
    R3 R;
    R.cx = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + z * w), 2.f * (x * z - y * w));
    R.cy = v3(2.f * (x * y - z * w), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + x * w));
    R.cz = v3(2.f * (x * z + y * w), 2.f * (y * z - x * w), 1.f - 2.f * (x * x + y * y));
    return R;
}
// isaacgym/torch_utils.py:71-81 quat_rotate_inverse (a - b + c form)
GRX_DEV V3 quat_rotate_inverse(V3 qv, float qw, V3 v) {
    float s = 2.0f * qw * qw - 1.0f;
    V3 cr = cross(qv, v);
    float d = dot(qv, v);
    return v3(v.x * s - cr.x * qw * 2.0f + qv.x * d * 2.0f, v.y * s - cr.y * qw * 2.0f + qv.y * d * 2.0f,
              v.z * s - cr.z * qw * 2.0f + qv.z * d * 2.0f);
}
// isaacgym/torch_utils.py:48-55 quat_apply
GRX_DEV V3 quat_apply(V3 qv, float qw, V3 b) {
    V3 t = cross(qv, b) * 2.0f;
    V3 u = cross(qv, t);
    return v3(b.x + qw * t.x + u.x, b.y + qw * t.y + u.y, b.z + qw * t.z + u.z);
}
// child rotation = parent rotation * Rot_axis(q); AX: 0 = x, 1 = y, 2 = z
template <int AX>
GRX_DEV R3 joint_rot(const R3& P, float c, float s) {
    R3 R;
    if (AX == 0) { R.cx = P.cx; R.cy = fma3(P.cy, c, P.cz * s); R.cz = fma3(P.cz, c, P.cy * (-s)); }
    else if (AX == 1) { R.cy = P.cy; R.cx = fma3(P.cx, c, P.cz * (-s)); R.cz = fma3(P.cz, c, P.cx * s); }
    else { R.cz = P.cz; R.cx = fma3(P.cx, c, P.cy * s); R.cy = fma3(P.cy, c, P.cx * (-s)); }
    return R;
}
template <int AX>
GRX_DEV V3 axis_of(const R3& R) { return AX == 0 ? R.cx : (AX == 1 ? R.cy : R.cz); }

// ---- lane <-> env mapping ------------------------------------------------------------------------------------------------
// GRX_LPE lanes per env.  2 (default): lane 2e = left leg, 2e + 1 = right leg.  4 (grx_quad.hip, <= 16 envs per CU): a lane PAIR
// per leg -- lanes 4e, 4e + 1 the left leg, 4e + 2, 4e + 3 the right one; the two lanes of a leg ("halves") split its work.
// Everything below is one DPP quad_perm step away: no LDS traffic.
#ifndef GRX_LPE
#define GRX_LPE 2
#endif
constexpr int LPE = GRX_LPE, LPL = GRX_LPE / 2;   // lanes per env, lanes per leg
static_assert(LPE == 2 || LPE == 4, "one lane or a lane pair per leg");
GRX_DEV int lane_env(int lane) { return lane / LPE; }
GRX_DEV int lane_side(int lane) { return (lane / LPL) & 1; }
GRX_DEV int lane_half(int lane) { return lane & (LPL - 1); }
template <int CTRL>
GRX_DEV float quad_perm(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
// the same half of the env's OTHER leg
GRX_DEV float pair_swap(float v) { return quad_perm<LPE == 2 ? 0xB1 : 0x4E>(v); }   // quad_perm [1,0,3,2] / [2,3,0,1]
// the other half of the same leg (LPE == 4)
GRX_DEV float half_swap(float v) { return LPE == 2 ? v : quad_perm<0xB1>(v); }
GRX_DEV float half_sum(float v) { return LPE == 2 ? v : v + quad_perm<0xB1>(v); }
GRX_DEV float pair_sum(float v) { return v + pair_swap(v); }
GRX_DEV float env_sum(float v) { return pair_sum(half_sum(v)); }   // over all lanes of the env
GRX_DEV V3 half_sum(V3 v) { return v3(half_sum(v.x), half_sum(v.y), half_sum(v.z)); }
GRX_DEV V3 half_swap(V3 v) { return v3(half_swap(v.x), half_swap(v.y), half_swap(v.z)); }
GRX_DEV V3 pair_sum(V3 v) { return v3(pair_sum(v.x), pair_sum(v.y), pair_sum(v.z)); }
GRX_DEV S3 pair_sum(const S3& a) {
    S3 r = {pair_sum(a.xx), pair_sum(a.xy), pair_sum(a.xz), pair_sum(a.yy), pair_sum(a.yz), pair_sum(a.zz)};
    return r;
}
GRX_DEV M3 pair_sum(const M3& a) {
    M3 r = {pair_sum(a.a00), pair_sum(a.a01), pair_sum(a.a02), pair_sum(a.a10), pair_sum(a.a11),
            pair_sum(a.a12), pair_sum(a.a20), pair_sum(a.a21), pair_sum(a.a22)};
    return r;
}
