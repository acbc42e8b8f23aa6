// Feasibility probe for "more lanes per env" (DESIGN.md section 8): the inertia half of the articulated-body recursion of one leg
// (5 joints: add the rigid inertia, U = I^A S, 1/d, rank-1 downdate, I^A c) as the step kernel does it -- ONE lane per leg --
// against the same arithmetic split by ROWS over two lanes of a quad (lane h = 0 owns the angular rows [A B], lane h = 1 the
// linear rows [B^T D]; U halves, the partial d and c halves cross with DPP quad_perm [2,3,0,1]).  One wave per block (the step
// kernel's situation), cycles per recursion by s_memtime, results compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../wiki-grx-gym_amd/csrc/grx_math.h"
constexpr int LEG = 5;
__device__ __forceinline__ float half_swap(float v) {   // lane ^ 2
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ V3 half_swap(V3 v) { return v3(half_swap(v.x), half_swap(v.y), half_swap(v.z)); }
// inputs of one leg's recursion: generated ONCE (outside the timed loop); the loop perturbs them with one cheap scale so that
// the compiler cannot hoist the recursion
struct In { V3 a[LEG], s[LEG], ca[LEG], cl[LEG]; S3 AK[LEG]; V3 h[LEG]; float m[LEG]; };
__device__ __forceinline__ void make_in(int leg, In& x) {
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        const float t = 0.37f * (k + 1) + 0.11f * leg;
        x.a[k] = v3(__cosf(t), __sinf(t) * 0.6f, 0.3f + 0.1f * k); x.s[k] = v3(0.2f * __sinf(2 * t), -0.1f + 0.05f * k, 0.3f * __cosf(3 * t));
        x.ca[k] = v3(0.01f * k, 0.02f, -0.03f * __sinf(t)); x.cl[k] = v3(-0.02f, 0.015f * k, 0.01f);
        x.AK[k] = S3{0.05f + 0.01f * k, 0.002f, -0.001f, 0.06f, 0.0015f, 0.03f + 0.005f * k};
        x.h[k] = v3(0.02f * k, -0.05f, -0.3f - 0.1f * k); x.m[k] = 1.0f + 0.7f * k;
    }
}
// ---- reference: one lane per leg (grx_wavepipe.h substep_p, inertia half)
__global__ __launch_bounds__(64) void k_one(float* out, long long* cyc, int iters) {
    const int leg = threadIdx.x;
    In x; make_in(leg, x);
    float acc = 0.f, g = 1.0f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        g = g * 1.0001f;
        asm volatile("" : "+v"(g));
        S3 A = {0, 0, 0, 0, 0, 0}, D = {0, 0, 0, 0, 0, 0};
        M3 B = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            A = A + x.AK[k];
            const V3 h = x.h[k] * g;
            B.a01 -= h.z; B.a02 += h.y; B.a10 += h.z; B.a12 -= h.x; B.a20 -= h.y; B.a21 += h.x;
            D.xx += x.m[k]; D.yy += x.m[k]; D.zz += x.m[k];
            const V3 ua = mul(A, x.a[k]) + mul(B, x.s[k]);
            const V3 ul = mulT(B, x.a[k]) + mul(D, x.s[k]);
            const float di = grx_rcp(dot(x.a[k], ua) + dot(x.s[k], ul));
            syr(A, ua, di); ger(B, ua, ul, di); syr(D, ul, di);
            const V3 Ica = mul(A, x.ca[k]) + mul(B, x.cl[k]);
            const V3 Icl = mulT(B, x.ca[k]) + mul(D, x.cl[k]);
            acc += Ica.x + Ica.y + Ica.z + Icl.x + Icl.y + Icl.z + ua.x + ul.z + di;
        }
        acc += A.xx + A.yz + B.a00 + B.a12 + B.a21 + D.xx + D.yz;
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// ---- rows split over two lanes of a quad: lane = 4 * env + 2 * half + side.  Each lane holds its three rows of I^A as [P | Q]
// (half 0: P = A, Q = B; half 1: P = D, Q = B^T) and its own halves of S and c (mine / other), set up ONCE outside the loop: the
// loop body is the same instruction stream for both halves, different data.
__global__ __launch_bounds__(64) void k_quad(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x, half = (lane >> 1) & 1, leg = (lane & 1) + 2 * (lane >> 2);
    In x; make_in(leg, x);
    V3 mine[LEG], other[LEG], cm[LEG], co[LEG], qh[LEG];
    S3 pk[LEG];
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        mine[k] = half == 0 ? x.a[k] : x.s[k]; other[k] = half == 0 ? x.s[k] : x.a[k];
        cm[k] = half == 0 ? x.ca[k] : x.cl[k]; co[k] = half == 0 ? x.cl[k] : x.ca[k];
        pk[k] = half == 0 ? x.AK[k] : S3{x.m[k], 0.f, 0.f, x.m[k], 0.f, x.m[k]};   // what the rigid body adds to P
        qh[k] = half == 0 ? x.h[k] : neg(x.h[k]);                                  // B -= skew(h)   /   B^T += skew(h)
    }
    float acc = 0.f, g = 1.0f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        g = g * 1.0001f;
        asm volatile("" : "+v"(g));
        S3 P = {0, 0, 0, 0, 0, 0};
        M3 Q = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            P = P + pk[k];
            const V3 h = qh[k] * g;
            Q.a01 -= h.z; Q.a02 += h.y; Q.a10 += h.z; Q.a12 -= h.x; Q.a20 -= h.y; Q.a21 += h.x;
            const V3 u = mul(P, mine[k]) + mul(Q, other[k]);   // ua on half 0, ul on half 1
            const float dpart = dot(mine[k], u);
            const float di = grx_rcp(dpart + half_swap(dpart));
            const V3 uo = half_swap(u);                        // the other half's U
            syr(P, u, di); ger(Q, u, uo, di);
            const V3 Ic = mul(P, cm[k]) + mul(Q, co[k]);       // I^a c: angular rows on half 0, linear rows on half 1
            acc += Ic.x + Ic.y + Ic.z + u.x * (half == 0 ? 1.f : 0.f) + u.z * (half == 0 ? 0.f : 1.f) + (half == 0 ? di : 0.f);
        }
        acc += P.xx + P.yz + (half == 0 ? Q.a00 + Q.a12 + Q.a21 : 0.f);
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks = 256, iters = 400;
    float* out; long long* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    float h1[64], h2[64]; long long c1, c2;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_one, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h1, out, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h2, out, sizeof h2, hipMemcpyDeviceToHost); hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost);
    // leg 0 / 1 of env 0: one-lane result = sum of the two halves' results
    double worst = 0;
    for (int leg = 0; leg < 2; ++leg) { const double a = h1[leg], b = (double)h2[leg] + (double)h2[2 + leg]; worst = fmax(worst, fabs(a - b) / fmax(1.0, fabs(a))); }
    printf("one lane per leg : %.0f cycles per recursion (64 legs per wave)\n", (double)c1 / iters);
    printf("two lanes per leg: %.0f cycles per recursion (32 legs per wave)   relative difference of the results %.2e\n", (double)c2 / iters, worst);
    return 0;
}
