// Probe for hand-packed fp32 (v_pk_fma_f32) on the articulated-inertia recursion (VERDICT r3 #3 ii): the inertia half of one leg
// (5 joints: add the parent's rigid inertia, U = I^A S, 1/d, rank-1 downdate) with ONE lane per leg, as the lane-pair layouts run it
// (grx_wavepipe.h substep_p / grx_kernels.hip substep) -- scalar V3 / S3 / M3 algebra on the 21 unique entries against the full 6 x 6
// kept as six COLUMNS of three float2 each: U = sum_j col_j S_j and col_j -= U (U_j / d) are float2 FMAs with a broadcast scalar
// (op_sel picks the half of a register pair: no moves).  Four waves per block (every SIMD of the CU busy: the step kernel's
// situation), cycles per recursion by s_memtime, results compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../wiki-grx-gym_amd/csrc/grx_math.h"
constexpr int LEG = 5;
typedef float f2 __attribute__((ext_vector_type(2)));
struct In { V3 a[LEG], s[LEG]; S3 AK[LEG]; V3 h[LEG]; float m[LEG]; };
__device__ __forceinline__ void make_in(int leg, In& x) {
#pragma unroll
    for (int k = 0; k < LEG; ++k) {
        const float t = 0.37f * (k + 1) + 0.11f * leg;
        x.a[k] = v3(__cosf(t), __sinf(t) * 0.6f, 0.3f + 0.1f * k); x.s[k] = v3(0.2f * __sinf(2 * t), -0.1f + 0.05f * k, 0.3f * __cosf(3 * t));
        x.AK[k] = S3{0.05f + 0.01f * k, 0.002f, -0.001f, 0.06f, 0.0015f, 0.03f + 0.005f * k};
        x.h[k] = v3(0.02f * k, -0.05f, -0.3f - 0.1f * k); x.m[k] = 1.0f + 0.7f * k;
    }
}
__global__ __launch_bounds__(256) void k_one(float* out, long long* cyc, int iters) {
    const int leg = threadIdx.x & 63;
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
            acc += ua.x + ua.y + ua.z + ul.x + ul.y + ul.z + di;
        }
        acc += A.xx + A.yz + B.a00 + B.a12 + B.a21 + D.xx + D.yz;
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// full 6 x 6, columns of three float2 (rows 01 | 23 | 45); S = (a; s) as three float2
struct C6 { f2 r01, r23, r45; };
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }
__global__ __launch_bounds__(256) void k_pk(float* out, long long* cyc, int iters) {
    const int leg = threadIdx.x & 63;
    In x; make_in(leg, x);
    float acc = 0.f, g = 1.0f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        g = g * 1.0001f;
        asm volatile("" : "+v"(g));
        C6 c[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { c[j].r01 = splat(0.f); c[j].r23 = splat(0.f); c[j].r45 = splat(0.f); }
#pragma unroll
        for (int k = LEG - 1; k >= 0; --k) {
            const S3 K = x.AK[k];
            const V3 h = x.h[k] * g;
            const float m = x.m[k];
            // rigid inertia [A B; B^T D], B = skew-like (B01 = -h.z, B02 = +h.y, B10 = +h.z, B12 = -h.x, B20 = -h.y, B21 = +h.x)
            c[0].r01 += f2{K.xx, K.xy}; c[0].r23.x += K.xz;                     c[0].r45 += f2{-h.z, h.y};
            c[1].r01 += f2{K.xy, K.yy}; c[1].r23 += f2{K.yz, h.z};              c[1].r45.y += -h.x;
            c[2].r01 += f2{K.xz, K.yz}; c[2].r23 += f2{K.zz, -h.y};             c[2].r45.x += h.x;
            c[3].r01.y += h.z;          c[3].r23 += f2{-h.y, m};
            c[4].r01.x += -h.z;         c[4].r23.x += h.x;                      c[4].r45.x += m;
            c[5].r01 += f2{h.y, -h.x};                                          c[5].r45.y += m;
            const f2 s01 = {x.a[k].x, x.a[k].y}, s23 = {x.a[k].z, x.s[k].x}, s45 = {x.s[k].y, x.s[k].z};
            const float sj[6] = {x.a[k].x, x.a[k].y, x.a[k].z, x.s[k].x, x.s[k].y, x.s[k].z};
            f2 u01 = c[0].r01 * splat(sj[0]), u23 = c[0].r23 * splat(sj[0]), u45 = c[0].r45 * splat(sj[0]);
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                u01 = __builtin_elementwise_fma(c[j].r01, splat(sj[j]), u01);
                u23 = __builtin_elementwise_fma(c[j].r23, splat(sj[j]), u23);
                u45 = __builtin_elementwise_fma(c[j].r45, splat(sj[j]), u45);
            }
            f2 t = s01 * u01;
            t = __builtin_elementwise_fma(s23, u23, t);
            t = __builtin_elementwise_fma(s45, u45, t);
            const float di = grx_rcp(t.x + t.y);
            const f2 w01 = u01 * splat(di), w23 = u23 * splat(di), w45 = u45 * splat(di);
            const float wj[6] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                c[j].r01 = __builtin_elementwise_fma(-u01, splat(wj[j]), c[j].r01);
                c[j].r23 = __builtin_elementwise_fma(-u23, splat(wj[j]), c[j].r23);
                c[j].r45 = __builtin_elementwise_fma(-u45, splat(wj[j]), c[j].r45);
            }
            acc += u01.x + u01.y + u23.x + u23.y + u45.x + u45.y + di;
        }
        acc += c[0].r01.x + c[1].r23.x + c[3].r01.x + c[5].r01.y + c[4].r23.x + c[3].r23.y + c[4].r45.y;   // A.xx + A.yz + B.a00 + B.a12 + B.a21 + D.xx + D.yz
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks = 256, iters = 400;
    float* out; long long* cyc; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    float h1[64], h2[64]; long long c1, c2;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_one, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h1, out, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h2, out, sizeof h2, hipMemcpyDeviceToHost); hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost);
    printf("scalar, 21 unique entries : %.0f cycles per recursion\n", (double)c1 / iters);
    printf("packed, six float2 columns: %.0f cycles per recursion   (results %.6f vs %.6f)\n", (double)c2 / iters, h1[0], h2[0]);
    return 0;
}
