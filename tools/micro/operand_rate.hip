// single-wave VALU issue-rate probe with REAL operand traffic: every FMA reads three distinct VGPRs (the step kernel's
// v_fmac / v_fma mix), scalar vs packed (v_pk_fma_f32), one or two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o operand_rate operand_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int ILP, int MODE>   // MODE 0: v_fma 3 VGPR; 1: v_pk_fma 3 VGPR pairs; 2: v_fma with 2 SGPR-like constants
__global__ void k(const float* in, float* out, long long* cyc, int iters) {
    f2 a[ILP], b[ILP], c[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        a[i].x = in[threadIdx.x + i]; a[i].y = in[threadIdx.x + 64 + i];
        b[i].x = in[threadIdx.x + 128 + i]; b[i].y = in[threadIdx.x + 192 + i];
        c[i].x = in[threadIdx.x + 256 + i]; c[i].y = in[threadIdx.x + 320 + i];
    }
    const float kb = 1.0001f, kc = 0.0003f;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (MODE == 1) { a[i] = __builtin_elementwise_fma(a[i], b[i], c[i]); c[i] = __builtin_elementwise_fma(b[i], a[i], c[i]); }
                else if (MODE == 0) {
                    a[i].x = fmaf(a[i].x, b[i].x, c[i].x); a[i].y = fmaf(a[i].y, b[i].y, c[i].y);
                    c[i].x = fmaf(b[i].x, a[i].x, c[i].x); c[i].y = fmaf(b[i].y, a[i].y, c[i].y);
                    asm volatile("" : "+v"(a[i].x), "+v"(a[i].y), "+v"(c[i].x), "+v"(c[i].y));
                } else {
                    a[i].x = fmaf(a[i].x, kb, kc); a[i].y = fmaf(a[i].y, kb, kc); c[i].x = fmaf(c[i].x, kb, kc); c[i].y = fmaf(c[i].y, kb, kc);
                    asm volatile("" : "+v"(a[i].x), "+v"(a[i].y), "+v"(c[i].x), "+v"(c[i].y));
                }
            }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i].x + a[i].y + c[i].x + c[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP, int MODE> void run(int threads) {
    float *in, *out; long long* cyc; const int blocks = 256;
    hipMalloc(&in, 4096 * 4); hipMemset(in, 0, 4096 * 4); hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const char* nm[3] = {"v_fma 3 VGPR   ", "v_pk_fma 3 VGPR", "v_fma 1 VGPR   "};
    printf("%s ILP %d, %d waves/SIMD: %.2f cycles per fp32 FMA per wave (%.2f per instruction)\n", nm[MODE], ILP, threads / 256, (double)h[0] / ((double)iters * 8 * ILP * 4),
           (double)h[0] / ((double)iters * 8 * ILP * (MODE == 1 ? 2 : 4)));
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main() {
    run<4, 2>(256); run<4, 0>(256); run<4, 1>(256); run<8, 0>(256); run<8, 1>(256);
    run<4, 2>(512); run<4, 0>(512); run<4, 1>(512); run<8, 0>(512); run<8, 1>(512);
    return 0;
}
