// single-wave issue-rate probe: v_fma_f32 vs v_pk_fma_f32 (two fp32 FMAs per lane and instruction), 1 wave per block
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int ILP, bool PK>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
    float2v a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = threadIdx.x * 0.002f + i; }
    float2v b = {1.0001f, 0.9999f}, c = {0.0003f, 0.0002f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (PK) a[i] = __builtin_elementwise_fma(a[i], b, c);
                else { a[i].x = fmaf(a[i].x, b.x, c.x); asm volatile("" : "+v"(a[i].x)); a[i].y = fmaf(a[i].y, b.y, c.y); asm volatile("" : "+v"(a[i].y)); }
            }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP, bool PK> void run(int blocks) {
    float* out; long long* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    int iters = 2000;
    hipLaunchKernelGGL((k<ILP, PK>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<ILP, PK>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s ILP %2d: %.2f cycles per PAIR of fp32 FMAs (wave64, one wave per SIMD)\n", PK ? "v_pk_fma_f32" : "2 x v_fma_f32", ILP, (double)h[0] / ((double)iters * 16 * ILP));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, false>(128); run<1, true>(128); run<2, false>(128); run<2, true>(128); run<4, false>(128); run<4, true>(128); run<8, false>(128); run<8, true>(128);
    return 0;
}
