// single-wave VALU issue-rate probe: dependent vs independent fp32 FMA chains, 1 wave per block
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
    float a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0001f, c = 0.0003f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = fmaf(a[i], b, c);
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP> void run(int blocks) {
    float* out; long long* cyc; hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    int iters = 2000;
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, cyc, 8 * (blocks < 4 ? blocks : 4), hipMemcpyDeviceToHost);
    printf("ILP %2d blocks %4d: %.2f cycles per FMA instruction (wave64)\n", ILP, blocks, (double)h[0] / ((double)iters * 16 * ILP));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(128); run<2>(128); run<4>(128); run<8>(128); run<16>(128);
    run<1>(1024); run<4>(1024); run<8>(2048); run<8>(8192);
    return 0;
}
