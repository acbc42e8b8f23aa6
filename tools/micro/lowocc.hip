// Control experiment for the sporadic 10-80 ms stalls seen with the step kernel at 4096 envs: a trivial
// latency-bound kernel (dependent FMA chain, no LDS, no memory traffic) at the SAME low occupancy
// (128 single-wave blocks) and at full occupancy, launched back to back, queue drained by spinning on an
// event.  Build/run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/lowocc tools/micro/lowocc.hip && /tmp/lowocc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void chain(float* out, int iters, float a) {
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) x = fmaf(x, a, 0.5f);
    if (x == 123.456f) out[0] = x;
}
static double run(int blocks, int iters, int launches) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f);
    hipEventRecord(ev, 0);
    while (hipEventQuery(ev) == hipErrorNotReady) {}
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipFree(d); hipEventDestroy(ev);
    return ms / launches;
}
int main() {
    for (int blocks : {128, 512, 8192}) {
        printf("blocks=%5d ms/launch:", blocks);
        for (int rep = 0; rep < 12; ++rep) printf(" %.4f", run(blocks, 20000, 500));
        printf("\n");
    }
    return 0;
}
