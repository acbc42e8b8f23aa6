// lds_handover.hip -- litmus test of the LDS hand-over the step kernels' wave pipelines rest on (csrc/grx_flags.h).
//
// Claim under test: a producer wave writes a payload with ds_write_b128 (all 64 lanes) and then raises a flag with ONE ds_write_b32
// from lane 0, with NO s_waitcnt between them (flag_set: the release fence is wavefront scope, i.e. it orders the compiler only);
// a consumer wave ON ANOTHER SIMD that sees the flag (flag_wait: relaxed spin + workgroup acquire on the LDS address space) then
// reads the complete payload of that hand-over -- because the LDS unit executes a wave's LDS instructions in program order.
//
// The kernel uses the product's own flag_set / flag_wait (included from csrc/grx_flags.h, compiled with the product's flags).
// Every block runs PAIRS producer/consumer waves (wave 2p produces for wave 2p + 1: different SIMDs, as waves are dealt round robin)
// in a ping-pong: the consumer acknowledges hand-over i through a second flag (also flag_set), the producer rewrites the payload only
// then -- the pipelines' single-buffer rule.  The payload of hand-over i is QUADS float4 per lane whose every word encodes (i, lane,
// quad, word); the consumer checks all of them and counts mismatches.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGRX_FLAG_FENCED] -I../../wiki-grx-gym_amd/csrc -o lds_handover lds_handover.hip
//   ./lds_handover [hand-overs per pair = 2000000] [blocks = 1024] [pairs per block = 4]
// prints one JSON line (tools/run_litmus.sh collects the unfenced and the fenced build into profiles/rNN_lds_handover_litmus.json).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define GRX_DEV __device__ __forceinline__
enum { FL_COUNT = 22 };
#include "grx_flags.h"

constexpr int QUADS = 6;        // float4 per lane and hand-over (the pipelines' records are 2..15 quads)
constexpr int MAXPAIRS = 4;

__device__ __forceinline__ uint32_t word_of(uint32_t i, int lane, int q, int w) { return i * 2654435761u + (uint32_t)(lane * 97 + q * 13 + w * 7 + 1); }

__global__ __launch_bounds__(64 * 2 * MAXPAIRS) void handover(int iters, int pairs, unsigned long long* errors, unsigned long long* done) {
    __shared__ uint4 s_pay[MAXPAIRS][QUADS][64];
    __shared__ int s_flag[MAXPAIRS][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pair = wave >> 1;
    if (threadIdx.x < MAXPAIRS * 2) (&s_flag[0][0])[threadIdx.x] = 0;
    __syncthreads();
    if (pair >= pairs) return;
    int* const f_data = &s_flag[pair][0];
    int* const f_ack = &s_flag[pair][1];
    unsigned long long bad = 0;
    if ((wave & 1) == 0) {   // producer
        for (int i = 0; i < iters; ++i) {
            flag_wait(f_ack, i);   // hand-over i - 1 has been consumed
#pragma unroll
            for (int q = 0; q < QUADS; ++q) s_pay[pair][q][lane] = make_uint4(word_of(i, lane, q, 0), word_of(i, lane, q, 1), word_of(i, lane, q, 2), word_of(i, lane, q, 3));
            flag_set(f_data, i + 1, lane);
        }
    } else {                 // consumer
        for (int i = 0; i < iters; ++i) {
            flag_wait(f_data, i + 1);
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                const uint4 v = s_pay[pair][q][lane];
                bad += (v.x != word_of(i, lane, q, 0)) + (v.y != word_of(i, lane, q, 1)) + (v.z != word_of(i, lane, q, 2)) + (v.w != word_of(i, lane, q, 3));
            }
            flag_set(f_ack, i + 1, lane);
        }
        if (bad) atomicAdd(errors, bad);
        if (lane == 0) atomicAdd(done, (unsigned long long)iters);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000000, blocks = argc > 2 ? atoi(argv[2]) : 1024, pairs = argc > 3 ? atoi(argv[3]) : MAXPAIRS;
    unsigned long long *d_err, *d_done, h_err = 0, h_done = 0;
    hipMalloc(&d_err, 8); hipMalloc(&d_done, 8);
    hipMemset(d_err, 0, 8); hipMemset(d_done, 0, 8);
    hipLaunchKernelGGL(handover, dim3(blocks), dim3(64 * 2 * pairs), 0, 0, 1000, pairs, d_err, d_done);   // warm-up
    hipDeviceSynchronize();
    hipMemset(d_err, 0, 8); hipMemset(d_done, 0, 8);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(handover, dim3(blocks), dim3(64 * 2 * pairs), 0, 0, iters, pairs, d_err, d_done);
    const hipError_t e = hipDeviceSynchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipMemcpy(&h_err, d_err, 8, hipMemcpyDeviceToHost); hipMemcpy(&h_done, d_done, 8, hipMemcpyDeviceToHost);
#ifdef GRX_FLAG_FENCED
    const char* variant = "workgroup-scope release fence before the flag (GRX_FLAG_FENCED)";
#else
    const char* variant = "wavefront-scope release (compiler order only): the product's flag_set";
#endif
    printf("{\"test\": \"lds_handover\", \"variant\": \"%s\", \"hip_error\": \"%s\", \"blocks\": %d, \"pairs_per_block\": %d, \"handovers_per_pair\": %d, "
           "\"quads_per_lane\": %d, \"handovers\": %llu, \"payload_words_checked\": %llu, \"mismatched_words\": %llu, \"seconds\": %.3f, \"ns_per_round_trip\": %.1f}\n",
           variant, hipGetErrorString(e), blocks, pairs, iters, QUADS, h_done, h_done * (unsigned long long)(QUADS * 4 * 64), h_err, dt, dt * 1e9 / iters);
    return (e == hipSuccess && h_err == 0 && h_done == (unsigned long long)iters * blocks * pairs) ? 0 : 1;
}
