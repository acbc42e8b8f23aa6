// grx_rng.h -- counter-based Philox4x32-10 for the HIP kernels.
// One generator keyed by (seed, global env index, step, stream id): results are independent of
// the number of GPUs/ranks (SURVEY.md Appendix B "Determinism").  The reference draws from the
// torch / numpy global generators (legged_robot.py:656-677, 725-777, 790-793, 481).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum {
    GRX_RNG_RESET_DOF = 1,
    GRX_RNG_RESET_ROOT = 2,
    GRX_RNG_CMD_TIME = 3,
    GRX_RNG_CMD_RESET = 4,
    GRX_RNG_PUSH = 5,
    GRX_RNG_NOISE = 6,
    GRX_RNG_CURRICULUM = 7,
    GRX_RNG_INIT_DR = 8,
    GRX_RNG_INIT_LEVEL = 9,
    GRX_RNG_NOISE_DOF_L = 10,   // obs noise of the left-leg dof terms: item = group*5 + k (group 0 pos, 1 vel, 2 action)
    GRX_RNG_NOISE_DOF_R = 11
};

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 grx_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    U4 o = {c0, c1, c2, c3};
    return o;
}

__host__ __device__ inline float grx_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// i-th uniform of a stream
__host__ __device__ inline float grx_rand(uint64_t seed, uint32_t genv, uint32_t step, uint32_t stream, uint32_t i) {
    U4 o = grx_philox4x32_10(genv, step, stream, i >> 2, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t k = i & 3u;
    uint32_t v = k == 0 ? o.x : (k == 1 ? o.y : (k == 2 ? o.z : o.w));
    return grx_u01(v);
}
