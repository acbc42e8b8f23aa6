/* TEST INFRASTRUCTURE (oracle): counter-based Philox4x32-10 (Salmon et al., SC'11, "Parallel
 * random numbers: as easy as 1, 2, 3"), restated from the published algorithm.  The reference
 * uses three global RNG streams (torch / numpy / torch-CPU); the build keys
 * one counter-based generator by (seed, global env index, step, stream id) so results do not
 * depend on the number of GPUs.  Known-answer vectors from the Random123 distribution are checked
 * in tests/test_philox.py. */
#ifndef GRO_PHILOX_H_
#define GRO_PHILOX_H_
#include <stdint.h>

static inline void gro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 24-bit uniform in [0,1) -- exactly representable in fp32 */
static inline float gro_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* RNG stream ids (shared with the HIP kernels: wiki-grx-gym_amd/csrc/grx_rng.h) */
enum {
    GRO_RNG_RESET_DOF = 1,
    GRO_RNG_RESET_ROOT = 2,
    GRO_RNG_CMD_TIME = 3,
    GRO_RNG_CMD_RESET = 4,
    GRO_RNG_PUSH = 5,
    GRO_RNG_NOISE = 6,
    GRO_RNG_CURRICULUM = 7,
    GRO_RNG_INIT_DR = 8,
    GRO_RNG_INIT_LEVEL = 9,
    GRO_RNG_NOISE_DOF_L = 10, /* obs noise of the left-leg dof terms: item = group*5 + k (0 pos, 1 vel, 2 action) */
    GRO_RNG_NOISE_DOF_R = 11
};

/* i-th uniform of stream `stream` for (global env, step) */
static inline float gro_rand(uint64_t seed, uint32_t genv, uint32_t step, uint32_t stream, uint32_t i) {
    uint32_t ctr[4] = {genv, step, stream, i >> 2};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    gro_philox4x32_10(ctr, key, out);
    return gro_u01(out[i & 3]);
}
#endif
