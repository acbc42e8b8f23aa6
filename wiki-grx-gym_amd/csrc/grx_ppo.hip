// grx_ppo.hip -- the PPO minibatch loss with its gradients, one pass over the batch (include/grx_ppo.h).
// Follows rsl_rl/algorithms/ppo.py:215-245 and torch.distributions.Normal (log_prob, entropy) term by term;
// the gradients are those torch.autograd produces for that expression (torch.max splits ties half / half,
// clamp passes the gradient on the closed interval).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grx_ppo.h"

namespace {
constexpr int MAXA = 32;   // the full-body GR1T1 has 32 actions
constexpr int NTHR = 256;
constexpr int NRED = 3 + MAXA;   // surrogate, value loss, KL, d_std[MAXA]

// The batch sums run in double: the surrogate loss is a mean of terms of both signs (normalised advantages) that
// cancels to ~1e-3 of their magnitude, so an fp32 sum in ANY order is only good to ~1e-4 of the result.
// torch.max / torch.clamp PROPAGATE NaN (fmaxf / fminf drop it): a NaN ratio or value must reach the loss, so that the
// update's NaN-skip (ppo.py:297-299) fires instead of a finite loss hiding NaN gradients
__device__ inline float nmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ inline float nmin(float a, float b) { return (a < b || a != a) ? a : b; }

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

template <int A>
__global__ __launch_bounds__(NTHR) void ppo_loss_kernel(int B, const float* __restrict__ mu, const float* __restrict__ stdp,
                                                         const float* __restrict__ value, const float* __restrict__ actions,
                                                         const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                                                         const float* __restrict__ old_sigma, const float* __restrict__ adv_,
                                                         const float* __restrict__ ret_, const float* __restrict__ tv_,
                                                         float clip, float vcoef, int use_clipped,
                                                         float* __restrict__ d_mu, float* __restrict__ d_value, double* __restrict__ partials) {
    const int i = blockIdx.x * NTHR + threadIdx.x;
    float red[3 + A];
#pragma unroll
    for (int k = 0; k < 3 + A; ++k) red[k] = 0.f;
    if (i < B) {
        const float invB = 1.0f / (float)B;
        float sg[A], df[A];
        float logp = 0.f, kl = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const float s = stdp[k], m = mu[(size_t)i * A + k];
            sg[k] = s;
            df[k] = actions[(size_t)i * A + k] - m;
            // Normal.log_prob: -((x - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi))
            logp += -(df[k] * df[k]) / (2.0f * s * s) - logf(s) - 0.9189385332046727f;
            const float os = old_sigma[(size_t)i * A + k], dm = old_mu[(size_t)i * A + k] - m;
            kl += logf(s / os + 1.e-5f) + (os * os + dm * dm) / (2.0f * s * s) - 0.5f;
        }
        const float adv = adv_[i];
        const float ratio = expf(logp - old_logp[i]);
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float rc = nmin(nmax(ratio, lo), hi);
        const float s1 = -adv * ratio, s2 = -adv * rc;
        const float surr = nmax(s1, s2);
        // d max(s1, s2): the larger one takes the gradient, a tie splits it; d clamp = 1 on [lo, hi]
        const float w1 = s1 > s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f), w2 = 1.f - w1;
        const float in_range = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        const float g_ratio = -adv * (w1 + w2 * in_range);
        const float g_logp = g_ratio * ratio * invB;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const float s = sg[k], inv2 = 1.0f / (s * s);
            d_mu[(size_t)i * A + k] = g_logp * df[k] * inv2;
            red[3 + k] = g_logp * (df[k] * df[k] * inv2 / s - 1.0f / s);
        }
        const float v = value[i], ret = ret_[i];
        float vl, gv;
        if (use_clipped) {
            const float tv = tv_[i];
            const float dv = v - tv;
            const float vc = tv + nmin(nmax(dv, -clip), clip);
            const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
            vl = nmax(l1, l2);
            const float u1 = l1 > l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f), u2 = 1.f - u1;
            const float vin = (dv >= -clip && dv <= clip) ? 1.f : 0.f;
            gv = u1 * 2.0f * (v - ret) + u2 * 2.0f * (vc - ret) * vin;
        } else {
            vl = (ret - v) * (ret - v);
            gv = -2.0f * (ret - v);
        }
        d_value[i] = vcoef * gv * invB;
        red[0] = surr; red[1] = vl; red[2] = kl;
    }
    __shared__ double s_red[NTHR / 64][3 + A];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3 + A; ++k) {
        const double t = wave_sum((double)red[k]);
        if (lane == 0) s_red[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 3 + A) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NTHR / 64; ++w) t += s_red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * NRED + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void ppo_loss_finalize(int B, int A, int nblk, const double* __restrict__ partials,
                                                         const float* __restrict__ stdp, float vcoef, float ecoef,
                                                         float* __restrict__ out, float* __restrict__ d_std) {
    const int k = threadIdx.x;
    double t = 0.0;
    constexpr int STAGE = 4096;   // doubles: the partials of a minibatch fit (41 blocks at 10^4 samples); fetched in parallel, added from LDS in the SAME order
    __shared__ double s_p[STAGE];
    const bool staged = nblk * NRED <= STAGE;
    if (staged) {
        for (int i = k; i < nblk * NRED; i += 64) s_p[i] = partials[i];
        __syncthreads();
    }
    if (k < 3 + A)
        for (int b = 0; b < nblk; ++b) t += staged ? s_p[b * NRED + k] : partials[(size_t)b * NRED + k];
    __shared__ double s_t[64];
    s_t[k] = t;   // 3 + MAXA <= 64
    __syncthreads();
    if (k >= 3 && k < 3 + A) d_std[k - 3] = (float)t - ecoef / stdp[k - 3];   // entropy: -ecoef * mean_b sum_k log(std_k) -> -ecoef / std_k
    if (k == 0) {
        const float surr = (float)(s_t[0] / (double)B), vl = (float)(s_t[1] / (double)B), klm = (float)(s_t[2] / (double)B);
        float ent = 0.f;   // Normal.entropy: 0.5 + 0.5 log(2 pi) + log(scale), summed over the actions
        for (int a = 0; a < A; ++a) ent += 1.4189385332046727f + logf(stdp[a]);
        out[0] = surr; out[1] = vl; out[2] = surr + vcoef * vl - ecoef * ent; out[3] = klm;
    }
}

template <int A>
void launch(int nblk, hipStream_t st, int B, const float* mu, const float* stdp, const float* value, const float* actions,
            const float* old_logp, const float* old_mu, const float* old_sigma, const float* adv, const float* ret, const float* tv,
            float clip, float vcoef, int use_clipped, float* d_mu, float* d_value, double* partials) {
    hipLaunchKernelGGL(ppo_loss_kernel<A>, dim3(nblk), dim3(NTHR), 0, st, B, mu, stdp, value, actions, old_logp, old_mu, old_sigma,
                       adv, ret, tv, clip, vcoef, use_clipped, d_mu, d_value, partials);
}

// ---- column sums of a row-major [rows][cols] matrix (bias gradients: db = sum over the batch of dY) -----------------
// torch's own column reduction (at::native reduce_kernel) is what the captured PPO step must avoid: replayed from a HIP
// graph with other GPU work in between it returned garbage for one 128-column case (tools/gpu_ppo_graph_check.py).
// Two deterministic passes: 256-row slabs -> partials [slab][col] (a wave reads 64 consecutive columns of a row), then
// one thread per column adds the slabs in order.
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_partial(int rows, int cols, const float* __restrict__ x, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float acc = 0.f;
    if (col < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four loads in flight per lane
        int r = r0 + wv;
        for (; r + 12 < r1; r += 16) {
            a0 += x[(size_t)r * cols + col]; a1 += x[(size_t)(r + 4) * cols + col];
            a2 += x[(size_t)(r + 8) * cols + col]; a3 += x[(size_t)(r + 12) * cols + col];
        }
        for (; r < r1; r += 4) a0 += x[(size_t)r * cols + col];
        acc = (a0 + a1) + (a2 + a3);
    }
    __shared__ float s_acc[4][64];
    s_acc[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && col < cols) partials[(size_t)blockIdx.y * cols + col] = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
}
// The same first pass for a hidden layer's backward: dz = dy * ELU'(y) from the layer's OUTPUT y (y > 0 ? 1 : y + 1, torch's
// elu_backward with is_result), written out for the two GEMMs that follow, and summed per column for the bias gradient --
// one read of dy instead of an elu_backward launch followed by colsum_partial.
__global__ __launch_bounds__(256) void elu_bwd_colsum_partial(int rows, int cols, const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dz, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float acc = 0.f;
    if (col < cols) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // (the same four interleaved accumulators as colsum_partial)
        auto one = [&](const int r) -> float {
            const size_t o = (size_t)r * cols + col;
            const float yy = y[o];
            const float g = dy[o] * (yy > 0.f ? 1.0f : yy + 1.0f);
            dz[o] = g;
            return g;
        };
        int r = r0 + wv;
        for (; r + 12 < r1; r += 16) { a0 += one(r); a1 += one(r + 4); a2 += one(r + 8); a3 += one(r + 12); }
        for (; r < r1; r += 4) a0 += one(r);
        acc = (a0 + a1) + (a2 + a3);
    }
    __shared__ float s_acc[4][64];
    s_acc[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && col < cols) partials[(size_t)blockIdx.y * cols + col] = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
}
// one wave per column: lane l adds slabs l, l + 64, ... (in order), then a fixed shuffle tree -- deterministic, one exposed
// load latency instead of nslab dependent ones
__global__ __launch_bounds__(256) void colsum_final(int cols, int nslab, const float* __restrict__ partials, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    float t = 0.f;
    for (int b = lane; b < nslab; b += 64) t += partials[(size_t)b * cols + col];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if (lane == 0) out[col] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// One layer of the policy / value MLP for the ROLLOUT (inference): Y = act(X W^T + b), fp32 in, fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain per output).  rsl_rl's MLP is
// Linear -> ELU -> ... -> Linear (mlp.py:7-42); a rollout step runs two of them at batch = num_envs, which through
// torch is 8 library GEMMs + 6 ELU kernels + glue (~150 us beside a 67 us env step).  Here a layer is one launch:
// 64 x 64 output tile per 256-thread block (four waves, a 32 x 32 MFMA tile each), K in chunks of 32 through LDS with the
// next chunk's global loads in flight during the current chunk's 16 MFMAs, bias + ELU in the epilogue.
// Operand maps (cdna_hip_programming.md): A: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31];
// D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ML_BM = 64, ML_BN = 64, ML_KC = 64, ML_LD = 65;   // LDS tiles are [k][row], row stride 65 (transposed stores without bank conflicts)

__device__ inline float elu1(float x) { return x > 0.f ? x : expm1f(x); }   // torch.nn.ELU(alpha = 1)

// 64 rows x 64 k of a row-major matrix -> registers (4 x float4 per thread: thread t, pass p -> row 16 p + t / 16, k 4 (t % 16)),
// zero beyond the matrix; VEC: rows are 16-byte aligned (ld % 4 == 0)
template <bool VEC>
__device__ inline void ml_fetch(const float* __restrict__ G, int rows, int ld, int r0, int k0, int tid, float4 v[4]) {
    const int kq = k0 + 4 * (tid & 15);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 16 * p + (tid >> 4);
        const float* g = G + (size_t)min(r, rows - 1) * ld + kq;
        float4 x = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (VEC) { if (kq < ld) x = *reinterpret_cast<const float4*>(g); }   // (ld % 4 == 0: a quad is inside or outside as a whole)
            else { if (kq < ld) x.x = g[0]; if (kq + 1 < ld) x.y = g[1]; if (kq + 2 < ld) x.z = g[2]; if (kq + 3 < ld) x.w = g[3]; }
        }
        v[p] = x;
    }
}
__device__ inline void ml_stash(float* __restrict__ S, int tid, const float4 v[4]) {
    const int kq = 4 * (tid & 15);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float* s = S + kq * ML_LD + 16 * p + (tid >> 4);
        s[0] = v[p].x; s[ML_LD] = v[p].y; s[2 * ML_LD] = v[p].z; s[3 * ML_LD] = v[p].w;
    }
}

// HEAD: the actor's output layer (N = number of actions <= 32) with the rollout's sampling and log-probability in the epilogue
// (actor_critic_mlp.py act() / get_actions_log_prob(), torch.distributions.Normal): mu = X W^T + b; a = mu + std * eps;
// logp = sum_k -(a - mu)^2 / (2 std^2) - log std - log sqrt(2 pi) -- a row's actions sit in the 32 lanes of a half wave.
struct HeadOut { const float* stdp; const float* eps; float* actions; float* logp; float* mu; float* sigma; };
template <bool ELU, bool VEC, bool HEAD = false>
__global__ __launch_bounds__(256) void mlp_layer_kernel(int M, int K, int N, const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y, HeadOut ho = HeadOut()) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mlp_layer_kernel keeps 65 KB of static LDS: gfx950 (160 KB per CU) only -- this library is built for MI355X, see the Makefile"
#endif
    __shared__ float Xs[2][ML_KC * ML_LD], Ws[2][ML_KC * ML_LD];   // double-buffered: chunk c + 1 lands while chunk c multiplies
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wy = wv >> 1, wx = wv & 1;
    const int m0 = blockIdx.x * ML_BM, n0 = blockIdx.y * ML_BN;
    float4 xr[4], wr[4];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ml_fetch<VEC>(X, M, K, m0, 0, tid, xr);
    ml_fetch<VEC>(W, N, K, n0, 0, tid, wr);
    ml_stash(Xs[0], tid, xr); ml_stash(Ws[0], tid, wr);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += ML_KC, buf ^= 1) {
        const bool more = k0 + ML_KC < K;
        if (more) { ml_fetch<VEC>(X, M, K, m0, k0 + ML_KC, tid, xr); ml_fetch<VEC>(W, N, K, n0, k0 + ML_KC, tid, wr); }   // in flight during the MFMAs
        const float* xa = Xs[buf] + (lane >> 5) * ML_LD + 32 * wy + (lane & 31);
        const float* wb = Ws[buf] + (lane >> 5) * ML_LD + 32 * wx + (lane & 31);
        const int nk = min(ML_KC, K - k0);
        if (nk == ML_KC) {
            // operands of the whole chunk into registers first: 64 LDS reads in flight, then 32 MFMAs back to back (read ->
            // wait -> MFMA one by one, as the compiler orders the naive loop, exposes the LDS latency 32 times per chunk)
            float av[ML_KC / 2], bv[ML_KC / 2];
#pragma unroll
            for (int kk = 0; kk < ML_KC / 2; ++kk) { av[kk] = xa[2 * kk * ML_LD]; bv[kk] = wb[2 * kk * ML_LD]; }
#pragma unroll
            for (int kk = 0; kk < ML_KC / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
        } else {
            for (int kk = 0; kk < (nk + 1) / 2; ++kk)   // (the tile is zero beyond K)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * kk * ML_LD], wb[2 * kk * ML_LD], acc, 0, 0, 0);
        }
        if (more) { ml_stash(Xs[buf ^ 1], tid, xr); ml_stash(Ws[buf ^ 1], tid, wr); }
        __syncthreads();
    }
    const int col = n0 + 32 * wx + (lane & 31);
    if (HEAD) {
        if (wx != 0) return;   // (N <= 32: the second column half of the tile is empty)
        const bool live = col < N;
        const float b = (live && bias) ? bias[col] : 0.f;
        const float sd = live ? ho.stdp[col] : 1.f;
        const float lsd = logf(sd), inv2 = 1.0f / (2.0f * sd * sd);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bool ok = live && row < M;
            const float m = acc[r] + b;
            float term = 0.f;
            if (ok) {
                const size_t o = (size_t)row * N + col;
                const float a = m + sd * ho.eps[o];
                const float d = a - m;
                term = -(d * d) * inv2 - lsd - 0.9189385332046727f;
                ho.actions[o] = a; ho.mu[o] = m; ho.sigma[o] = sd;
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) term += __shfl_xor(term, off);   // over the 32 lanes (columns) of this row
            if ((lane & 31) == 0 && row < M) ho.logp[row] = term;
        }
        return;
    }
    if (col < N) {
        const float b = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) { const float y = acc[r] + b; Y[(size_t)row * N + col] = ELU ? elu1(y) : y; }
        }
    }
}

}  // namespace


// ---------------------------------------------------------------------------------------------------------------------
// The minibatch of one PPO step, gathered in ONE launch: rows idx[0..mb) of up to GRX_PPO_GATHER_MAX row-major fp32 tensors
// (observations, privileged observations, actions, values, advantages, returns, log-probs, mu, sigma: rollout_storage.py:82-112)
// into their static minibatch buffers -- nine index_select launches otherwise, 200 times per update.
namespace {
struct GatherArgs { const float* src[GRX_PPO_GATHER_MAX]; float* dst[GRX_PPO_GATHER_MAX]; int width[GRX_PPO_GATHER_MAX]; };
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a, const long long* __restrict__ idx, int mb) {
    const int t = blockIdx.y, w = a.width[t];
    const long long n = (long long)mb * w;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int row = (int)(e / w), col = (int)(e - (long long)row * w);
        a.dst[t][e] = a.src[t][idx[row] * w + col];
    }
}
}  // namespace

extern "C" int grx_ppo_gather_rows(int n_tensors, const float* const* src, float* const* dst, const int* widths, const long long* idx, int mb, void* stream) {
    if (n_tensors < 1 || n_tensors > GRX_PPO_GATHER_MAX || mb < 1 || !src || !dst || !widths || !idx) return -1;
    GatherArgs a;
    int wmax = 1;
    for (int t = 0; t < GRX_PPO_GATHER_MAX; ++t) {
        const int u = t < n_tensors ? t : 0;
        if (!src[u] || !dst[u] || widths[u] < 1) return -1;
        a.src[t] = src[u]; a.dst[t] = dst[u]; a.width[t] = widths[u];
        if (widths[u] > wmax) wmax = widths[u];
    }
    const long long nmax = (long long)mb * wmax;
    const int bx = (int)((nmax + 255) / 256 < 1024 ? (nmax + 255) / 256 : 1024);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, a, idx, mb);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int grx_mlp_policy_head(int M, int K, int A, const float* X, const float* W, const float* bias, const float* std,
                                   const float* eps, float* actions, float* logp, float* mu, float* sigma, void* stream) {
    if (M < 1 || K < 1 || A < 1 || A > 32 || !X || !W || !std || !eps || !actions || !logp || !mu || !sigma) return -1;
    const HeadOut ho = {std, eps, actions, logp, mu, sigma};
    const bool vec = K % 4 == 0 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)W % 16 == 0);
    const dim3 grid((M + ML_BM - 1) / ML_BM, 1);
    if (vec) hipLaunchKernelGGL((mlp_layer_kernel<false, true, true>), grid, dim3(256), 0, (hipStream_t)stream, M, K, A, X, W, bias, (float*)nullptr, ho);
    else hipLaunchKernelGGL((mlp_layer_kernel<false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, M, K, A, X, W, bias, (float*)nullptr, ho);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int grx_mlp_layer(int M, int K, int N, const float* X, const float* W, const float* bias, float* Y, int elu, void* stream) {
    if (M < 1 || K < 1 || N < 1 || !X || !W || !Y) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = K % 4 == 0 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)W % 16 == 0);
#define GRX_ML_LAUNCH(KERNEL, GRID, BLOCK)                                                                              \
    do {                                                                                                                \
        if (elu) { if (vec) hipLaunchKernelGGL((KERNEL<true, true>), GRID, BLOCK, 0, st, M, K, N, X, W, bias, Y, HeadOut());       \
                   else hipLaunchKernelGGL((KERNEL<true, false>), GRID, BLOCK, 0, st, M, K, N, X, W, bias, Y, HeadOut()); }        \
        else { if (vec) hipLaunchKernelGGL((KERNEL<false, true>), GRID, BLOCK, 0, st, M, K, N, X, W, bias, Y, HeadOut());          \
               else hipLaunchKernelGGL((KERNEL<false, false>), GRID, BLOCK, 0, st, M, K, N, X, W, bias, Y, HeadOut()); }           \
    } while (0)
    GRX_ML_LAUNCH(mlp_layer_kernel, dim3((M + ML_BM - 1) / ML_BM, (N + ML_BN - 1) / ML_BN), dim3(256));
#undef GRX_ML_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- the tail of a minibatch step (include/grx_ppo.h grx_ppo_step_tail)
namespace {
constexpr int TAIL_CHUNK = 4096, TAIL_THR = 256, TAIL_SPLIT = 4;   // (the apply kernel: TAIL_SPLIT blocks per chunk of the norm kernel -- 4 elements per thread)
struct TailWhere { int t; long long begin, count; };   // the tensor a block works on and its chunk of it
__device__ inline TailWhere tail_locate(const grx_ppo_tail_tensors& T, int block) {
    int b = block;
    for (int t = 0; t < T.n; ++t) {
        const int nb = (int)((T.numel[t] + TAIL_CHUNK - 1) / TAIL_CHUNK);
        if (b < nb) { const long long beg = (long long)b * TAIL_CHUNK; return TailWhere{t, beg, min((long long)TAIL_CHUNK, T.numel[t] - beg)}; }
        b -= nb;
    }
    return TailWhere{-1, 0, 0};
}
__device__ inline bool tail_bad(const grx_ppo_tail_args& A) { return !isfinite(*A.loss) || (A.bad_flag && *A.bad_flag != 0.f); }
__global__ __launch_bounds__(TAIL_THR) void tail_norm_kernel(const grx_ppo_tail_tensors T, const grx_ppo_tail_args A) {
    __shared__ float red[TAIL_THR];
    const TailWhere w = tail_locate(T, blockIdx.x);
    float s = 0.f;
    if (w.t >= 0) {
        const float* g = T.grad[w.t] + w.begin;
        for (long long i = threadIdx.x; i < w.count; i += TAIL_THR) s += g[i] * g[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = TAIL_THR / 2; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) {
        A.partials[blockIdx.x] = red[0];
        if (w.t >= 0 && w.begin == 0 && !tail_bad(A)) *T.step[w.t] += 1.f;   // (torch: _foreach_add_(steps, 1), taken back under found_inf)
        if (blockIdx.x == 0 && A.adaptive) {   // update_learning_rate (ppo.py:205-213), the branches of PPO._device_lr_update
            const float lr = *A.lr, kl = *A.kl;
            const float down = fmaxf(lr * (1.0f / 1.5f), A.lr_min), up = fminf(lr * 1.5f, A.lr_max);   // (torch divides a device tensor by a host scalar as a product with its fp32 reciprocal: the same bits as PPO._device_lr_update)
            *A.lr = kl > A.desired_kl * 2.0f ? down : ((kl < A.desired_kl / 2.0f && kl > 0.0f) ? up : lr);
        }
    }
}
__global__ __launch_bounds__(TAIL_THR) void tail_apply_kernel(const grx_ppo_tail_tensors T, const grx_ppo_tail_args A, int nblocks) {
    __shared__ float s_clip;
    __shared__ float s_part[1024];
    const bool bad = tail_bad(A);
    for (int i = threadIdx.x; i < nblocks && i < 1024; i += TAIL_THR) s_part[i] = A.partials[i];   // (one parallel fetch; the sums below then run on LDS)
    __syncthreads();
    if (threadIdx.x == 0) {
        // clip_grad_norm_: per-tensor 2-norms, the 2-norm of those, max_norm / (total + 1e-6) clamped to 1 -- every block adds the same
        // partials in the same order
        float tot = 0.f;
        int b = 0;
        for (int t = 0; t < T.n; ++t) {
            const int nb = (int)((T.numel[t] + TAIL_CHUNK - 1) / TAIL_CHUNK);
            float st = 0.f;
            for (int k = 0; k < nb; ++k) st += (b + k < 1024 ? s_part[b + k] : A.partials[b + k]);
            b += nb;
            const float nt = sqrtf(st);
            tot += nt * nt;
        }
        const float total = sqrtf(tot);
        s_clip = fminf(A.max_grad_norm / (total + 1e-6f), 1.0f);
        if (blockIdx.x == 0 && A.sums) {
            const float ok = bad ? 0.f : 1.f;
            A.sums[0] += *A.value_loss * ok; A.sums[1] += *A.surrogate_loss * ok; A.sums[2] = *A.kl;
        }
    }
    __syncthreads();
    if (bad) return;
    TailWhere w = tail_locate(T, blockIdx.x / TAIL_SPLIT);
    if (w.t < 0) return;
    {   // this block's part of the chunk
        const long long sub = (TAIL_CHUNK / TAIL_SPLIT) * (long long)(blockIdx.x % TAIL_SPLIT);
        if (sub >= w.count) return;
        w.begin += sub; w.count = min((long long)(TAIL_CHUNK / TAIL_SPLIT), w.count - sub);
    }
    const float clip = s_clip;
    const double lr = (double)*A.lr, beta1 = A.beta1, beta2 = A.beta2, eps = A.eps;
    const float stepc = *T.step[w.t];                                  // (already counted by tail_norm_kernel)
    const float bias_correction1 = (float)(1.0 - pow(beta1, (double)stepc));
    const float bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)stepc));
    float* p = T.param[w.t] + w.begin; const float* g = T.grad[w.t] + w.begin;
    float* m = T.exp_avg[w.t] + w.begin; float* v = T.exp_avg_sq[w.t] + w.begin;
    for (long long i = threadIdx.x; i < w.count; i += TAIL_THR) {
        float param = p[i];
        const float grad = g[i] * clip;
        float exp_avg = m[i], exp_avg_sq = v[i];
        exp_avg = (float)(beta1 * exp_avg + (1 - beta1) * grad);               // (fused_adam_utils.cuh adam_math: double constants, float state)
        exp_avg_sq = (float)(beta2 * exp_avg_sq + (1 - beta2) * grad * grad);
        const float step_size = (float)(lr / bias_correction1);
        const float denom = (float)((sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps);
        param -= step_size * exp_avg / denom;
        p[i] = param; m[i] = exp_avg; v[i] = exp_avg_sq;
    }
}
}  // namespace
extern "C" int grx_ppo_step_tail_blocks(const grx_ppo_tail_tensors* t) {
    if (!t || t->n < 1 || t->n > GRX_PPO_TAIL_MAX) return -1;
    long long nb = 0;
    for (int i = 0; i < t->n; ++i) { if (t->numel[i] < 1) return -1; nb += (t->numel[i] + TAIL_CHUNK - 1) / TAIL_CHUNK; }
    return nb > 65535 ? -1 : (int)nb;
}
extern "C" int grx_ppo_step_tail(const grx_ppo_tail_tensors* t, const grx_ppo_tail_args* a, void* stream) {
    const int nb = grx_ppo_step_tail_blocks(t);
    if (nb < 1 || !a || !a->loss || !a->kl || !a->lr || !a->partials) return -1;
    hipLaunchKernelGGL(tail_norm_kernel, dim3(nb), dim3(TAIL_THR), 0, (hipStream_t)stream, *t, *a);
    hipLaunchKernelGGL(tail_apply_kernel, dim3(nb * TAIL_SPLIT), dim3(TAIL_THR), 0, (hipStream_t)stream, *t, *a, nb);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int grx_ppo_colsum_partials_size(int rows, int cols) { return (rows < 1 || cols < 1) ? 0 : ((rows + CS_ROWS - 1) / CS_ROWS) * cols; }

extern "C" int grx_ppo_colsum(int rows, int cols, const float* x, float* out, float* partials, void* stream) {
    if (rows < 1 || cols < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nslab = (rows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_partial, dim3((cols + 63) / 64, nslab), dim3(256), 0, st, rows, cols, x, partials);
    hipLaunchKernelGGL(colsum_final, dim3((cols + 3) / 4), dim3(256), 0, st, cols, nslab, partials, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int grx_ppo_elu_backward_colsum(int rows, int cols, const float* dy, const float* y, float* dz, float* out, float* partials, void* stream) {
    if (rows < 1 || cols < 1 || !dy || !y || !dz || !out || !partials) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nslab = (rows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(elu_bwd_colsum_partial, dim3((cols + 63) / 64, nslab), dim3(256), 0, st, rows, cols, dy, y, dz, partials);
    hipLaunchKernelGGL(colsum_final, dim3((cols + 3) / 4), dim3(256), 0, st, cols, nslab, partials, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// in floats (the buffer holds doubles: 2 floats each; the caller's allocation must be 8-byte aligned)
extern "C" int grx_ppo_loss_partials_size(int batch) { return batch < 1 ? 0 : ((batch + NTHR - 1) / NTHR) * NRED * 2; }

extern "C" int grx_ppo_loss(int batch, int num_actions, const float* mu, const float* std, const float* value,
                            const float* actions, const float* old_logp, const float* old_mu, const float* old_sigma,
                            const float* advantages, const float* returns, const float* target_values,
                            float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                            float* out, float* d_mu, float* d_std, float* d_value, float* partials, void* stream) {
    if (batch < 1 || num_actions < 1 || num_actions > MAXA) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (batch + NTHR - 1) / NTHR;
#define GRX_PPO_CASE(A) case A: launch<A>(nblk, st, batch, mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, \
                                          target_values, clip_param, value_loss_coef, use_clipped_value_loss, d_mu, d_value, (double*)partials); break;
    switch (num_actions) {
        GRX_PPO_CASE(1) GRX_PPO_CASE(2) GRX_PPO_CASE(3) GRX_PPO_CASE(4) GRX_PPO_CASE(5) GRX_PPO_CASE(6) GRX_PPO_CASE(7) GRX_PPO_CASE(8)
        GRX_PPO_CASE(9) GRX_PPO_CASE(10) GRX_PPO_CASE(11) GRX_PPO_CASE(12) GRX_PPO_CASE(13) GRX_PPO_CASE(14) GRX_PPO_CASE(15) GRX_PPO_CASE(16)
        GRX_PPO_CASE(17) GRX_PPO_CASE(18) GRX_PPO_CASE(19) GRX_PPO_CASE(20) GRX_PPO_CASE(21) GRX_PPO_CASE(22) GRX_PPO_CASE(23) GRX_PPO_CASE(24)
        GRX_PPO_CASE(25) GRX_PPO_CASE(26) GRX_PPO_CASE(27) GRX_PPO_CASE(28) GRX_PPO_CASE(29) GRX_PPO_CASE(30) GRX_PPO_CASE(31) GRX_PPO_CASE(32)
    }
#undef GRX_PPO_CASE
    hipLaunchKernelGGL(ppo_loss_finalize, dim3(1), dim3(64), 0, st, batch, num_actions, nblk, (const double*)partials, std, value_loss_coef, entropy_coef, out, d_std);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- grx_ppo_store_transition: one rollout step's bookkeeping in one launch -----------------------------------------
// Rows: the (N, k) blocks are contiguous, so the copies are flat; every thread walks the concatenated index space
// [obs | pri | actions | mu | sigma] with a grid stride, then the per-env scalars.
struct StoreArgs {
    int N, no, np, na;
    const float *obs, *pri, *actions, *mu, *sigma, *values, *logp, *rewards;
    const unsigned char *dones, *time_outs;
    float gamma;
    float *st_obs, *st_pri, *st_actions, *st_mu, *st_sigma, *st_values, *st_logp, *st_rewards;
    unsigned char* st_dones;
    float *cur_rew, *cur_len, *done_rew, *done_len;
};
__global__ __launch_bounds__(256) void store_transition_kernel(StoreArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_obs = (size_t)a.N * a.no, n_pri = a.pri ? (size_t)a.N * a.np : 0, n_act = (size_t)a.N * a.na;
    for (size_t i = tid; i < n_obs; i += stride) a.st_obs[i] = a.obs[i];
    for (size_t i = tid; i < n_pri; i += stride) a.st_pri[i] = a.pri[i];
    for (size_t i = tid; i < n_act; i += stride) { a.st_actions[i] = a.actions[i]; a.st_mu[i] = a.mu[i]; a.st_sigma[i] = a.sigma[i]; }
    for (size_t i = tid; i < (size_t)a.N; i += stride) {
        const float v = a.values[i], r = a.rewards[i];
        const bool done = a.dones[i] != 0, to = a.time_outs && a.time_outs[i] != 0;
        a.st_values[i] = v;
        a.st_logp[i] = a.logp[i];
        a.st_rewards[i] = r + (to ? a.gamma * v : 0.0f);   // rewards + gamma * values * time_outs (ppo.py:190-191)
        a.st_dones[i] = done ? 1 : 0;
        if (a.cur_rew) {   // on_policy_runner.py:170-181 without the nonzero() + .cpu() per step
            const float cr = a.cur_rew[i] + r, cl = a.cur_len[i] + 1.0f;
            if (done) { a.done_rew[i] = cr; a.done_len[i] = cl; }
            a.cur_rew[i] = done ? 0.0f : cr;
            a.cur_len[i] = done ? 0.0f : cl;
        }
    }
}
extern "C" int grx_ppo_store_transition(int N, int num_obs, int num_pri, int num_actions,
                                        const float* obs, const float* pri, const float* actions, const float* mu, const float* sigma,
                                        const float* values, const float* logp, const float* rewards, const unsigned char* dones,
                                        const unsigned char* time_outs, float gamma,
                                        float* st_obs, float* st_pri, float* st_actions, float* st_mu, float* st_sigma, float* st_values,
                                        float* st_logp, float* st_rewards, unsigned char* st_dones,
                                        float* cur_rew, float* cur_len, float* done_rew, float* done_len, void* stream) {
    if (N < 1 || num_obs < 1 || num_actions < 1 || (pri && num_pri < 1)) return -1;
    if ((cur_rew != nullptr) != (cur_len != nullptr) || (cur_rew != nullptr) != (done_rew != nullptr) || (cur_rew != nullptr) != (done_len != nullptr)) return -1;
    StoreArgs a = {N, num_obs, num_pri, num_actions, obs, pri, actions, mu, sigma, values, logp, rewards, dones, time_outs, gamma,
                   st_obs, st_pri, st_actions, st_mu, st_sigma, st_values, st_logp, st_rewards, st_dones, cur_rew, cur_len, done_rew, done_len};
    const size_t work = (size_t)N * (size_t)(num_pri > num_obs ? num_pri : num_obs);
    int blocks = (int)((work + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(store_transition_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
