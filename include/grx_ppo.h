/* grx_ppo.h -- C ABI of libgrx_ppo.so: the PPO minibatch loss, forward and gradients in one pass (gfx950).
 *
 * Replaces, for the training loop that sits on top of the env step, the ~100 element-wise kernels that
 * rsl_rl/algorithms/ppo.py:215-245 (log-prob, ratio, clipped surrogate, clipped value loss, entropy, KL) and
 * their autograd backward expand to.  Plain device pointers and sizes, no torch types; everything is fp32.
 * Deterministic: per-block partial sums are combined in block order by a second kernel.
 * Also here: the column sum behind every bias gradient of the two MLPs (grx_ppo_colsum).
 */
#ifndef GRX_PPO_H
#define GRX_PPO_H
#ifdef __cplusplus
extern "C" {
#endif

/* number of floats the caller must provide in `partials` (8-byte aligned scratch) for a batch of `batch` samples */
int grx_ppo_loss_partials_size(int batch);

/* One minibatch.
 *   mu [batch][num_actions], std [num_actions], value [batch]: the networks' outputs (row-major, contiguous)
 *   actions, old_mu, old_sigma [batch][num_actions]; old_logp, advantages, returns, target_values [batch]
 *   out[4]      = { surrogate loss, value loss, total loss, mean KL(old || new) }
 *   d_mu [batch][num_actions], d_std [num_actions], d_value [batch] = d(total loss)/d(.)
 *   total loss = surrogate + value_loss_coef * value_loss - entropy_coef * mean entropy   (ppo.py:243)
 * `stream` is a hipStream_t (0 = the null stream).  Returns 0, or a negative number for invalid arguments
 * (num_actions outside 1..32, batch < 1) -- nothing is launched then.
 */
int grx_ppo_loss(int batch, int num_actions, const float* mu, const float* std, const float* value,
                 const float* actions, const float* old_logp, const float* old_mu, const float* old_sigma,
                 const float* advantages, const float* returns, const float* target_values,
                 float clip_param, float value_loss_coef, float entropy_coef, int use_clipped_value_loss,
                 float* out, float* d_mu, float* d_std, float* d_value, float* partials, void* stream);

/* Column sums of a row-major fp32 matrix x [rows][cols] -> out [cols] (the bias gradient of a linear layer: the sum of
 * dY over the batch), deterministic (256-row slabs added in order).  `partials`: scratch of
 * grx_ppo_colsum_partials_size(rows, cols) floats.  Returns 0, negative for rows < 1 or cols < 1. */
int grx_ppo_colsum_partials_size(int rows, int cols);
int grx_ppo_colsum(int rows, int cols, const float* x, float* out, float* partials, void* stream);

/* Backward of a hidden layer's ELU(alpha 1) fused with that layer's bias gradient: dz [rows][cols] = dy * (y > 0 ? 1 : y + 1)
 * from the layer's OUTPUT y (torch's elu_backward on the result), out [cols] = column sums of dz (as grx_ppo_colsum:
 * deterministic, same slab order).  `partials`: grx_ppo_colsum_partials_size(rows, cols) floats. */
int grx_ppo_elu_backward_colsum(int rows, int cols, const float* dy, const float* y, float* dz, float* out, float* partials, void* stream);

/* One rollout step's bookkeeping in ONE launch (rsl_rl: PPO.process_env_step ppo.py:184-197 + RolloutStorage.add_transitions
 * rollout_storage.py:23-59 + the runner's running episode reward / length, on_policy_runner.py:170-181 -- ~25 small torch
 * kernels per env step otherwise).  All pointers are device pointers; N envs.
 *   in : obs (N, num_obs), pri (N, num_pri) or NULL, actions / mu / sigma (N, num_actions), values / logp / rewards (N),
 *        dones / time_outs (N) uint8 (time_outs may be NULL), gamma
 *   out: the storage rows of this step, each contiguous: st_obs (N, num_obs), st_pri (N, num_pri) or NULL, st_actions / st_mu /
 *        st_sigma (N, num_actions), st_values / st_logp / st_rewards (N), st_dones (N) uint8.
 *        st_rewards = rewards + gamma * values * time_outs  (bootstrap on time-outs, ppo.py:190-191)
 *   logging (all four may be NULL): cur_rew / cur_len (N) running sums, updated in place and zeroed where done; done_rew /
 *        done_len (N): the finished episode's totals where done (left untouched elsewhere).
 * Returns 0, negative for invalid sizes. */
int grx_ppo_store_transition(int N, int num_obs, int num_pri, int num_actions,
                             const float* obs, const float* pri, const float* actions, const float* mu, const float* sigma,
                             const float* values, const float* logp, const float* rewards, const unsigned char* dones,
                             const unsigned char* time_outs, float gamma,
                             float* st_obs, float* st_pri, float* st_actions, float* st_mu, float* st_sigma, float* st_values,
                             float* st_logp, float* st_rewards, unsigned char* st_dones,
                             float* cur_rew, float* cur_len, float* done_rew, float* done_len, void* stream);

/* One layer of an MLP at inference: Y [M][N] = act(X [M][K] . W^T + bias), W [N][K] row-major as torch.nn.Linear.weight,
 * bias [N] or NULL, act = ELU(alpha 1) when `elu` != 0 -- the rollout's policy / value forward (rsl_rl modules/mlp.py:7-42,
 * actor_critic_mlp.py act() / evaluate()) as one launch per layer: f32-input MFMA (v_mfma_f32_32x32x2_f32, exact f32:
 * every output is a k-ordered fmaf chain), bias and activation in the epilogue.  Any M, K, N >= 1; layers narrower than 32
 * outputs (action means, value) take a lane-per-row path.  Contiguous row-major fp32 everywhere.  Returns 0, negative for
 * invalid arguments / a failed launch. */
/* The minibatch of one PPO step in one launch: dst[t][r][:] = src[t][idx[r]][:] for t < n_tensors (<= GRX_PPO_GATHER_MAX), r < mb;
 * src[t] row-major fp32 with widths[t] columns, idx int64 on the device (RolloutStorage.mini_batch_generator's permutation
 * slice, rollout_storage.py:82-112).  src / dst / widths are HOST arrays of device pointers / ints. */
/* The TAIL of one PPO minibatch step in two launches: everything rsl_rl/algorithms/ppo.py:264-311 does between loss.backward() and the
 * next minibatch -- the adaptive learning rate from the minibatch KL (ppo.py:205-213), the NaN-skip (ppo.py:297-299),
 * nn.utils.clip_grad_norm_ and Adam.step() -- which PyTorch runs as ~20 small kernels (comparisons, clamps, wheres, a foreach norm, a
 * foreach multiply, the fused-Adam kernel's 42 us multi-tensor launch, statistics adds): 145 us of a 790 us captured step at the GR1T1
 * train shape, all of it on the step's critical path (profiles/r06_ppo_update_kernel_stats.txt).
 *   launch 1: per-chunk sums of squares of the gradients -> partials (deterministic order); block 0: lr <- adaptive rule(kl);
 *             the chunk-0 block of every tensor: its Adam step counter += 1 unless the loss is not finite
 *   launch 2: every block adds the partials in a fixed order -> per-tensor norms -> total norm -> clip coefficient (clip_grad_norm_'s
 *             formulas); unless the loss is not finite: Adam on its chunk with the CLIPPED gradient -- the arithmetic of ATen's
 *             fused_adam_utils.cuh adam_math (double constants, float state), bias corrections from the step counter; block 0: the
 *             update's running statistics sums[0] += value_loss, sums[1] += surrogate_loss (finite steps only), sums[2] = kl.
 * weight_decay = 0, no amsgrad, no maximize (the reference's optimizer).  Tensor pointers travel BY VALUE in the launch arguments: a captured
 * graph keeps the addresses of its capture (PyTorch's graph pool keeps gradient buffers where they were).  All pointers: device memory. */
#define GRX_PPO_TAIL_MAX 24
typedef struct grx_ppo_tail_tensors {
    int n, pad;
    float* param[GRX_PPO_TAIL_MAX]; const float* grad[GRX_PPO_TAIL_MAX]; float* exp_avg[GRX_PPO_TAIL_MAX]; float* exp_avg_sq[GRX_PPO_TAIL_MAX];
    float* step[GRX_PPO_TAIL_MAX];          /* the optimizer's per-parameter step counter (a float scalar on the device, torch's capturable Adam) */
    long long numel[GRX_PPO_TAIL_MAX];
} grx_ppo_tail_tensors;
typedef struct grx_ppo_tail_args {
    const float* loss;                      /* total loss of the minibatch: a non-finite one skips the step */
    const float* bad_flag;                  /* optional: != 0 skips the step too (the multi-rank path's collective decision) */
    const float* kl;                        /* minibatch mean KL */
    float* lr;                              /* learning rate, in / out */
    const float *value_loss, *surrogate_loss;
    float* sums;                            /* [3] running statistics of the update (may be NULL) */
    float* partials;                        /* scratch: grx_ppo_step_tail_blocks(tensors) floats */
    int adaptive, pad;
    float desired_kl, lr_min, lr_max, max_grad_norm;
    double beta1, beta2, eps;
} grx_ppo_tail_args;
int grx_ppo_step_tail_blocks(const grx_ppo_tail_tensors* t);
int grx_ppo_step_tail(const grx_ppo_tail_tensors* t, const grx_ppo_tail_args* a, void* stream);

#define GRX_PPO_GATHER_MAX 12
int grx_ppo_gather_rows(int n_tensors, const float* const* src, float* const* dst, const int* widths, const long long* idx, int mb, void* stream);

int grx_mlp_layer(int M, int K, int N, const float* X, const float* W, const float* bias, float* Y, int elu, void* stream);

/* The actor's output layer fused with the rollout's sampling and log-probability (actor_critic_mlp.py act() /
 * get_actions_log_prob() -> torch.distributions.Normal): mu = X . W^T + bias, actions = mu + std * eps,
 * logp = sum_k -(a - mu)^2 / (2 std^2) - log(std) - log(sqrt(2 pi)); sigma = std broadcast to (M, A).
 * X [M][K], W [A][K], std [A], eps [M][A] standard-normal draws supplied by the caller.  1 <= A <= 32. */
int grx_mlp_policy_head(int M, int K, int A, const float* X, const float* W, const float* bias, const float* std,
                        const float* eps, float* actions, float* logp, float* mu, float* sigma, void* stream);

#ifdef __cplusplus
}
#endif
#endif
