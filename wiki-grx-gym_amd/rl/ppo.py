"""PPO with rsl_rl semantics (reference: rsl_rl/algorithms/ppo.py:10-333): clipped surrogate, clipped
value loss, entropy bonus, adaptive-KL learning rate (x / 1.5 outside [kl/2, 2 kl], clamped),
NaN-skip, grad-norm clip 1.0, one Adam step per minibatch, time-out bootstrap.

Multi-GPU (the one addition): envs are sharded over ranks, every rank holds a replica of
the 436 885-parameter model.  Per optimizer step ONE flat fp32 bucket (gradients + the minibatch KL
appended as the last element) is all-reduced over RCCL/xGMI and averaged, so every rank takes the
same adaptive-LR decision and the same Adam step: replicas stay bit-identical without broadcasting.
At 1.75 MB the collective is latency-bound; a single fused bucket keeps it at one launch."""
import contextlib
import os

import torch
import torch.distributed as dist
from .fused_loss import fused_ppo_loss, mlp_can_fuse, mlp_forward, policy_act
import torch.nn as nn
import torch.optim as optim

from .modules import ActorCriticMLP  # noqa: F401
from .storage import RolloutStorage


def _collective_path():
    """True when gradients go through the flat bucket + all-reduce: more than one rank, or GRX_PPO_FORCE_BUCKET=1 with
    an initialised process group (lets a single-GPU box exercise the RCCL path: the all-reduce is then an identity)."""
    return _world() > 1 or (os.environ.get("GRX_PPO_FORCE_BUCKET") == "1" and dist.is_available() and dist.is_initialized())


def _capture_mode():
    """With a process group alive, ProcessGroupNCCL's watchdog thread polls its events (hipEventQuery) at any time; under the
    default "global" capture mode that call from ANOTHER thread invalidates a capture in progress and takes the process
    down.  "thread_local" confines the legality check to the capturing thread."""
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@contextlib.contextmanager
def _graph_capture(graph, stream):
    """torch.cuda.graph(...) with Python's cyclic garbage collector held off for the duration of the capture.  A collection that
    fires inside a capture -- on this thread, or on autograd's worker thread, which runs the Python backward of FusedPPOLoss /
    _TrainLinear -- may finalise an object that owns device memory (an env handle: grx_destroy -> hipFree / hipHostFree, a torch
    tensor's storage, an event): a free is not a capturable operation, the capture is invalidated and the HIP runtime or the
    autograd thread aborts the process.  Seen once in three runs of the full GPU suite in round 4 (129 tests' worth of garbage
    before the capture in test_rccl_bucket_path_matches_reference); a training job that drops an env while the update graph is being
    built would hit the same."""
    import gc
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode=_capture_mode()):
            yield
    finally:
        if was_enabled:
            gc.enable()


class PPO:
    def __init__(self, actor_critic=None, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, learning_rate_min=1e-5, learning_rate_max=1e-2,
                 weight_decay=0.0, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01,
                 device="cpu", storage_class="RolloutStorage", **kwargs):
        if kwargs:
            print("PPO.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        self.device = device
        self.desired_kl, self.schedule, self.mean_kl = desired_kl, schedule, 0.0
        self.learning_rate, self.learning_rate_min, self.learning_rate_max = learning_rate, learning_rate_min, learning_rate_max
        self.actor_critic = actor_critic.to(device)
        self.storage_class = storage_class
        self.storage, self.transition = None, None
        # On a HIP device the whole update runs without host synchronisation: the adaptive learning rate lives in a
        # device scalar consumed by the fused Adam kernel, the NaN-skip uses Adam's found_inf hook, and the loss
        # statistics are read back once per update().  (The reference does three .item() per minibatch, ppo.py:264,308-309.)
        self._device_lr = torch.device(device).type == "cuda"
        # ... and, single rank, the whole minibatch step (forward, losses, backward, adaptive LR, clip, fused Adam) is
        # captured ONCE in a HIP graph and replayed 200x per update (GRX_PPO_GRAPH=0: eager).  The captured step equals the
        # eager one bit for bit, rollouts in between included (tests/test_ppo_gpu.py) -- provided no torch column reduction
        # is inside it: see rl/modules.py:_TrainLinear.
        self._use_graph = self._device_lr and os.environ.get("GRX_PPO_GRAPH", "1") not in ("0", "")
        if self._use_graph:
            from . import modules
            if modules._TRAIN_LINEAR != "colsum":   # torch's column reduction goes wrong under graph replay
                print("PPO: GRX_PPO_LINEAR=" + modules._TRAIN_LINEAR + " -> the minibatch step runs eagerly (GRX_PPO_GRAPH=0)")
                self._use_graph = False
        # ... and everything between the networks' outputs and their gradients is one HIP kernel (rl/fused_loss.py)
        self._fused_loss = self._device_lr and os.environ.get("GRX_PPO_FUSED_LOSS", "1") != "0"
        self._fused_store = self._device_lr and os.environ.get("GRX_PPO_FUSED_STORE", "1") != "0"
        # ... and everything behind loss.backward() -- adaptive learning rate, NaN-skip, gradient clip, Adam -- is two launches of
        # libgrx_ppo.so (rl/fused_loss.py StepTail) instead of ~20 small torch kernels on the step's critical path (GRX_PPO_FUSED_TAIL=0: torch)
        self._fused_tail = self._device_lr and os.environ.get("GRX_PPO_FUSED_TAIL", "1") != "0"
        self._tail = None
        self._graph, self._graph_mb, self._static, self._sums, self._restore_opt = None, None, None, None, None
        # ... and so is the rollout's policy step (GRX_PPO_ACT_GRAPH=0: eager)
        self._use_act_graph = self._device_lr and os.environ.get("GRX_PPO_ACT_GRAPH", "1") not in ("0", "")
        self._act_graph, self._act_key, self._act_in, self._act_out = None, None, None, None
        # inside the captured step the critic runs on a second stream (GRX_PPO_TWO_STREAMS=0: one stream); eagerly the extra
        # stream bookkeeping costs more than the overlap gives
        self._two_streams = self._use_graph and os.environ.get("GRX_PPO_TWO_STREAMS", "1") != "0"
        self._aux_stream = None
        if self._device_lr:
            # rsl_rl's `Normal.set_default_validate_args = False` (actor_critic.py) is an assignment, not a call, so the
            # reference validates (and host-syncs) on every Normal(); here the validation really is off
            torch.distributions.Distribution.set_default_validate_args(False)
            self._lr_t = torch.tensor(float(learning_rate), device=device)
            self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=self._lr_t, weight_decay=weight_decay, fused=True,
                                        capturable=True)
        else:
            self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate, weight_decay=weight_decay)
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.symmetry_coef = 0
        self.num_updates = 0
        self._params = [p for p in self.actor_critic.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self._params)
        # multi-rank: every .grad is a VIEW into one flat fp32 bucket [grads | minibatch KL | non-finite-loss flag], so the
        # single all-reduce per optimizer step needs no per-parameter copies in either direction
        # (each view starts on a 256-byte boundary like a separately allocated .grad would: torch's multi-tensor kernels pick
        #  their vector width -- and with it the summation order of the gradient norm -- from the pointer alignment)
        n = sum(-(-p.numel() // 64) * 64 for p in self._params)
        self._nflat = n
        self._bucket = torch.zeros(n + 2, device=device) if _collective_path() else None
        self._mid = torch.zeros(2, device=device) if _collective_path() else None   # value / surrogate loss between the two halves
        self._graph_back = None
        if self._bucket is not None:
            self._install_flat_grads()

    def _install_flat_grads(self):
        for p, o in zip(self._params, self._offsets()):
            p.grad = self._bucket[o:o + p.numel()].view_as(p)

    def init_storage(self, num_envs, num_transitions_per_env, **_):
        ac = self.actor_critic
        self.transition = RolloutStorage.Transition()
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, [ac.num_actor_input], [ac.num_critic_input],
                                      [ac.num_actor_output], self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def _act_eager(self, actor_observations, critic_observations, capturable=False):
        ac = self.actor_critic
        if capturable:   # Normal.sample() = torch.normal(mean, std) checks std >= 0 on the host: not capturable
            ac.update_distribution(actor_observations)
            actions = (ac.action_mean + ac.action_std * torch.randn_like(ac.action_mean)).detach()
        else:
            actions = ac.act(actor_observations).detach()
        values = ac.evaluate(critic_observations).detach()
        logp = ac.get_actions_log_prob(actions).detach()
        return actions, values, logp, ac.action_mean.detach(), ac.action_std.detach()

    def _act_graphed(self, actor_observations, critic_observations):
        """The rollout's policy step replayed from HIP graphs on static input / output buffers -- TWO of them: the actor
        (MLP forward, sample, log-prob: what env.step() waits for) on the current stream, the critic (MLP forward: only the
        storage needs it) on a side stream, where it overlaps the env step that follows (the step kernel leaves half of the
        chip idle at 4096 envs).  process_env_step() waits for the critic before it stores the transition.  The outputs are
        consumed (env.step, the storage kernel) before the next call overwrites them.  The sampling draws from torch's default
        generator, which CUDA graphs advance per replay."""
        key = (tuple(actor_observations.shape), tuple(critic_observations.shape))
        ac = self.actor_critic
        cur = torch.cuda.current_stream(self.device)
        if self._act_graph is None or self._act_key != key:
            # (not under inference_mode, where the runner calls act(): tensors created there -- the static buffers, the
            #  generator's graph state -- would be inference tensors that later captures / replays may not update)
            with torch.inference_mode(False), torch.no_grad():
                self._act_in = (torch.zeros_like(actor_observations), torch.zeros_like(critic_observations))

                # GRX_PPO_FUSED_MLP (default on): the two MLP forwards through libgrx_ppo.so -- one MFMA launch per layer with
                # bias + ELU in the epilogue, the actor's output layer fused with the sampling and the log-probability
                # (rl/fused_loss.py: mlp_forward / policy_act) -- instead of 8 library GEMMs, 6 ELU kernels and ~20 small
                # distribution kernels per rollout step
                fuse = (os.environ.get("GRX_PPO_FUSED_MLP", "1") != "0" and mlp_can_fuse(ac.actor, self._act_in[0])
                        and mlp_can_fuse(ac.critic, self._act_in[1]) and ac.num_actor_output <= 32)
                self._act_fused = fuse

                def actor_part():
                    if fuse:
                        eps = torch.randn(self._act_in[0].shape[0], ac.num_actor_output, device=self.device)
                        # (fixed_std: update_distribution -- and so the update's log-prob -- uses init_noise_std, not the parameter)
                        std = ac.init_std.to(ac.std.device).clone() if getattr(ac, "fixed_std", False) else ac.std.detach()
                        return policy_act(ac.actor, std, self._act_in[0], eps)
                    ac.update_distribution(self._act_in[0])   # Normal.sample() checks std >= 0 on the host: not capturable
                    actions = (ac.action_mean + ac.action_std * torch.randn_like(ac.action_mean)).detach()
                    return actions, ac.get_actions_log_prob(actions).detach(), ac.action_mean.detach(), ac.action_std.detach()

                def critic_part():
                    if fuse:
                        return mlp_forward(ac.critic, self._act_in[1])
                    return ac.evaluate(self._act_in[1]).detach()
                side = torch.cuda.Stream(device=self.device)
                self._act_side = torch.cuda.Stream(device=self.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        actor_part(); critic_part()
                cur.wait_stream(side)
                self._act_graph = torch.cuda.CUDAGraph()
                with _graph_capture(self._act_graph, side):
                    a_out = actor_part()
                self._critic_graph = torch.cuda.CUDAGraph()
                with _graph_capture(self._critic_graph, side):
                    v_out = critic_part()
                self._act_out = (a_out[0], v_out, a_out[1], a_out[2], a_out[3])
                self._ev_obs, self._ev_critic = torch.cuda.Event(), torch.cuda.Event()
            self._act_key = key
        self._ev_obs.record(cur)                       # this step's observations are final on the current stream
        self._act_side.wait_event(self._ev_obs)
        with torch.cuda.stream(self._act_side):
            self._act_in[1].copy_(critic_observations)
            self._critic_graph.replay()
            self._ev_critic.record(self._act_side)
        self._critic_pending = True
        self._act_in[0].copy_(actor_observations)
        self._act_graph.replay()
        return self._act_out

    def _join_critic(self):
        """values of the current policy step are ready on the current stream from here on"""
        if getattr(self, "_critic_pending", False):
            torch.cuda.current_stream(self.device).wait_event(self._ev_critic)
            self._critic_pending = False

    def act(self, actor_observations, critic_observations):
        t = self.transition
        if self._use_act_graph and actor_observations.is_cuda and not torch.is_grad_enabled():
            t.actions, t.values, t.actions_log_prob, t.action_mean, t.action_sigma = self._act_graphed(actor_observations, critic_observations)
        else:
            t.actions, t.values, t.actions_log_prob, t.action_mean, t.action_sigma = self._act_eager(actor_observations, critic_observations)
        t.observations, t.critic_observations = actor_observations, critic_observations
        return t.actions

    def act_inference(self, obs):
        return self.actor_critic.act_inference(obs)

    def process_env_step(self, rewards, dones, infos, log=None):
        """ppo.py:184-197.  On a HIP device everything here -- the time-out bootstrap, the nine row copies of
        RolloutStorage.add_transitions and (log = (cur_rew, cur_len, done_rew_row, done_len_row)) the runner's running episode
        reward / length -- is ONE kernel (include/grx_ppo.h grx_ppo_store_transition)."""
        t = self.transition
        st = self.storage
        self._join_critic()
        if self._fused_store and rewards.is_cuda and t.observations is not None and t.observations.is_contiguous():
            if st.step >= st.num_transitions_per_env:
                raise AssertionError("Rollout buffer overflow")
            from .fused_loss import store_transition
            pri = t.critic_observations if st.pri_observations is not None else None
            store_transition(st, st.step, t.observations, pri, t.actions, t.action_mean, t.action_sigma, t.values, t.actions_log_prob,
                             rewards if rewards.is_contiguous() else rewards.contiguous(), dones, infos.get("time_outs"), self.gamma, log)
            st.step += 1
            t.clear()
            self.actor_critic.reset(dones)
            return
        if log is not None:   # the torch spelling of the runner's bookkeeping (on_policy_runner.py:170-181)
            cur_rew, cur_len, done_rew, done_len = log
            cur_rew += rewards; cur_len += 1
            d = dones > 0
            done_rew.copy_(cur_rew); done_len.copy_(cur_len)
            cur_rew *= ~d; cur_len *= ~d
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:   # bootstrap on time-outs (ppo.py:190-191)
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        self._join_critic()
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def update_learning_rate(self, kl_mean):
        if kl_mean > self.desired_kl * 2.0:
            self.learning_rate = max(self.learning_rate_min, self.learning_rate / 1.5)
        elif self.desired_kl / 2.0 > kl_mean > 0.0:
            self.learning_rate = min(self.learning_rate_max, self.learning_rate * 1.5)

    def _sync_gradients(self, kl_mean, bad=None):
        """ONE collective per optimizer step: [flat grads | KL | bad] summed over ranks, averaged.  The gradients already
        live in the bucket (backward accumulated into the views after _zero_flat_grads).  Returns (KL, any-rank-bad)."""
        b, n = self._bucket, self._nflat
        b[n] = kl_mean
        b[n + 1] = 0.0 if bad is None else bad.to(b.dtype)
        dist.all_reduce(b)
        b /= _world()
        return b[n], b[n + 1] > 0

    def _zero_flat_grads(self):
        if any(p.grad is None or p.grad.data_ptr() != self._bucket.data_ptr() + 4 * o for p, o in zip(self._params, self._offsets())):
            self._install_flat_grads()   # (someone ran zero_grad(set_to_none=True) in between)
        self._bucket.zero_()

    def _offsets(self):
        o = 0
        for p in self._params:
            yield o
            o += -(-p.numel() // 64) * 64

    def _device_lr_update(self, kl_mean):
        """update_learning_rate() on the device (same branches as ppo.py:205-213)."""
        lr = self._lr_t
        down = torch.clamp(lr / 1.5, min=self.learning_rate_min)
        up = torch.clamp(lr * 1.5, max=self.learning_rate_max)
        new = torch.where(kl_mean > self.desired_kl * 2.0, down,
                          torch.where((kl_mean < self.desired_kl / 2.0) & (kl_mean > 0.0), up, lr))
        lr.copy_(new)

    def update(self):
        if self._device_lr:
            # For these GEMM shapes (batch ~10^4 rows, 39..512 columns, fp32) rocBLAS's kernel choices beat hipBLASLt's by 2x
            # on the weight-gradient products dY^T X (27-48 us against 66-73 us, tools/gpu_gemm_probe.py).  torch's BLAS
            # preference is process-global, so it is switched for the update only (GRX_PPO_BLAS=hipblaslt leaves it alone).
            prev_blas = torch.backends.cuda.preferred_blas_library()
            if os.environ.get("GRX_PPO_BLAS", "rocblas") == "rocblas":
                torch.backends.cuda.preferred_blas_library("cublas")
            try:
                if self._use_graph:
                    return self._update_graphed()
                return self._update_device()
            finally:
                torch.backends.cuda.preferred_blas_library(prev_blas)
        mean_value_loss, mean_surrogate_loss = 0.0, 0.0
        ac, multi = self.actor_critic, _collective_path()
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        for (obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, _, _) in \
                self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            ac.act(obs)
            logp = ac.get_actions_log_prob(actions)
            value = ac.evaluate(cobs)
            mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
            kl_mean = None
            if adaptive:
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5) + (old_sigma.square() + (old_mu - mu).square())
                                   / (2.0 * sigma.square()) - 0.5, axis=-1)
                    kl_mean = kl.mean()
                if not multi:
                    self._apply_kl(kl_mean.item())
            ratio = torch.exp(logp - torch.squeeze(old_logp))
            adv = torch.squeeze(advantages)
            surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
            if self.use_clipped_value_loss:
                clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value - returns).pow(2), (clipped - returns).pow(2)).mean()
            else:
                value_loss = (returns - value).pow(2).mean()
            loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
            if not multi and torch.isnan(loss):
                continue
            if multi:
                self._zero_flat_grads()
            else:
                self.optimizer.zero_grad()
            loss.backward()
            if multi:
                kl_avg, bad = self._sync_gradients(kl_mean if kl_mean is not None else torch.zeros((), device=self.device),
                                                   ~torch.isfinite(loss.detach()))
                if adaptive:
                    self._apply_kl(kl_avg.item())
                if bool(bad) or not bool(torch.isfinite(self._bucket[:self._nflat]).all()):
                    continue   # NaN-skip must be a collective decision: the averaged bucket is identical on every rank
            nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm)
            self.optimizer.step()
            mean_value_loss += value_loss.item()
            mean_surrogate_loss += surrogate_loss.item()
        self.num_updates = self.num_learning_epochs * self.num_mini_batches
        return mean_value_loss / self.num_updates, mean_surrogate_loss / self.num_updates

    def _losses(self, obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma):
        """(surrogate_loss, value_loss, loss, kl_mean) of one minibatch -- ppo.py:215-245.

        On a HIP device: the actor / critic forward in torch, everything after it in ONE kernel that also produces the
        gradients (rl/fused_loss.py -> libgrx_ppo.so; GRX_PPO_FUSED_LOSS=0 keeps the torch expression below)."""
        ac = self.actor_critic
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        if self._fused_loss and not ac.fixed_std and ac.num_actor_output <= 32:   # the kernel's action-count limit
            if self._two_streams:
                # actor and critic are independent until the loss: the critic's forward (and, through autograd's stream
                # bookkeeping, its backward) runs on a second stream
                cur = torch.cuda.current_stream(self.device)
                if self._aux_stream is None:
                    self._aux_stream = torch.cuda.Stream(device=self.device)
                self._aux_stream.wait_stream(cur)
                with torch.cuda.stream(self._aux_stream):
                    value = ac.evaluate(cobs)
                mu = ac.actor(obs)
                cur.wait_stream(self._aux_stream)
                value.record_stream(cur)
            else:
                mu = ac.actor(obs)
                value = ac.evaluate(cobs)
            out = fused_ppo_loss(mu, ac.std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                                 self.clip_param, self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
            kl_mean = out[3].detach() if adaptive else torch.zeros((), device=self.device)
            return out[0], out[1], out[2], kl_mean
        ac.update_distribution(obs)
        logp = ac.get_actions_log_prob(actions)
        value = ac.evaluate(cobs)
        mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
        kl_mean = torch.zeros((), device=self.device)
        if adaptive:
            with torch.no_grad():
                kl_mean = torch.sum(torch.log(sigma / old_sigma + 1.e-5) + (old_sigma.square() + (old_mu - mu).square())
                                    / (2.0 * sigma.square()) - 0.5, axis=-1).mean()
        ratio = torch.exp(logp - torch.squeeze(old_logp))
        adv = torch.squeeze(advantages)
        surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        if self.use_clipped_value_loss:
            clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).pow(2), (clipped - returns).pow(2)).mean()
        else:
            value_loss = (returns - value).pow(2).mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
        return surrogate_loss, value_loss, loss, kl_mean

    def _update_device(self):
        """Same arithmetic as update(), no host round-trips inside the minibatch loop."""
        ac, multi = self.actor_critic, _collective_path()
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        sums = torch.zeros(3, device=self.device)        # value loss, surrogate loss, last KL
        for (obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, _, _) in \
                self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            surrogate_loss, value_loss, loss, kl_mean = self._losses(obs, cobs, actions, target_values, advantages, returns,
                                                                      old_logp, old_mu, old_sigma)
            if multi:
                self._zero_flat_grads()
            else:
                self.optimizer.zero_grad(set_to_none=False)
            loss.backward()
            if not multi and self._step_tail(loss, kl_mean, value_loss, surrogate_loss, sums, adaptive):
                continue
            with torch.no_grad():
                bad = ~torch.isfinite(loss)
                if multi:
                    kl_mean, bad = self._sync_gradients(kl_mean, bad)
                    bad = bad | ~torch.isfinite(self._bucket[:self._nflat]).all()
            if multi and self._step_tail(value_loss, kl_mean, value_loss, surrogate_loss, sums, adaptive, bad_flag=bad.float()):
                continue
            if adaptive:
                self._device_lr_update(kl_mean)
            with torch.no_grad():
                # NaN-skip (ppo.py:297-299) without a sync: the fused Adam kernel leaves parameters and moments
                # untouched when found_inf is set (the GradScaler hook)
                self.optimizer.found_inf = bad.float().reshape(())
                self.optimizer.grad_scale = None
            nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm, foreach=True)
            self.optimizer.step()
            with torch.no_grad():
                ok = (~bad).float()
                sums[0] += value_loss.detach() * ok
                sums[1] += surrogate_loss.detach() * ok
                sums[2] = kl_mean
        self.num_updates = self.num_learning_epochs * self.num_mini_batches
        host = sums.tolist()                           # the only device->host transfer of the update
        self.mean_kl = host[2]
        self.learning_rate = float(self._lr_t.item())
        return host[0] / self.num_updates, host[1] / self.num_updates

    # ------------------------------------------------------------------ HIP-graph minibatch step
    def _minibatch_step(self, batch, sums):
        """One PPO minibatch step on static tensors (the arithmetic of _update_device)."""
        obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma = batch
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        surrogate_loss, value_loss, loss, kl_mean = self._losses(obs, cobs, actions, target_values, advantages, returns,
                                                                  old_logp, old_mu, old_sigma)
        # grads dropped, not zero-filled: backward then WRITES each .grad (from the graph's private pool on replay) instead of
        # accumulating into a zeroed one -- one fill and one add kernel less per parameter and step (GRX_PPO_GRAD_NONE=0: fill)
        self.optimizer.zero_grad(set_to_none=os.environ.get("GRX_PPO_GRAD_NONE", "1") != "0")
        loss.backward()
        if self._step_tail(loss, kl_mean, value_loss, surrogate_loss, sums, adaptive):
            return
        if adaptive:
            self._device_lr_update(kl_mean)
        with torch.no_grad():
            bad = ~torch.isfinite(loss)
            self.optimizer.found_inf = bad.float().reshape(())   # NaN-skip (ppo.py:297-299) through fused Adam's hook
            self.optimizer.grad_scale = None
        nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm, foreach=True)
        self.optimizer.step()
        with torch.no_grad():
            ok = (~bad).float()
            sums[0] += value_loss.detach() * ok
            sums[1] += surrogate_loss.detach() * ok
            sums[2] = kl_mean

    def _step_tail(self, loss, kl_mean, value_loss, surrogate_loss, sums, adaptive, bad_flag=None):
        """Adaptive learning rate, NaN-skip, clip_grad_norm_, Adam.step() and the update's statistics through libgrx_ppo.so (two launches);
        False: not available here (GRX_PPO_FUSED_TAIL=0, an optimizer configuration it does not cover) -- the caller runs the torch tail."""
        if not self._fused_tail:
            return False
        if self._tail is None:
            from .fused_loss import StepTail
            if not StepTail.supported(self.optimizer, self._params) or any(p.grad is None for p in self._params):
                self._fused_tail = False
                return False
            self._tail = StepTail(self.optimizer, self._params, self._lr_t)
        with torch.no_grad():
            self._tail(loss.detach(), kl_mean.detach(), value_loss.detach(), surrogate_loss.detach(), sums, adaptive, self.desired_kl,
                       self.learning_rate_min, self.learning_rate_max, self.max_grad_norm, bad_flag=bad_flag)
        return True

    # ... and with more than one rank the step is captured as TWO halves around the one eager RCCL all-reduce:
    #   front  = zero the flat bucket, forward, fused loss, backward (accumulating into the bucket's views), KL and the
    #            non-finite flag into the bucket's tail
    #   (dist.all_reduce(bucket) -- enqueued on the same stream between the two replays; no host synchronisation)
    #   back   = average, adaptive learning rate from the AVERAGED KL, collective NaN-skip, clip, fused Adam, statistics
    def _mb_front(self, batch):
        obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma = batch
        surrogate_loss, value_loss, loss, kl_mean = self._losses(obs, cobs, actions, target_values, advantages, returns,
                                                                  old_logp, old_mu, old_sigma)
        self._zero_flat_grads()
        loss.backward()
        with torch.no_grad():
            b, n = self._bucket, self._nflat
            b[n] = kl_mean
            b[n + 1] = (~torch.isfinite(loss)).float()
            self._mid[0] = value_loss.detach()
            self._mid[1] = surrogate_loss.detach()

    def _mb_back(self, sums):
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        b, n = self._bucket, self._nflat
        with torch.no_grad():
            b /= _world()
            kl_mean = b[n]
            bad = (b[n + 1] > 0) | ~torch.isfinite(b[:n]).all()
        if self._step_tail(self._mid[0], kl_mean, self._mid[0], self._mid[1], sums, adaptive, bad_flag=bad.float()):
            return   # (the collective NaN decision travels as the flag; _mid[0], the value loss, stands in for the total loss: not finite -> bad anyway)
        if adaptive:
            self._device_lr_update(kl_mean)
        with torch.no_grad():
            self.optimizer.found_inf = bad.float().reshape(())
            self.optimizer.grad_scale = None
        nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm, foreach=True)
        self.optimizer.step()
        with torch.no_grad():
            ok = (~bad).float()
            sums[0] += self._mid[0] * ok
            sums[1] += self._mid[1] * ok
            sums[2] = kl_mean

    def _build_graph(self, mb):
        st, dev = self.storage, self.device
        widths = [st.observations.shape[-1], (st.pri_observations if st.pri_observations is not None else st.observations).shape[-1],
                  st.actions.shape[-1], 1, 1, 1, 1, st.mu.shape[-1], st.sigma.shape[-1]]
        self._static = [torch.zeros(mb, w, device=dev) for w in widths]
        self._sums = torch.zeros(3, device=dev)
        # warm-up runs real optimizer steps (allocator / lazy-state initialisation): snapshot and restore everything they touch
        ac_state = [p.detach().clone() for p in self.actor_critic.parameters()]   # (not load_state_dict: it rewrites std, AC:116-134)
        lr0 = self._lr_t.clone()
        self._static[8].fill_(1.0)   # sigma > 0 for the dry runs
        # The GEMM kernels are chosen when the graph is captured.  For these shapes (batch ~10^4 rows, 39..512 columns,
        # fp32) rocBLAS's choices beat hipBLASLt's by 2x on the weight-gradient products dY^T X (27-48 us against
        # 66-73 us, tools/gpu_gemm_probe.py).  The preference is process-global in torch, so it is switched for the
        # dry runs (the library initialises itself there, outside the capture) and the capture only, then put back.
        prev_blas = torch.backends.cuda.preferred_blas_library()
        if os.environ.get("GRX_PPO_BLAS", "rocblas") == "rocblas":
            torch.backends.cuda.preferred_blas_library("cublas")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        multi = _collective_path()
        with torch.cuda.stream(side):
            for _ in range(3):
                if multi:   # (every rank builds its graphs at the same point of the run: the dry-run collectives pair up)
                    self._mb_front(self._static)
                    dist.all_reduce(self._bucket)
                    self._mb_back(self._sums)
                else:
                    self._minibatch_step(self._static, self._sums)
        torch.cuda.current_stream(dev).wait_stream(side)
        if os.environ.get("GRX_PPO_GRAPH", "1") == "2":   # debugging aid: static buffers, eager replay
            class _Eager:
                def __init__(s, fn): s.fn = fn
                def replay(s): s.fn()
            if multi:
                self._graph = _Eager(lambda: self._mb_front(self._static))
                self._graph_back = _Eager(lambda: self._mb_back(self._sums))
            else:
                self._graph = _Eager(lambda: self._minibatch_step(self._static, self._sums))
        elif multi:
            self._graph, self._graph_back = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with _graph_capture(self._graph, side):
                self._mb_front(self._static)
            with _graph_capture(self._graph_back, side):
                self._mb_back(self._sums)
        else:
            self._graph = torch.cuda.CUDAGraph()
            # capture on the stream the dry runs used: autograd's AccumulateGrad nodes remember the stream they were
            # created on, and one that differs from the capture stream runs (and allocates) outside the capture
            with _graph_capture(self._graph, side):
                self._minibatch_step(self._static, self._sums)
        torch.backends.cuda.preferred_blas_library(prev_blas)
        # restore: parameters, Adam moments/step counters, learning rate
        with torch.no_grad():
            for p, v in zip(self.actor_critic.parameters(), ac_state):
                p.copy_(v)
        for stt in self.optimizer.state.values():
            for v in stt.values():
                if torch.is_tensor(v):
                    v.zero_()
        self._lr_t.copy_(lr0)
        if self._restore_opt is not None:
            self._load_opt_tensors(self._restore_opt)
        self._graph_mb = mb

    def _opt_tensors(self):
        return [{k: v.clone() for k, v in stt.items() if torch.is_tensor(v)} for stt in self.optimizer.state.values()]

    def _load_opt_tensors(self, saved):
        for stt, sv in zip(self.optimizer.state.values(), saved):
            for k, v in sv.items():
                stt[k].copy_(v)

    def _update_graphed(self):
        st = self.storage
        batch = st.num_envs * st.num_transitions_per_env
        mb = batch // self.num_mini_batches
        if self._graph is None or self._graph_mb != mb:
            # capture after the optimizer already ran (resume / later iterations): keep its moments
            self._restore_opt = self._opt_tensors() if len(self.optimizer.state) else None
            self._build_graph(mb)
        flat = lambda x: x.flatten(0, 1)
        cobs = st.pri_observations if st.pri_observations is not None else st.observations
        srcs = [flat(x) for x in (st.observations, cobs, st.actions, st.values, st.advantages, st.returns, st.actions_log_prob, st.mu, st.sigma)]
        indices = torch.randperm(self.num_mini_batches * mb, requires_grad=False, device=self.device)   # RS:63-112: one permutation, reused
        self._sums.zero_()
        multi = _collective_path()
        gather = None
        if self._fused_store:   # (libgrx_ppo.so is in use)
            key = tuple(s.data_ptr() for s in srcs) + tuple(b.data_ptr() for b in self._static)
            if getattr(self, "_gather_key", None) != key:
                from .fused_loss import RowGather
                self._gather, self._gather_key = RowGather(srcs, self._static), key
            gather = self._gather
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                if gather is not None:
                    gather(idx)   # the nine minibatch tensors in one launch
                else:
                    for buf, src in zip(self._static, srcs):
                        torch.index_select(src, 0, idx, out=buf)
                self._graph.replay()
                if multi:
                    dist.all_reduce(self._bucket)
                    self._graph_back.replay()
        self.num_updates = self.num_learning_epochs * self.num_mini_batches
        host = self._sums.tolist()                     # the only device->host transfer of the update
        self.mean_kl = host[2]
        self.learning_rate = float(self._lr_t.item())
        return host[0] / self.num_updates, host[1] / self.num_updates

    def _apply_kl(self, kl_value):
        self.mean_kl = kl_value
        self.update_learning_rate(kl_value)
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate

    def load_optimizer_state(self, state_dict):
        """optimizer.load_state_dict() for the device-resident update.  torch replaces param_groups[*]['lr'] with a NEW tensor
        (or, from a reference rsl_rl checkpoint, a float with capturable=False / fused=None): the adaptive learning rate would
        then be written to an orphaned `_lr_t`, the NaN-skip hook ignored, and graphs captured earlier would keep updating the
        old Adam moments.  Re-attach `_lr_t`, force the fused capturable configuration, move the step counters to the device
        and drop every captured graph."""
        self.optimizer.load_state_dict(state_dict)
        if self._device_lr:
            lr = self.optimizer.param_groups[0]["lr"]
            with torch.no_grad():
                self._lr_t.copy_(torch.as_tensor(float(lr), device=self.device))
            for g in self.optimizer.param_groups:
                g["lr"] = self._lr_t
                g["fused"], g["capturable"], g["foreach"] = True, True, False
            for stt in self.optimizer.state.values():
                for k, v in list(stt.items()):
                    if torch.is_tensor(v) and (v.device != self._lr_t.device or (k == "step" and v.dtype != torch.float32)):
                        stt[k] = v.to(device=self.device, dtype=torch.float32 if k == "step" else v.dtype)
                    elif k == "step" and not torch.is_tensor(v):
                        stt[k] = torch.tensor(float(v), device=self.device)
            self.learning_rate = float(lr)
        else:
            self.learning_rate = float(self.optimizer.param_groups[0]["lr"])
        self.invalidate_graphs()

    def invalidate_graphs(self):
        """captured policy / minibatch graphs refer to the tensors they were captured with: recapture on next use"""
        self._graph, self._graph_back, self._graph_mb, self._static, self._restore_opt = None, None, None, None, None
        self._act_graph, self._act_key, self._critic_pending = None, None, False

    def clear_storage(self):
        self.storage.clear()
