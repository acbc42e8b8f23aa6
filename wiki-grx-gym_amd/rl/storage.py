"""Rollout storage with rsl_rl semantics (reference: rsl_rl/storage/base_storage.py:27-198,
rollout_storage.py:23-112): (T, N, .) buffers, reverse GAE scan, advantages normalised with the
GLOBAL mean / unbiased std, one randperm reused for every epoch, tail of the batch dropped.

Multi-GPU: with envs sharded over ranks the advantage moments must be global: the three
sums (n, sum, sum of squares) are all-reduced -- 12 bytes per iteration."""
import torch
import torch.distributed as dist


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.clear()

        def clear(self):
            self.observations = self.critic_observations = self.actions = self.rewards = self.dones = None
            self.values = self.actions_log_prob = self.action_mean = self.action_sigma = None

    def __init__(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, actions_shape, device, **kwargs):
        if kwargs:
            print("RolloutStorage.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        self.device = device
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, **k: torch.zeros(T, N, *s, device=device, **k)
        self.observations = z(*actor_obs_shape)
        self.pri_observations = z(*critic_obs_shape) if critic_obs_shape[0] is not None else None
        self.actions, self.mu, self.sigma = z(*actions_shape), z(*actions_shape), z(*actions_shape)
        self.rewards, self.values, self.returns, self.advantages, self.actions_log_prob = z(1), z(1), z(1), z(1), z(1)
        self.dones = z(1, dtype=torch.uint8)
        self.num_transitions_per_env, self.num_envs, self.step = T, N, 0

    def add_transitions(self, t):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        i = self.step
        self.observations[i].copy_(t.observations)
        if self.pri_observations is not None:
            self.pri_observations[i].copy_(t.critic_observations)
        self.actions[i].copy_(t.actions)
        self.rewards[i].copy_(t.rewards.view(-1, 1))
        self.dones[i].copy_(t.dones.view(-1, 1))
        self.values[i].copy_(t.values)
        self.actions_log_prob[i].copy_(t.actions_log_prob.view(-1, 1))
        self.mu[i].copy_(t.action_mean)
        self.sigma[i].copy_(t.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        adv = 0
        for step in reversed(range(self.num_transitions_per_env)):
            nxt = last_values if step == self.num_transitions_per_env - 1 else self.values[step + 1]
            alive = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + alive * gamma * nxt - self.values[step]
            adv = delta + alive * gamma * lam * adv
            self.returns[step] = adv + self.values[step]
        a = self.returns - self.values
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            m = torch.stack([torch.tensor(float(a.numel()), device=a.device), a.sum(), (a * a).sum()]).double()
            dist.all_reduce(m)
            n, mean = m[0], m[1] / m[0]
            var = (m[2] - n * mean * mean) / (n - 1)          # unbiased, like torch.std
            self.advantages = (a - mean.float()) / (var.clamp_min(0).sqrt().float() + 1e-8)
        else:
            self.advantages = (a - a.mean()) / (a.std() + 1e-8)

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        batch = self.num_envs * self.num_transitions_per_env
        mb = batch // num_mini_batches
        indices = torch.randperm(num_mini_batches * mb, requires_grad=False, device=self.device)
        flat = lambda x: x.flatten(0, 1)
        obs = flat(self.observations)
        cobs = flat(self.pri_observations) if self.pri_observations is not None else obs
        cols = [flat(x) for x in (self.actions, self.values, self.advantages, self.returns, self.actions_log_prob, self.mu, self.sigma)]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = indices[i * mb:(i + 1) * mb]
                yield (obs[idx], cobs[idx], *[c[idx] for c in cols], (None, None), None)
