"""Actor-critic MLP policy with the semantics of rsl_rl's ActorCriticMLP (reference:
rsl_rl/modules/actor_critic_mlp.py:10-231, mlp.py:7-42): separate actor / critic MLPs with ELU
(or the named activation) between hidden layers and a linear output, one learnable std per action,
``Normal(mean, mean*0 + std)``, log-prob / entropy summed over actions, default PyTorch init.

State-dict layout (``actor.model.<i>.weight`` ..., ``critic.model.<i>...``, ``std``) equals the
reference's so checkpoints (``model_<it>.pt``) interchange; ``load_state_dict`` reproduces the
reference quirk of overwriting ``std`` with ``set_noise_std`` unless ``set_std=False``
(actor_critic_mlp.py:116-134)."""
import torch
import torch.nn as nn
from torch.distributions import Normal

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
                "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def get_activation(name):
    """rsl_rl/utils/utils.py:231-254"""
    if name not in _ACTIVATIONS:
        print("invalid activation function!")
        return None
    return _ACTIVATIONS[name]()


import os

_FUSED_ELU_FORWARD = os.environ.get("GRX_PPO_FUSED_FWD", "1") != "0"   # ... and their forward through grx_mlp_layer (bias + ELU epilogue)
_FUSED_ELU_BACKWARD = os.environ.get("GRX_PPO_FUSED_ELU", "1") != "0"   # hidden layers: ELU backward + bias gradient in one pass (_TrainLinearELU)
_TRAIN_LINEAR = os.environ.get("GRX_PPO_LINEAR", "colsum")   # "torch": plain nn.Linear autograd on a HIP device too


class _TrainLinear(torch.autograd.Function):
    """nn.Linear as PPO trains it on a HIP device: y = x W^T + b, with two departures from torch's autograd formula.

    * The bias gradient is libgrx_ppo.so's deterministic column sum (`fused_loss.colsum`), not torch's column reduction:
      replayed from a captured HIP graph with other GPU work between replays, at::native's reduce kernel returned garbage
      for one [10485, 128] -> [128] case (the critic's last hidden bias) while every other tensor matched an fp64
      reference to 1e-7 (tools/gpu_ppo_graph_check.py).  With this function the captured step and the eager step agree
      bit for bit, rollouts in between included.
    * A one-column layer (the value head) pins its three GEMV-shaped products to hipBLASLt: PPO prefers rocBLAS for the
      duration of update() (2x faster weight-gradient GEMMs at these shapes), but rocBLAS's pick for [1, B] x [B, H]
      takes 570 us at B = 10^4 against hipBLASLt's 32 us (tools/gpu_gemm_probe.py).  torch's BLAS preference is a
      process-global flag read at call time, so it is flipped around those products only."""

    @staticmethod
    def _blas(pin):
        prev = torch.backends.cuda.preferred_blas_library()
        if pin:
            torch.backends.cuda.preferred_blas_library("cublaslt")
        return prev

    @staticmethod
    def forward(ctx, x, weight, bias):
        prev = _TrainLinear._blas(weight.shape[0] == 1)
        try:
            y = torch.addmm(bias, x, weight.t())
        finally:
            torch.backends.cuda.preferred_blas_library(prev)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .fused_loss import colsum
        x, weight = ctx.saved_tensors
        prev = _TrainLinear._blas(weight.shape[0] == 1)
        try:
            dx = dy @ weight if ctx.needs_input_grad[0] else None   # (the first layer's input is data: no dX product)
            return dx, dy.t() @ x, colsum(dy)
        finally:
            torch.backends.cuda.preferred_blas_library(prev)


class _TrainLinearELU(torch.autograd.Function):
    """ELU(x W^T + b) of a hidden layer as PPO trains it on a HIP device: _TrainLinear's products, with the ELU's backward and the
    bias gradient in ONE pass over dY (fused_loss.elu_backward_colsum: dZ = dY * ELU'(Y) from the saved OUTPUT, column sums of dZ)
    instead of an elu_backward launch followed by the column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if _FUSED_ELU_FORWARD and x.is_contiguous() and weight.is_contiguous():
            from .fused_loss import linear_elu
            y = linear_elu(x, weight, bias)   # libgrx_ppo.so: f32 MFMA, bias + ELU in the epilogue (one launch)
        else:
            prev = _TrainLinear._blas(weight.shape[0] == 1)
            try:
                y = torch.nn.functional.elu(torch.addmm(bias, x, weight.t()))
            finally:
                torch.backends.cuda.preferred_blas_library(prev)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .fused_loss import elu_backward_colsum
        x, weight, y = ctx.saved_tensors
        dz, db = elu_backward_colsum(dy, y)
        prev = _TrainLinear._blas(weight.shape[0] == 1)
        try:
            dx = dz @ weight if ctx.needs_input_grad[0] else None   # (the first layer's input is data: no dX product)
            return dx, dz.t() @ x, db
        finally:
            torch.backends.cuda.preferred_blas_library(prev)


class MLP(nn.Module):
    def __init__(self, input_size, output_size, hidden_dims=(256, 256, 256), activation="relu", **_):
        super().__init__()
        self.input_size, self.output_size, self.hidden_dims = input_size, output_size, list(hidden_dims)
        dims = [input_size] + list(hidden_dims)
        layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [nn.Linear(a, b), get_activation(activation)]
        layers.append(nn.Linear(dims[-1], output_size))
        self.model = nn.Sequential(*layers)

    @torch.jit.unused
    def _forward_train(self, x):
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU) and mods[i + 1].alpha == 1.0 \
                    and _FUSED_ELU_BACKWARD:
                x = _TrainLinearELU.apply(x, m.weight, m.bias)
                i += 2
                continue
            x = _TrainLinear.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
            i += 1
        return x

    def forward(self, x):
        if not torch.jit.is_scripting():   # (export_policy_as_jit scripts this module: the plain path)
            if x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.float32 and _TRAIN_LINEAR == "colsum":
                return self._forward_train(x)   # training on a HIP device: see _TrainLinear
        return self.model(x)


class ActorCriticMLP(nn.Module):
    def __init__(self, actor_num_input, critic_num_input, actor_num_output, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation="elu", fixed_std=False, init_noise_std=1.0,
                 set_std=True, set_noise_std=1.0, actor_output_activation=None, critic_output_activation=None, actor_output_gain=1.0, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        super().__init__()
        self.num_actor_input, self.num_actor_output = actor_num_input, actor_num_output
        self.num_critic_input, self.num_critic_output = critic_num_input, 1
        self.actor = MLP(actor_num_input, actor_num_output, actor_hidden_dims, activation)
        self.critic = MLP(critic_num_input, 1, critic_hidden_dims, activation)
        if actor_output_gain != 1.0:   # (not in the reference, whose layers keep PyTorch's default init: 1.0.  The build-defined 32-DOF task starts its
            with torch.no_grad():      #  policy near the zero action -- the PD targets' default pose --: envs/config.py GR1T1FullBodyCfgPPO)
                last = [m for m in self.actor.model if isinstance(m, nn.Linear)][-1]
                last.weight.mul_(float(actor_output_gain)); last.bias.mul_(float(actor_output_gain))
        self.fixed_std, self.init_noise_std = fixed_std, init_noise_std
        # (a number, as in the reference -- actor_critic_mlp.py -- or one value per action: the 32-DOF task starts its upper-body joints quieter)
        self.std = nn.Parameter(torch.as_tensor(init_noise_std, dtype=torch.float32) * torch.ones(actor_num_output))
        # (a scalar or one value per action -- GR1T1FullBodyCfgPPO --: a buffer, so that the fixed_std paths see a tensor on the module's device)
        self.register_buffer("init_std", torch.as_tensor(init_noise_std, dtype=torch.float32) * torch.ones(actor_num_output), persistent=False)
        self.set_std, self.set_noise_std = set_std, set_noise_std
        self.distribution = None
        Normal.set_default_validate_args = False

    is_recurrent = False

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        if self.set_std:
            state_dict["std"] = torch.ones_like(state_dict["std"]) * self.set_noise_std
        # (the reference rebinds self.std.data here, actor_critic_mlp.py:116-134; copying IN PLACE gives the same values and keeps
        #  the storage the captured act / update graphs and the optimizer state point at)
        if self.fixed_std:
            with torch.no_grad():
                self.std.copy_(torch.as_tensor(self.init_noise_std, dtype=torch.float32) * torch.ones_like(self.std))
            self.std.requires_grad = False
            state_dict["std"] = self.std.detach().clone()
        return super().load_state_dict(state_dict, strict)

    def reset(self, dones=None):
        pass

    def forward(self, *a, **k):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        std = self.init_std.to(mean.device) if self.fixed_std else self.std.to(mean.device)
        self.distribution = Normal(mean, mean * 0.0 + std)

    def act(self, observations, **_):
        self.update_distribution(observations)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations=None, **_):
        return self.critic(critic_observations)
