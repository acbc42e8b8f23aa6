"""PPO minibatch loss through libgrx_ppo.so (include/grx_ppo.h): forward value and gradients from one HIP kernel.

The expression is rsl_rl/algorithms/ppo.py:215-245 with torch.distributions.Normal's log_prob / entropy; the
torch spelling of the same arithmetic lives in ppo.py (`_loss_torch`) and is what CPU tensors use.  On a GPU the
~100 element-wise kernels that expression and its autograd backward expand to are one launch plus a 64-thread
finalize; the result enters autograd through `FusedPPOLoss`, whose backward hands out the stored gradients.
"""
import ctypes as C
import os

import torch

_LIB = None


def load_ppo_library():
    """libgrx_ppo.so next to the step library (built in-tree by __graft_entry__.build() / make -C csrc)."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("GRX_PPO_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "libgrx_ppo.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no silent fallback for CUDA tensors; set GRX_PPO_FUSED_LOSS=0 to use the torch expression)")
        lib = C.CDLL(path)
        fp = C.c_void_p
        lib.grx_ppo_loss.restype = C.c_int
        lib.grx_ppo_loss.argtypes = [C.c_int, C.c_int] + [fp] * 10 + [C.c_float, C.c_float, C.c_float, C.c_int] + [fp] * 5 + [C.c_void_p]
        lib.grx_ppo_loss_partials_size.restype = C.c_int
        lib.grx_ppo_loss_partials_size.argtypes = [C.c_int]
        lib.grx_ppo_colsum.restype = C.c_int
        lib.grx_ppo_colsum.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_void_p]
        lib.grx_ppo_colsum_partials_size.restype = C.c_int
        lib.grx_ppo_colsum_partials_size.argtypes = [C.c_int, C.c_int]
        lib.grx_ppo_store_transition.restype = C.c_int
        lib.grx_ppo_store_transition.argtypes = [C.c_int] * 4 + [fp] * 10 + [C.c_float] + [fp] * 13 + [C.c_void_p]
        lib.grx_ppo_elu_backward_colsum.restype = C.c_int
        lib.grx_ppo_elu_backward_colsum.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_void_p]
        lib.grx_ppo_gather_rows.restype = C.c_int
        lib.grx_ppo_gather_rows.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), fp, C.c_int, C.c_void_p]
        lib.grx_mlp_layer.restype = C.c_int
        lib.grx_mlp_layer.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_int, C.c_void_p]
        lib.grx_mlp_policy_head.restype = C.c_int
        lib.grx_mlp_policy_head.argtypes = [C.c_int] * 3 + [fp] * 9 + [C.c_void_p]
        lib.grx_ppo_step_tail.restype = C.c_int
        lib.grx_ppo_step_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.grx_ppo_step_tail_blocks.restype = C.c_int
        lib.grx_ppo_step_tail_blocks.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


TAIL_MAX = 24


class _TailTensors(C.Structure):   # include/grx_ppo.h grx_ppo_tail_tensors
    _fields_ = [("n", C.c_int), ("pad", C.c_int), ("param", C.c_void_p * TAIL_MAX), ("grad", C.c_void_p * TAIL_MAX),
                ("exp_avg", C.c_void_p * TAIL_MAX), ("exp_avg_sq", C.c_void_p * TAIL_MAX), ("step", C.c_void_p * TAIL_MAX),
                ("numel", C.c_longlong * TAIL_MAX)]


class _TailArgs(C.Structure):      # grx_ppo_tail_args
    _fields_ = [("loss", C.c_void_p), ("bad_flag", C.c_void_p), ("kl", C.c_void_p), ("lr", C.c_void_p), ("value_loss", C.c_void_p),
                ("surrogate_loss", C.c_void_p), ("sums", C.c_void_p), ("partials", C.c_void_p), ("adaptive", C.c_int), ("pad", C.c_int),
                ("desired_kl", C.c_float), ("lr_min", C.c_float), ("lr_max", C.c_float), ("max_grad_norm", C.c_float),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


class StepTail:
    """grx_ppo_step_tail for one torch.optim.Adam (fused, capturable, one param group, no weight decay / amsgrad / maximize): the adaptive
    learning rate, the NaN-skip, clip_grad_norm_ and Adam.step() of a PPO minibatch step in two launches.  The optimizer's own state tensors
    (exp_avg, exp_avg_sq, step) are updated in place -- created here, as torch creates them lazily, if the optimizer has not stepped yet --,
    so optimizer.state_dict() / load_state_dict() and checkpoints are those of the torch path."""

    @staticmethod
    def supported(optimizer, params):
        g = optimizer.param_groups
        return (len(g) == 1 and len(params) <= TAIL_MAX and not g[0].get("amsgrad") and not g[0].get("maximize") and g[0].get("weight_decay", 0) == 0
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params))

    def __init__(self, optimizer, params, lr_t):
        self.lib = load_ppo_library()
        self.opt, self.params, self.lr_t = optimizer, list(params), lr_t
        self.t, self.a = _TailTensors(), _TailArgs()
        self.key, self.partials = None, None

    def _state(self, p):
        st = self.opt.state[p]
        if "exp_avg" not in st:   # torch's lazy initialisation (optim/adam.py _init_group), capturable / fused: the step counter is a device scalar
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def __call__(self, loss, kl, value_loss, surrogate_loss, sums, adaptive, desired_kl, lr_min, lr_max, max_grad_norm, bad_flag=None):
        t, a = self.t, self.a
        sts = [self._state(p) for p in self.params]
        key = tuple(x.data_ptr() for p, st in zip(self.params, sts) for x in (p, st["exp_avg"], st["exp_avg_sq"], st["step"]))
        if key != self.key:
            t.n = len(self.params)
            for i, (p, st) in enumerate(zip(self.params, sts)):
                if not (st["step"].is_cuda and st["step"].dtype == torch.float32 and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("StepTail: the optimizer state is not torch's fused / capturable layout")
                t.param[i], t.exp_avg[i], t.exp_avg_sq[i], t.step[i], t.numel[i] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel()
            nb = self.lib.grx_ppo_step_tail_blocks(C.byref(t))
            if nb < 1:
                raise RuntimeError("grx_ppo_step_tail_blocks failed")
            self.partials = torch.empty(nb, device=self.params[0].device, dtype=torch.float32)
            self.key = key
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                raise RuntimeError("StepTail: every parameter needs a contiguous fp32 .grad")
            t.grad[i] = g.data_ptr()
        g0 = self.opt.param_groups[0]
        a.loss, a.kl, a.lr = loss.data_ptr(), kl.data_ptr(), self.lr_t.data_ptr()
        a.bad_flag = bad_flag.data_ptr() if bad_flag is not None else None
        a.value_loss, a.surrogate_loss = value_loss.data_ptr(), surrogate_loss.data_ptr()
        a.sums = sums.data_ptr() if sums is not None else None
        a.partials = self.partials.data_ptr()
        a.adaptive = int(bool(adaptive))
        a.desired_kl, a.lr_min, a.lr_max, a.max_grad_norm = float(desired_kl or 0.0), float(lr_min), float(lr_max), float(max_grad_norm)
        a.beta1, a.beta2, a.eps = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"])
        dev = self.params[0].device
        with torch.cuda.device(dev):
            rc = self.lib.grx_ppo_step_tail(C.byref(t), C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"grx_ppo_step_tail failed ({rc})")


def _f32c(x):
    return x if (x.dtype == torch.float32 and x.is_contiguous()) else x.contiguous().float()


class FusedPPOLoss(torch.autograd.Function):
    """(mu [B, A], std [A], value [B] or [B, 1]; data...) -> tensor [surrogate, value_loss, total_loss, mean_kl]."""

    @staticmethod
    def forward(ctx, mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss):
        lib = load_ppo_library()
        B, A = mu.shape
        mu_c, std_c, value_c = _f32c(mu), _f32c(std), _f32c(value).reshape(-1)
        data = [_f32c(t) for t in (actions, old_logp, old_mu, old_sigma, advantages, returns, target_values)]
        out = torch.empty(4, device=mu.device, dtype=torch.float32)
        d_mu = torch.empty_like(mu_c)
        d_std = torch.empty_like(std_c)
        d_value = torch.empty_like(value_c)
        partials = torch.empty(lib.grx_ppo_loss_partials_size(B), device=mu.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(mu.device).cuda_stream
        with torch.cuda.device(mu.device):
            rc = lib.grx_ppo_loss(B, A, mu_c.data_ptr(), std_c.data_ptr(), value_c.data_ptr(), *[t.data_ptr() for t in data],
                                  float(clip_param), float(value_loss_coef), float(entropy_coef), int(bool(use_clipped_value_loss)),
                                  out.data_ptr(), d_mu.data_ptr(), d_std.data_ptr(), d_value.data_ptr(), partials.data_ptr(), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"grx_ppo_loss failed ({rc}): batch {B}, num_actions {A}")
        ctx.save_for_backward(d_mu, d_std, d_value)
        ctx.value_shape = value.shape
        return out

    @staticmethod
    def backward(ctx, g):
        d_mu, d_std, d_value = ctx.saved_tensors
        gl = g[2]   # only the total loss is differentiable
        return (d_mu * gl, d_std * gl, (d_value * gl).reshape(ctx.value_shape)) + (None,) * 11


def fused_ppo_loss(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                   clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss):
    return FusedPPOLoss.apply(mu, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, target_values,
                              clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss)


def colsum(x):
    """Sum over dim 0 of a CUDA fp32 matrix [rows, cols] through grx_ppo_colsum (deterministic; HIP-graph safe)."""
    lib = load_ppo_library()
    x = _f32c(x)
    rows, cols = x.shape
    out = torch.empty(cols, device=x.device, dtype=torch.float32)
    partials = torch.empty(lib.grx_ppo_colsum_partials_size(rows, cols), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        rc = lib.grx_ppo_colsum(rows, cols, x.data_ptr(), out.data_ptr(), partials.data_ptr(),
                                C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"grx_ppo_colsum failed ({rc}): {rows} x {cols}")
    return out


def elu_backward_colsum(dy, y):
    """(dz, db): dz = dy * ELU'(y) from the layer's output y, db = dz.sum(0) -- one pass (grx_ppo_elu_backward_colsum)"""
    lib = load_ppo_library()
    dy, y = _f32c(dy), _f32c(y)
    rows, cols = dy.shape
    dz = torch.empty_like(dy)
    out = torch.empty(cols, device=dy.device, dtype=torch.float32)
    partials = torch.empty(lib.grx_ppo_colsum_partials_size(rows, cols), device=dy.device, dtype=torch.float32)
    with torch.cuda.device(dy.device):
        rc = lib.grx_ppo_elu_backward_colsum(rows, cols, dy.data_ptr(), y.data_ptr(), dz.data_ptr(), out.data_ptr(), partials.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"grx_ppo_elu_backward_colsum failed ({rc}): {rows} x {cols}")
    return dz, out


def store_transition(storage, step, obs, pri, actions, mu, sigma, values, logp, rewards, dones, time_outs, gamma, log=None):
    """One rollout step's bookkeeping through grx_ppo_store_transition (include/grx_ppo.h): the storage rows of `step`,
    the time-out bootstrap, and -- log = (cur_rew, cur_len, done_rew_row, done_len_row) -- the runner's episode sums."""
    lib = load_ppo_library()
    N = storage.num_envs
    ptr = lambda t: t.data_ptr() if t is not None else None
    for t in (obs, pri, actions, mu, sigma, values, logp, rewards):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("store_transition needs contiguous float32 CUDA tensors")
    def as_u8(t):   # the kernel reads one byte per env: bool is reinterpreted, any other dtype (a reference-style long reset_buf) converted
        if t.dtype == torch.bool:
            t = t.view(torch.uint8)
        elif t.dtype != torch.uint8:
            t = (t != 0).to(torch.uint8)
        if not (t.is_cuda and t.is_contiguous() and t.numel() == N):
            raise RuntimeError("store_transition needs contiguous CUDA dones / time_outs with one entry per env")
        return t
    d8 = as_u8(dones)
    t8 = None if time_outs is None else as_u8(time_outs)
    sp = storage.pri_observations
    lg = log if log is not None else (None, None, None, None)
    with torch.cuda.device(obs.device):
        rc = lib.grx_ppo_store_transition(
            N, obs.shape[1], pri.shape[1] if (pri is not None and sp is not None) else 0, actions.shape[1],
            ptr(obs), ptr(pri) if sp is not None else None, ptr(actions), ptr(mu), ptr(sigma), ptr(values), ptr(logp), ptr(rewards), ptr(d8), ptr(t8),
            float(gamma), ptr(storage.observations[step]), ptr(sp[step]) if sp is not None else None, ptr(storage.actions[step]), ptr(storage.mu[step]),
            ptr(storage.sigma[step]), ptr(storage.values[step]), ptr(storage.actions_log_prob[step]), ptr(storage.rewards[step]), ptr(storage.dones[step]),
            ptr(lg[0]), ptr(lg[1]), ptr(lg[2]), ptr(lg[3]), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"grx_ppo_store_transition failed ({rc})")


def _linears_of(mlp):
    """(weight, bias) of every nn.Linear of an rl.modules.MLP if it is Linear -> ELU(1) -> ... -> Linear, else None."""
    mods = list(mlp.model)
    lin = []
    for i, m in enumerate(mods):
        if i % 2 == 0:
            if not isinstance(m, torch.nn.Linear):
                return None
            lin.append((m.weight, m.bias))
        elif not (isinstance(m, torch.nn.ELU) and m.alpha == 1.0):
            return None
    return lin if len(mods) % 2 == 1 else None


def mlp_can_fuse(mlp, x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and _linears_of(mlp) is not None


def _layer(lib, x, w, b, elu, stream):
    y = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
    rc = lib.grx_mlp_layer(x.shape[0], x.shape[1], w.shape[0], x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                           y.data_ptr(), int(elu), stream)
    if rc:
        raise RuntimeError(f"grx_mlp_layer failed ({rc})")
    return y


def linear_elu(x, weight, bias):
    """ELU(x W^T + b) through grx_mlp_layer (one launch)"""
    lib = load_ppo_library()
    with torch.cuda.device(x.device):
        return _layer(lib, x, weight, bias, True, torch.cuda.current_stream(x.device).cuda_stream)


def mlp_forward(mlp, x):
    """Inference forward of an rl.modules.MLP through libgrx_ppo.so: one MFMA launch per layer (bias + ELU in the epilogue)."""
    lib = load_ppo_library()
    lin = _linears_of(mlp)
    x = _f32c(x)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        for i, (w, b) in enumerate(lin):
            x = _layer(lib, x, w, b, i + 1 < len(lin), stream)
    return x


def policy_act(mlp, std, x, eps):
    """actor forward + sample + log-prob: (actions, logp [M], mu, sigma) -- the hidden layers as in mlp_forward, the output
    layer fused with `mu + std * eps` and Normal.log_prob summed over the actions."""
    lib = load_ppo_library()
    lin = _linears_of(mlp)
    x = _f32c(x)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        for w, b in lin[:-1]:
            x = _layer(lib, x, w, b, True, stream)
        w, b = lin[-1]
        M, A = x.shape[0], w.shape[0]
        actions, mu, sigma = (torch.empty(M, A, device=x.device, dtype=torch.float32) for _ in range(3))
        logp = torch.empty(M, device=x.device, dtype=torch.float32)
        rc = lib.grx_mlp_policy_head(M, x.shape[1], A, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                     _f32c(std).data_ptr(), _f32c(eps).data_ptr(), actions.data_ptr(), logp.data_ptr(),
                                     mu.data_ptr(), sigma.data_ptr(), stream)
        if rc:
            raise RuntimeError(f"grx_mlp_policy_head failed ({rc})")
    return actions, logp, mu, sigma


class RowGather:
    """dst[t] = src[t][idx] for a fixed set of (src, dst) pairs in one launch (grx_ppo_gather_rows): the pointer tables are
    built once, a call passes the index slice."""

    def __init__(self, srcs, dsts):
        self.lib = load_ppo_library()
        n = len(srcs)
        assert n == len(dsts) and all(s.dtype == torch.float32 and s.is_contiguous() and d.is_contiguous() and d.dtype == torch.float32
                                      and s.shape[1:] == d.shape[1:] for s, d in zip(srcs, dsts))
        self.n, self.mb, self.device = n, dsts[0].shape[0], srcs[0].device
        self.src = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
        self.dst = (C.c_void_p * n)(*[d.data_ptr() for d in dsts])
        self.w = (C.c_int * n)(*[int(s[0].numel()) for s in srcs])
        self.keep = (srcs, dsts)

    def __call__(self, idx):
        assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == self.mb
        with torch.cuda.device(self.device):
            rc = self.lib.grx_ppo_gather_rows(self.n, self.src, self.dst, self.w, idx.data_ptr(), self.mb,
                                              torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            raise RuntimeError(f"grx_ppo_gather_rows failed ({rc})")
