"""ctypes binding of the C ABI (include/grx.h) + zero-copy tensor views.

``HipSim`` loads ``csrc/libgrx_hip.so`` -- the ONLY simulation backend of the product.  There is
no CPU fallback: if the library is missing, or no HIP device is visible, construction raises
(the reference equally raises when its binding is missing, gymapi.py:100-101).

``SimHandle`` is backend-agnostic over a bound entry-point table so that the test-suite can drive
the CPU oracle (oracle/binding.py) through the very same code path.

Tensor views follow the gymtorch.wrap_tensor model (gymtorch.py:61-73, gymtorch.cpp:33-158):
non-owning, zero-copy, possibly strided.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("GRX_HIP_LIB") or os.path.join(_HERE, "csrc", "libgrx_hip.so")   # GRX_HIP_LIB: A/B builds (tools/)

_TORCH_DTYPE = {_capi.DTYPE_F32: torch.float32, _capi.DTYPE_U8: torch.uint8,
                _capi.DTYPE_I32: torch.int32, _capi.DTYPE_I64: torch.int64}
_NP_DTYPE = {_capi.DTYPE_F32: np.float32, _capi.DTYPE_U8: np.uint8, _capi.DTYPE_I32: np.int32,
             _capi.DTYPE_I64: np.int64}
_TYPESTR = {_capi.DTYPE_F32: "<f4", _capi.DTYPE_U8: "|u1", _capi.DTYPE_I32: "<i4", _capi.DTYPE_I64: "<i8"}


class GrxError(RuntimeError):
    pass


class _DeviceArray:
    """Minimal __cuda_array_interface__ carrier (works for HIP device memory in ROCm torch)."""

    def __init__(self, ptr, shape, strides_bytes, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "strides": tuple(strides_bytes), "version": 3}


def _wrap_device(ptr, dtype, shape, stride, device_index):
    ext = _load_torch_ext()
    if ext is not None:
        return ext.wrap(int(ptr), int(dtype), list(shape), list(stride), int(device_index))
    item = np.dtype(_NP_DTYPE[dtype]).itemsize
    arr = _DeviceArray(ptr, shape, [s * item for s in stride], _TYPESTR[dtype])
    return torch.as_tensor(arr, device=torch.device("cuda", device_index))


def _wrap_host(ptr, dtype, shape, stride):
    npdt = np.dtype(_NP_DTYPE[dtype])
    span = 1 + sum((n - 1) * s for n, s in zip(shape, stride)) if all(n > 0 for n in shape) else 0
    if span == 0:
        return torch.zeros(tuple(shape), dtype=_TORCH_DTYPE[dtype])
    buf = (C.c_char * (span * npdt.itemsize)).from_address(ptr)
    flat = np.frombuffer(buf, dtype=npdt)
    view = np.lib.stride_tricks.as_strided(flat, shape=tuple(shape), strides=tuple(s * npdt.itemsize for s in stride))
    return torch.from_numpy(view)


_ext_cache = [False, None]


def _load_torch_ext():
    """The native gymtorch successor (csrc/grx_torch.cpp -> _grxtorch*.so), if it was built."""
    if _ext_cache[0]:
        return _ext_cache[1]
    _ext_cache[0] = True
    try:
        import importlib.util
        import glob
        cands = glob.glob(os.path.join(_HERE, "csrc", "_grxtorch*.so"))
        if cands:
            spec = importlib.util.spec_from_file_location("_grxtorch", cands[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _ext_cache[1] = mod
    except Exception:  # pragma: no cover - fall back to __cuda_array_interface__
        _ext_cache[1] = None
    return _ext_cache[1]


class SimHandle:
    """One simulation handle behind the C ABI.  ``device``: torch.device the buffers live on."""

    def __init__(self, api, cfg_struct, device, device_id=0, keepalive=()):
        self._api = api
        self._keep = list(keepalive) + [cfg_struct]
        self.device = torch.device(device)
        self.num_envs = int(cfg_struct.num_envs)
        self.num_dofs = int(cfg_struct.model.num_bodies) - 1
        self._h = C.c_void_p()
        self._check(api["create"](C.byref(cfg_struct), int(device_id), C.byref(self._h)), "create")
        self._views = {}
        # tensors this handle publishes ON REFRESH (grx_publish_mode): tensor() brings them up to date before it hands out the view
        self._on_refresh = {name for name, field in (("RIGID_BODY_STATES", "publish_rigid_body_states"), ("MEASURED_HEIGHTS", "publish_measured_heights"))
                            if int(getattr(cfg_struct, field)) == _capi.PUBLISH_ON_REFRESH}
        self.last_stats_slot = 0   # row of EPISODE_STATS_HISTORY holding the episode statistics of the last step
        self.last_stats_seq = 0    # ... and that step's launch number (grx_step_args.stats_seq)

    # -- errors
    def _check(self, rc, what):
        if rc != 0:
            msg = self._api["last_error"]()
            raise GrxError(f"grx {what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    # -- API
    def tensor(self, name):
        """Zero-copy torch view of a library buffer (cached)."""
        if name == "EPISODE_STATS":
            self.flush_stats()   # a step's statistics are reduced by the NEXT launch (include/grx.h grx_flush_stats)
        if name in self._on_refresh and self._api["refresh"](self._h, _capi.T[name], self._stream()) != 0:
            self._on_refresh.discard(name)   # (the library publishes it differently for this handle -- the one-lane generic kernel has no on-demand link frames: the descriptor below says so)
        if name in self._views:
            return self._views[name]
        d = _capi.TensorDesc()
        self._check(self._api["tensor"](self._h, _capi.T[name], C.byref(d)), f"tensor({name})")
        if not d.data:
            raise GrxError(f"tensor {name} is not published by this handle (see the publish_* switches of grx_config)")
        shape = [int(d.shape[i]) for i in range(d.ndim)]
        stride = [int(d.stride[i]) for i in range(d.ndim)]
        if self.device.type == "cuda":
            t = _wrap_device(d.data, d.dtype, shape, stride, self.device.index or 0)
        else:
            t = _wrap_host(d.data, d.dtype, shape, stride)
        self._views[name] = t
        return t

    def refresh(self, name):
        """gym.refresh_*_tensor for a tensor published ON REFRESH (include/grx.h grx_refresh); a no-op for those a step keeps current."""
        self._check(self._api["refresh"](self._h, _capi.T[name], self._stream()), f"refresh({name})")

    def reset_all(self):
        self._check(self._api["reset_all"](self._h, self._stream()), "reset_all")

    def step(self, actions, delay_substeps, common_step_counter, noise_uniform=None, obs_out=None, pri_obs_out=None):
        """actions: contiguous float32 (N, nd) tensor on self.device (borrowed for the call)."""
        if actions is not None:
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device.type != self.device.type:
                raise GrxError("actions must be a contiguous float32 tensor on the simulation device")  # gymtorch.py:98-99
            if tuple(actions.shape) != (self.num_envs, self.num_dofs):
                raise GrxError(f"actions shape {tuple(actions.shape)} != ({self.num_envs}, {self.num_dofs})")
        a = _capi.StepArgs()
        a.actions = actions.data_ptr() if actions is not None else None
        a.delay_substeps = float(delay_substeps)
        a.common_step_counter = int(common_step_counter)
        if noise_uniform is not None:
            if noise_uniform.dtype != torch.float32 or not noise_uniform.is_contiguous():
                raise GrxError("noise_uniform must be contiguous float32")
            a.noise_uniform = noise_uniform.data_ptr()
        for name, t in (("obs_out", obs_out), ("pri_obs_out", pri_obs_out)):
            if t is not None:
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != self.device.type or t.shape[0] != self.num_envs:
                    raise GrxError(f"{name} must be a contiguous float32 (N, k) tensor on the simulation device")
                setattr(a, name, t.data_ptr())
        self._check(self._api["step"](self._h, C.byref(a), self._stream()), "step")
        self.last_stats_slot = int(a.stats_slot)
        self.last_stats_seq = int(a.stats_seq)
        return self.last_stats_slot

    def flush_stats(self):
        """Reduce the episode statistics of the last enqueued step now (they are otherwise folded into the next launch)."""
        self._check(self._api["flush_stats"](self._h, self._stream()), "flush_stats")

    def _ids(self, env_ids):
        ids = torch.as_tensor(env_ids, device=self.device).to(torch.int32).contiguous().reshape(-1)
        return ids, (ids.data_ptr() if ids.numel() else None), int(ids.numel())

    def reset_idx(self, env_ids):
        """LeggedRobot.reset_idx(env_ids) (legged_robot.py:377-440) outside a step."""
        ids, ptr, n = self._ids(env_ids)
        self._check(self._api["reset_idx"](self._h, ptr, n, self._stream()), "reset_idx")

    def set_state(self, root_states=None, dof_pos=None, dof_vel=None):
        def ptr(t, cols):
            if t is None:
                return None
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (self.num_envs, cols):
                raise GrxError("set_state tensors must be contiguous float32 (N, k)")
            return t.data_ptr()
        self._check(self._api["set_state"](self._h, ptr(root_states, 13), ptr(dof_pos, self.num_dofs),
                                           ptr(dof_vel, self.num_dofs), self._stream()), "set_state")

    def set_state_indexed(self, env_ids, root_states=None, dof_pos=None, dof_vel=None):
        """set_dof_state_tensor_indexed / set_actor_root_state_tensor_indexed (legged_robot.py:737-740, 782-784): rows env_ids of
        the FULL (N, k) tensors."""
        def ptr(t, cols):
            if t is None:
                return None
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (self.num_envs, cols):
                raise GrxError("set_state tensors must be contiguous float32 (N, k)")
            return t.data_ptr()
        ids, p, n = self._ids(env_ids)
        self._check(self._api["set_state_indexed"](self._h, p, n, ptr(root_states, 13), ptr(dof_pos, self.num_dofs),
                                                   ptr(dof_vel, self.num_dofs), self._stream()), "set_state_indexed")

    def debug_post_physics(self, states, apply_reset=False, common_step_counter=1, noise_uniform=None):
        """TEST-ONLY (include/grx.h grx_debug_post_physics): post_physics_step of all envs on injected state.
        states: ctypes array of _capi.PipelineState, one per env."""
        if "debug_post_physics" not in self._api:
            raise GrxError("this backend has no grx_debug_post_physics")
        if len(states) != self.num_envs:
            raise GrxError("debug_post_physics needs one record per env")
        a = _capi.StepArgs()
        a.common_step_counter = int(common_step_counter)
        if noise_uniform is not None:
            if noise_uniform.dtype != torch.float32 or not noise_uniform.is_contiguous():
                raise GrxError("noise_uniform must be contiguous float32")
            a.noise_uniform = noise_uniform.data_ptr()
        self._check(self._api["debug_post_physics"](self._h, states, int(bool(apply_reset)), C.byref(a), self._stream()), "debug_post_physics")

    def debug_terrain(self, xy):
        """TEST-ONLY (include/grx.h grx_debug_terrain): (n, 3) height and gradient of the physics terrain surface under the (n, 2) points."""
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        out = np.zeros((xy.shape[0], 3), dtype=np.float32)
        self._check(self._api["debug_terrain"](self._h, xy.ctypes.data_as(C.POINTER(C.c_float)), int(xy.shape[0]),
                                               out.ctypes.data_as(C.POINTER(C.c_float)), self._stream()), "debug_terrain")
        return out

    def debug_wall(self, xyzr):
        """TEST-ONLY (include/grx.h grx_debug_wall): (n, 3) overlap x direction of the spheres (n, 4: x, y, z, r) with the vertical faces of the trimesh next to them."""
        xyzr = np.ascontiguousarray(xyzr, dtype=np.float32)
        out = np.zeros((xyzr.shape[0], 3), dtype=np.float32)
        self._check(self._api["debug_wall"](self._h, xyzr.ctypes.data_as(C.POINTER(C.c_float)), int(xyzr.shape[0]),
                                            out.ctypes.data_as(C.POINTER(C.c_float)), self._stream()), "debug_wall")
        return out

    def episode_stats(self):
        out = (C.c_float * (_capi.NUM_REWARD_TERMS + 2))()   # means, [NT] episodes that ended, [NT + 1] mean terrain level
        self._check(self._api["episode_stats"](self._h, out, self._stream()), "episode_stats")
        return np.array(out[:], dtype=np.float32)

    def wait_idle(self):
        """Spin until every step enqueued so far has finished on the GPU (reads the library's pinned progress word;
        does not go through the HIP runtime's event/signal machinery)."""
        if "wait_idle" in self._api:
            self._check(self._api["wait_idle"](self._h), "wait_idle")

    def kernel_time_ms(self, enable=True):
        """(average ms, launches timed) since the last call; enable: False/0 stop, True/1 every launch, n every n-th."""
        if "kernel_time_ms" not in self._api:
            return 0.0, 0
        ms, n = C.c_float(0), C.c_int64(0)
        self._check(self._api["kernel_time_ms"](self._h, int(enable), C.byref(ms), C.byref(n)), "kernel_time_ms")
        return float(ms.value), int(n.value)

    def stats_seq(self):
        """Launch number of the handle's last statistics-writing launch (include/grx.h grx_stats_seq)."""
        if "stats_seq" not in self._api:
            return self.last_stats_seq
        v = C.c_int64(0)
        self._check(self._api["stats_seq"](self._h, C.byref(v)), "stats_seq")
        return int(v.value)

    def spin_report(self):
        """(code, bounded): code != 0 when a bounded LDS spin of the wave pipelines expired (-DGRX_SPIN_LIMIT builds, csrc/grx_flags.h);
        bounded: whether the loaded library is such a build."""
        code, bounded = C.c_uint64(0), C.c_int(0)
        self._check(self._api["debug_spin_report"](self._h, C.byref(code), C.byref(bounded)), "debug_spin_report")
        return int(code.value), bool(bounded.value)

    def layout(self):
        """What step() launches (include/grx.h grx_layout): dict(lanes_per_env, waves_per_block, envs_per_block, num_blocks, kernel)."""
        if "layout" not in self._api:
            return None
        li = _capi.LayoutInfo()
        self._check(self._api["layout"](self._h, C.byref(li)), "layout")
        return {"lanes_per_env": int(li.lanes_per_env), "waves_per_block": int(li.waves_per_block), "envs_per_block": int(li.envs_per_block),
                "num_blocks": int(li.num_blocks), "kernel": li.kernel.decode()}

    def close(self):
        if self._h:
            self._views.clear()
            self._api["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hip_api = None


def load_hip_library():
    """dlopen csrc/libgrx_hip.so and bind every include/grx.h entry point; raises if absent."""
    global _hip_api
    if _hip_api is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise GrxError(f"{HIP_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(HIP_LIB_PATH)
        _hip_api = _capi.bind(lib, "grx_")
        if _hip_api["abi_version"]() != _capi.GRX_ABI_VERSION:
            raise GrxError("libgrx_hip.so ABI version mismatch")
        for name, (sid, cls) in _capi.STRUCT_IDS.items():   # a short mirror of grx_step_args would be overrun by grx_step's OUT fields
            if _hip_api["sizeof"](sid) != C.sizeof(cls):
                raise GrxError(f"libgrx_hip.so: sizeof(grx_{name.lower()}) = {_hip_api['sizeof'](sid)} but the ctypes mirror has {C.sizeof(cls)} bytes")
    return _hip_api


class HipSim(SimHandle):
    def __init__(self, cfg_struct, device="cuda:0", keepalive=()):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise GrxError(f"the GRx simulation runs on an MI355X (HIP) device only; got sim device '{device}'. "
                           "No CPU pipeline is shipped (the reference's CPU pipeline is PhysX-CPU, which this build does not emulate).")
        if not torch.cuda.is_available():
            raise GrxError("no HIP device visible (torch.cuda.is_available() is False); the GRx step has no CPU fallback")
        super().__init__(load_hip_library(), cfg_struct, dev, device_id=dev.index or 0, keepalive=keepalive)
