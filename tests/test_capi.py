"""The C-ABI library loads, exports every symbol include/grx.h declares, and refuses to run without
a HIP device (no CPU fallback).  No compute calls here (CPU box)."""
import ctypes as C
import os
import re

import pytest
import torch

from wiki_grx_gym_amd import _capi, sim
from wiki_grx_gym_amd.envs import build_config
from tests.helpers import make_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    if not os.path.exists(sim.HIP_LIB_PATH):
        import __graft_entry__ as g
        g._run(["make", "-C", g.CSRC, "libgrx_hip.so"])
    return C.CDLL(sim.HIP_LIB_PATH)


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "grx.h")).read()
    declared = set(re.findall(r"\b(grx_[a-z_]+)\s*\(", hdr))
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    lib = _lib()
    for s in declared:
        assert hasattr(lib, s), s


def test_ppo_header_symbols_are_exported():
    """libgrx_ppo.so (the fused PPO minibatch loss) exports what include/grx_ppo.h declares."""
    from wiki_grx_gym_amd.rl import fused_loss
    hdr = open(os.path.join(ROOT, "include", "grx_ppo.h")).read()
    declared = set(re.findall(r"\b(grx_(?:ppo|mlp)_[a-z_]+)\s*\(", hdr))
    assert declared == {"grx_ppo_loss", "grx_ppo_loss_partials_size", "grx_ppo_colsum", "grx_ppo_colsum_partials_size", "grx_ppo_store_transition", "grx_ppo_gather_rows", "grx_ppo_elu_backward_colsum", "grx_mlp_layer", "grx_mlp_policy_head", "grx_ppo_step_tail", "grx_ppo_step_tail_blocks"}
    path = os.path.join(os.path.dirname(sim.HIP_LIB_PATH), "libgrx_ppo.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g._run(["make", "-C", g.CSRC, "libgrx_ppo.so"])
    lib = fused_loss.load_ppo_library()
    for s_ in declared:
        assert hasattr(lib, s_), s_
    assert lib.grx_ppo_loss_partials_size(1000) == 4 * 35 * 2 and lib.grx_ppo_loss_partials_size(0) == 0
    assert lib.grx_ppo_colsum_partials_size(10485, 128) == 41 * 128 and lib.grx_ppo_colsum_partials_size(0, 5) == 0
    # the step tail's two argument structs travel by value: the ctypes mirrors (rl/fused_loss.py) against the C compiler's view of the header
    import ctypes as C, subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include <stddef.h>\n#include "grx_ppo.h"\nsize_t a(void){return sizeof(grx_ppo_tail_tensors);} size_t b(void){return sizeof(grx_ppo_tail_args);}\n'
                             'size_t c(void){return offsetof(grx_ppo_tail_tensors, numel);} size_t e(void){return offsetof(grx_ppo_tail_args, beta1);} int f(void){return GRX_PPO_TAIL_MAX;}\n')
        so = os.path.join(d, "sz.so")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", so])
        z = C.CDLL(so)
        for f_ in (z.a, z.b, z.c, z.e):
            f_.restype = C.c_size_t
        assert z.a() == C.sizeof(fused_loss._TailTensors) and z.b() == C.sizeof(fused_loss._TailArgs)
        assert z.c() == fused_loss._TailTensors.numel.offset and z.e() == fused_loss._TailArgs.beta1.offset and z.f() == fused_loss.TAIL_MAX
    t = fused_loss._TailTensors(); t.n = 0
    assert lib.grx_ppo_step_tail_blocks(C.byref(t)) == -1       # (argument checks only: nothing is launched without a GPU)
    t.n = 2; t.numel[0] = 5000; t.numel[1] = 4096
    assert lib.grx_ppo_step_tail_blocks(C.byref(t)) == 3


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsets of the ctypes mirror == the C compiler's view of include/grx.h."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stddef.h>\n#include "grx.h"\n'
                   'size_t a(void){return sizeof(grx_config);} size_t b(void){return sizeof(grx_model);}\n'
                   'size_t c(void){return offsetof(grx_config, reward_scale);} size_t d(void){return offsetof(grx_config, height_samples);}\n'
                   'size_t e(void){return offsetof(grx_config, terrain_origins);} size_t f(void){return sizeof(grx_step_args);}\n'
                   'size_t g(void){return sizeof(grx_tensor_desc);} int h(void){return GRX_NUM_TENSORS;} int i(void){return GRX_NUM_REWARD_TERMS;}\n'
                   'size_t j(void){return sizeof(grx_pipeline_state);} size_t k(void){return offsetof(grx_pipeline_state, episode_length);}\n')
    so = tmp_path / "sz.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    for fn in "abcdefgjk":
        getattr(lib, fn).restype = C.c_size_t
    assert lib.a() == C.sizeof(_capi.Config) and lib.b() == C.sizeof(_capi.Model)
    assert lib.c() == _capi.Config.reward_scale.offset and lib.d() == _capi.Config.height_samples.offset
    assert lib.e() == _capi.Config.terrain_origins.offset
    assert lib.f() == C.sizeof(_capi.StepArgs) and lib.g() == C.sizeof(_capi.TensorDesc)
    assert lib.j() == C.sizeof(_capi.PipelineState) and lib.k() == _capi.PipelineState.episode_length.offset
    assert lib.h() == len(_capi.TENSOR_IDS) and lib.i() == _capi.NUM_REWARD_TERMS


def test_reward_term_names_and_abi_version():
    lib = _lib()
    api = _capi.bind(lib)
    assert api["abi_version"]() == _capi.GRX_ABI_VERSION
    assert [api["reward_term_name"](t).decode() for t in range(_capi.NUM_REWARD_TERMS)] == list(_capi.REWARD_TERMS)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_no_cpu_fallback():
    cfg = make_cfg()
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 8)
    with pytest.raises(sim.GrxError, match="MI355X|HIP"):
        sim.HipSim(c, "cpu", keep)
    with pytest.raises(sim.GrxError, match="no HIP device|no CPU fallback"):
        sim.HipSim(c, "cuda:0", keep)
    # the C entry point itself reports the missing device instead of computing on the host
    api = sim.load_hip_library()
    h = C.c_void_p()
    rc = api["create"](C.byref(c), 0, C.byref(h))
    assert rc == -4 and b"no HIP device" in api["last_error"]()
    c.struct_size = 12
    assert api["create"](C.byref(c), 0, C.byref(h)) == -6


def test_the_integration_stub_matches_the_library():
    """INTEGRATION.md 1a is what a reference maintainer pastes: its struct mirrors are EXECUTED here (the snippet's class statements,
    verbatim) and their sizes and fields compared with libgrx_hip.so's own sizeof (grx_sizeof) and with the tested mirror of
    _capi.py -- round 4 shipped an ABI-3 grx_step_args there (no stats_seq), which grx_step would have overrun by 8 bytes."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = md[md.index("```python\n# legged_gym/legged_gym/envs/base/grx_backend.py"):]
    code = code[len("```python\n"):code.index("```", 10)]
    # keep the struct declarations only: the CDLL / torch lines need the library on the loader path and a GPU-side module
    keep, take = [], False
    for line in code.splitlines():
        if line.startswith("class "):
            take = True
        elif line and not line.startswith(" ") and not line.startswith("class "):
            take = False
        if take:
            keep.append(line)
    ns = {"C": C}
    exec("\n".join(keep), ns)
    assert {"StepArgs", "TensorDesc"} <= set(ns)
    api = _capi.bind(_lib())
    assert api["sizeof"](_capi.STRUCT_IDS["STEP_ARGS"][0]) == C.sizeof(ns["StepArgs"]) == C.sizeof(_capi.StepArgs)
    assert api["sizeof"](_capi.STRUCT_IDS["TENSOR_DESC"][0]) == C.sizeof(ns["TensorDesc"]) == C.sizeof(_capi.TensorDesc)
    assert [(n, C.sizeof(t)) for n, t in ns["StepArgs"]._fields_] == [(n, C.sizeof(t)) for n, t in _capi.StepArgs._fields_]
    assert [(n, C.sizeof(t)) for n, t in ns["TensorDesc"]._fields_] == [(n, C.sizeof(t)) for n, t in _capi.TensorDesc._fields_]
    # and every struct of the full mirror against the library (sim.load_hip_library does this once per process)
    for name, (sid, cls) in _capi.STRUCT_IDS.items():
        assert api["sizeof"](sid) == C.sizeof(cls), name
    assert api["sizeof"](99) == -1
    assert "grx_sizeof(1) == C.sizeof(StepArgs)" in code      # the stub itself carries the check
    hdr = open(os.path.join(ROOT, "include", "grx.h")).read()
    assert f"#define GRX_ABI_VERSION {_capi.GRX_ABI_VERSION}" in hdr and f"ABI {_capi.GRX_ABI_VERSION})" in code
