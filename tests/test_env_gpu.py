"""The product path end to end on the GPU: task_registry.make_env("GR1T1") -> GR1T1 (HipSim) -> PPO."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_make_env_step_and_short_training(tmp_path):
    from wiki_grx_gym_amd.envs import GR1T1, GR1T1Cfg, GR1T1CfgPPO
    from wiki_grx_gym_amd.utils import get_args, task_registry
    args = get_args(["--task", "GR1T1", "--headless", "--num_envs", "512", "--seed", "1"])
    cfg = GR1T1Cfg()
    cfg.terrain.mesh_type = "heightfield"
    env, _ = task_registry.make_env("GR1T1", args=args, env_cfg=cfg)
    assert isinstance(env, GR1T1) and env.device == "cuda:0" and env.obs_buf.is_cuda
    obs, pri = env.reset()
    assert obs.shape == (512, 39) and pri.shape == (512, 168)
    for _ in range(5):
        o, p, r, d, ex = env.step(torch.zeros(512, 10, device="cuda"))
    assert torch.isfinite(o).all() and torch.isfinite(p).all() and torch.isfinite(r).all()
    assert d.dtype == torch.bool and "time_outs" in ex and "terrain_level" in ex["episode"]
    # every URDF link frame, published by the step kernel (GRX_T_RIGID_BODY_STATES): a zero-copy view, consistent with what the
    # kernel's own walk published for the feet and with the root state
    rbs = env.rigid_body_states
    assert rbs.shape == (512, env.num_bodies, 13) and rbs is env.rigid_body_states and rbs.data_ptr() == env._sim.tensor("RIGID_BODY_STATES").data_ptr()
    live = ~d
    feet_pos = env._sim.tensor("FEET_POS")                                                   # (N, 2, 3) from the step kernel
    assert (rbs[live][:, env.feet_indices, 0:3] - feet_pos[live]).abs().max() < 1e-4   # fp32 ulps of world coordinates up to 170 m
    assert (rbs[live][:, 0, 0:3] - env.root_states[live][:, 0:3]).abs().max() == 0
    assert (rbs[:, env.feet_indices, 3:7].norm(dim=-1) - 1).abs().max() < 1e-5
    # the body index sets of gr1t1.py:18-113 (case-sensitive substring match: GR1T1 names its IMU link "IMU_link", so imu_indices is empty there as in the reference)
    for name, n in (("torso", 1), ("forehead", 1), ("imu", 0), ("waist", 3), ("head", 3), ("thigh", 6), ("shank", 2), ("feet", 2), ("sole", 0),
                    ("upper_arm", 6), ("lower_arm", 2), ("hand", 6), ("arm_base", 0), ("arm_end", 0)):
        idx = getattr(env, name + "_indices")
        assert idx.dtype == torch.long and len(idx) == n, (name, idx)
        assert all(getattr(env.cfg.asset, ("foot" if name == "feet" else name) + "_name") in env.body_names[i] for i in idx.tolist())
    env.reset_idx(torch.tensor([3, 4, 100], device="cuda"))                                # callable outside step()
    assert env.reset_buf[[3, 4, 100]].all() and (env.episode_length_buf[[3, 4, 100]] == 0).all()
    tcfg = GR1T1CfgPPO()
    tcfg.runner.num_steps_per_env = 16
    runner, _ = task_registry.make_alg_runner(env, name="GR1T1", args=args, train_cfg=tcfg, log_root=str(tmp_path))
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
    assert runner.current_learning_iteration == 3
    assert all(torch.isfinite(p).all() for p in runner.algorithm.actor_critic.parameters())


def test_create_refuses_env_counts_beyond_32_bit_column_offsets():
    """The step kernels form a column element's byte offset in 32 bits (csrc/grx_kernels.hip GCOL): grx_create has to refuse an env
    count whose tallest column table (GRX_MAX_HEIGHT_POINTS rows of floats) would pass 4 GiB -- before it allocates anything."""
    import ctypes as C
    from wiki_grx_gym_amd import sim
    from wiki_grx_gym_amd.envs import build_config
    from tests.helpers import make_cfg
    cfg = make_cfg()
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 8)
    api = sim.load_hip_library()
    h = C.c_void_p()
    c.num_envs = (1 << 32) // (128 * 4)       # 8 388 608: the first count that does not fit
    assert api["create"](C.byref(c), 0, C.byref(h)) != 0 and b"32-bit column offsets" in api["last_error"]()
    c.num_envs = 0
    assert api["create"](C.byref(c), 0, C.byref(h)) != 0 and b"num_envs < 1" in api["last_error"]()


def test_native_library_is_what_runs():
    """The tensors are zero-copy views of libgrx_hip.so's device memory (no silent torch fallback)."""
    import ctypes
    from tests.helpers import make_cfg, make_sims
    from wiki_grx_gym_amd import sim
    maps = open("/proc/self/maps").read()
    hip, _ = make_sims(make_cfg(), 64)
    maps = open("/proc/self/maps").read()
    assert "libgrx_hip.so" in maps
    t = hip.tensor("DOF_POS")
    assert t.is_cuda and t.stride() == (1, 64)
    t.fill_(0.25)
    hip.set_state(None, None, None)
    assert float(hip.tensor("DOF_POS").mean()) == 0.25


def test_steps_recorded_into_a_hip_graph_neither_pace_nor_hang():
    """grx_step's run-ahead pacing spins on a progress word that only advances when steps EXECUTE.  Steps
    recorded under stream capture do not execute, so recording more than the pacing window must not wait for them; a
    replayed graph then advances the state exactly like the same steps issued eagerly."""
    from tests.helpers import make_cfg, make_sims
    hip, _ = make_sims(make_cfg(), 256)
    ref, _ = make_sims(make_cfg(), 256)
    act = torch.zeros(256, hip.num_dofs, device="cuda:0")
    for s in (hip, ref):
        s.reset_all(); s.step(act, 0.0, 1)
    torch.cuda.synchronize()
    n = 300                                               # > the run-ahead window
    import gc
    gc.collect(); gc.disable()                            # (a handle finalised inside a capture frees device memory there: not capturable)
    hip.flush_stats()                                     # capture starts from reduced statistics (include/grx.h grx_flush_stats)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(n):
            hip.step(act, 0.0, 2 + k)
    gc.enable()
    g.replay()
    hip.wait_idle()                                       # only eager steps hold tickets: returns at once
    for k in range(n):
        ref.step(act, 0.0, 2 + k)
    ref.wait_idle()
    torch.cuda.synchronize()
    for name in ("DOF_POS", "ROOT_STATES", "OBS", "REW", "EPISODE_LENGTH"):
        assert torch.equal(hip.tensor(name), ref.tensor(name)), name
    hip.step(act, 0.0, 2 + n); hip.wait_idle()            # eager steps after a replay still pace and complete


@pytest.mark.parametrize("steps_in_graph", [1, 3])
def test_episode_statistics_survive_graph_replays(steps_in_graph):
    """a step recorded into a HIP graph cannot leave its episode statistics to 'the next launch' -- on a replay its
    predecessor in execution order is itself.  Recorded steps carry their own reduction: after any number of replays of a
    one-step (or odd-length) graph GRX_T_EPISODE_STATS and the step's history row equal those of the same steps issued eagerly."""
    from tests.helpers import make_cfg, make_sims
    cfg = make_cfg()
    cfg.env.episode_length_s = 0.1                       # 5 steps: time-outs (finished episodes) inside the window
    hip, _ = make_sims(cfg, 256)
    ref, _ = make_sims(cfg, 256)
    act = torch.full((256, hip.num_dofs), 0.05, device="cuda:0")
    for s in (hip, ref):
        s.reset_all(); s.step(act, 0.0, 1)
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.disable()
    hip.flush_stats()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(steps_in_graph):
            hip.step(act, 0.0, 2 + k)
    gc.enable()
    seen = set()
    for rep in range(5):
        g.replay()
        for k in range(steps_in_graph):
            ref.step(act, 0.0, 2 + k)
        torch.cuda.synchronize()
        a, b = hip.tensor("EPISODE_STATS").cpu(), ref.tensor("EPISODE_STATS").cpu()      # (ref: flushed; hip: current by the recorded reduction)
        assert torch.equal(a, b), (rep, a, b)
        assert torch.equal(hip.tensor("EPISODE_STATS_HISTORY")[hip.last_stats_slot].cpu(), a)
        seen.add(tuple(a.tolist()))
    NT = hip.tensor("EPISODE_STATS").numel() - 2
    assert float(ref.tensor("EPISODE_STATS")[NT]) > 0 and len(seen) >= 2      # episodes did end, and the means moved between replays
    hip.step(act, 0.0, 50); ref.step(act, 0.0, 50)                             # eager launches after the replays keep agreeing
    assert torch.equal(hip.tensor("EPISODE_STATS").cpu(), ref.tensor("EPISODE_STATS").cpu())
    for name in ("DOF_POS", "OBS", "EPISODE_LENGTH", "EPISODE_SUMS"):
        assert torch.equal(hip.tensor(name), ref.tensor(name)), name


def test_a_recorded_step_that_never_runs_leaves_the_statistics_alone():
    """(i) a capture that begins with unreduced statistics is refused -- recorded, the reduction of the last eager launch's
    rows would run at replay time on whichever launch then holds that parity; (ii) a step that was recorded but never replayed changes
    nothing: the statistics stay those of the last eager step, and the eager steps that follow agree with a handle that never captured."""
    from tests.helpers import make_cfg, make_sims
    from wiki_grx_gym_amd.sim import GrxError
    cfg = make_cfg()
    cfg.env.episode_length_s = 0.1                       # 5 steps: time-outs inside the window
    hip, _ = make_sims(cfg, 256)
    ref, _ = make_sims(cfg, 256)
    act = torch.full((256, hip.num_dofs), 0.05, device="cuda:0")
    for s in (hip, ref):
        s.reset_all()
        for k in range(7):
            s.step(act, 0.0, 1 + k)
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.disable()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.default_stream())
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            try:
                with pytest.raises(GrxError, match="grx_flush_stats"):
                    hip.step(act, 0.0, 8)                 # (i) statistics of step 7 are still unreduced
            finally:
                g.capture_end()
        before = hip.tensor("EPISODE_STATS").clone()      # (flushes)
        with torch.cuda.stream(side):
            g2 = torch.cuda.CUDAGraph()
            g2.capture_begin()
            hip.step(act, 0.0, 8)                         # (ii) recorded, never replayed
            g2.capture_end()
    finally:
        gc.enable()
    torch.cuda.synchronize()
    assert torch.equal(hip.tensor("EPISODE_STATS"), before)
    assert torch.equal(hip.tensor("EPISODE_STATS").cpu(), ref.tensor("EPISODE_STATS").cpu())
    for k in range(6):                                    # eager steps behind the dead recording: same rows, same means as without it
        hip.step(act, 0.0, 8 + k); ref.step(act, 0.0, 8 + k)
        torch.cuda.synchronize()
        assert torch.equal(hip.tensor("EPISODE_STATS").cpu(), ref.tensor("EPISODE_STATS").cpu()), k
    for name in ("DOF_POS", "OBS", "EPISODE_LENGTH", "EPISODE_SUMS"):
        assert torch.equal(hip.tensor(name), ref.tensor(name)), name


def test_pipelines_on_the_bounded_spin_build():
    """the wave pipelines hand over through LDS flags and spin on them; a lost hand-over would hang the GPU.  The
    -DGRX_SPIN_LIMIT build of the same sources (csrc/variants/libgrx_spinlimit.so, csrc/grx_flags.h) bounds every spin and, on
    expiry, reports the flag and traps.  Every pipelined layout runs 2000 policy steps on it (rough terrain, falls and resets
    included) in a process of its own: no spin expires.  (The protocol itself -- payload then flag without a wait in between -- is
    checked by tools/micro/lds_handover.hip: profiles/r04_lds_handover_litmus.json, 1.2e9 hand-overs, no mismatch.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "wiki-grx-gym_amd", "csrc", "variants", "libgrx_spinlimit.so")
    assert os.path.exists(lib), "build it: make -C wiki-grx-gym_amd/csrc variants/libgrx_spinlimit.so (build() does)"
    prog = r"""
import json, os, sys, torch
sys.path.insert(0, %r)
from tests.helpers import make_cfg, make_sims, random_actions
out = {}
for name, env in (("pair4", {"GRX_LANES_PER_ENV": "2", "GRX_WAVES_PER_BLOCK": "4"}), ("pair8", {"GRX_LANES_PER_ENV": "2", "GRX_WAVES_PER_BLOCK": "8"}),
                  ("quad4", {"GRX_LANES_PER_ENV": "4", "GRX_QUAD_WAVES": "4"}), ("quad8", {"GRX_LANES_PER_ENV": "4", "GRX_QUAD_WAVES": "8"})):
    for k in ("GRX_LANES_PER_ENV", "GRX_WAVES_PER_BLOCK", "GRX_QUAD_WAVES"): os.environ.pop(k, None)
    os.environ.update(env)
    cfg = make_cfg(terrain="heightfield", noise=True, dr=True, push=True)
    hip, _ = make_sims(cfg, 512, seed=1)
    hip.reset_all()
    gen = torch.Generator().manual_seed(0)
    acts = [random_actions(cfg, 512, gen, 1.0).cuda() for _ in range(8)]
    for i in range(2000): hip.step(acts[i %% 8], 5.0, i + 1)
    hip.wait_idle(); torch.cuda.synchronize()
    code, bounded = hip.spin_report()
    out[name] = {"code": code, "bounded": bounded, "layout": hip.layout()["kernel"], "obs": float(hip.tensor("OBS").double().sum()), "resets": int(hip.tensor("EPISODE_LENGTH").lt(100).sum()),
                 "finite": bool(torch.isfinite(hip.tensor("OBS")).all())}
    hip.close()
print("RESULT " + json.dumps(out))
""" % root
    env = dict(os.environ)
    env["GRX_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    want = {"pair4": "grx_step_kernel<true, 4, false>", "pair8": "grx_step_kernel<true, 8, false>",
            "quad4": "grx_step_kernel_quad<true, 4, false>", "quad8": "grx_step_kernel_quad<true, 8, false>"}
    for name, b in res.items():
        assert b["bounded"] and b["code"] == 0 and b["finite"] and b["resets"] > 0 and b["layout"] == want[name], (name, b)


def test_bench_line_has_the_contract_fields():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command): one JSON line with BASELINE.json's metric, the
    roofline object (HBM fraction + the VALU issue fraction that actually bounds the kernel) and the cpu_baseline object."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-steps", "3"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"] == "env-steps/sec GR1T1 rough-terrain @4096 envs" and j["unit"] == "env-steps/s" and j["n_gpus"] == 1
    # the line names what RAN: grx_layout() of the handle, not a guess from the batch size
    lay = j["config"]["layout"]
    assert j["roofline"]["kernel"] == lay["kernel"] == "grx_step_kernel_quad<true, 8, false>" and lay["lanes_per_env"] == 4 and lay["waves_per_block"] == 8
    assert "4 lanes per env, 8 waves per 16-env block" in j["roofline"]["note"] and j["config"]["on_demand_tensors"] == "on_refresh" and j["config"]["product_default"] is True
    assert j["steps"] == 20 and j["warmup"] == 5 and j["higher_is_better"] and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and j["config"]["finite_outputs"]
    assert abs(j["value"] - 4096 * 20 / (j["ms_per_step"] * 20e-3)) < 1e-3 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel_ms"] > 0 and r["launches_timed"] >= 2 and abs(r["achieved"] - 2476.0 * 4096 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["valu_issue_frac"] is None or 0.01 < r["valu_issue_frac"] < 1.0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["cpu_model"] and "note" in c
    assert c["reference_stage"] is None or c["reference_stage"]["value"] > 0
    assert j["value"] > 20e6      # an MI355X does not fall below this even inside a 20-step window
    fi = j["full_iteration"]      # the reference's own metric (on_policy_runner.py:235): whole PPO iterations on the product-default env
    assert "error" not in fi and fi["iters"] >= 1 and fi["n_gpus"] == 1 and fi["envs_per_gpu"] == 4096 and fi["num_steps_per_env"] == 64
    assert fi["env_steps_per_s"] > 2e5 and abs(fi["env_steps_per_s"] - 64 * 4096 / (fi["iteration_ms"] * 1e-3)) < 1e-6 * fi["env_steps_per_s"]
    assert 0 < fi["collection_ms"] < fi["iteration_ms"] and 0 < fi["learn_ms"] < fi["iteration_ms"]


def test_bench_runs_under_torchrun_with_a_real_rccl_group():
    """bench.py's multi-rank path -- RCCL process-group init, the barriers around the timed window, the MAX all-reduce of the
    elapsed time -- as the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), on the
    one GPU this box has: world size 1, every collective executes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "10", "--envs-per-gpu", "1024", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 50 and out["value"] > 1e6 and out["config"]["finite_outputs"]


def test_refresh_after_graph_replays():
    """grx_refresh keyed its 'already current' shortcut on the host's launch counter; a step replayed through a graph advances the
    simulation without touching it, so the second refresh between replays returned without launching and RIGID_BODY_STATES / MEASURED_HEIGHTS
    kept the first replay's data.  From the first recorded launch on the shortcut is off: after every replay the on-refresh tensors equal those
    of a handle that ran the same steps eagerly and writes them every step."""
    import gc
    from tests.helpers import make_cfg, make_terrain
    from wiki_grx_gym_amd.envs import build_config
    from wiki_grx_gym_amd.sim import HipSim
    cfg = make_cfg(terrain="heightfield")
    N = 256
    ter = make_terrain(cfg, N, 1)

    def make(mode):
        c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
        c.publish_rigid_body_states = mode; c.publish_measured_heights = mode
        s = HipSim(c, "cuda:0", keep)
        s.reset_all()
        return s
    from wiki_grx_gym_amd import _capi
    hip, ref = make(_capi.PUBLISH_ON_REFRESH), make(_capi.PUBLISH_EVERY_STEP)
    act = torch.full((N, hip.num_dofs), 0.1, device="cuda:0")
    for s in (hip, ref):
        s.step(act, 0.0, 1)
    hip.tensor("EPISODE_STATS")                              # (flushes the statistics: a capture refuses to begin on unreduced rows)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.default_stream())
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            hip.step(act, 0.0, 2)
            g.capture_end()
    finally:
        gc.enable()
    torch.cuda.synchronize()
    for k in range(3):
        g.replay(); ref.step(act, 0.0, 2)
        torch.cuda.synchronize()
        from tests.test_kinematics import rbs_err                  # (positions, quaternions up to sign, velocities)
        nl = int(hip._keep[-1].model.num_links)
        a, b = hip.tensor("RIGID_BODY_STATES").cpu()[:, :nl], ref.tensor("RIGID_BODY_STATES").cpu()[:, :nl]     # (sim.tensor refreshes an on-refresh tensor)
        ep, eq, ev = rbs_err(a, b)
        assert ep <= 2e-5 and eq <= 2e-5 and ev <= 2e-4, (k, ep, eq, ev)
        assert torch.equal(hip.tensor("MEASURED_HEIGHTS").cpu(), ref.tensor("MEASURED_HEIGHTS").cpu()), k
        if k:
            assert float((a[..., :3] - first).abs().max()) > 1e-4, "the robot moved between the replays"
        first = a[..., :3].clone()
