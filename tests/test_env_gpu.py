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
    # all-link kinematics on demand (torch, envs/kinematics.py) agree with what the kernel's own walk published
    rbs = env.rigid_body_states
    assert rbs.shape == (512, env.num_bodies, 13) and rbs is env.rigid_body_states        # cached per step
    live = ~d
    feet_pos = env._sim.tensor("FEET_POS")                                                   # (N, 2, 3) from the step kernel
    assert (rbs[live][:, env.feet_indices, 0:3] - feet_pos[live]).abs().max() < 1e-4   # fp32 ulps of world coordinates up to 170 m
    assert (rbs[:, 0, 0:3] - env.root_states[:, 0:3]).abs().max() == 0
    assert (rbs[:, env.feet_indices, 3:7].norm(dim=-1) - 1).abs().max() < 1e-5
    tcfg = GR1T1CfgPPO()
    tcfg.runner.num_steps_per_env = 16
    runner, _ = task_registry.make_alg_runner(env, name="GR1T1", args=args, train_cfg=tcfg, log_root=str(tmp_path))
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
    assert runner.current_learning_iteration == 3
    assert all(torch.isfinite(p).all() for p in runner.algorithm.actor_critic.parameters())


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
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(n):
            hip.step(act, 0.0, 2 + k)
    g.replay()
    hip.wait_idle()                                       # only eager steps hold tickets: returns at once
    for k in range(n):
        ref.step(act, 0.0, 2 + k)
    ref.wait_idle()
    torch.cuda.synchronize()
    for name in ("DOF_POS", "ROOT_STATES", "OBS", "REW", "EPISODE_LENGTH"):
        assert torch.equal(hip.tensor(name), ref.tensor(name)), name
    hip.step(act, 0.0, 2 + n); hip.wait_idle()            # eager steps after a replay still pace and complete
