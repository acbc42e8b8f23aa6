"""The fused PPO minibatch loss (libgrx_ppo.so, include/grx_ppo.h) against the torch expression it replaces
(rsl_rl/algorithms/ppo.py:215-245 spelled with torch.distributions.Normal): values and gradients."""
import os

import pytest
import torch

from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
from wiki_grx_gym_amd.rl.ppo import PPO

pytestmark = pytest.mark.gpu


def _alg(fused, use_clipped=True):
    os.environ["GRX_PPO_FUSED_LOSS"] = "1" if fused else "0"
    os.environ["GRX_PPO_GRAPH"] = "0"
    try:
        torch.manual_seed(3)
        ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[64, 32], critic_hidden_dims=[64, 32], activation="elu", init_noise_std=0.7)
        return PPO(ac, clip_param=0.2, value_loss_coef=1.3, entropy_coef=0.01, use_clipped_value_loss=use_clipped,
                   schedule="adaptive", desired_kl=0.03, device="cuda:0")
    finally:
        del os.environ["GRX_PPO_FUSED_LOSS"], os.environ["GRX_PPO_GRAPH"]


def _batch(B):
    g = torch.Generator(device="cuda:0").manual_seed(11)
    r = lambda *s, scale=1.0: torch.randn(*s, device="cuda:0", generator=g) * scale
    obs, cobs = r(B, 39), r(B, 168)
    actions, old_mu = r(B, 10), r(B, 10, scale=0.5)
    old_sigma = 0.5 + torch.rand(B, 10, device="cuda:0", generator=g)
    # old_logp spread so that ratios land inside, below and above the clip interval; exact ties (ratio inside) abound
    old_logp = r(B, 1, scale=3.0) - 12.0
    return obs, cobs, actions, r(B, 1), r(B, 1, scale=2.0), r(B, 1), old_logp, old_mu, old_sigma


@pytest.mark.parametrize("B", [1, 255, 10485])
@pytest.mark.parametrize("use_clipped", [True, False])
def test_fused_loss_matches_torch(B, use_clipped):
    ref, fus = _alg(False, use_clipped), _alg(True, use_clipped)
    assert fus._fused_loss and not ref._fused_loss
    batch = _batch(B)
    outs = []
    for alg in (ref, fus):
        s, v, loss, kl = alg._losses(*batch)
        alg.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        outs.append((s.item(), v.item(), loss.item(), kl.item(), [p.grad.clone() for p in alg.actor_critic.parameters()]))
    (s0, v0, l0, k0, g0), (s1, v1, l1, k1, g1) = outs
    for a, b in ((s0, s1), (v0, v1), (l0, l1), (k0, k1)):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (a, b)
    for (n, _), a, b in zip(ref.actor_critic.named_parameters(), g0, g1):
        tol = 2e-5 * max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() <= tol, (n, (a - b).abs().max().item(), tol)


def test_graphed_update_with_fused_loss_tracks_torch_expression():
    """Two full update() calls: the default path (HIP graph captured with rocBLAS preferred, fused loss) against the
    eager device path with the torch expression and against the graph with the torch expression -- same data, seeds."""
    res = []
    for fused, graph in ((False, "0"), (False, "1"), (True, "1")):
        os.environ["GRX_PPO_FUSED_LOSS"] = "1" if fused else "0"
        os.environ["GRX_PPO_GRAPH"] = graph
        try:
            torch.manual_seed(0)
            ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[64, 32], critic_hidden_dims=[64, 32], activation="elu", init_noise_std=0.2)
            alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, entropy_coef=0.01, learning_rate=1e-4,
                      schedule="adaptive", desired_kl=0.03, device="cuda:0")
        finally:
            del os.environ["GRX_PPO_FUSED_LOSS"], os.environ["GRX_PPO_GRAPH"]
        alg.init_storage(64, 8)
        st = alg.storage
        g = torch.Generator(device="cuda:0").manual_seed(1)
        for x in (st.observations, st.pri_observations, st.actions, st.rewards, st.values, st.returns, st.advantages, st.mu):
            x.copy_(torch.randn(x.shape, device="cuda:0", generator=g) * 0.3)
        st.sigma.fill_(0.2); st.actions_log_prob.fill_(-1.0)
        st.step = 8
        torch.manual_seed(5)
        out = [alg.update() for _ in range(2)]
        res.append((out, torch.cat([p.detach().flatten() for p in ac.parameters()]), alg.learning_rate))
    (o0, w0, lr0) = res[0]
    assert torch.isfinite(w0).all() and all(abs(v) > 1e-6 for o in o0 for v in o)   # a skipped (NaN) step would report zeros
    for (o1, w1, lr1) in res[1:]:
        assert lr0 == lr1
        assert torch.isfinite(w1).all()
        assert (w0 - w1).abs().max().item() < 2e-4 * w0.abs().max().item()
        for a, b in zip(o0, o1):
            assert abs(a[0] - b[0]) < 1e-3 * max(1.0, abs(a[0])) and abs(a[1] - b[1]) < 1e-3 * max(1.0, abs(a[1]))


def test_full_size_graphed_updates_stay_finite_and_match_the_eager_path():
    """The train-config shapes (4096 envs x 24 steps, [512, 256, 128] networks, 4 minibatches of 24576): three updates
    through the captured graph equal the eager device path's, and nothing goes non-finite on the way."""
    res = []
    for graph in ("0", "1"):
        os.environ["GRX_PPO_GRAPH"] = graph
        try:
            torch.manual_seed(0)
            ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=0.2)
            alg = PPO(ac, num_learning_epochs=5, num_mini_batches=4, clip_param=0.2, entropy_coef=0.01, learning_rate=1e-4,
                      schedule="adaptive", desired_kl=0.03, device="cuda:0")
        finally:
            del os.environ["GRX_PPO_GRAPH"]
        alg.init_storage(4096, 24)
        st = alg.storage
        g = torch.Generator(device="cuda:0").manual_seed(1)
        for x in (st.observations, st.pri_observations, st.actions, st.rewards, st.values, st.returns, st.advantages, st.mu):
            x.copy_(torch.randn(x.shape, device="cuda:0", generator=g) * 0.3)
        st.sigma.fill_(0.2); st.actions_log_prob.fill_(-1.0)
        st.step = 24
        torch.manual_seed(5)
        out = [alg.update() for _ in range(3)]
        res.append((out, torch.cat([p.detach().flatten() for p in ac.parameters()])))
    (o0, w0), (o1, w1) = res
    assert torch.isfinite(w0).all() and torch.isfinite(w1).all()
    assert (w0 - w1).abs().max().item() < 5e-3 * w0.abs().max().item()   # Adam amplifies last-bit GEMM differences
    for a, b in zip(o0, o1):
        assert abs(a[0]) > 1e-6 and abs(a[0] - b[0]) < 2e-3 * max(1.0, abs(a[0])) and abs(a[1] - b[1]) < 2e-3 * max(1.0, abs(a[1]))


def test_captured_step_equals_eager_step_with_rollouts_in_between():
    """The regression behind rl/modules.py:_TrainLinear: four updates with a synthetic rollout (alg.act / process_env_step /
    compute_returns, i.e. eager GPU work and fresh storage contents) between them.  With torch's own column reduction in
    the captured step the second update already differed (one bias gradient wrong by 100 %, tools/gpu_ppo_graph_check.py);
    with the library's column sum the HIP-graph path and the eager device path agree to the last bit."""
    N, T = 512, 24
    res = {}
    for graph in ("0", "1"):
        os.environ["GRX_PPO_GRAPH"] = graph
        try:
            torch.manual_seed(0)
            ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=0.2)
            alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, entropy_coef=0.01, learning_rate=1e-4,
                      schedule="adaptive", desired_kl=0.03, device="cuda:0")
        finally:
            del os.environ["GRX_PPO_GRAPH"]
        assert alg._use_graph == (graph == "1")
        alg.init_storage(N, T)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        snaps = []
        for u in range(4):
            torch.manual_seed(50 + u)
            with torch.inference_mode():
                for _ in range(T):
                    o = torch.randn(N, 39, device="cuda:0", generator=g); c = torch.randn(N, 168, device="cuda:0", generator=g)
                    alg.act(o, c)
                    r = torch.randn(N, device="cuda:0", generator=g) * 0.1
                    d = torch.rand(N, device="cuda:0", generator=g) < 0.02
                    alg.process_env_step(r, d, {"time_outs": torch.zeros(N, device="cuda:0", dtype=torch.bool)})
                alg.compute_returns(c)
            out = alg.update()
            alg.clear_storage()
            assert all(abs(v) > 1e-7 for v in out)
            snaps.append(torch.cat([p.detach().flatten() for p in ac.parameters()]).clone())
        res[graph] = snaps
    for u in range(4):
        a, b = res["0"][u], res["1"][u]
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-6, (u, float((a - b).abs().max()))


def test_colsum_matches_torch():
    from wiki_grx_gym_amd.rl.fused_loss import colsum
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for rows, cols in ((1, 1), (7, 10), (10485, 128), (10485, 512), (24576, 1), (300, 700)):
        x = torch.randn(rows, cols, device="cuda:0", generator=g)
        ref = x.double().sum(0)
        got = colsum(x)
        assert got.shape == (cols,) and float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
        assert torch.equal(got, colsum(x))   # deterministic


def test_graphed_policy_step_matches_eager():
    """PPO.act through the captured policy step: same means / values / sigma as the eager call, log-prob consistent with the
    sampled actions, fresh noise on every replay, and the current parameters (they change under Adam) are what it reads."""
    torch.manual_seed(0)
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[64, 32], critic_hidden_dims=[64, 32], activation="elu", init_noise_std=0.3)
    alg = PPO(ac, device="cuda:0")
    assert alg._use_act_graph
    alg.init_storage(128, 4)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    prev_actions = None
    for it in range(3):
        o = torch.randn(128, 39, device="cuda:0", generator=g); c = torch.randn(128, 168, device="cuda:0", generator=g)
        with torch.inference_mode():
            a = alg.act(o, c).clone()
            t = alg.transition
            mu, sg, v, lp = t.action_mean.clone(), t.action_sigma.clone(), t.values.clone(), t.actions_log_prob.clone()
            mu_e, v_e = ac.actor(o), ac.critic(c)
        assert alg._act_graph is not None
        assert torch.allclose(mu, mu_e, atol=1e-6) and torch.allclose(v, v_e, atol=1e-6) and torch.allclose(sg, ac.std.expand_as(mu))
        ref_lp = torch.distributions.Normal(mu, sg).log_prob(a).sum(-1)
        assert torch.allclose(lp.reshape(-1), ref_lp, atol=1e-4)
        z = (a - mu) / sg
        assert 0.8 < float(z.std()) < 1.2 and abs(float(z.mean())) < 0.15
        assert prev_actions is None or not torch.equal(z, prev_actions)
        prev_actions = z
        with torch.no_grad():   # an optimizer step in between: the graph must see the new weights
            for p in ac.parameters():
                p.add_(0.01 * torch.randn_like(p))


def test_fused_transition_store_matches_the_torch_bookkeeping():
    """grx_ppo_store_transition (one kernel per rollout step) against PPO.process_env_step + RolloutStorage.add_transitions +
    the runner's running episode sums spelled in torch (rsl_rl ppo.py:184-197, rollout_storage.py:23-59,
    on_policy_runner.py:170-181): identical storage rows and logging buffers, three steps, time-out bootstrap included."""
    from wiki_grx_gym_amd.rl.ppo import PPO
    from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
    dev = "cuda:0"
    N, T, no, npri, na = 300, 3, 39, 168, 10
    out = {}
    for fused in ("1", "0"):
        os.environ["GRX_PPO_FUSED_STORE"] = fused
        torch.manual_seed(0)
        ac = ActorCriticMLP(no, npri, na, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=0.2)
        alg = PPO(actor_critic=ac, gamma=0.99, device=dev)
        alg._use_act_graph = False
        alg.init_storage(N, T)
        g = torch.Generator(device=dev).manual_seed(1)
        cur_rew, cur_len = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        done_rew, done_len = torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev)
        with torch.inference_mode():
            for t in range(T):
                obs = torch.randn(N, no, device=dev, generator=g); pri = torch.randn(N, npri, device=dev, generator=g)
                torch.manual_seed(10 + t)
                alg.act(obs, pri)
                rew = torch.randn(N, device=dev, generator=g)
                dones = torch.rand(N, device=dev, generator=g) < 0.2
                to = dones & (torch.rand(N, device=dev, generator=g) < 0.5)
                alg.process_env_step(rew, dones, {"time_outs": to}, log=(cur_rew, cur_len, done_rew[t], done_len[t]))
        st = alg.storage
        assert st.step == T
        out[fused] = {k: getattr(st, k).clone() for k in ("observations", "pri_observations", "actions", "mu", "sigma", "values", "actions_log_prob", "rewards", "dones")}
        m = st.dones.squeeze(-1).bool()
        out[fused].update(cur_rew=cur_rew.clone(), cur_len=cur_len.clone(), done_rew=done_rew[m].clone(), done_len=done_len[m].clone())
    os.environ.pop("GRX_PPO_FUSED_STORE", None)
    assert out["1"]["dones"].sum() > 10
    for k in out["1"]:
        assert torch.equal(out["1"][k], out["0"][k]), k


def _toy_alg(seed=0):
    torch.manual_seed(seed)
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[64, 32], critic_hidden_dims=[64, 32], activation="elu", init_noise_std=0.2, set_std=False)
    alg = PPO(actor_critic=ac, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, gamma=0.99, lam=0.95, value_loss_coef=1.0,
              entropy_coef=0.01, learning_rate=1e-5, learning_rate_min=1e-6, learning_rate_max=1e-2, max_grad_norm=1.0,
              use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.03, device="cuda:0")
    alg.init_storage(256, 8)
    return alg


def _toy_rollout(alg, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    with torch.inference_mode():
        for t in range(8):
            obs = torch.randn(256, 39, device="cuda:0", generator=g); pri = torch.randn(256, 168, device="cuda:0", generator=g)
            torch.manual_seed(100 * seed + t)
            alg.act(obs, pri)
            alg.process_env_step(torch.randn(256, device="cuda:0", generator=g), torch.rand(256, device="cuda:0", generator=g) < 0.05, {})
        alg.compute_returns(torch.randn(256, 168, device="cuda:0", generator=g))


def test_resume_keeps_the_adaptive_learning_rate_alive(tmp_path):
    """runner.load() on a HIP device: after optimizer.load_state_dict the learning rate Adam uses must still be the
    device scalar the adaptive-KL rule writes, captured graphs must be dropped, std must be copied in place.  A resumed
    trainer continues bit for bit like the one that never stopped -- and its learning rate keeps moving."""
    a = _toy_alg()
    for it in range(2):
        _toy_rollout(a, it); torch.manual_seed(7 + it); a.update(); a.clear_storage()
    ckpt = {"model": {k: v.clone() for k, v in a.actor_critic.state_dict().items()}, "opt": a.optimizer.state_dict()}
    torch.save(ckpt, tmp_path / "c.pt")
    lr_at_save = a.learning_rate
    for it in range(2, 4):
        _toy_rollout(a, it); torch.manual_seed(7 + it); a.update(); a.clear_storage()
    b = _toy_alg(seed=5)                                   # different initial weights: everything must come from the checkpoint
    _toy_rollout(b, 9); b.update(); b.clear_storage()     # ... and graphs captured BEFORE the load must not survive it
    std_storage = b.actor_critic.std.data_ptr()
    ck = torch.load(tmp_path / "c.pt", map_location="cuda:0", weights_only=False)
    b.actor_critic.load_state_dict(ck["model"]); b.invalidate_graphs(); b.load_optimizer_state(ck["opt"])
    assert b.actor_critic.std.data_ptr() == std_storage
    assert b.optimizer.param_groups[0]["lr"] is b._lr_t and abs(b.learning_rate - lr_at_save) < 1e-12
    assert b._graph is None and b._act_graph is None
    _toy_rollout(b, 99); b.clear_storage()                 # (recapture the policy graphs here: the capturing call draws its noise
                                                           #  at a different Philox offset than a replay does)
    for it in range(2, 4):
        _toy_rollout(b, it); torch.manual_seed(7 + it); b.update(); b.clear_storage()
    assert b.learning_rate != lr_at_save                   # the adaptive rule still reaches the optimizer
    assert abs(b.learning_rate - a.learning_rate) < 1e-12
    for (n, p), q in zip(a.actor_critic.named_parameters(), b.actor_critic.parameters()):
        assert torch.equal(p, q), n
    # a reference-style checkpoint (python-float lr, not capturable) is coerced as well
    sd = a.optimizer.state_dict()
    sd["param_groups"][0].update(lr=3e-4, capturable=False, fused=None)
    for st in sd["state"].values():
        st["step"] = st["step"].cpu() if torch.is_tensor(st["step"]) else st["step"]
    b.load_optimizer_state(sd)
    g = b.optimizer.param_groups[0]
    assert g["lr"] is b._lr_t and abs(float(b._lr_t) - 3e-4) < 1e-10 and g["capturable"] and g["fused"]
    _toy_rollout(b, 11); b.update()
    assert all(torch.isfinite(p).all() for p in b.actor_critic.parameters())


def _bucket_worker(out, force, port):
    """one process, one rank: the multi-rank update (flat bucket, two captured halves around an RCCL all-reduce) when `force`"""
    import os, sys, time
    sys.stderr = sys.stdout = open(out + ".log", "w", buffering=1)
    os.dup2(sys.stderr.fileno(), 2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["GRX_PPO_FORCE_BUCKET"] = "1"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    a = _toy_alg()
    assert (a._bucket is not None) == bool(force)
    times = []
    for it in range(4):
        _toy_rollout(a, it); torch.manual_seed(7 + it)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a.update()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        a.clear_storage()
    if force:
        assert a._graph_back is not None                       # the captured halves ran, not the eager loop
        assert all(p.grad.data_ptr() == a._bucket.data_ptr() + 4 * o for p, o in zip(a._params, a._offsets()))
    torch.save({"params": [p.detach().cpu() for p in a.actor_critic.parameters()], "lr": a.learning_rate, "t": min(times[1:])}, out)
    dist.destroy_process_group()


def test_multi_rank_update_is_the_captured_step_around_one_all_reduce(tmp_path):
    """with a process group the update keeps the HIP graph -- gradients accumulate straight into a flat
    bucket (views, no per-parameter copies), ONE RCCL all-reduce between two captured halves.  On one rank (world 1,
    GRX_PPO_FORCE_BUCKET=1) it must reproduce the single-process captured step bit for bit.  (No multi-GPU box is
    available to this suite: the N>1 arithmetic is covered by the gloo world-2 tests in test_distributed.py.)"""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = []
    for force in (0, 1):
        out = str(tmp_path / f"b{force}.pt")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        p = ctx.Process(target=_bucket_worker, args=(out, force, port)); p.start(); p.join(300)
        assert p.exitcode == 0, open(out + ".log").read()[-3000:]
        res.append(torch.load(out))
    assert res[0]["lr"] == res[1]["lr"]
    for x, y in zip(res[0]["params"], res[1]["params"]):
        assert torch.equal(x, y)
    print(f"update: plain graph {res[0]['t']*1e3:.2f} ms, bucket + all-reduce between captured halves {res[1]['t']*1e3:.2f} ms")


def test_fused_loss_propagates_nan_like_torch():
    """torch.max / torch.clamp propagate NaN; the kernel must too, so that a NaN ratio or value makes the LOSS NaN
    and the update's NaN-skip fires (a finite loss over NaN gradients would poison every parameter)."""
    from wiki_grx_gym_amd.rl.fused_loss import fused_ppo_loss
    torch.manual_seed(0)
    B, A = 512, 10
    dev = "cuda:0"
    mk = lambda *s: torch.randn(*s, device=dev)
    base = dict(mu=mk(B, A) * 0.1, std=torch.full((A,), 0.3, device=dev), value=mk(B, 1), actions=mk(B, A) * 0.3,
                old_logp=mk(B, 1) * 0.1 - 2, old_mu=mk(B, A) * 0.1, old_sigma=torch.full((B, A), 0.3, device=dev),
                adv=mk(B, 1), ret=mk(B, 1), tv=mk(B, 1))
    def run(**over):
        d = {k: v.clone() for k, v in base.items()}
        for k, v in over.items():
            d[k][7] = v
        out = fused_ppo_loss(d["mu"].requires_grad_(), d["std"].requires_grad_(), d["value"].requires_grad_(), d["actions"],
                             d["old_logp"], d["old_mu"], d["old_sigma"], d["adv"], d["ret"], d["tv"], 0.2, 1.0, 0.01, True)
        return out
    assert all(torch.isfinite(o).all() for o in run()[:3])
    assert torch.isnan(run(mu=float("nan"))[2])          # NaN ratio -> NaN surrogate -> NaN loss
    assert torch.isnan(run(value=float("nan"))[2])       # NaN value -> NaN value loss -> NaN loss
    assert torch.isnan(run(old_logp=float("nan"))[2])
    # ... and the update then leaves every parameter untouched
    a = _toy_alg()
    _toy_rollout(a, 0)
    a.storage.values[3, 5] = float("nan")
    before = [p.detach().clone() for p in a.actor_critic.parameters()]
    a.update()
    after = list(a.actor_critic.parameters())
    assert all(torch.isfinite(p).all() for p in after)


@pytest.mark.parametrize("M,K,N", [(4096, 39, 512), (4096, 168, 512), (4096, 512, 256), (4096, 256, 128), (4100, 130, 96), (70, 33, 64),
                                   (4096, 128, 10), (513, 128, 1)])
def test_mlp_layer_kernel_matches_torch(M, K, N):
    """grx_mlp_layer (f32-input MFMA, bias + ELU epilogue; narrow output layers on a masked tile) against torch, asymmetric operands,
    ragged M / K / N."""
    from wiki_grx_gym_amd.rl.fused_loss import load_ppo_library, _layer
    lib = load_ppo_library()
    g = torch.Generator(device="cuda:0").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda:0", generator=g)
    w = torch.randn(N, K, device="cuda:0", generator=g) / K ** 0.5
    w[:, 0] += torch.arange(N, device="cuda:0") * 0.01          # (row / column swaps must not cancel)
    b = torch.randn(N, device="cuda:0", generator=g)
    st = torch.cuda.current_stream().cuda_stream
    for elu in (True, False):
        y = _layer(lib, x, w, b, elu, st)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        ref = torch.nn.functional.elu(ref) if elu else ref
        assert (y.double() - ref).abs().max() < 2e-5 * (1 + ref.abs().max())
    y0 = _layer(lib, x, w, None, False, st)
    assert (y0.double() - x.double() @ w.double().t()).abs().max() < 2e-5 * (1 + ref.abs().max())


def test_fused_policy_step_matches_the_torch_distribution_path():
    """The rollout's policy step through libgrx_ppo.so (hidden layers on the matrix cores, output layer + sample + log-prob in
    one kernel; critic likewise) against rsl_rl's spelling: Normal(mu, std), a = mu + std * eps, log_prob summed over actions."""
    from wiki_grx_gym_amd.rl.fused_loss import mlp_can_fuse, mlp_forward, policy_act
    torch.manual_seed(3)
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu",
                        init_noise_std=0.2).to("cuda:0")
    with torch.no_grad():
        ac.std.mul_(torch.linspace(0.5, 1.5, 10, device="cuda:0"))
    obs, pri = torch.randn(4096, 39, device="cuda:0"), torch.randn(4096, 168, device="cuda:0")
    eps = torch.randn(4096, 10, device="cuda:0")
    assert mlp_can_fuse(ac.actor, obs) and mlp_can_fuse(ac.critic, pri)
    with torch.no_grad():
        actions, logp, mu, sigma = policy_act(ac.actor, ac.std.detach(), obs, eps)
        value = mlp_forward(ac.critic, pri)
        ac.update_distribution(obs)
        a_ref = ac.action_mean + ac.action_std * eps
        assert (mu - ac.action_mean).abs().max() < 1e-5 and torch.equal(sigma, ac.action_std)
        assert (actions - a_ref).abs().max() < 1e-5
        assert (logp - ac.get_actions_log_prob(actions)).abs().max() < 1e-4
        assert (value - ac.evaluate(pri)).abs().max() < 1e-5
    # ... and PPO.act() uses it inside its captured graphs: same transition tensors as the eager distribution path up to rounding
    alg = _toy_alg()
    assert alg._use_act_graph
    o, p = torch.randn(256, 39, device="cuda:0"), torch.randn(256, 168, device="cuda:0")
    with torch.inference_mode():
        torch.manual_seed(11); alg.act(o, p); alg._join_critic()
        t = alg.transition
        assert alg._act_fused
        got = [x.clone() for x in (t.actions, t.values, t.actions_log_prob, t.action_mean, t.action_sigma)]
        torch.manual_seed(11); alg.act(o, p); alg._join_critic()      # replay: fresh noise, same means
        assert torch.equal(t.action_mean, got[3]) and not torch.equal(t.actions, got[0])
        ac2 = alg.actor_critic
        ac2.update_distribution(o)
        assert (got[3] - ac2.action_mean).abs().max() < 1e-5 and (got[1].reshape(-1) - ac2.evaluate(p).reshape(-1)).abs().max() < 1e-5
        assert (got[2].reshape(-1) - ac2.get_actions_log_prob(got[0])).abs().max() < 1e-4


def test_fused_step_tail_tracks_the_torch_tail(monkeypatch):
    """libgrx_ppo.so's step tail (grx_ppo_step_tail: adaptive learning rate, NaN-skip, clip_grad_norm_, Adam.step() in two launches) against
    the torch tail it replaces (ppo.py:264-311 through PPO._device_lr_update, nn.utils.clip_grad_norm_, torch.optim.Adam fused / capturable):
    same rollouts, three updates each -- the learning rate takes the same decisions bit for bit, the step counters agree, parameters and
    moments agree to rounding (the two Adam kernels contract their multiply-adds differently); a minibatch with a non-finite loss leaves
    parameters, moments and step counters untouched in both."""
    algs = {}
    for tail in ("1", "0"):
        monkeypatch.setenv("GRX_PPO_FUSED_TAIL", tail)
        a = _toy_alg()
        assert a._fused_tail == (tail == "1")
        for it in range(3):
            _toy_rollout(a, it); torch.manual_seed(7 + it); a.update(); a.clear_storage()
        algs[tail] = a
    f, t = algs["1"], algs["0"]
    assert f._tail is not None and t._tail is None
    assert f.learning_rate == t.learning_rate and f.learning_rate != 1e-5
    for (n, p), q in zip(f.actor_critic.named_parameters(), t.actor_critic.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), (n, float((p - q).abs().max()))
    for p, q in zip(f.actor_critic.parameters(), t.actor_critic.parameters()):
        sf, st = f.optimizer.state[p], t.optimizer.state[q]
        assert float(sf["step"]) == float(st["step"]) == 24.0          # 3 updates x 2 epochs x 4 minibatches
        assert torch.allclose(sf["exp_avg"], st["exp_avg"], rtol=1e-4, atol=1e-8) and torch.allclose(sf["exp_avg_sq"], st["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    # NaN-skip: poison the returns of one rollout -> every minibatch loss is non-finite -> nothing moves
    for a in (f, t):
        before = [p.detach().clone() for p in a.actor_critic.parameters()]
        steps = [float(a.optimizer.state[p]["step"]) for p in a.actor_critic.parameters()]
        _toy_rollout(a, 5); a.storage.returns.fill_(float("nan")); a.update(); a.clear_storage()
        for p, b, s in zip(a.actor_critic.parameters(), before, steps):
            assert torch.equal(p, b) and float(a.optimizer.state[p]["step"]) == s
