"""PPO side (stays PyTorch): this build's rl/ package against the reference rsl_rl run on the same
seeded inputs (tests/golden/ppo.npz, SURVEY G-10): GAE returns, advantage normalisation,
time-out bootstrap, log-probs, the minibatch index stream, and one full update() (8 optimizer
steps incl. adaptive-KL learning rate) ending in the same weights."""
import os

import numpy as np
import pytest
import torch

from wiki_grx_gym_amd.rl import PPO, ActorCriticMLP, OnPolicyRunner

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(d, device="cpu"):
    N, T, no, npri, na = 16, 8, 39, 168, 10
    ac = ActorCriticMLP(no, npri, na, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=0.2)
    with torch.no_grad():
        for k, v in ac.state_dict().items():
            v.copy_(torch.tensor(d["w0_" + k]))
    alg = PPO(actor_critic=ac, num_learning_epochs=2, num_mini_batches=4, clip_param=0.2, gamma=0.99, lam=0.95,
              value_loss_coef=1.0, entropy_coef=0.01, learning_rate=1e-4, learning_rate_min=1e-5, learning_rate_max=1e-3,
              max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.03, device=device)
    alg.init_storage(N, T)
    alg._use_act_graph = False   # the rollout below injects its noise by patching Normal.sample: not capturable
    return ac, alg, N, T


@pytest.mark.gpu
def test_device_update_path_matches_reference(monkeypatch):
    """The sync-free HIP update path (device-side adaptive LR, fused Adam, found_inf NaN-skip, HIP graph), with the loss
    spelled in torch so that its rounding is the reference's (see `loose` above)."""
    monkeypatch.setenv("GRX_PPO_FUSED_LOSS", "0")
    _check_against_reference("cuda")


@pytest.mark.gpu
def test_device_update_path_with_fused_loss_matches_reference():
    """The default HIP path: the same, with the loss and its gradients from libgrx_ppo.so (tests/test_ppo_gpu.py pins
    that kernel against the torch expression at 2e-5; here the whole update against the reference's)."""
    _check_against_reference("cuda", loose=True)


@pytest.mark.gpu
def test_rccl_bucket_path_matches_reference(monkeypatch):
    """The multi-GPU update path (flat [grads | KL] bucket, ONE RCCL all-reduce per optimizer step) on a one-rank
    process group: the collective is an identity, so the result must still equal the reference's update."""
    import torch.distributed as dist
    monkeypatch.setenv("GRX_PPO_FORCE_BUCKET", "1")
    monkeypatch.setenv("GRX_PPO_FUSED_LOSS", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        _check_against_reference("cuda")
    finally:
        dist.destroy_process_group()


def test_rollout_returns_and_update_match_reference():
    _check_against_reference("cpu")


def _check_against_reference(device, loose=False):
    d = np.load(os.path.join(G, "ppo.npz"))
    ac, alg, N, T = _build(d, device)
    assert sorted(ac.state_dict()) == sorted(k[3:] for k in d.files if k.startswith("w0_"))   # checkpoint key layout
    dv = lambda k: torch.tensor(d[k]).to(device)
    obs, pri, rew, done, tos, eps = dv("obs"), dv("pri"), dv("rew"), dv("done"), dv("time_outs"), dv("eps")
    with torch.inference_mode():
        for t in range(T):
            orig = torch.distributions.Normal.sample
            torch.distributions.Normal.sample = lambda self, _e=eps[t]: self.mean + self.stddev * _e
            try:
                a = alg.act(obs[t], pri[t])
            finally:
                torch.distributions.Normal.sample = orig
            np.testing.assert_allclose(a.cpu().numpy(), d["actions"][t], rtol=1e-5, atol=1e-6)
            alg.process_env_step(rew[t].clone(), done[t], {"time_outs": tos[t]})
        alg.compute_returns(pri[T])
    st = alg.storage
    for name, got in (("values", st.values), ("rewards", st.rewards), ("returns", st.returns), ("advantages", st.advantages),
                      ("log_prob", st.actions_log_prob), ("mu", st.mu), ("sigma", st.sigma)):
        np.testing.assert_allclose(got.cpu().numpy(), d[name], rtol=2e-5, atol=2e-6, err_msg=name)
    torch.manual_seed(123)
    np.testing.assert_array_equal(torch.randperm(4 * (N * T // 4)).numpy(), d["perm"])
    perm = torch.tensor(d["perm"]).to(device)
    orig_rp = torch.randperm
    torch.randperm = lambda *a, **k: perm.clone()      # same minibatch stream on any device
    try:
        vl, sl = alg.update()
    finally:
        torch.randperm = orig_rp
    # loose: in the 7th of the 8 minibatch steps of this fixture one sample's probability ratio sits 9 ulp below the
    # upper clip bound 1 + clip_param, where the surrogate's gradient jumps from -A to 0.  An implementation whose
    # log-prob rounds differently in the last bit (the fused HIP loss sums the ten action terms in another order than
    # torch) lands on the other side for that sample; both are correct, but the last step then differs at 1e-3.
    assert vl == pytest.approx(float(d["value_loss"]), rel=1e-3 if loose else 1e-4)
    assert sl == pytest.approx(float(d["surrogate_loss"]), rel=5e-3 if loose else 1e-4, abs=1e-6)
    assert alg.learning_rate == pytest.approx(float(d["lr_after"]), rel=1e-6)   # the device path keeps the LR in fp32
    for k, v in ac.state_dict().items():
        np.testing.assert_allclose(v.detach().cpu().numpy(), d["w1_" + k], rtol=2e-4 if device == "cuda" else 1e-4,
                                   atol=2.5e-3 if loose else 2e-6, err_msg=k)   # loose: one Adam step at lr 1e-3


def test_checkpoint_roundtrip_and_std_quirk(tmp_path):
    """model_<it>.pt keys as the reference (on_policy_runner.py:297-331); loading overwrites std with
    set_noise_std = 1.0 unless set_std=False (actor_critic_mlp.py:116-134)."""
    d = np.load(os.path.join(G, "ppo.npz"))
    ac, alg, _, _ = _build(d)
    path = tmp_path / "model_0.pt"
    torch.save({"model_state_dict": ac.state_dict(), "optimizer_state_dict": alg.optimizer.state_dict(), "iter": 7, "infos": None}, path)
    loaded = torch.load(path, weights_only=False)
    assert set(loaded) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    ac2 = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], init_noise_std=0.2)
    ac2.load_state_dict(loaded["model_state_dict"])
    assert torch.allclose(ac2.std.data, torch.ones(10))                  # the quirk
    assert torch.equal(ac2.actor.model[0].weight, ac.actor.model[0].weight)
    ac3 = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], init_noise_std=0.5, set_std=False)
    ac3.load_state_dict(loaded["model_state_dict"])
    assert torch.allclose(ac3.std.data, torch.full((10,), 0.2))


def test_parameter_count_of_the_registered_policy():
    ac = ActorCriticMLP(39, 168, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], init_noise_std=0.2)
    assert sum(p.numel() for p in ac.parameters()) == 436885          # 1 747 540 B fp32


def test_actor_critic_init_options_of_the_32_dof_task():
    """GR1T1FullBodyCfgPPO's two policy options (not in the reference, default = the reference's behaviour): one initial action noise per action, and an
    output-layer gain that starts the mean action near zero.  Defaults leave the reference's initialisation untouched (same RNG stream, same weights)."""
    from wiki_grx_gym_amd.envs import GR1T1FullBodyCfgPPO
    from wiki_grx_gym_amd.envs.config import class_to_dict
    from wiki_grx_gym_amd.rl.modules import ActorCriticMLP
    pol = class_to_dict(GR1T1FullBodyCfgPPO())["policy"]
    assert pol["init_noise_std"] == [0.2] * 12 + [0.05] * 20 and pol["actor_output_gain"] == 0.01
    torch.manual_seed(3)
    a = ActorCriticMLP(105, 234, 32, **pol)
    torch.manual_seed(3)
    ref = ActorCriticMLP(105, 234, 32, **{**pol, "init_noise_std": 0.2, "actor_output_gain": 1.0})
    assert torch.equal(a.std.detach(), torch.tensor([0.2] * 12 + [0.05] * 20))
    la, lr = [m for m in a.actor.model if isinstance(m, torch.nn.Linear)], [m for m in ref.actor.model if isinstance(m, torch.nn.Linear)]
    for x, y in zip(la[:-1], lr[:-1]):
        assert torch.equal(x.weight, y.weight) and torch.equal(x.bias, y.bias)
    assert torch.allclose(la[-1].weight, lr[-1].weight * 0.01) and torch.allclose(la[-1].bias, lr[-1].bias * 0.01)
    for x, y in zip(a.critic.model, ref.critic.model):
        if isinstance(x, torch.nn.Linear):
            assert torch.equal(x.weight, y.weight)
    obs = torch.randn(16, 105)
    assert a.act_inference(obs).abs().max() < 0.02 < ref.act_inference(obs).abs().max()
    # with fixed_std the per-action list reaches the distribution as a tensor (it used to be added to one as a Python list)
    f = ActorCriticMLP(105, 234, 32, **{**pol, "fixed_std": True})
    f.update_distribution(obs)
    assert torch.equal(f.action_std[0], torch.tensor([0.2] * 12 + [0.05] * 20)) and f.act(obs).shape == (16, 32)
