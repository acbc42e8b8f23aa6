"""Physics pins of the oracle (the reference's PhysX is unobtainable: parity unpinned).
Invariants: ABA vs an independent RNEA, energy / momentum conservation, free fall, static stance,
no tunnelling, joint limits."""
import numpy as np
import pytest
import torch

from tests.helpers import make_cfg
from wiki_grx_gym_amd.envs import build_config

G9 = 9.81


def make(N=4, precision="f64", gravity=True, task="GR1T1", **grx):
    from oracle.binding import OracleSim
    cfg = make_cfg(task=task)
    if not gravity:
        cfg.sim.gravity = [0.0, 0.0, 0.0]
    for k, v in grx.items():
        setattr(cfg.sim.grx, k, v)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    return OracleSim(c, precision, keep), cfg, meta


def random_state(sim, seed, z=5.0, vel=1.0):
    g = torch.Generator().manual_seed(seed)
    N, nd = sim.num_envs, sim.num_dofs
    root = torch.zeros(N, 13)
    root[:, 2] = z
    q = torch.randn(N, 4, generator=g)
    root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    root[:, 7:13] = torch.randn(N, 6, generator=g) * vel
    q0 = torch.tensor([0, 0, -0.2618, 0.5236, -0.2618] * 2)[:nd] if nd == 10 else torch.zeros(nd)
    dq = q0 + (torch.rand(N, nd, generator=g) - 0.5) * 0.15      # inside the joint limits
    dqd = torch.randn(N, nd, generator=g) * 3 * vel
    sim.set_state(root.contiguous(), dq.contiguous(), dqd.contiguous())


@pytest.mark.parametrize("task", ["GR1T1", "GR1T2"])
def test_aba_matches_rnea(task):
    """ID(FD(tau)) == tau and the free-floating base carries no residual wrench (fp64, <= 1e-9)."""
    sim, _, _ = make(task=task)
    random_state(sim, 0)
    rng = np.random.RandomState(0)
    for e in range(sim.num_envs):
        tau = rng.randn(sim.num_dofs) * 30
        qdd, acc = sim.forward_dynamics(e, tau)
        tau2, wrench = sim.inverse_dynamics(e, qdd, acc)
        np.testing.assert_allclose(tau2, tau, atol=1e-9, rtol=1e-10)
        assert np.abs(wrench).max() < 1e-8


def _fall(dt, n, seed=1, vel=0.3):
    from oracle.binding import OracleSim
    cfg = make_cfg()
    cfg.sim.dt = dt
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 4)
    sim = OracleSim(c, "f64", keep)
    random_state(sim, seed, z=50.0, vel=vel)
    out = []
    for i in range(sim.num_envs):
        e0 = sim.energy(i)
        sim.substeps(i, np.zeros(sim.num_dofs), n, contact=False)
        e1 = sim.energy(i)
        out.append((e0, e1))
    return out


def test_free_fall_and_momentum():
    """Zero torque, no contact: dP = M g t; horizontal momentum and energy are conserved up to the
    first-order integration error of the generalized-coordinate Euler step, which shrinks with dt."""
    coarse, fine = _fall(0.002, 250), _fall(0.0005, 1000)
    for (e0, e1), (f0, f1) in zip(coarse, fine):
        M = e1["mass"]
        dP, dPf = e1["P"] - e0["P"], f1["P"] - f0["P"]
        want = np.array([0, 0, -M * G9 * 0.5])
        assert np.abs(dP - want).max() < 0.02 * M                  # < 0.02 m/s of COM velocity
        assert np.abs(dPf - want).max() < 0.5 * np.abs(dP - want).max() + 1e-9
        E0, E1 = e0["KE"] + e0["PE"], e1["KE"] + e1["PE"]
        # free fall under semi-implicit Euler lags by g t dt / 2 in height (M g * that in energy): first order in dt
        assert abs(E1 - E0) < 1.5 * M * G9 * (0.5 * G9 * 0.5 * 0.002) + 0.05 * e0["KE"]
        assert abs((f1["KE"] + f1["PE"]) - (f0["KE"] + f0["PE"])) < 0.5 * abs(E1 - E0) + 1e-9


def test_energy_without_gravity_is_conserved():
    sim, _, _ = make(gravity=False, k_limit=0.0)
    random_state(sim, 2, vel=0.5)
    for i in range(sim.num_envs):
        E0 = sim.energy(i)["KE"]
        sim.substeps(i, np.zeros(sim.num_dofs), 500, contact=False)   # 1 s
        E1 = sim.energy(i)["KE"]
        assert abs(E1 - E0) / E0 < 2e-2


def test_static_stance_supports_the_weight():
    """PD at the default pose on the plane: sum of foot normal forces = m g +- 1 %, no chatter."""
    sim, cfg, meta = make(N=2)
    sim.reset_all()
    root = sim.tensor("ROOT_STATES").clone()
    root[:, 3:7] = torch.tensor([0, 0, 0, 1.0])
    root[:, 2] = 0.90
    sim.set_state(root.contiguous(), None, None)
    act = torch.zeros(2, 10)
    fz = []
    for i in range(40):
        sim.step(act, 0.0, i + 1)
        fz.append(sim.tensor("FEET_CONTACT_FORCE")[0, :, 2].sum().item())
    fz = np.array(fz[15:35])
    W = meta["model"].total_mass() * G9
    assert abs(fz.mean() - W) / W < 0.01
    assert fz.std() / W < 0.02
    assert not sim.tensor("RESET")[0].item()
    assert sim.tensor("FEET_CONTACT").all()


def test_no_tunnelling_from_reset_height():
    sim, cfg, _ = make(N=8)
    sim.reset_all()
    act = torch.zeros(8, 10)
    zmin = 10.0
    for i in range(60):
        sim.step(act, 0.0, i + 1)
        zmin = min(zmin, sim.tensor("FEET_POS")[:, :, 2].min().item())
    assert zmin > 0.055 - 0.02        # sole never sinks more than 2 cm below the ground plane


def test_joint_limits_hold():
    sim, cfg, meta = make(N=1)
    random_state(sim, 3, z=5.0, vel=0.0)
    tau = np.array(meta["model"].dof_effort) * 1.0                     # saturate every motor
    sim.substeps(0, tau, 1500, contact=False)
    q = sim.tensor("DOF_POS")[0].numpy()
    assert (q <= meta["model"].dof_upper + 0.1).all() and (q >= meta["model"].dof_lower - 0.1).all()
    assert np.isfinite(q).all()


def test_f32_tracks_f64_over_one_policy_step():
    s64, cfg, _ = make(N=16, precision="f64")
    s32, _, _ = make(N=16, precision="f32")
    s64.reset_all(); s32.reset_all()
    g = torch.Generator().manual_seed(0)
    for i in range(12):    # includes the landing
        a = (torch.rand(16, 10, generator=g) - 0.5) * 0.6
        # re-synchronise, then compare one step
        s32.set_state(s64.tensor("ROOT_STATES").clone().contiguous(), s64.tensor("DOF_POS").clone().contiguous(), s64.tensor("DOF_VEL").clone().contiguous())
        s64.set_state(s64.tensor("ROOT_STATES").clone().contiguous(), s64.tensor("DOF_POS").clone().contiguous(), s64.tensor("DOF_VEL").clone().contiguous())
        s64.step(a, 3.0, i + 1); s32.step(a, 3.0, i + 1)
        np.testing.assert_allclose(s32.tensor("DOF_POS").numpy(), s64.tensor("DOF_POS").numpy(), atol=2e-4)
        np.testing.assert_allclose(s32.tensor("ROOT_STATES").numpy(), s64.tensor("ROOT_STATES").numpy(), atol=2e-4, rtol=1e-4)
