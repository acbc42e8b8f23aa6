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


# ------------------------------------------------------------------ self-collision, restitution (round 2)
def _leg_squeeze(self_collisions, precision="f64", steps=400):
    """Free-floating robot far above the ground, no gravity: the hip-roll motors drive the legs into each other."""
    from oracle.binding import OracleSim
    cfg = make_cfg()
    cfg.sim.gravity = [0.0, 0.0, 0.0]
    cfg.asset.self_collisions = 0 if self_collisions else 1      # the reference's flag: 0 = collide (legged_robot_config.py:121)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, 1)
    sim = OracleSim(c, precision, keep)
    rm = meta["model"]
    root = torch.zeros(1, 13); root[0, 2] = 50.0; root[0, 6] = 1.0
    q0 = torch.tensor([[0, 0, -0.2618, 0.5236, -0.2618] * 2], dtype=torch.float32)
    sim.set_state(root.contiguous(), q0.contiguous(), torch.zeros(1, 10))
    tau = np.zeros(10)
    names = rm.dof_names
    for j, n in enumerate(names):
        if "hip_roll" in n:
            tau[j] = (-1.0 if n.startswith("l") else 1.0) * rm.dof_effort[j]     # adduct both legs at the effort limit
    P0 = sim.energy(0)["P"].copy()
    m = c.model

    def overlap():   # deepest overlap over the listed sphere pairs, from the oracle's own kinematics
        deep = 0.0
        for k in range(m.num_pairs):
            a, b = m.pair_a[k], m.pair_b[k]
            Ra, pa = sim.body_pose(0, m.sph_body[a]); Rb, pb = sim.body_pose(0, m.sph_body[b])
            ca = pa + Ra @ np.array(m.sph_pos[a][:]); cb = pb + Rb @ np.array(m.sph_pos[b][:])
            deep = max(deep, m.sph_radius[a] + m.sph_radius[b] - np.linalg.norm(ca - cb))
        return deep
    deepest, lf_best = 0.0, np.zeros((37, 3))
    for _ in range(steps // 5):
        sim.substeps(0, tau, 5, contact=True)
        deepest = max(deepest, overlap())
        lf = sim.link_forces(0)
        if np.abs(lf).max() > np.abs(lf_best).max():
            lf_best = lf
    P1 = sim.energy(0)["P"].copy()
    return deepest, P1 - P0, lf_best, sim, meta


def test_self_collision_keeps_the_legs_apart():
    """legged_robot_config.py:121 self_collisions = 0 (enabled): under saturated adduction torques the thighs press on
    each other and stop at the compliant sink (F / kn: a few millimetres); with the filter on they pass through."""
    deep_on, dP_on, lf, sim, meta = _leg_squeeze(True)
    deep_off, _, lf_off, _, _ = _leg_squeeze(False)
    assert deep_off > 0.04, deep_off              # without self-collision the thighs interpenetrate by centimetres
    assert 0.0 < deep_on < 0.012, deep_on        # with it: the compliant sink only
    # the contact is an internal force pair: the link forces sum to zero and no momentum is created
    assert np.abs(lf).max() > 20.0 and np.abs(lf.sum(0)).max() < 1e-6 * np.abs(lf).max()
    M = meta["model"].total_mass()
    assert np.abs(dP_on).max() < 1e-3 * M      # < 1 mm/s of COM velocity over 0.8 s of motor-driven squeezing: the Euler step's own error
    assert np.abs(lf_off).max() == 0.0
    names = meta["model"].body_names
    loaded = {names[i] for i in np.nonzero(np.abs(lf).sum(1) > 1.0)[0]}
    assert {"left_thigh_pitch_link", "right_thigh_pitch_link"} <= loaded, loaded
    assert all(("thigh" in n or "shank" in n or "foot" in n) for n in loaded), loaded      # (the feet meet as well)


def test_self_collision_pairs_come_from_the_model_table():
    """Every listed sphere pair sits on two different, non-adjacent moving bodies (PhysX filters jointed links)."""
    for task in ("GR1T1", "GR1T2", "GR1T1Full"):
        cfg = make_cfg(task=task)
        c, _, meta = build_config.build(cfg, cfg.sim.dt, 1)
        m = c.model
        assert m.num_pairs > 0 and c.self_collisions == 1
        for k in range(m.num_pairs):
            ba, bb = m.sph_body[m.pair_a[k]], m.sph_body[m.pair_b[k]]
            assert ba != bb and m.parent[ba] != bb and m.parent[bb] != ba


def _drop(restitution, vdown=1.5, n=400, z=0.93):
    """Zero-torque drop of the robot onto the plane; per sub-step: sum of the foot links' normal forces, foot-sphere approach marker."""
    from oracle.binding import OracleSim
    cfg = make_cfg(dr=False)
    cfg.domain_rand.randomize_restitution = True
    cfg.domain_rand.restitution_range = [restitution, restitution]
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, 1)
    sim = OracleSim(c, "f64", keep)
    root = torch.zeros(1, 13); root[0, 2] = z; root[0, 6] = 1.0; root[0, 9] = -vdown
    q = torch.tensor([[0, 0, -0.2618, 0.5236, -0.2618] * 2], dtype=torch.float32)
    sim.set_state(root.contiguous(), q.contiguous(), torch.zeros(1, 10))
    feet = meta["feet_links"]
    fz = []
    for _ in range(n):
        sim.substeps(0, np.zeros(10), 1, contact=True)
        fz.append(sim.link_forces(0)[feet, 2].sum())
    return np.array(fz)


def test_restitution_raises_the_rebound_above_the_bounce_threshold():
    """Per-env restitution (legged_robot.py:565-575) with PhysX's bounce threshold (legged_robot_config.py:48): after a
    1.5 m/s landing the separating foot keeps only (1 - e) of its contact damping, so from the first separating sub-step
    on the ground pushes harder with e = 0.5 than with e = 0; a 0.2 m/s landing (below the 0.5 m/s threshold) is unaffected."""
    f0, f5 = _drop(0.0), _drop(0.5)
    assert f0.max() > 100.0
    diff = np.nonzero(f0 != f5)[0]
    assert diff.size > 0 and diff[0] > np.nonzero(f0 > 0)[0][0]      # identical through the compression phase
    assert f5[diff[0]] > f0[diff[0]], (diff[0], f0[diff[0]], f5[diff[0]])
    l0, l5 = _drop(0.0, vdown=0.2, n=60, z=0.90), _drop(0.5, vdown=0.2, n=60, z=0.90)      # touches at ~0.35 m/s
    assert l0.max() > 50.0 and np.array_equal(l0, l5)


# ------------------------------------------------------------------ terrain normal, vertical faces of 'trimesh' (round 2)
class _FakeTerrain:
    """A raster handed to build_config in place of utils.terrain.Terrain (only what build_config reads)."""

    def __init__(self, hs, rows=1, cols=1):
        self.heightsamples = np.ascontiguousarray(hs, dtype=np.int16)
        self.env_origins = np.zeros((rows, cols, 3), dtype=np.float32)


def _terrain_sim(hs, mesh_type, friction=None, N=1):
    from oracle.binding import OracleSim
    cfg = make_cfg(terrain=mesh_type)
    cfg.terrain.border_size = 0.0
    cfg.terrain.num_rows = cfg.terrain.num_cols = 1
    cfg.terrain.curriculum = False
    if friction is not None:
        cfg.terrain.static_friction = cfg.terrain.dynamic_friction = friction
        cfg.domain_rand.randomize_friction = True
        cfg.domain_rand.friction_range = [friction, friction]
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N, terrain=_FakeTerrain(hs))
    return OracleSim(c, "f64", keep), cfg, meta


def test_stair_riser_is_a_ramp_on_the_heightfield_and_a_wall_on_the_trimesh():
    """mesh_type 'trimesh' collides against the slope-corrected mesh (isaacgym terrain_utils.py:286-350,
    legged_robot.py:903-921): a raster step steeper than slope_treshold (0.75) is a vertical face at the HIGH vertex.
    On 'heightfield' the same step is the one-cell ramp PhysX's heightfield triangles make of it."""
    hs = np.zeros((40, 40), np.int16)
    hs[20:, :] = 40                      # a 0.2 m step (vertical_scale 0.005) between rows 19 and 20: x in [1.9, 2.0]
    hf, _, _ = _terrain_sim(hs, "heightfield")
    tm, _, _ = _terrain_sim(hs, "trimesh")
    for x, want_hf, want_tm in ((1.85, 0.0, 0.0), (1.92, 0.04, 0.0), (1.95, 0.10, 0.0), (1.974, 0.148, 0.0), (1.99, 0.18, 0.0), (1.9999, 0.1998, 0.0), (2.0001, 0.2, 0.2), (2.02, 0.2, 0.2)):
        np.testing.assert_allclose(hf.terrain(x, 1.0)[0], want_hf, atol=1e-6)
        np.testing.assert_allclose(tm.terrain(x, 1.0)[0], want_tm, atol=1e-6)
    # gradient: 2 on the heightfield ramp; the trimesh ground is level on both sides of the face ...
    assert abs(hf.terrain(1.95, 1.0)[1] - 2.0) < 1e-6 and tm.terrain(1.95, 1.0)[1] == 0.0 and tm.terrain(1.99, 1.0)[1] == 0.0
    # ... and the face itself is a contact of its own: a sphere of radius 4 cm, 1.5 cm in front of it, overlaps it by 2.5 cm along -x;
    # above the face's top (z = 0.2) the contact is with its upper edge; from the upper level there is no face
    np.testing.assert_allclose(tm.wall(1.985, 1.0, 0.05, 0.04), [0.025, -1, 0, 0], atol=1e-6)
    d = np.hypot(0.015, 0.02)
    np.testing.assert_allclose(tm.wall(1.985, 1.0, 0.22, 0.04), [0.04 - d, -0.015 / d, 0, 0.02 / d], atol=1e-6)
    assert tm.wall(1.95, 1.0, 0.05, 0.04)[0] == 0.0 and tm.wall(2.01, 1.0, 0.22, 0.04)[0] == 0.0 and hf.wall(1.985, 1.0, 0.05, 0.04)[0] == 0.0
    # a gentle slope (0.4) is below the threshold: identical in both modes
    ramp = (np.arange(40)[:, None] * 8 * np.ones((1, 40))).astype(np.int16)      # 8 units = 0.04 m per 0.1 m
    a, _, _ = _terrain_sim(ramp, "heightfield"); b, _, _ = _terrain_sim(ramp, "trimesh")
    np.testing.assert_allclose(a.terrain(1.234, 2.0), b.terrain(1.234, 2.0))
    np.testing.assert_allclose(a.terrain(1.234, 2.0), [0.4 * 1.234, 0.4, 0.0], atol=1e-6)


def test_a_foot_is_stopped_by_a_trimesh_riser():
    """Stair-edge foot contact: a robot sliding feet-first into a 0.2 m riser.  On the trimesh the feet meet a wall
    (horizontal contact force against the motion, the base stops short of the step); on the heightfield raster the same
    feet ride up the one-cell ramp."""
    hs = np.zeros((80, 40), np.int16)
    hs[30:, :] = 40                      # step face at x = 3.0 (trimesh) / ramp over x in [2.9, 3.0] (heightfield)
    out = {}
    for mesh in ("heightfield", "trimesh"):
        sim, cfg, meta = _terrain_sim(hs, mesh, friction=0.05)
        root = torch.zeros(1, 13); root[0, 0] = 2.6; root[0, 1] = 2.0; root[0, 2] = 0.90; root[0, 6] = 1.0; root[0, 7] = 1.5
        q = torch.tensor([[0, 0, -0.2618, 0.5236, -0.2618] * 2], dtype=torch.float32)
        sim.set_state(root.contiguous(), q.contiguous(), torch.zeros(1, 10))
        fx_min, x_feet = 0.0, []
        for i in range(25):
            sim.step(torch.zeros(1, 10), 0.0, i + 1)
            fx_min = min(fx_min, sim.tensor("FEET_CONTACT_FORCE")[0, :, 0].min().item())
            x_feet.append(sim.tensor("FEET_POS")[0, :, 0].max().item())
        out[mesh] = (fx_min, max(x_feet), sim.tensor("FEET_POS")[0, :, 2].max().item())
    toe = 0.15          # foremost sole sphere centre ahead of the foot link origin (URDF: x = 0.05 + 0.12 - 0.02)
    tm_toe, hf_toe = out["trimesh"][1] + toe, out["heightfield"][1] + toe
    # trimesh: the toe sphere (radius 3 cm) runs on the low ground until it touches the face at x = 3.0 and stops there
    assert 2.96 < tm_toe < 3.0 and out["trimesh"][0] < -20.0, out   # (the feet, already braked by the legs, lose their last 0.3 m/s against the face: -30 N at a policy step's end)
    # heightfield: the one-cell ramp starts at x = 2.9: the toe is caught there, 7-8 cm earlier, and lifted
    assert hf_toe < tm_toe - 0.05 and out["heightfield"][0] < -50.0 and out["heightfield"][2] > out["trimesh"][2] + 0.02, out


def test_frictionless_slope_slides_downhill():
    """Terrain normal from the gradient of the patch: on a frictionless 0.3 slope the robot accelerates downhill at about
    g sin(theta) cos(theta) horizontally (a vertical-only contact force would leave it standing)."""
    slope = 0.3
    hs = (np.arange(120)[:, None] * (slope * 0.1 / 0.005) * np.ones((1, 40))).astype(np.int16)
    sim, cfg, meta = _terrain_sim(hs, "heightfield", friction=0.0)
    x0 = 6.0
    root = torch.zeros(1, 13); root[0, 0] = x0; root[0, 1] = 2.0; root[0, 2] = slope * x0 + 0.90; root[0, 6] = 1.0
    q = torch.tensor([[0, 0, -0.2618, 0.5236, -0.2618] * 2], dtype=torch.float32)
    sim.set_state(root.contiguous(), q.contiguous(), torch.zeros(1, 10))
    for i in range(8):                    # 0.16 s (the level-torso stance tips over on the slope soon after)
        sim.step(torch.zeros(1, 10), 0.0, i + 1)
    e = sim.energy(0)
    vx = e["P"][0] / e["mass"]            # horizontal velocity of the centre of mass
    th = np.arctan(slope)
    want = -G9 * np.sin(th) * np.cos(th) * 0.16
    assert 1.15 * want < vx < 0.6 * want, (vx, want)     # downhill (-x), the right size


def test_single_pendulum_period_of_one_leg_link():
    """SURVEY 8c physics pin: one leg link swinging about its joint with everything else locked.  The free-floating model has no
    fixed joint, so the lock is inertial -- base and the other chain bodies 1e8 times heavier -- and the restoring torque is the
    joint's own PD spring (legged_robot.py:679-715 with kd = 0) in zero gravity: the foot about the ankle axis is then the
    textbook torsional pendulum, omega^2 = kp / I_axis with I_axis = I_yy(com) + m (c_x^2 + c_z^2).  Semi-implicit Euler at dt
    shifts the frequency to omega_d = (2 / dt) asin(omega dt / 2) exactly; the measured period (zero crossings over ~8
    periods, fp64) must equal 2 pi / omega_d to 1e-6 and the continuous-time period to first order in (omega dt)^2 / 24."""
    from oracle.binding import OracleSim
    cfg = make_cfg()
    cfg.control.decimation = 1
    cfg.sim.gravity = [0.0, 0.0, 0.0]
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 1)
    m = c.model
    ankle = 5                                              # body index of left_foot (leaf of the left chain); its dof is 4
    for b in range(m.num_bodies):
        if b != ankle:
            m.mass[b] *= 1e8
            for k in range(6):
                m.inertia[b][k] *= 1e8
    m.base_link_mass *= 1e8; m.base_rest_mass *= 1e8
    for k in range(6):
        m.base_link_inertia[k] *= 1e8; m.base_rest_inertia[k] *= 1e8
    kp = 4.0
    for j in range(10):
        c.kd[j] = 0.0
        c.default_dof_pos[j] = 0.0
        c.kp[j] = kp if j == ankle - 1 else 0.0
    I = m.inertia[ankle][3] + m.mass[ankle] * (m.com[ankle][0] ** 2 + m.com[ankle][2] ** 2)
    w = np.sqrt(kp / I)
    dt = c.sim_dt
    assert 5.0 < w < 60.0 and w * dt < 0.2
    sim = OracleSim(c, "f64", keep)
    sim.reset_all()
    root = torch.zeros(1, 13); root[0, 2] = 5.0; root[0, 6] = 1.0
    q = torch.zeros(1, 10); q[0, ankle - 1] = 0.1
    sim.set_state(root, q, torch.zeros(1, 10))
    a = torch.zeros(1, 10)
    n = int(8.5 * 2 * np.pi / w / dt)
    traj = np.empty(n)
    for i in range(n):
        sim.step(a, 0.0, i + 1)
        traj[i] = float(sim.tensor("DOF_POS")[0, ankle - 1])
    assert abs(traj).max() <= 0.1 * (1 + 1e-3) and abs(traj).max() > 0.0999          # undamped, bounded
    others = sim.tensor("DOF_POS")[0].numpy().copy(); others[ankle - 1] = 0
    assert np.abs(others).max() < 1e-6 and abs(float(sim.tensor("ROOT_STATES")[0, 2]) - 5.0) < 1e-6   # the rest IS locked
    up = [i + traj[i] / (traj[i] - traj[i + 1]) for i in range(n - 1) if traj[i] < 0 <= traj[i + 1]]   # upward zero crossings, interpolated
    assert len(up) >= 7
    T = (up[-1] - up[0]) / (len(up) - 1) * dt
    wd = 2.0 / dt * np.arcsin(w * dt / 2.0)
    # (the tensors are published in fp32: a crossing time is good to ~1e-6 s)
    assert abs(T - 2 * np.pi / wd) < 2e-6 * T, (T, 2 * np.pi / wd)
    assert abs(T - 2 * np.pi / w) < 1.5 * (w * dt) ** 2 / 24 * T + 2e-6 * T


def test_import_state_continues_bit_for_bit():
    """gro_debug_import_state (test support of the oracle: the inverse of its publish step for tests.helpers.STATE_TENSORS) -- a twin
    oracle that imports the state of another one reproduces that one's next policy step BIT FOR BIT, through contact, resets and
    the curriculum; nudged by 1e-6 it does not (what tests/test_hip_parity.py's sensitivity twins rely on)."""
    import tests.test_hip_parity as hp
    from tests.helpers import CMP_EXACT, CMP_TENSORS, make_cfg, make_sims, random_actions
    cfg = make_cfg(terrain="heightfield", dr=True, push=True, noise=True)
    cfg.env.episode_length_s = 0.5
    _, ora = make_sims(cfg, 48, hip=False)
    ora.reset_all()
    exact, nudged = hp.oracle_twin(ora), hp.oracle_twin(ora)
    gen, pg = torch.Generator().manual_seed(0), torch.Generator().manual_seed(1)
    moved = 0.0
    for s in range(30):
        keep, hp.PERT = hp.PERT, 0.0
        try:
            hp.perturbed_copy(ora, exact, pg)
        finally:
            hp.PERT = keep
        hp.perturbed_copy(ora, nudged, pg)
        a = random_actions(cfg, 48, gen, 0.5)
        for sim in (ora, exact, nudged):
            sim.step(a, 5.0, s + 1)
        for name in CMP_TENSORS + CMP_EXACT:
            assert torch.equal(exact.tensor(name), ora.tensor(name)), (s, name)
        moved = max(moved, float((nudged.tensor("DOF_VEL") - ora.tensor("DOF_VEL")).abs().max()))
    assert int(ora.tensor("EPISODE_LENGTH").min()) < 25 and moved > 1e-5      # resets happened; the nudge is felt
