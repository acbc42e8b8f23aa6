"""Pin the CPU oracle's env pipeline against golden vectors produced by the REFERENCE Python
(tools/gen_golden.py imported /root/reference in the build container; fixtures in tests/golden/).

Covers SURVEY 8c G-2..G-6: clip_actions/_compute_torques, post_physics_step state update, feet
timers, termination thresholds, the 24 active reward terms + totals + episode sums over two
consecutive steps (stale base_heights_offset quirk Q4, last_last_actions quirk), obs/pri_obs with
injected noise (x25 height scaling Q5, pre-noise copy Q6), and every implemented reward term."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import make_cfg
from wiki_grx_gym_amd import _capi
from wiki_grx_gym_amd.envs import build_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_oracle(N, precision, noise=True):
    from oracle.binding import OracleSim
    cfg = make_cfg(noise=noise, dr=False)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    return OracleSim(c, precision, keep), cfg, meta


def fill_state(ps, d, pre, i):
    from oracle.binding import PipelineState  # noqa: F401
    def put(dst, src):
        src = np.asarray(src, dtype=np.float64).reshape(-1)
        for k in range(src.size):
            dst[k] = float(src[k])
    put(ps.q, d[pre + "dof_pos"][i]); put(ps.qd, d[pre + "dof_vel"][i]); put(ps.root, d[pre + "root"][i])
    put(ps.actions, d[pre + "actions"][i]); put(ps.last_actions, d[pre + "last_actions"][i])
    put(ps.last_last_actions, d[pre + "last_last_actions"][i]); put(ps.last_dof_vel, d[pre + "last_dof_vel"][i])
    put(ps.torques, d[pre + "torques"][i]); put(ps.commands, d[pre + "commands"][i])
    put(ps.air_time, d[pre + "air_time"][i]); put(ps.land_time, d[pre + "land_time"][i])
    for f in range(2):
        ps.contact_last[f] = int(d[pre + "contact_last"][i][f])
        for k in range(3):
            ps.feet_force[f][k] = float(d[pre + "feet_force"][i][f][k])
            ps.feet_pos[f][k] = float(d[pre + "feet_pos"][i][f][k])
            ps.avg_speed[f][k] = float(d[pre + "avg_speed"][i][f][k])
        ps.avg_force[f] = float(d[pre + "avg_force"][i][f])
    put(ps.torso_R, quat_to_R(d[pre + "torso_quat"][i]))
    ps.base_heights_offset = float(d[pre + "base_heights_offset"][i])
    ps.episode_length = int(d[pre + "episode_length"][i])
    ps.term_contact = int(d[pre + "term_contact"][i])


def states_from(d, pre, N):
    """One PipelineState record per env of a golden fixture (keys `pre` + name)."""
    from oracle.binding import PipelineState
    arr = (PipelineState * N)()
    for i in range(N):
        fill_state(arr[i], d, pre, i)
    return arr


def inject(sim, arr, **kw):
    """Run post_physics_step on the injected records: per env through the oracle's gro_debug_post_physics, all at
    once through libgrx_hip.so's grx_debug_post_physics (same record type, include/grx.h)."""
    if hasattr(sim, "post_physics"):
        for i in range(len(arr)):
            sim.post_physics(i, arr[i], apply_reset=False, **kw)
    else:
        nz = kw.pop("noise_uniform", None)
        sim.debug_post_physics(arr, apply_reset=False, noise_uniform=None if nz is None else nz.to(sim.device), **kw)
        torch.cuda.synchronize()


def T_(sim, name):
    return sim.tensor(name).detach().cpu()


def check_pipeline_two_steps(sim, cfg, meta, tol):
    """reference post_physics_step x2 (tests/golden/pipeline.npz: legged_robot.py:269-481, legged_robot_fftai.py:90-167,
    gr1t1.py:281-589 on synthetic state incl. the threshold rows) vs `sim` (oracle or HIP)."""
    d = np.load(os.path.join(G, "pipeline.npz"))
    N = d["s0_in_root"].shape[0]
    names = list(d["reward_names"])
    assert names == meta["active_terms"], "active reward terms / their (alphabetical) order differ from the reference"
    term_idx = [_capi.REWARD_TERMS.index(n) for n in names]
    ever_resampled = np.zeros(N, bool)
    for s in range(2):
        pre, post = f"s{s}_in_", f"s{s}_out_"
        noise = torch.tensor(d[f"s{s}_noise_u"]).contiguous()
        inject(sim, states_from(d, pre, N), common_step_counter=s + 1, noise_uniform=noise)
        resampled = np.any(d[post + "commands_after"] != d[pre + "commands"], axis=1)
        keep = ~resampled
        ever_resampled |= resampled   # their reward history differs (RNG), so do the cumulative sums
        assert keep.sum() >= N - 4

        def close(name, got, want, rows=keep):
            got, want = np.asarray(got, dtype=np.float64)[rows], np.asarray(want, dtype=np.float64)[rows]
            err = np.abs(got - want)
            lim = tol + tol * np.abs(want)
            assert (err <= lim).all(), f"step {s} {name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        allrows = np.ones(N, bool)
        np.testing.assert_array_equal(T_(sim, "RESET").numpy().astype(bool), d[post + "reset"].astype(bool))
        np.testing.assert_array_equal(T_(sim, "TIME_OUT").numpy().astype(bool), d[post + "time_out"].astype(bool))
        np.testing.assert_array_equal(T_(sim, "FEET_CONTACT").numpy().astype(bool), d[post + "feet_contact"].astype(bool))
        np.testing.assert_array_equal(T_(sim, "EPISODE_LENGTH").numpy(), d[post + "episode_length_after"])
        close("base_lin_vel", T_(sim, "BASE_LIN_VEL"), d[post + "base_lin_vel"], allrows)
        close("base_ang_vel", T_(sim, "BASE_ANG_VEL"), d[post + "base_ang_vel"], allrows)
        close("projected_gravity", T_(sim, "PROJECTED_GRAVITY"), d[post + "projected_gravity"], allrows)
        close("air_time", T_(sim, "FEET_AIR_TIME"), d[post + "air_time_after"], allrows)
        close("land_time", T_(sim, "FEET_LAND_TIME"), d[post + "land_time_after"], allrows)
        close("feet_height", T_(sim, "FEET_HEIGHT"), d[post + "feet_height"], allrows)
        close("base_heights_offset", T_(sim, "BASE_HEIGHTS_OFFSET"), d[post + "base_heights_offset_after"], allrows)
        close("last_actions", T_(sim, "LAST_ACTIONS"), d[post + "last_actions_after"], allrows)
        close("rew", T_(sim, "REW"), d[post + "rew"])
        close("obs", T_(sim, "OBS"), d[post + "obs"])
        close("pri_obs", T_(sim, "PRI_OBS"), d[post + "pri_obs"])
        es = T_(sim, "EPISODE_SUMS").numpy()[term_idx]          # (24, N)
        close("episode_sums", es.T, d[post + "episode_sums"].T, ~ever_resampled)
        # quirk: after a step last_last_actions == last_actions == actions (fftai:94 after legged_robot.py:299)
        np.testing.assert_array_equal(d[post + "last_last_actions_after"], d[post + "last_actions_after"])
    # the fixture's edge rows really are edge rows (F_z = 1.0 exactly, |g_z| either side of 0.33, ep_len 1000 / 1001,
    # last_last_actions != last_actions): they are what the threshold comparisons above were fed
    assert d["s0_in_feet_force"][3, 0, 2] == 1.0 and not d["s0_out_feet_contact"][3, 0] and d["s0_out_feet_contact"][4, 1]
    assert d["s0_out_reset"][8] and not d["s0_out_time_out"][8] and d["s0_out_time_out"][7] and not d["s0_out_time_out"][6]
    assert np.abs(d["s0_in_last_last_actions"] - d["s0_in_last_actions"]).max() > 0.1
    sc = np.array([cfg_scale for cfg_scale in d["reward_scales_dt"]])
    mine = np.array([getattr(cfg.rewards.scales, n) * meta["dt"] for n in names])
    np.testing.assert_allclose(mine, sc, rtol=1e-6)


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_pipeline_two_steps(precision, tol):
    sim, cfg, meta = make_oracle(np.load(os.path.join(G, "pipeline.npz"))["s0_in_root"].shape[0], precision)
    check_pipeline_two_steps(sim, cfg, meta, tol)


def check_every_reward_term(sim, cfg, tol):
    """The 34 FF/G1 reward terms the reference can evaluate (legged_robot_fftai.py:180-352, gr1t1.py:338-589; active or
    not -- limits_actions has no sigma in the reference) on one synthetic state: the ACTIVE ones, scaled."""
    d = np.load(os.path.join(G, "reward_terms.npz"))
    N = d["in_root"].shape[0]
    names = list(d["names"])
    assert len(names) == 34 and {"limits_dof_tor", "limits_dof_vel", "on_the_air", "pose_offset"} <= set(names)
    inject(sim, states_from(d, "in_", N))
    got_active = T_(sim, "REWARD_TERMS").numpy()
    dt = cfg.control.decimation * cfg.sim.dt
    # the fixture evaluates the terms on the injected commands; a full post_physics_step (the HIP kernel) first redraws
    # the commands of rows whose episode length hits the resampling interval (legged_robot.py:315-317)
    rows = (d["in_episode_length"] + 1) % int(cfg.commands.resampling_command_interval_s / dt) != 0
    assert rows.sum() >= N - 2
    checked = 0
    for n in names:
        t = _capi.REWARD_TERMS.index(n)
        scale = getattr(cfg.rewards.scales, n, 0.0)
        if scale == 0:
            continue
        want = d["values"][names.index(n)][rows] * scale * dt
        err = np.abs(got_active[t][rows] - want)
        assert (err <= tol + tol * np.abs(want)).all(), f"{n}: max err {err.max():.3e}"
        checked += 1
    assert checked == 24   # every active term of the registered GR1T1 task is pinned individually


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_every_reward_term(precision, tol):
    sim, cfg, _ = make_oracle(64, precision, noise=False)
    check_every_reward_term(sim, cfg, tol)


def inactive_terms_cfg():
    d = np.load(os.path.join(G, "reward_terms.npz"))
    names = list(d["names"])
    cfg = make_cfg(noise=False, dr=False)
    inactive = [n for n in names if getattr(cfg.rewards.scales, n, 0.0) == 0 and n != "termination"]
    assert "dof_vel_new" in inactive and "action_diff_knee" in inactive
    for n in inactive:
        setattr(cfg.rewards.scales, n, 1.0)
    cfg.rewards.scales.termination = 1.0
    return cfg, inactive


def check_inactive_reward_terms(sim, cfg, inactive, tol):
    d = np.load(os.path.join(G, "reward_terms.npz"))
    N = d["in_root"].shape[0]
    names = list(d["names"])
    inject(sim, states_from(d, "in_", N))
    got = T_(sim, "REWARD_TERMS").numpy()
    dt = cfg.control.decimation * cfg.sim.dt
    rows = (d["in_episode_length"] + 1) % int(cfg.commands.resampling_command_interval_s / dt) != 0
    for n in inactive + ["termination"]:
        want = d["values"][names.index(n)][rows] * 1.0 * dt
        err = np.abs(got[_capi.REWARD_TERMS.index(n)][rows] - want)
        assert (err <= tol + tol * np.abs(want)).all(), f"{n}: max err {err.max():.3e}"


def test_inactive_reward_terms_formulas():
    """Terms with zero scale in the registered config: enabled (scale 1) in the oracle."""
    from oracle.binding import OracleSim
    cfg, inactive = inactive_terms_cfg()
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 64)
    check_inactive_reward_terms(OracleSim(c, "f64", keep), cfg, inactive, 2e-6)


@pytest.mark.parametrize("precision,tol", [("f64", 1e-6), ("f32", 1e-4)])
def test_clip_actions_and_torques(precision, tol):
    from oracle.binding import OracleSim
    d = np.load(os.path.join(G, "torques.npz"))
    N = d["actions"].shape[0]
    cfg = make_cfg(noise=False, dr=False)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    np.testing.assert_allclose(np.array(c.kp[:10]), d["p_gains"], rtol=1e-6)
    np.testing.assert_allclose(np.array(c.kd[:10]), d["d_gains"], rtol=1e-6)
    np.testing.assert_allclose(np.array(c.default_dof_pos[:10]), d["default_dof_pos"].reshape(-1), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.array(c.clip_actions_min[:10]), d["clip_min"], rtol=1e-6)
    np.testing.assert_allclose(np.array(c.clip_actions_max[:10]), d["clip_max"], rtol=1e-6)
    np.testing.assert_allclose(np.array(c.model.dof_effort[:10]), d["torque_limits"], rtol=1e-6)
    np.testing.assert_allclose(np.array(c.model.dof_vel_limit[:10]), d["dof_vel_limits"], rtol=1e-6)
    np.testing.assert_allclose(build_config.soft_dof_pos_limits(meta["model"], cfg.rewards.soft_dof_pos_limit), d["dof_pos_limits"], rtol=1e-5, atol=1e-6)
    sim = OracleSim(c, precision, keep)
    sim.set_state(None, torch.tensor(d["dof_pos"]).contiguous(), torch.tensor(d["dof_vel"]).contiguous())
    # motor strength is a creation-time constant of the oracle (DR off -> 1): fold it into the expectation
    clipped, tq = sim.torques(d["actions"])
    np.testing.assert_allclose(clipped, d["clipped"], rtol=tol, atol=tol)
    base = d["p_gains"] * (d["clipped"] * 1.0 + d["default_dof_pos"] - d["dof_pos"]) - d["d_gains"] * d["dof_vel"]
    want = np.clip(base, -d["torque_limits"], d["torque_limits"])
    np.testing.assert_allclose(tq, want, rtol=tol, atol=tol * 10)
    # and the reference's own torques are consistent with that formula times its strength factors
    ref = np.clip(base * d["strength"], -d["torque_limits"], d["torque_limits"])
    np.testing.assert_allclose(d["torques"], ref, rtol=1e-5, atol=1e-4)


def test_noise_vector_matches_reference():
    d = np.load(os.path.join(G, "torques.npz"))
    cfg = make_cfg(noise=True)
    n, s = cfg.noise.noise_scales, cfg.normalization.obs_scales
    lv = cfg.noise.noise_level
    vec = np.concatenate([np.zeros(3), np.full(3, n.ang_vel * lv * s.ang_vel), np.full(3, n.gravity * lv * s.gravity),
                          np.full(10, n.dof_pos * lv * s.dof_pos), np.full(10, n.dof_vel * lv * s.dof_vel),
                          np.full(10, n.action * lv * s.action)])
    np.testing.assert_allclose(vec, d["noise_vec"], rtol=1e-6)


def quat_states(N):
    """PipelineState records carrying the G-1 fixture (tests/golden/quat.npz: isaacgym torch_utils.py:48-81, legged_gym math.py:38-55
    evaluated by the reference): root quaternion q_i, linear AND angular velocity v_i; everything else at rest."""
    from oracle.binding import PipelineState
    d = np.load(os.path.join(G, "quat.npz"))
    arr = (PipelineState * N)()
    for i in range(N):
        ps = arr[i]
        ps.root[2] = 1.0
        for k in range(4):
            ps.root[3 + k] = float(d["q"][i][k])
        for k in range(3):
            ps.root[7 + k] = float(d["v"][i][k]); ps.root[10 + k] = float(d["v"][i][k])
            for f in range(2):
                ps.feet_pos[f][k] = 0.0
        for k in range(9):
            ps.torso_R[k] = float(k % 4 == 0)
    return arr, d


def check_quat_rotate_inverse(sim, tol):
    """G-1 on the env step itself: base_lin_vel / base_ang_vel = quat_rotate_inverse(base_quat, v) (legged_robot.py:284-286) and
    projected_gravity = quat_rotate_inverse(base_quat, (0, 0, -1)) = -(third row of R(q)), 256 random orientations."""
    arr, d = quat_states(256)
    inject(sim, arr, common_step_counter=1)
    want = d["rotate_inverse"].astype(np.float64)
    for name in ("BASE_LIN_VEL", "BASE_ANG_VEL"):
        got = T_(sim, name).double().numpy()
        assert (np.abs(got - want) <= tol + tol * np.abs(want)).all(), name
    # quat_rotate(q, v) is the inverse map: R(q) R(q)^T v = v ties `rotate` and `rotate_inverse` of the fixture together
    for i in range(0, 256, 17):
        R = quat_to_R(d["q"][i].astype(np.float64))
        np.testing.assert_allclose(R @ d["rotate_inverse"][i], d["v"][i], atol=2e-5)
        np.testing.assert_allclose(R @ d["v"][i], d["rotate"][i], atol=2e-5)
        np.testing.assert_allclose(R @ d["v"][i], d["apply"][i], atol=2e-5)
    g = T_(sim, "PROJECTED_GRAVITY").double().numpy()
    wantg = np.stack([-quat_to_R(d["q"][i].astype(np.float64))[2] for i in range(256)])
    assert (np.abs(g - wantg) <= 10 * tol).all()


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_quat_fixture_on_the_oracle(precision, tol):
    sim, _, _ = make_oracle(256, precision, noise=False)
    check_quat_rotate_inverse(sim, tol)


def test_quat_fixture_reset_yaw_and_wrap():
    """The remaining G-1 columns the path uses: the reset orientation quat_from_euler_xyz(0, 0, yaw) (legged_robot.py:765-767) is the
    oracle's / kernels' closed form (0, 0, sin(yaw/2), cos(yaw/2)); quat_apply_yaw keeps z and rotates xy by the quaternion's yaw.
    wrap_to_pi (heading commands, math.py:45-48) is NOT on the path: heading_command is False for the GRx tasks and build_config
    rejects True."""
    d = np.load(os.path.join(G, "quat.npz"))
    e = d["euler"].astype(np.float64)
    # general Euler -> quaternion (torch_utils.py:176-190) against the fixture, then the yaw-only special case used at reset
    cr, sr, cp, sp, cy, sy = np.cos(e[:, 0] / 2), np.sin(e[:, 0] / 2), np.cos(e[:, 1] / 2), np.sin(e[:, 1] / 2), np.cos(e[:, 2] / 2), np.sin(e[:, 2] / 2)
    q = np.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp, cy * cr * cp + sy * sr * sp], 1)
    np.testing.assert_allclose(q, d["from_euler"], atol=2e-6)
    yaw_only = np.stack([0 * sy, 0 * sy, sy, cy], 1)
    np.testing.assert_allclose(yaw_only[np.abs(sr) + np.abs(sp) < 1e-9], q[np.abs(sr) + np.abs(sp) < 1e-9], atol=1e-12)
    # quat_apply_yaw: z untouched, xy rotated by atan2-free yaw of the normalised (0, 0, qz, qw)
    qz, qw = d["q"][:, 2].astype(np.float64), d["q"][:, 3].astype(np.float64)
    n = np.sqrt(qz * qz + qw * qw)
    c, s_ = (qw * qw - qz * qz) / (n * n), 2 * qz * qw / (n * n)
    v = d["v"].astype(np.float64)
    np.testing.assert_allclose(np.stack([c * v[:, 0] - s_ * v[:, 1], s_ * v[:, 0] + c * v[:, 1], v[:, 2]], 1), d["apply_yaw"], atol=2e-5)
    np.testing.assert_allclose(np.mod(d["ang"].astype(np.float64) + np.pi, 2 * np.pi) - np.pi, d["wrap_to_pi"], atol=2e-5)


# ---- the reference options the GRx tasks leave off (tests/golden/control_modes.npz, tools/gen_golden.py gen_control_modes) ---------
def control_torque_formula(d, ct, scale=1.0):
    """legged_robot.py:693-707 on the fixture's inputs (before the motor-strength ratio and the clip)."""
    from wiki_grx_gym_amd.envs.config import GR1T1LowerLimbCfg  # noqa: F401
    t = np.load(os.path.join(G, "torques.npz"))
    kp, kd, q0 = t["p_gains"].astype(np.float64), t["d_gains"].astype(np.float64), t["default_dof_pos"].astype(np.float64)
    a, q, qd, qdl = (d[k].astype(np.float64) for k in ("clipped", "dof_pos", "dof_vel", "last_dof_vel"))
    if ct == "P":
        return kp * (a * scale + q0 - q) - kd * qd
    if ct == "V":
        return kp * (a * scale - qd) - kd * (qd - qdl) / 0.002
    return a * scale


def check_heading_command(sim, tol):
    """legged_robot.py:320-326 through one injected post_physics_step: commands[:, 2] of EVERY env is the clipped half heading error
    against commands_heading = 0; the yaw command is not drawn at the time-based resample (legged_robot.py:668-676)."""
    d = np.load(os.path.join(G, "control_modes.npz"))
    N = d["h_in_root"].shape[0]
    inject(sim, states_from(d, "h_in_", N), common_step_counter=1, noise_uniform=torch.tensor(d["h_noise_u"]).contiguous())
    got = T_(sim, "COMMANDS").numpy()
    want = d["h_out_commands"]
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=tol, atol=tol)
    np.testing.assert_allclose(T_(sim, "OBS").numpy()[:, 2], d["h_out_obs"][:, 2], rtol=tol, atol=tol)
    lo, hi = d["h_yaw_range"]
    assert (want[:, 2] == lo).any() and (want[:, 2] == hi).any() and ((want[:, 2] > lo) & (want[:, 2] < hi)).sum() > N // 4   # clipped and free rows
    assert np.abs(want[:, 2] - d["h_in_commands"][:, 2]).max() > 0.1                                                          # the rule did overwrite the command


@pytest.mark.parametrize("precision,tol", [("f64", 1e-6), ("f32", 1e-4)])
@pytest.mark.parametrize("ct", ["P", "V", "T"])
def test_control_types_match_the_reference(ct, precision, tol):
    """cfg.control.control_type 'P' / 'V' / 'T' (legged_robot.py:693-707) in the oracle against the reference's own _compute_torques."""
    from oracle.binding import OracleSim, PipelineState
    d = np.load(os.path.join(G, "control_modes.npz"))
    N = d["actions"].shape[0]
    cfg = make_cfg(noise=False, dr=False)
    cfg.control.control_type = ct
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    sim = OracleSim(c, precision, keep)
    # last_dof_vel: post_physics_step leaves last_dof_vel = dof_vel (legged_robot.py:300) -- inject records whose dof_vel is the fixture's last_dof_vel
    arr = (PipelineState * N)()
    for i in range(N):
        arr[i].root[2] = 1.0; arr[i].root[6] = 1.0
        for k in range(9):
            arr[i].torso_R[k] = 1.0 if k % 4 == 0 else 0.0
        for j in range(10):
            arr[i].qd[j] = float(d["last_dof_vel"][i][j])
    inject(sim, arr)
    sim.set_state(None, torch.tensor(d["dof_pos"]).contiguous(), torch.tensor(d["dof_vel"]).contiguous())
    clipped, tq = sim.torques(d["actions"])
    np.testing.assert_allclose(clipped, d["clipped"], rtol=tol, atol=tol)
    base = control_torque_formula(d, ct, cfg.control.action_scale)
    lim = d["torque_limits"]
    # (the velocity law divides a difference of two velocities by sim_dt: fp32 keeps ~1e-7 x 20 x 500 of it)
    np.testing.assert_allclose(tq, np.clip(base, -lim, lim), rtol=tol, atol=tol * (100 if ct == "V" else 10))
    # ... and the reference's own torques are that formula times its strength factors
    np.testing.assert_allclose(d["torques_" + ct], np.clip(base * d["strength"], -lim, lim), rtol=1e-5, atol=2e-3 if ct == "V" else 1e-4)
    sat = np.abs(d["torques_" + ct]) >= lim - 1e-6
    assert ct == "T" or (sat.any() and (~sat).any())


def test_unknown_control_type_raises_like_the_reference():
    cfg = make_cfg(noise=False, dr=False)
    cfg.control.control_type = "X"
    with pytest.raises(NameError):   # legged_robot.py:707
        build_config.build(cfg, cfg.sim.dt, 4)


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_heading_command_matches_the_reference(precision, tol):
    from oracle.binding import OracleSim
    cfg = make_cfg(noise=True, dr=False)
    cfg.commands.heading_command = True
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, 64)
    check_heading_command(OracleSim(c, precision, keep), tol)


# ---- the reference's rough-terrain raster: _get_heights and one post_physics_step ON it (tests/golden/terrain.npz, pipeline_rough.npz) ----
def reference_raster_terrain():
    """The reference's own 10 x 20 curriculum raster (seed 1) and tile origins as build_config.build's `terrain` argument."""
    import types
    d = np.load(os.path.join(G, "terrain.npz"))
    return types.SimpleNamespace(heightsamples=d["heightsamples"].copy(), env_origins=d["env_origins"].astype(np.float32).copy()), d


def rough_cfg(noise=True):
    return make_cfg(noise=noise, dr=False, terrain="heightfield")


def get_heights_states(d):
    """terrain.npz's 64 root poses (incl. negative coordinates and poses beyond the map: the clamps of legged_robot.py:1262-1265)."""
    from oracle.binding import PipelineState
    N = d["root"].shape[0]
    arr = (PipelineState * N)()
    for i in range(N):
        for k in range(13):
            arr[i].root[k] = float(d["root"][i][k])
        arr[i].torso_R[0] = arr[i].torso_R[4] = arr[i].torso_R[8] = 1.0
    return arr


def check_get_heights(sim, d, max_frac):
    """MEASURED_HEIGHTS of `sim` (oracle or HIP) on terrain.npz's poses against the reference's _get_heights (legged_robot.py:1235-1274).
    Heights are quantised (min of three raster corners at a truncated cell index): a scan point within fp32 rounding of a cell edge may
    read the neighbouring cell -- every differing point must BE such a point (fp64 distance to the edge < 1e-3 cells), and there may be
    at most max_frac of them."""
    inject(sim, get_heights_states(d))
    got = T_(sim, "MEASURED_HEIGHTS").numpy()
    bad = np.abs(got - d["heights"]) > 1e-6
    cfg = rough_cfg()
    hp = np.asarray(sim_height_points(cfg), dtype=np.float64)                      # (121, 2) base-frame points
    q = d["root"][:, 3:7].astype(np.float64)
    n2 = q[:, 2] ** 2 + q[:, 3] ** 2                                               # quat_apply_yaw: yaw of the normalised (0, 0, z, w)
    c, s = (q[:, 3] ** 2 - q[:, 2] ** 2) / n2, 2 * q[:, 2] * q[:, 3] / n2
    x = c[:, None] * hp[None, :, 0] - s[:, None] * hp[None, :, 1] + d["root"][:, 0:1].astype(np.float64)
    y = s[:, None] * hp[None, :, 0] + c[:, None] * hp[None, :, 1] + d["root"][:, 1:2].astype(np.float64)
    f = np.stack([x, y], -1) + cfg.terrain.border_size
    f = f / cfg.terrain.horizontal_scale
    edge = np.abs(f - np.rint(f)).min(-1) < 1e-3
    assert not (bad & ~edge).any(), f"{(bad & ~edge).sum()} height samples differ away from any cell edge"
    assert bad.mean() <= max_frac, f"{bad.sum()} of {bad.size} height samples differ"
    # the fixture's clamp rows are clamp rows: env 0 sits below index 0 on both axes, env 1 beyond dim - 2
    assert d["root"][0, 0] < -25 and d["root"][1, 0] > 105 and (d["root"][:, 0] < 0).sum() >= 2
    return int(bad.sum())


def sim_height_points(cfg):
    """measured_points_x x measured_points_y in meshgrid order (legged_robot.py:1219-1233)."""
    xs, ys = cfg.terrain.measured_points_x, cfg.terrain.measured_points_y
    return [(x, y) for x in xs for y in ys]


def test_get_heights_fixture_on_the_oracle():
    from oracle.binding import OracleSim
    ter, d = reference_raster_terrain()
    cfg = rough_cfg()
    N = d["root"].shape[0]
    for prec, frac in (("f64", 0.0), ("f32", 2e-3)):
        c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
        check_get_heights(OracleSim(c, prec, keep), d, frac)


def check_pipeline_rough(sim, cfg, meta, tol):
    """pipeline_rough.npz: one reference post_physics_step() on the rough raster, heights from the reference's own _get_heights
    inside the step: the 121-entry height block of pri_obs (gr1t1.py:281-313: clip(z - target - h_k) x 5 x 5,
    Q5), feet_height = mean_k(z_foot - h_k) (legged_robot_fftai.py:118-124), base_heights_offset, rewards that read them."""
    d = np.load(os.path.join(G, "pipeline_rough.npz"))
    N = d["in_root"].shape[0]
    names = list(d["reward_names"])
    assert names == meta["active_terms"]
    inject(sim, states_from(d, "in_", N), common_step_counter=1, noise_uniform=torch.tensor(d["noise_u"]).contiguous())
    keep = ~np.any(d["out_commands_after"] != d["in_commands"], axis=1)     # rows whose commands the step redrew (RNG)
    assert keep.sum() >= N - 3

    def close(name, got, want, rows=np.ones(N, bool), t=tol):
        got, want = np.asarray(got, dtype=np.float64)[rows], np.asarray(want, dtype=np.float64)[rows]
        err = np.abs(got - want)
        assert (err <= t + t * np.abs(want)).all(), f"{name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    mh = T_(sim, "MEASURED_HEIGHTS").numpy()
    np.testing.assert_allclose(mh, d["out_measured_heights"], atol=1e-6, rtol=0)        # every scan point reads the reference's cell
    assert d["out_measured_heights"].std(axis=1).max() > 0.05                             # ... on terrain that is not flat
    close("feet_height", T_(sim, "FEET_HEIGHT"), d["out_feet_height"])
    close("base_heights_offset", T_(sim, "BASE_HEIGHTS_OFFSET"), d["out_base_heights_offset_after"])
    pri = T_(sim, "PRI_OBS").numpy()
    close("pri_obs height block", pri[:, 47:168], d["out_pri_obs"][:, 47:168])
    hb = d["out_pri_obs"][:, 47:168]
    assert (np.abs(hb) == 25.0).any() and (np.abs(hb) < 24.0).any()                      # clipped (x25: Q5) and free entries both occur
    close("pri_obs", pri, d["out_pri_obs"], keep)
    close("obs", T_(sim, "OBS"), d["out_obs"], keep)
    close("rew", T_(sim, "REW"), d["out_rew"], keep)
    np.testing.assert_array_equal(T_(sim, "RESET").numpy().astype(bool), d["out_reset"].astype(bool))
    np.testing.assert_array_equal(T_(sim, "FEET_CONTACT").numpy().astype(bool), d["out_feet_contact"].astype(bool))
    term_idx = [_capi.REWARD_TERMS.index(n) for n in names]
    close("episode_sums", T_(sim, "EPISODE_SUMS").numpy()[term_idx].T, d["out_episode_sums"].T, keep)


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_pipeline_on_the_rough_raster(precision, tol):
    from oracle.binding import OracleSim
    ter, _ = reference_raster_terrain()
    cfg = rough_cfg()
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, 64, terrain=ter)
    check_pipeline_rough(OracleSim(c, precision, keep), cfg, meta, tol)


# ---- the robots the round 1-5 fixtures did not cover: GR1T2 (20-body termination set) and the 32-DOF full body (VERDICT r5 #4b) ----
OTHER_ROBOTS = {"gr1t2": ("pipeline_gr1t2.npz", "GR1T2", 10), "full_body": ("pipeline_full_body.npz", "GR1T1Full", 32)}


def check_pipeline_other_robot(sim, cfg, meta, tol, which):
    """tests/golden/pipeline_gr1t2.npz / pipeline_full_body.npz (tools/gen_golden.py gen_pipeline_other_robots): one post_physics_step() of
    the reference's GR1T2 class with GR1T2LowerLimbCfg, and of its GR1T1 class on the 32-DOF full body with the build's task values
    (gr1t1.py:18-113 index sets, 281-336 observation columns: 105 / 234; legged_robot.py:336-353 termination) against `sim` (oracle or HIP)."""
    fixture, _, nd = OTHER_ROBOTS[which]
    d = np.load(os.path.join(G, fixture))
    N = d["in_root"].shape[0]
    assert d["in_dof_pos"].shape == (N, nd) and d["out_obs"].shape == (N, 9 + 3 * nd) and d["out_pri_obs"].shape == (N, 9 + 3 * nd + 129)
    names = list(d["reward_names"])
    assert names == meta["active_terms"], "active reward terms / their (alphabetical) order differ from the reference"
    term_idx = [_capi.REWARD_TERMS.index(n) for n in names]
    inject(sim, states_from(d, "in_", N), common_step_counter=1, noise_uniform=torch.tensor(d["noise_u"]).contiguous())
    keep = ~np.any(d["out_commands_after"] != d["in_commands"], axis=1)   # rows whose commands the reference redrew (its RNG)
    assert keep.sum() >= N - 4

    def close(name, got, want, rows=keep):
        got, want = np.asarray(got, dtype=np.float64)[rows], np.asarray(want, dtype=np.float64)[rows]
        err = np.abs(got - want)
        assert (err <= tol + tol * np.abs(want)).all(), f"{which} {name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    allrows = np.ones(N, bool)
    np.testing.assert_array_equal(T_(sim, "RESET").numpy().astype(bool), d["out_reset"].astype(bool))
    np.testing.assert_array_equal(T_(sim, "TIME_OUT").numpy().astype(bool), d["out_time_out"].astype(bool))
    np.testing.assert_array_equal(T_(sim, "FEET_CONTACT").numpy().astype(bool), d["out_feet_contact"].astype(bool))
    np.testing.assert_array_equal(T_(sim, "EPISODE_LENGTH").numpy(), d["out_episode_length_after"])
    close("base_lin_vel", T_(sim, "BASE_LIN_VEL"), d["out_base_lin_vel"], allrows)
    close("base_ang_vel", T_(sim, "BASE_ANG_VEL"), d["out_base_ang_vel"], allrows)
    close("projected_gravity", T_(sim, "PROJECTED_GRAVITY"), d["out_projected_gravity"], allrows)
    close("air_time", T_(sim, "FEET_AIR_TIME"), d["out_air_time_after"], allrows)
    close("land_time", T_(sim, "FEET_LAND_TIME"), d["out_land_time_after"], allrows)
    close("feet_height", T_(sim, "FEET_HEIGHT"), d["out_feet_height"], allrows)
    close("last_actions", T_(sim, "LAST_ACTIONS"), d["out_last_actions_after"], allrows)
    dt = cfg.control.decimation * cfg.sim.dt
    terms = T_(sim, "REWARD_TERMS").numpy()[term_idx]                      # every active term on its own, scaled
    want_terms = d["out_term_values"] * np.array([getattr(cfg.rewards.scales, n) * dt for n in names])[:, None]
    for k, n in enumerate(names):
        if n != "termination":
            close("term " + n, terms[k], want_terms[k])
    close("rew", T_(sim, "REW"), d["out_rew"])
    close("obs", T_(sim, "OBS"), d["out_obs"])
    close("pri_obs", T_(sim, "PRI_OBS"), d["out_pri_obs"])
    close("episode_sums", T_(sim, "EPISODE_SUMS").numpy()[term_idx].T, d["out_episode_sums"].T)
    np.testing.assert_allclose(np.array([getattr(cfg.rewards.scales, n) * meta["dt"] for n in names]), d["reward_scales_dt"], rtol=1e-6)
    # the rows the fixture was built for: eight different terminating bodies reset, a body outside the set does not; GR1T2: imu_link does
    assert d["out_reset"][16:24].all() and not d["out_reset"][24] and d["in_term_contact"][16:24].all() and not d["in_term_contact"][24]
    if which == "gr1t2":
        assert "imu_link" in list(d["termination_bodies"]) and len(d["termination_bodies"]) == 20 and d["out_reset"][25] and d["in_term_contact"][25]


def make_other_robot(which, N, precision=None, noise=True):
    cfg = make_cfg(OTHER_ROBOTS[which][1], noise=noise, dr=False)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    if precision is None:
        return cfg, c, keep, meta
    from oracle.binding import OracleSim
    return OracleSim(c, precision, keep), cfg, meta


@pytest.mark.parametrize("which", sorted(OTHER_ROBOTS))
@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_pipeline_of_the_other_robots(which, precision, tol):
    sim, cfg, meta = make_other_robot(which, 64, precision)
    check_pipeline_other_robot(sim, cfg, meta, tol, which)
