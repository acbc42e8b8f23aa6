"""The reference's golden fixtures presented DIRECTLY to the HIP kernels (`-m gpu`).

tests/golden/pipeline.npz and reward_terms.npz hold inputs and outputs of the reference's own post_physics_step
(legged_robot.py:269-481, legged_robot_fftai.py:90-167, gr1t1.py:281-589; tools/gen_golden.py imports the reference in
the build container).  libgrx_hip.so's test-only entry grx_debug_post_physics (include/grx.h) runs the step kernel's
post-physics half on those injected states, so the edge rows the fixture was built for -- F_z = 1.0 exactly,
|g_z| = 0.33 +- eps, episode_length 1000 / 1001, last_last_actions != last_actions, the stale base_heights_offset --
hit the HIP code itself, at the north star's 1e-4.  The same checkers run against the CPU oracle in
tests/test_oracle_golden.py; the `oracle pin` tests below re-run those on the GPU box so that one GPUTEST record shows
golden -> oracle and golden -> HIP side by side."""
import numpy as np
import pytest
import torch

from tests import test_oracle_golden as og
from tests.helpers import make_cfg
from wiki_grx_gym_amd.envs import build_config

pytestmark = pytest.mark.gpu

# The fixtures go through the post-physics half of EVERY step kernel the library launches (grx_debug_post_physics runs the DBG
# instantiation of the handle's own layout): the one-wave kernel of large batches, and the pipelines BASELINE.json's configs 2-4
# run -- lane pairs with four / eight waves (8192-16384 envs per GPU), lane quads with four / eight waves (<= 4096 envs per GPU) --
# where the reward inputs cross LDS to two reward waves, reset_idx's draws come from the foot wave, the termination flag from the
# base-lump wave and the height block from the helper waves.
# "tree": the lower-limb model forced through the tree kernel (GRX_FORCE_GENERIC) -- the post-physics code BASELINE.json's config 5 runs
# (csrc/grx_tree.h; the reference registers no full-body task, so its fixtures reach that code through the 10-dof robot).
LAYOUTS = pytest.mark.parametrize("layout", [1, 4, 8, "quad4", "quad", "tree", "tree16"])
KERNEL_OF = {1: ("grx_step_kernel<", 2, 1), 4: ("grx_step_kernel<", 2, 4), 8: ("grx_step_kernel<", 2, 8),
             "quad4": ("grx_step_kernel_quad<", 4, 4), "quad": ("grx_step_kernel_quad<", 4, 8), "tree": ("grx_step_tree<", 8, 2), "tree16": ("grx_step_tree16<", 16, 2)}


def pick_layout(monkeypatch, layout):
    from tests.test_hip_parity import set_layout
    if layout in ("tree", "tree16"):
        monkeypatch.setenv("GRX_FORCE_GENERIC", "1")
        monkeypatch.setenv("GRX_TREE", "1")
        monkeypatch.setenv("GRX_TREE_G", "16" if layout == "tree16" else "8")
    else:
        monkeypatch.delenv("GRX_FORCE_GENERIC", raising=False)
        set_layout(monkeypatch, layout)


def make_hip(cfg, N=64, layout=None, monkeypatch=None):
    from wiki_grx_gym_amd.sim import HipSim
    if layout is not None:
        pick_layout(monkeypatch, layout)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N)
    sim = HipSim(c, "cuda:0", keep)
    if layout is not None:   # the handle really runs (and debug-injects into) the kernel this case names
        lay = sim.layout()
        name, lpe, waves = KERNEL_OF[layout]
        assert lay["kernel"].startswith(name) and lay["lanes_per_env"] == lpe and lay["waves_per_block"] == waves, lay
    return sim, meta


@LAYOUTS
def test_pipeline_two_steps_on_the_hip_kernel(layout, monkeypatch):
    cfg = make_cfg(noise=True, dr=False)
    sim, meta = make_hip(cfg, layout=layout, monkeypatch=monkeypatch)
    og.check_pipeline_two_steps(sim, cfg, meta, 1e-4)


@LAYOUTS
def test_every_active_reward_term_on_the_hip_kernel(layout, monkeypatch):
    cfg = make_cfg(noise=False, dr=False)
    sim, _ = make_hip(cfg, layout=layout, monkeypatch=monkeypatch)
    og.check_every_reward_term(sim, cfg, 1e-4)


@LAYOUTS
def test_inactive_reward_terms_on_the_hip_kernel(layout, monkeypatch):
    cfg, inactive = og.inactive_terms_cfg()
    sim, _ = make_hip(cfg, layout=layout, monkeypatch=monkeypatch)
    og.check_inactive_reward_terms(sim, cfg, inactive, 1e-4)


@LAYOUTS
def test_injected_resets_are_applied_like_the_oracle(layout, monkeypatch):
    """apply_reset = 1: rows the fixture resets (tilt, time-out, terminating contact) get the masked in-kernel
    reset_idx; compared with the oracle's reset on the same records (same Philox streams)."""
    from oracle.binding import OracleSim
    d = np.load(og.os.path.join(og.G, "pipeline.npz"))
    N = d["s0_in_root"].shape[0]
    cfg = make_cfg(noise=False, dr=False)
    sim, _ = make_hip(cfg, N, layout=layout, monkeypatch=monkeypatch)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N)
    ora = OracleSim(c, "f32", keep)
    arr = og.states_from(d, "s0_in_", N)
    for i in range(N):
        ora.post_physics(i, arr[i], apply_reset=True, common_step_counter=1)
    sim.debug_post_physics(arr, apply_reset=True, common_step_counter=1)
    torch.cuda.synchronize()
    assert d["s0_out_reset"].sum() >= 3
    # the oracle's debug entry does not redraw commands on the resampling interval (the kernel does: legged_robot.py:315-317)
    rows = torch.tensor((d["s0_in_episode_length"] + 1) % int(cfg.commands.resampling_command_interval_s / 0.02) != 0)
    for name in ("RESET", "TIME_OUT", "EPISODE_LENGTH", "FEET_CONTACT"):
        assert torch.equal(sim.tensor(name).cpu().to(torch.int64), ora.tensor(name).to(torch.int64)), name
    for name in ("DOF_POS", "DOF_VEL", "ROOT_STATES", "COMMANDS", "OBS", "PRI_OBS", "REW", "LAST_ACTIONS", "FEET_AIR_TIME", "EPISODE_SUMS"):
        a, b = sim.tensor(name).cpu().double(), ora.tensor(name).double()
        if name == "EPISODE_SUMS":   # (the oracle's debug entry leaves the sums of reset rows in place; its step zeroes them)
            was_reset = torch.tensor(d["s0_out_reset"].astype(bool))
            assert float(a[:, was_reset].abs().max()) == 0.0 and float(b[:, was_reset].abs().max()) > 0.0
            a, b = a[:, rows & ~was_reset], b[:, rows & ~was_reset]
        else:
            a, b = a[rows], b[rows]
        assert ((a - b).abs() <= 1e-4 + 1e-4 * b.abs()).all(), name


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_oracle_pin_pipeline_on_this_box(precision, tol):
    og.test_pipeline_two_steps(precision, tol)


@pytest.mark.parametrize("precision,tol", [("f64", 2e-6), ("f32", 1e-4)])
def test_oracle_pin_reward_terms_on_this_box(precision, tol):
    og.test_every_reward_term(precision, tol)
    if precision == "f64":
        og.test_inactive_reward_terms_formulas()


@pytest.mark.parametrize("precision,tol", [("f64", 1e-6), ("f32", 1e-4)])
def test_oracle_pin_torques_on_this_box(precision, tol):
    og.test_clip_actions_and_torques(precision, tol)


@LAYOUTS
def test_quat_fixture_on_the_hip_kernel(layout, monkeypatch):
    """G-1 (tests/golden/quat.npz) through the step kernel's post-physics half: quat_rotate_inverse on 256 random orientations."""
    cfg = make_cfg(noise=False, dr=False)
    sim, _ = make_hip(cfg, 256, layout=layout, monkeypatch=monkeypatch)
    og.check_quat_rotate_inverse(sim, 1e-4)


@LAYOUTS
def test_quat_apply_yaw_fixture_selects_the_height_scan_cells(layout, monkeypatch):
    """G-1's quat_apply_yaw column on the HIP height scan (legged_robot.py:1235-1274 -> math.py:38-42): env i carries the fixture's
    quaternion q_i and the scan's point i is the fixture's v_i (x, y), so measured_heights[i, i] reads the raster cell under
    root_xy + quat_apply_yaw(q_i, v_i).  The raster encodes its own indices (h[r, c] = r * 181 + c, increasing both ways: the min of
    the three corners the reference takes is h[r, c] itself), so the kernel's cell choice is compared EXACTLY with the one the
    reference's arithmetic makes from the fixture's apply_yaw output."""
    import types
    from wiki_grx_gym_amd.sim import HipSim
    d = np.load(og.os.path.join(og.G, "quat.npz"))
    N, nh, A = 128, 121, 181
    cfg = make_cfg(noise=False, dr=False, terrain="heightfield", curriculum=False)
    rows = cols = 180
    hs = (np.arange(rows)[:, None] * A + np.arange(cols)[None, :]).astype(np.int16)
    ter = types.SimpleNamespace(heightsamples=hs, env_origins=np.zeros((cfg.terrain.num_rows, cfg.terrain.num_cols, 3), np.float32))
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    assert c.num_height_points == nh
    scale = 3.0 / np.abs(d["v"][:nh, :2]).max()            # keep every point within 3 m of the base
    for k in range(nh):
        c.height_points[k][0], c.height_points[k][1] = float(d["v"][k, 0] * scale), float(d["v"][k, 1] * scale)
    c.border_size = 0.0
    pick_layout(monkeypatch, layout)   # (heightfield kernels: the scan runs over one wave, or over the four / seven waves of the pipelines)
    sim = HipSim(c, "cuda:0", keep)
    assert sim.layout()["waves_per_block"] == KERNEL_OF[layout][2] and sim.layout()["kernel"].startswith(KERNEL_OF[layout][0] + "true")
    arr, _ = og.quat_states(N)
    for i in range(N):
        arr[i].root[0], arr[i].root[1], arr[i].root[2] = 9.0, 9.0, 100.0
    sim.debug_post_physics(arr, apply_reset=False, common_step_counter=1)
    torch.cuda.synchronize()
    mh = sim.tensor("MEASURED_HEIGHTS").cpu().numpy()
    code = np.rint(mh / np.float32(c.vertical_scale)).astype(np.int64)
    checked = 0
    for i in range(nh):
        # the reference's float32 arithmetic on ITS OWN quat_apply_yaw output (legged_robot.py:1262-1268)
        ay = (d["apply_yaw"][i, :2] * np.float32(scale)).astype(np.float32)    # quat_apply_yaw is linear in the vector
        p = (ay + np.float32(9.0)) + np.float32(0.0)
        f = p / np.float32(c.horizontal_scale)
        if np.abs(f - np.rint(f)).min() < 2e-3:          # within rounding of a cell edge: either neighbour is legitimate
            continue
        want = int(f[0]) * A + int(f[1])
        assert code[i, i] == want, (i, code[i, i], want)
        checked += 1
    assert checked >= 100


def test_reference_torques_on_the_hip_kernel():
    """G-2 (tests/golden/torques.npz: legged_robot_fftai.py:171-177 clip_actions, legged_robot.py:679-715 _compute_torques with the
    reference's own motor-strength draws) on the HIP step kernel: decimation = 1 makes GRX_T_TORQUES the torque of the FIRST
    (only) sub-step, computed from the injected (dof_pos, dof_vel); the fixture's strength factors go into the writable
    MOTOR_STRENGTH view."""
    d = np.load(og.os.path.join(og.G, "torques.npz"))
    N = d["actions"].shape[0]
    cfg = make_cfg(noise=False, dr=False)
    cfg.control.decimation = 1
    sim, _ = make_hip(cfg, N)
    sim.reset_all()
    root = torch.zeros(N, 13); root[:, 2] = 5.0; root[:, 6] = 1.0          # in the air: no contact in the way
    sim.set_state(root.cuda(), torch.tensor(d["dof_pos"]).cuda().contiguous(), torch.tensor(d["dof_vel"]).cuda().contiguous())
    sim.tensor("MOTOR_STRENGTH").copy_(torch.tensor(d["strength"]).cuda())
    sim.step(torch.tensor(d["actions"]).cuda().contiguous(), 0.0, 1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sim.tensor("ACTIONS").cpu().numpy(), d["clipped"])
    np.testing.assert_allclose(sim.tensor("TORQUES").cpu().numpy(), d["torques"], rtol=1e-4, atol=1e-3)
    lim = d["torque_limits"]
    assert (np.abs(d["torques"]) >= lim - 1e-6).any() and (np.abs(d["torques"]) < lim - 1).any()   # saturated and unsaturated rows


@pytest.mark.gpu
@pytest.mark.parametrize("ct", ["P", "V", "T"])
def test_control_types_on_the_hip_kernel(ct):
    """cfg.control.control_type 'P' / 'V' / 'T' (legged_robot.py:693-707; tests/golden/control_modes.npz holds the reference's own
    _compute_torques for each) on the HIP step kernel: decimation = 1 makes GRX_T_TORQUES the first (only) sub-step's torque; the
    fixture's last_dof_vel and motor-strength draws go into the writable LAST_DOF_VEL / MOTOR_STRENGTH views.  'V' and 'T' handles run
    the general one-wave layout (include/grx.h, ABI 5)."""
    d = np.load(og.os.path.join(og.G, "control_modes.npz"))
    N = d["actions"].shape[0]
    cfg = make_cfg(noise=False, dr=False)
    cfg.control.decimation = 1
    cfg.control.control_type = ct
    sim, _ = make_hip(cfg, N)
    lay = sim.layout()
    assert ct == "P" or (lay["kernel"].startswith("grx_step_kernel<") and lay["waves_per_block"] == 1), lay
    sim.reset_all()
    root = torch.zeros(N, 13); root[:, 2] = 5.0; root[:, 6] = 1.0          # in the air: no contact in the way
    sim.set_state(root.cuda(), torch.tensor(d["dof_pos"]).cuda().contiguous(), torch.tensor(d["dof_vel"]).cuda().contiguous())
    sim.tensor("MOTOR_STRENGTH").copy_(torch.tensor(d["strength"]).cuda())
    sim.tensor("LAST_DOF_VEL").copy_(torch.tensor(d["last_dof_vel"]).cuda())
    sim.step(torch.tensor(d["actions"]).cuda().contiguous(), 0.0, 1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sim.tensor("ACTIONS").cpu().numpy(), d["clipped"])
    # (the velocity law divides a velocity difference by sim_dt: fp32 keeps ~1e-7 x 0.4 x 500 x d_gain of it)
    np.testing.assert_allclose(sim.tensor("TORQUES").cpu().numpy(), d["torques_" + ct], rtol=1e-4, atol=5e-3 if ct == "V" else 1e-3)
    sim.close()


@pytest.mark.gpu
def test_heading_command_on_the_hip_kernel():
    """legged_robot.py:320-326 through the HIP kernel's post-physics half (grx_debug_post_physics): the reference's commands[:, 2]."""
    cfg = make_cfg(noise=True, dr=False)
    cfg.commands.heading_command = True
    sim, _ = make_hip(cfg, 64)
    lay = sim.layout()
    assert lay["kernel"].startswith("grx_step_kernel<") and lay["waves_per_block"] == 1, lay
    og.check_heading_command(sim, 1e-4)
    sim.close()


# ---- the reference's rough raster on the HIP kernels --------------------------------------------------
def make_hip_on_reference_raster(N, layout, monkeypatch, noise=True):
    from wiki_grx_gym_amd.sim import HipSim
    ter, d = og.reference_raster_terrain()
    cfg = og.rough_cfg(noise)
    pick_layout(monkeypatch, layout)
    c, keep, meta = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    sim = HipSim(c, "cuda:0", keep)
    lay = sim.layout()
    name, lpe, waves = KERNEL_OF[layout]
    assert lay["kernel"].startswith(name + "true") and lay["lanes_per_env"] == lpe and lay["waves_per_block"] == waves, lay
    return sim, cfg, meta, d


@LAYOUTS
def test_get_heights_fixture_on_the_hip_kernel(layout, monkeypatch):
    """SURVEY G-7 on the HIP height scan itself: tests/golden/terrain.npz -- the reference's raster, 64 root poses incl. negative
    coordinates (`.long()` truncates towards zero) and poses beyond the map (index clamps 0 / dim - 2), the reference's
    _get_heights output (legged_robot.py:1235-1274) -- through grx_debug_post_physics on every layout: MEASURED_HEIGHTS exact,
    except scan points that sit within fp32 rounding of a cell edge (each differing point is verified to be one)."""
    d = np.load(og.os.path.join(og.G, "terrain.npz"))
    sim, _, _, _ = make_hip_on_reference_raster(d["root"].shape[0], layout, monkeypatch, noise=False)
    n_edge = og.check_get_heights(sim, d, 2e-3)
    print("height samples on a cell edge within rounding:", n_edge, "of", d["heights"].size)
    sim.close()


@LAYOUTS
def test_pipeline_on_the_rough_raster_on_the_hip_kernel(layout, monkeypatch):
    """tests/golden/pipeline_rough.npz (one reference post_physics_step on the rough raster, heights from the reference's own
    _get_heights): the height block of pri_obs, feet_height, base_heights_offset, obs, rewards on every layout at 1e-4."""
    sim, cfg, meta, _ = make_hip_on_reference_raster(64, layout, monkeypatch)
    og.check_pipeline_rough(sim, cfg, meta, 1e-4)
    sim.close()


# ---- round 6 (VERDICT r5 #4b): reference-generated fixtures of the robots beyond the 10-DOF GR1T1 -- the GR1T2 lower limb (20-body termination
# set incl. imu_link) on the fused kernels it runs on, and the 32-DOF full body (the reference's GR1T1 class fed the full-body names: 105 / 234
# observation columns, 32-joint reward sums, its own joint index sets) on the tree kernels BASELINE.json's config 5 runs
@pytest.mark.parametrize("layout", [1, 8, "quad"])
def test_gr1t2_pipeline_fixture_on_the_hip_kernel(layout, monkeypatch):
    pick_layout(monkeypatch, layout)
    from wiki_grx_gym_amd.sim import HipSim
    cfg, c, keep, meta = og.make_other_robot("gr1t2", 64)
    sim = HipSim(c, "cuda:0", keep)
    name, lpe, waves = KERNEL_OF[layout]
    lay = sim.layout()
    assert lay["kernel"].startswith(name) and lay["lanes_per_env"] == lpe and lay["waves_per_block"] == waves, lay
    og.check_pipeline_other_robot(sim, cfg, meta, 1e-4, "gr1t2")


@pytest.mark.parametrize("group", [8, 16])
def test_full_body_pipeline_fixture_on_the_tree_kernel(group, monkeypatch):
    monkeypatch.delenv("GRX_FORCE_GENERIC", raising=False)
    monkeypatch.setenv("GRX_TREE", "1")
    monkeypatch.setenv("GRX_TREE_G", str(group))
    from wiki_grx_gym_amd.sim import HipSim
    cfg, c, keep, meta = og.make_other_robot("full_body", 64)
    sim = HipSim(c, "cuda:0", keep)
    lay = sim.layout()
    assert lay["kernel"].startswith("grx_step_tree16<" if group == 16 else "grx_step_tree<") and lay["lanes_per_env"] == group, lay
    og.check_pipeline_other_robot(sim, cfg, meta, 1e-4, "full_body")
