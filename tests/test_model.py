"""URDF-derived model tables -> lumped dynamics model (known answers: SURVEY.md Appendix A)."""
import numpy as np
import pytest

from wiki_grx_gym_amd.model import RobotModel


def test_gr1t1_lower_limb_lumps():
    m = RobotModel("gr1t1_lower_limb")
    assert m.num_links == 37 and m.num_dofs == 10 and m.num_bodies == 11
    assert m.dof_names[:5] == ["left_hip_roll_joint", "left_hip_yaw_joint", "left_hip_pitch_joint", "left_knee_pitch_joint", "left_ankle_pitch_joint"]
    assert m.total_mass() == pytest.approx(52.8268, abs=1e-4)
    assert m.mass[0] == pytest.approx(21.5948, abs=1e-4)            # base lump: 23 links
    np.testing.assert_allclose(m.com[0], [-0.01147, -0.00085, 0.22238], atol=1e-4)
    np.testing.assert_allclose(np.diag(m.inertia[0]), [1.13969, 0.74417, 0.46885], atol=1e-4)
    assert m.mass[3] == pytest.approx(7.99) and m.mass[4] == pytest.approx(1.93)
    assert m.mass[5] == pytest.approx(1.076)                        # foot_pitch + foot_roll
    np.testing.assert_allclose(m.com[5], [0.0284, 0.0012, -0.0293], atol=1e-4)
    np.testing.assert_allclose(np.diag(m.inertia[5]), [0.00079, 0.00464, 0.00495], atol=2e-5)
    assert m.parent == [-1, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9]
    np.testing.assert_allclose(m.joint_pos[1], [-0.0025, 0.105, -0.0276])
    np.testing.assert_allclose(m.joint_pos[6], [-0.0025, -0.105, -0.0276])
    np.testing.assert_allclose(m.joint_axis[1:6], [[1, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0], [0, 1, 0]])
    np.testing.assert_allclose(m.dof_effort, [48, 66, 130, 130, 15] * 2)
    np.testing.assert_allclose(m.dof_vel_limit, [12.15, 16.76, 37.38, 37.38, 20.32] * 2)


def test_gr1t2_and_full_body():
    m2 = RobotModel("gr1t2_lower_limb")
    assert m2.total_mass() == pytest.approx(56.91, abs=1e-2)
    assert m2.mass[0] == pytest.approx(22.29, abs=1e-2) and m2.mass[5] == pytest.approx(2.26, abs=1e-2)
    full = RobotModel("gr1t1")
    assert full.num_dofs == 32 and full.total_mass() == pytest.approx(52.8268, abs=1e-4)
    # "left leg, right leg, waist, head, left arm, right arm" (gr1t1_config.py:284-299)
    groups = [n.split("_")[0] + "_" + n.split("_")[1] for n in full.dof_names]
    assert full.dof_names[12].startswith("waist") and full.dof_names[15].startswith("head")
    assert full.dof_names[18].startswith("left_shoulder") and full.dof_names[25].startswith("right_shoulder")


def test_contact_spheres_and_index_sets():
    m = RobotModel("gr1t1_lower_limb")
    assert len(m.spheres) == 27
    feet = m.links_containing("foot_roll")
    assert [m.body_names[i] for i in feet] == ["left_foot_roll_link", "right_foot_roll_link"]
    foot_spheres = [s for s in m.spheres if s[3] == feet[0]]
    assert len(foot_spheres) == 4
    for body, pos, r, _ in foot_spheres:                      # sole = z -0.055 in the ankle frame
        assert body == 5 and pos[2] - r == pytest.approx(-0.055, abs=1e-6)
    xs = sorted(round(float(p[0]), 4) for _, p, _, _ in foot_spheres)
    assert xs == [-0.05, -0.05, 0.15, 0.15]
    term = set()
    for n in ["imu", "torso", "head_pitch", "waist", "upper_arm", "lower_arm", "hand"]:
        term.update(m.links_containing(n))
    assert len(term) == 19                                   # 'imu' does not match 'IMU_link' 
    assert len({s[3] for s in m.spheres} & term) == 6        # torso, head_pitch, 2 upper_arm_yaw, 2 hand_yaw
