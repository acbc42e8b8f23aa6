"""utils/logger.py: the interface play.py uses (reference legged_gym/utils/logger.py:36-137), headless."""
import json

import numpy as np
import torch

import wiki_grx_gym_amd.envs  # noqa: F401  (package import order: envs first, as in the scripts)
from wiki_grx_gym_amd.utils import Logger


def test_logger_accumulates_like_the_reference_and_renders_headless(tmp_path):
    lg = Logger(0.02)
    for i in range(30):
        lg.log_states({"dof_pos": 0.1 * i, "dof_pos_target": 0.1 * i + 0.01, "dof_vel": 1.0 - 0.1 * i, "dof_torque": 2.0 * i,
                       "command_x": 0.5, "command_y": 0.0, "command_yaw": 0.1, "base_vel_x": 0.4, "base_vel_y": 0.01,
                       "base_vel_z": -0.02, "base_vel_yaw": 0.09, "contact_forces_z": np.array([100.0 + i, 90.0 - i])})
    assert len(lg.state_log["dof_pos"]) == 30 and lg.state_log["contact_forces_z"][3].shape == (2,)
    # log_rewards: only keys containing 'rew', each weighted by the number of episodes that ended on that step
    lg.log_rewards({"rew_alive": torch.tensor(0.5), "rew_x": 2.0, "terrain_level": torch.tensor(3.0)}, 2)
    lg.log_rewards({"rew_alive": torch.tensor(1.5), "rew_x": 0.0, "terrain_level": torch.tensor(3.0)}, 6)
    assert lg.num_episodes == 8 and set(lg.rew_log) == {"rew_alive", "rew_x"}
    avg = lg.average_rewards()
    assert abs(avg["rew_alive"] - (0.5 * 2 + 1.5 * 6) / 8) < 1e-12 and abs(avg["rew_x"] - 0.5) < 1e-12
    lg.print_rewards()
    png = lg.plot_states(str(tmp_path / "s.png"))
    dumped = json.load(open(tmp_path / "s.json"))
    assert dumped["num_episodes"] == 8 and len(dumped["states"]["contact_forces_z"]) == 30
    if png is not None:   # matplotlib present: nine panels rendered without a display
        assert (tmp_path / "s.png").stat().st_size > 10000
    lg.reset()
    assert not lg.state_log and not lg.rew_log
