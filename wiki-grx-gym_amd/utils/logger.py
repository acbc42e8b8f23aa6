"""Logger -- play.py's state / reward log (interface of the reference's legged_gym/utils/logger.py:36-137:
log_state, log_states, log_rewards, reset, plot_states, print_rewards).

Headless by construction (no viewer on a GPU box): `plot_states` renders the same nine panels into a PNG with the Agg
backend (`save_path`, default ./play_states.png) instead of opening a window from a child process, and `dump` writes
the raw logs as JSON.  matplotlib is optional: without it `plot_states` only writes the JSON next to `save_path`."""
import json
import os
from collections import defaultdict

import numpy as np

# (row, column, title, x label, y label, [(log key, legend label)], x-axis log key or None = time)
_PANELS = (
    (0, 0, "Base velocity x", "time [s]", "base lin vel [m/s]", (("base_vel_x", "measured"), ("command_x", "commanded")), None),
    (0, 1, "Base velocity y", "time [s]", "base lin vel [m/s]", (("base_vel_y", "measured"), ("command_y", "commanded")), None),
    (0, 2, "Base velocity yaw", "time [s]", "base ang vel [rad/s]", (("base_vel_yaw", "measured"), ("command_yaw", "commanded")), None),
    (1, 0, "DOF Position", "time [s]", "Position [rad]", (("dof_pos", "measured"), ("dof_pos_target", "target")), None),
    (1, 1, "Joint Velocity", "time [s]", "Velocity [rad/s]", (("dof_vel", "measured"), ("dof_vel_target", "target")), None),
    (1, 2, "Base velocity z", "time [s]", "base lin vel [m/s]", (("base_vel_z", "measured"),), None),
    (2, 0, "Vertical Contact forces", "time [s]", "Forces z [N]", (("contact_forces_z", "force"),), None),
    (2, 1, "Torque/velocity curves", "Joint vel [rad/s]", "Joint Torque [Nm]", (("dof_torque", "measured"),), "dof_vel"),
    (2, 2, "Torque", "time [s]", "Joint Torque [Nm]", (("dof_torque", "measured"),), None),
)


class Logger:
    def __init__(self, dt):
        self.dt = dt
        self.state_log = defaultdict(list)
        self.rew_log = defaultdict(list)
        self.num_episodes = 0

    # ---- recording
    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, states):
        for key, value in states.items():
            self.log_state(key, value)

    def log_rewards(self, episode_infos, num_episodes):
        """`episode_infos`: extras["episode"] of a step on which `num_episodes` envs finished (means over those envs)."""
        for key, value in episode_infos.items():
            if "rew" in key:
                self.rew_log[key].append(float(value) * num_episodes)
        self.num_episodes += num_episodes

    def reset(self):
        self.state_log.clear()
        self.rew_log.clear()

    # ---- reporting
    def average_rewards(self):
        n = max(self.num_episodes, 1)
        return {key: float(np.sum(values)) / n for key, values in self.rew_log.items()}

    def print_rewards(self):
        print("Average rewards per second:")
        for key, mean in self.average_rewards().items():
            print(f" - {key}: {mean}")
        print(f"Total number of episodes: {self.num_episodes}")

    def dump(self, path):
        def plain(v):
            return np.asarray(v).tolist()
        with open(path, "w") as f:
            json.dump({"dt": self.dt, "states": {k: plain(v) for k, v in self.state_log.items()},
                       "num_episodes": self.num_episodes, "average_rewards_per_second": self.average_rewards()}, f)
        return path

    def plot_states(self, save_path="play_states.png"):
        self.dump(os.path.splitext(save_path)[0] + ".json")
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return None
        steps = max((len(v) for v in self.state_log.values()), default=0)
        t = np.linspace(0.0, steps * self.dt, steps)
        fig, axs = plt.subplots(3, 3, figsize=(15, 10))
        for r, c, title, xl, yl, series, xkey in _PANELS:
            ax, drawn = axs[r, c], False
            for key, label in series:
                y = np.asarray(self.state_log.get(key, []), dtype=np.float64)
                if y.size == 0:
                    continue
                if xkey is not None:
                    x = np.asarray(self.state_log.get(xkey, []), dtype=np.float64)
                    if x.size != y.shape[0]:
                        continue
                    ax.plot(x, y, "x", label=label)
                elif y.ndim == 2:
                    for i in range(y.shape[1]):
                        ax.plot(t[:y.shape[0]], y[:, i], label=f"{label} {i}")
                else:
                    ax.plot(t[:y.shape[0]], y, label=label)
                drawn = True
            ax.set(title=title, xlabel=xl, ylabel=yl)
            if drawn:
                ax.legend()
        fig.tight_layout()
        fig.savefig(save_path, dpi=80)
        plt.close(fig)
        return save_path
