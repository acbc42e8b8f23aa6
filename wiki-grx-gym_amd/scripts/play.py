"""play.py -- load the last checkpoint, export the actor as TorchScript and roll the policy
(reference legged_gym/scripts/play.py:42-137).

Same overrides (:45-56), same export path (:67-82, `logs/<experiment>/exported/policy_jit.pt`), same loop (:96-98) and the
same Logger calls (:86-137; utils/logger.py here is headless: a PNG instead of a window).  No viewer on a GPU box, so
RECORD_FRAMES and MOVE_CAMERA are out of scope.  Files under `exported/`:

  * `play_states.jsonl`  -- one record per step for the first `stop_state_log` steps with exactly the keys of the dict
    passed to `logger.log_states` (:110-126), for robot 0 / joint 1;
  * `play_states.png` / `.json` -- `logger.plot_states()` at step `stop_state_log` (:128-129);
  * `play_rewards.json`  -- what `logger.log_rewards` accumulates and `print_rewards` shows (:131-137): per reward term, the
    episode means weighted by the number of episodes that ended, over the first `max_episode_length` steps.
"""
import json
import os

import torch

from wiki_grx_gym_amd.envs import *  # noqa: F401,F403
from wiki_grx_gym_amd.utils import Logger, export_policy_as_jit, get_args, task_registry
from wiki_grx_gym_amd.utils.task_registry import LEGGED_GYM_ROOT_DIR

EXPORT_POLICY = True

# what play.py:110-126 hands to Logger.log_states for (robot r, joint j): key -> value from the env's tensors
_SCALARS = (
    ("dof_pos", lambda e, r, j: e.dof_pos[r, j]),
    ("dof_vel", lambda e, r, j: e.dof_vel[r, j]),
    ("dof_torque", lambda e, r, j: e.torques[r, j]),
    ("command_x", lambda e, r, j: e.commands[r, 0]),
    ("command_y", lambda e, r, j: e.commands[r, 1]),
    ("command_yaw", lambda e, r, j: e.commands[r, 2]),
    ("base_vel_x", lambda e, r, j: e.base_lin_vel[r, 0]),
    ("base_vel_y", lambda e, r, j: e.base_lin_vel[r, 1]),
    ("base_vel_z", lambda e, r, j: e.base_lin_vel[r, 2]),
    ("base_vel_yaw", lambda e, r, j: e.base_ang_vel[r, 2]),
)


def _state_record(env, actions, r, j):
    rec = {"dof_pos_target": float(actions[r, j]) * env.cfg.control.action_scale}
    rec.update((key, float(get(env, r, j))) for key, get in _SCALARS)
    rec["contact_forces_z"] = env.contact_forces[r, env.feet_indices, 2].cpu().tolist()
    return rec


def play(args, steps=None, log_root="default"):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)

    # override some parameters for testing (play.py:45-56)
    env_cfg.env.episode_length_s = 600.0
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 50)
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False

    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()

    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=log_root)
    policy = ppo_runner.get_inference_policy(device=env.device)

    exp_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name) if log_root == "default" else log_root
    out_dir = os.path.join(exp_root, "exported")
    exported = None
    if EXPORT_POLICY:
        exported = export_policy_as_jit(ppo_runner.algorithm.actor_critic, out_dir)
        print(f"EXPORT_POLICY: Exported policy as jit script to: {exported}")
    os.makedirs(out_dir, exist_ok=True)

    robot_index = 0      # which robot is used for logging
    joint_index = 1      # which joint is used for logging
    stop_state_log = 100                               # number of steps the states are logged for
    stop_rew_log = int(env.max_episode_length) + 1     # number of steps before the average episode rewards are printed
    total = steps if steps is not None else 10 * int(env.max_episode_length)
    logger = Logger(env.dt)
    states_path = os.path.join(out_dir, "play_states.jsonl")
    rewards_path = os.path.join(out_dir, "play_rewards.json")

    def dump_rewards():
        with open(rewards_path, "w") as f:
            json.dump({"num_episodes": logger.num_episodes, "average_per_second": logger.average_rewards()}, f, indent=1)
        logger.print_rewards()

    with open(states_path, "w") as sf:
        for i in range(total):
            actions = policy(obs.detach())
            obs, _, rews, dones, infos = env.step(actions.detach())

            if i < stop_state_log:
                rec = _state_record(env, actions, robot_index, joint_index)
                logger.log_states(rec)
                sf.write(json.dumps(rec) + "\n")
            elif i == stop_state_log:
                sf.flush()
                logger.plot_states(os.path.join(out_dir, "play_states.png"))

            if 0 < i < stop_rew_log:
                if infos["episode"]:
                    num_episodes = int(torch.sum(env.reset_buf).item())
                    if num_episodes > 0:
                        logger.log_rewards(infos["episode"], num_episodes)
            elif i == stop_rew_log:
                dump_rewards()
    if total <= stop_rew_log:
        dump_rewards()
    if total <= stop_state_log:
        logger.plot_states(os.path.join(out_dir, "play_states.png"))
    return dict(env=env, runner=ppo_runner, exported=exported, states=states_path, rewards=rewards_path, logger=logger)


if __name__ == "__main__":
    play(get_args())
