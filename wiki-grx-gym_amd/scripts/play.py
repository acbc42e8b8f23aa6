"""play.py -- load the last checkpoint and roll the policy (reference legged_gym/scripts/play.py:42-137,
without the viewer / matplotlib logger, which are out of scope)."""
import os

import torch

from wiki_grx_gym_amd.envs import *  # noqa: F401,F403
from wiki_grx_gym_amd.utils import export_policy_as_jit, get_args, task_registry
from wiki_grx_gym_amd.utils.task_registry import LEGGED_GYM_ROOT_DIR

EXPORT_POLICY = True


def play(args, steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 50)          # play.py:45-54 overrides
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name, "exported", "policies")
        export_policy_as_jit(ppo_runner.algorithm.actor_critic, path)
        print("Exported policy as jit script to: ", path)
    total = steps if steps is not None else 10 * int(env.max_episode_length)
    rew = 0.0
    for i in range(total):
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        rew += rews.mean().item()
    print(f"mean reward per step over {total} steps: {rew / total:.4f}")


if __name__ == "__main__":
    play(get_args())
