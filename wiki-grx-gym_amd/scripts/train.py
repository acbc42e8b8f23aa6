"""train.py -- same three lines of logic as the reference (legged_gym/scripts/train.py:40-43).

    python -m wiki_grx_gym_amd.scripts.train --task GR1T1 --headless --num_envs 4096
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m wiki_grx_gym_amd.scripts.train --task GR1T1 --headless
"""
import os

import torch
import torch.distributed as dist

from wiki_grx_gym_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from wiki_grx_gym_amd.utils import get_args, task_registry


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == "__main__":
    args = get_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group("nccl" if args.sim_device_type == "cuda" else "gloo")
    train(args)
