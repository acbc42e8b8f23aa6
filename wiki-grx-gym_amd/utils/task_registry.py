"""Task registry: name -> (env class, env cfg, train cfg); same surface as the reference
(legged_gym/utils/task_registry.py:45-160): ``register``, ``get_task_class``, ``get_cfgs``,
``make_env(name, args=None, env_cfg=None) -> (env, env_cfg)``,
``make_alg_runner(env, name=None, args=None, train_cfg=None, log_root="default") -> (runner, train_cfg)``.

Multi-GPU (an addition; the reference is single process): when torch.distributed is initialised,
``make_env`` gives this rank the env shard ``[rank*n, (rank+1)*n)`` of ``world*n`` global envs."""
import os
from datetime import datetime

import torch.distributed as dist

from ..envs.config import class_to_dict
from ..rl import OnPolicyRunner
from .helpers import get_args, get_load_path, parse_sim_params, set_seed, update_cfg_from_args

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LEGGED_GYM_ROOT_DIR = ROOT_DIR


class TaskRegistry:
    def __init__(self):
        self.task_classes, self.env_cfgs, self.train_cfgs = {}, {}, {}

    def register(self, name, task_class, env_cfg, train_cfg):
        self.task_classes[name], self.env_cfgs[name], self.train_cfgs[name] = task_class, env_cfg, train_cfg

    def get_task_class(self, name):
        return self.task_classes[name]

    def get_cfgs(self, name):
        env_cfg, train_cfg = self.env_cfgs[name], self.train_cfgs[name]
        env_cfg.seed = train_cfg.seed
        return env_cfg, train_cfg

    def make_env(self, name, args=None, env_cfg=None):
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg, _ = self.get_cfgs(name)
        env_cfg, _ = update_cfg_from_args(env_cfg, None, args)
        if not hasattr(env_cfg, "seed"):
            env_cfg.seed = self.train_cfgs[name].seed
        set_seed(env_cfg.seed)
        # GRX_T_REWARD_TERMS (the per-term reward table, a debugging tensor the reference has no counterpart of) is only
        # written when asked for.  rigid_body_states and measured_heights are published ON REFRESH (grx_publish_mode, ABI 6): the env's properties
        # refresh them when read, a caller that keeps the raw view must call env._sim.refresh / read the property again after a step
        # (env.publish_* = "every_step" restores the step-written tensors of rounds 3-4); every other reference attribute is always current
        if not hasattr(env_cfg.env, "publish_reward_terms"):
            env_cfg.env.publish_reward_terms = False
        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        shard = {}
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            n = env_cfg.env.num_envs
            shard = dict(env_offset=dist.get_rank() * n, total_envs=dist.get_world_size() * n)
        env = task_class(cfg=env_cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                         sim_device=args.sim_device, headless=args.headless, **shard)
        return env, env_cfg

    def make_alg_runner(self, env, name=None, args=None, train_cfg=None, log_root="default"):
        if args is None:
            args = get_args()
        if train_cfg is None:
            if name is None:
                raise ValueError("Either 'name' or 'train_cfg' must be not None")
            _, train_cfg = self.get_cfgs(name)
        elif name is not None:
            print(f"'train_cfg' provided -> Ignoring 'name={name}'")
        _, train_cfg = update_cfg_from_args(None, train_cfg, args)
        stamp = datetime.now().strftime("%b%d_%H-%M-%S") + "_" + train_cfg.runner.run_name
        if log_root == "default":
            log_root = os.path.join(LEGGED_GYM_ROOT_DIR, "logs", train_cfg.runner.experiment_name)
            log_dir = os.path.join(log_root, stamp)
        elif log_root is None:
            log_dir = None
        else:
            log_dir = os.path.join(log_root, stamp)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), log_dir, device=args.rl_device)
        if train_cfg.runner.resume:
            resume_path = get_load_path(log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
            print(f"Loading model from: {resume_path}")
            runner.load(resume_path)
        return runner, train_cfg


task_registry = TaskRegistry()
