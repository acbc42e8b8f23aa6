"""CLI flags, seeding, checkpoint lookup, policy export -- the surface of the reference's
legged_gym/utils/helpers.py (get_args HP:159-185 incl. the gymutil flags GU:298-370, set_seed
HP:70-80, get_load_path HP:108-130, update_cfg_from_args HP:133-156, export_policy_as_jit HP:188-201)
without any isaacgym import."""
import argparse
import copy
import os
import random

import numpy as np
import torch

from ..envs.config import class_to_dict  # noqa: F401  (re-exported like the reference)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def parse_device_str(device_str):
    """gymutil.parse_device_str: 'cuda:1' -> ('cuda', 1); 'cpu' -> ('cpu', 0)"""
    parts = device_str.split(":")
    kind = parts[0].lower()
    if kind not in ("cpu", "cuda", "gpu"):
        raise ValueError(f'Invalid device string "{device_str}"')
    return ("cuda" if kind == "gpu" else kind), (int(parts[1]) if len(parts) > 1 else 0)


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="GR1T1", help="Registered task name (GR1T1, GR1T2)")
    p.add_argument("--resume", action="store_true", default=False, help="Resume training from a checkpoint")
    p.add_argument("--experiment_name", type=str, help="Name of the experiment to run or load. Overrides config file if provided.")
    p.add_argument("--run_name", type=str, help="Name of the run. Overrides config file if provided.")
    p.add_argument("--load_run", type=str, help="Name of the run to load when resume=True. If -1: will load the last run.")
    p.add_argument("--checkpoint", type=int, help="Saved model checkpoint number. If -1: will load the last checkpoint.")
    p.add_argument("--headless", action="store_true", default=False, help="Force display off at all times")
    p.add_argument("--horovod", action="store_true", default=False, help="(unused in the reference too, HP:169)")
    p.add_argument("--rl_device", type=str, default="cuda:0", help="Device used by the RL algorithm")
    p.add_argument("--num_envs", type=int, help="Number of environments to create. Overrides config file if provided.")
    p.add_argument("--seed", type=int, help="Random seed. Overrides config file if provided.")
    p.add_argument("--max_iterations", type=int, help="Maximum number of training iterations.")
    # gymutil.parse_arguments flags (GU:305-315)
    p.add_argument("--sim_device", type=str, default="cuda:0", help="Physics Device in PyTorch-like syntax")
    p.add_argument("--pipeline", type=str, default="gpu", help="Tensor API pipeline (cpu/gpu)")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--num_threads", type=int, default=0, help="accepted for compatibility (PhysX CPU threads)")
    p.add_argument("--subscenes", type=int, default=0, help="accepted for compatibility")
    p.add_argument("--slices", type=int)
    p.add_argument("--terrain", type=str, choices=["plane", "heightfield", "trimesh"], help="override cfg.terrain.mesh_type")
    args = p.parse_args(argv)
    args.sim_device_type, args.compute_device_id = parse_device_str(args.sim_device)
    args.use_gpu_pipeline = args.pipeline.lower() in ("gpu", "cuda")
    args.use_gpu = args.sim_device_type == "cuda"
    args.physics_engine = "SIM_HIP"          # the reference passes gymapi.SIM_PHYSX
    args.sim_device_id = args.compute_device_id
    args.sim_device = args.sim_device_type + (f":{args.sim_device_id}" if args.sim_device_type == "cuda" else "")
    if "LOCAL_RANK" in os.environ and args.sim_device_type == "cuda":   # torchrun: one process per GPU
        lr = int(os.environ["LOCAL_RANK"])
        args.sim_device = args.rl_device = f"cuda:{lr}"
        args.sim_device_id = lr
    return args


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None:
        if args.num_envs is not None:
            env_cfg.env.num_envs = args.num_envs
        if getattr(args, "terrain", None):
            env_cfg.terrain.mesh_type = args.terrain
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.max_iterations is not None:
            cfg_train.runner.max_iterations = args.max_iterations
        if args.resume:
            cfg_train.runner.resume = args.resume
        for name in ("experiment_name", "run_name", "load_run", "checkpoint"):
            if getattr(args, name) is not None:
                setattr(cfg_train.runner, name, getattr(args, name))
    return env_cfg, cfg_train


def parse_sim_params(args, cfg):
    """HP:83-105 returns a gymapi.SimParams; here a plain namespace with the same field names."""
    sim = dict(cfg.get("sim", {}))
    ns = argparse.Namespace(**{k: v for k, v in sim.items() if not isinstance(v, dict)})
    ns.use_gpu_pipeline = args.use_gpu_pipeline
    ns.physx = argparse.Namespace(**sim.get("physx", {}))
    ns.physx.use_gpu = args.use_gpu
    if getattr(args, "num_threads", 0) > 0:
        ns.physx.num_threads = args.num_threads
    return ns


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = sorted(os.listdir(root))
        if "exported" in runs:
            runs.remove("exported")
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = [f for f in os.listdir(load_run) if "model" in f]
        models.sort(key=lambda m: "{0:0>15}".format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def export_policy_as_jit(actor_critic, path):
    os.makedirs(path, exist_ok=True)
    path = os.path.join(path, "policy_jit.pt")
    model = copy.deepcopy(actor_critic.actor).to("cpu")
    torch.jit.script(model).save(path)
    return path
