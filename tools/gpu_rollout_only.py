import sys; sys.path.insert(0, ".")
import torch, wiki_grx_gym_amd.envs
from wiki_grx_gym_amd.utils import get_args, task_registry
args = get_args(["--task", "GR1T1", "--headless", "--num_envs", "4096", "--seed", "1"])
env, _ = task_registry.make_env("GR1T1", args=args)
runner, _ = task_registry.make_alg_runner(env, name="GR1T1", args=args, log_root=None)
alg = runner.algorithm
obs = env.get_observations(); pri = env.get_privileged_observations()
with torch.inference_mode():
    for it in range(6):
        for _ in range(64):
            a = alg.act(obs, pri)
            obs, pri, r, d, i = env.step(a)
            alg.process_env_step(r, d, i)
        alg.clear_storage()
torch.cuda.synchronize()
