"""Host-side cost of a rollout step: enqueue time (no sync) vs completed time, and a cProfile of 10 iterations."""
import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import torch
import wiki_grx_gym_amd.envs  # noqa
from wiki_grx_gym_amd.utils import get_args, task_registry
args = get_args(["--task", "GR1T1", "--headless", "--num_envs", "4096", "--seed", "1"])
env, _ = task_registry.make_env("GR1T1", args=args)
runner, _ = task_registry.make_alg_runner(env, name="GR1T1", args=args, log_root=None)
alg = runner.algorithm
T = runner.num_steps_per_env
obs = env.get_observations(); pri = env.get_privileged_observations()
def full():
    global obs, pri
    with torch.inference_mode():
        for _ in range(T):
            a = alg.act(obs, pri)
            obs, pri, r, d, i = env.step(a)
            alg.process_env_step(r, d, i)
    alg.clear_storage()
for _ in range(3): full()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); full(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"us/step: host enqueue {(t1 - t0) / T * 1e6:.1f}, completed {(t2 - t0) / T * 1e6:.1f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): full()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
