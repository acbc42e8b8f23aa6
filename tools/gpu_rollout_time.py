"""Where a training iteration's rollout time goes (GR1T1, 4096 envs): env.step alone, + policy, + storage bookkeeping."""
import sys, time; sys.path.insert(0, ".")
import torch
import wiki_grx_gym_amd.envs  # noqa
from wiki_grx_gym_amd.utils import get_args, task_registry
args = get_args(["--task", "GR1T1", "--headless", "--num_envs", "4096", "--seed", "1"])
env, _ = task_registry.make_env("GR1T1", args=args)
runner, _ = task_registry.make_alg_runner(env, name="GR1T1", args=args, log_root=None)
alg = runner.algorithm
T = runner.num_steps_per_env
obs = env.get_observations(); pri = env.get_privileged_observations()
def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / T * 1e6)
    return [round(t, 1) for t in ts]
acts = torch.zeros(env.num_envs, env.num_actions, device="cuda:0")
def only_env():
    for _ in range(T): env.step(acts)
def only_act():
    with torch.inference_mode():
        for _ in range(T): alg.actor_critic.act(obs); alg.actor_critic.evaluate(pri)
def act_env():
    global obs, pri
    with torch.inference_mode():
        for _ in range(T):
            a = alg.actor_critic.act(obs)
            obs, pri, r, d, i = env.step(a)
def full():
    global obs, pri
    with torch.inference_mode():
        for _ in range(T):
            a = alg.act(obs, pri)
            obs, pri, r, d, i = env.step(a)
            alg.process_env_step(r, d, i)
    alg.clear_storage()
print("steps per iteration:", T)
print("us/step  env.step only        :", timed(only_env))
print("us/step  policy act+evaluate  :", timed(only_act))
print("us/step  act + env.step       :", timed(act_env))
print("us/step  full rollout step    :", timed(full))
def upd():
    global obs, pri
    full_no_clear()
def full_no_clear():
    global obs, pri
    with torch.inference_mode():
        for _ in range(T):
            a = alg.act(obs, pri)
            obs, pri, r, d, i = env.step(a)
            alg.process_env_step(r, d, i)
        alg.compute_returns(pri)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); full_no_clear(); torch.cuda.synchronize(); t1 = time.perf_counter()
    alg.update(); alg.clear_storage(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iteration {k}: rollout {t1 - t0:.3f} s, update {t2 - t1:.3f} s")
