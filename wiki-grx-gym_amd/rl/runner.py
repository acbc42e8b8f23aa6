"""On-policy runner with the rsl_rl surface (reference: rsl_rl/runners/on_policy_runner.py:16-345):
``OnPolicyRunner(env, train_cfg_dict, log_dir, device)``, ``learn(num_learning_iterations,
init_at_random_ep_len)``, ``save/load`` with the reference's checkpoint dict keys
(``model_state_dict, optimizer_state_dict, iter, infos``), ``get_inference_policy``, the same
TensorBoard scalar names (Episode/*, Loss/*, Perf/*, Train/*, Policy/*).

Differences, all on the host side: the per-step reward/length bookkeeping stays on the device
(the reference does nonzero() + .cpu() every step, on_policy_runner.py:177-179) and is read back
once per iteration; TensorBoard is optional (a JSON-lines scalar log is always written); with
torch.distributed initialised only rank 0 logs and saves."""
import json
import os
import statistics
import time
from collections import deque

import torch
import torch.distributed as dist

from .modules import ActorCriticMLP
from .ppo import PPO
from .storage import RolloutStorage  # noqa: F401

_POLICIES = {"ActorCriticMLP": ActorCriticMLP, "ActorCritic": ActorCriticMLP}
_ALGORITHMS = {"PPO": PPO}


class _ScalarLog:
    """SummaryWriter when tensorboard is importable, always a scalars.jsonl next to it."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir, flush_secs=10)
        except Exception:
            self._tb = None

    def add_scalar(self, tag, value, step):
        value = float(value)
        self._f.write(json.dumps({"tag": tag, "value": value, "step": step}) + "\n")
        self._f.flush()
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.cfg, self.algorithm_cfg, self.policy_cfg = train_cfg["runner"], train_cfg["algorithm"], train_cfg["policy"]
        self.device, self.env = device, env
        critic_in = env.num_pri_obs if env.num_pri_obs is not None else env.num_obs
        policy_cls = _POLICIES[self.cfg["policy_class_name"]]
        if dist.is_available() and dist.is_initialized():
            torch.manual_seed(int(train_cfg.get("seed", 1)))      # identical replicas on every rank
        actor_critic = policy_cls(env.num_obs, critic_in, env.num_actions, **self.policy_cfg).to(device)
        self.algorithm = _ALGORITHMS[self.cfg["algorithm_class_name"]](actor_critic=actor_critic, device=device, **self.algorithm_cfg)
        self.alg = self.algorithm
        self.num_steps_per_env, self.save_interval = self.cfg["num_steps_per_env"], self.cfg["save_interval"]
        self.algorithm.init_storage(env.num_envs, self.num_steps_per_env)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            torch.manual_seed(int(train_cfg.get("seed", 1)) + 1000 * dist.get_rank())   # different action noise per shard
        self.env.reset()
        self.log_dir, self.writer = log_dir, None
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.is_main = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # Perf/collection time and Perf/learning_time of the last iteration (on_policy_runner.py:235's inputs), also without a log directory;
        # sync_timers: drain the device before each clock is read (bench.py's full_iteration leg: an iteration otherwise ends on the
        # update's .item(), which is a synchronisation too, but the split between the two halves is the host's enqueue time)
        self.last_collection_time = self.last_learn_time = 0.0
        self.sync_timers = False

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg = self.env, self.algorithm
        if self.log_dir is not None and self.writer is None and self.is_main:
            self.writer = _ScalarLog(self.log_dir)
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs = env.get_observations()
        pri = env.get_privileged_observations()
        critic_obs = pri if pri is not None else obs
        obs, critic_obs = obs.to(self.device), critic_obs.to(self.device)
        alg.actor_critic.train()
        ep_infos = []
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        cur_rew = torch.zeros(env.num_envs, dtype=torch.float, device=self.device)
        cur_len = torch.zeros(env.num_envs, dtype=torch.float, device=self.device)
        done_rew = torch.zeros(self.num_steps_per_env, env.num_envs, dtype=torch.float, device=self.device)
        done_len = torch.zeros(self.num_steps_per_env, env.num_envs, dtype=torch.float, device=self.device)
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            with torch.inference_mode():
                for t_ in range(self.num_steps_per_env):
                    actions = alg.act(obs, critic_obs)
                    obs, pri, rewards, dones, infos = env.step(actions)
                    critic_obs = pri if pri is not None else obs
                    obs, critic_obs, rewards, dones = obs.to(self.device), critic_obs.to(self.device), rewards.to(self.device), dones.to(self.device)
                    if self.log_dir is not None:
                        if "episode" in infos:
                            ep_infos.append(infos["episode"])
                        # running episode reward / length and the finished episodes' totals: inside process_env_step (one kernel on a
                        # HIP device), read back once per iteration instead of nonzero() + .cpu() per step (on_policy_runner.py:177-179)
                        alg.process_env_step(rewards, dones, infos, log=(cur_rew, cur_len, done_rew[t_], done_len[t_]))
                    else:
                        alg.process_env_step(rewards, dones, infos)
                if self.log_dir is not None:   # one device->host transfer per iteration
                    m = alg.storage.dones.squeeze(-1).bool()
                    rewbuffer.extend(done_rew[m].cpu().tolist())
                    lenbuffer.extend(done_len[m].cpu().tolist())
                if self.sync_timers:
                    torch.cuda.synchronize()
                collection_time = time.time() - start
                start = time.time()
                alg.compute_returns(critic_obs)
            mean_value_loss, mean_surrogate_loss = alg.update()
            alg.clear_storage()
            if self.sync_timers:
                torch.cuda.synchronize()
            learn_time = time.time() - start
            self.last_collection_time, self.last_learn_time = collection_time, learn_time
            if self.log_dir is not None and self.is_main:
                self.log(locals())
            if self.log_dir is not None and self.is_main and it % self.save_interval == 0:
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
            ep_infos.clear()
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None and self.is_main:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def log(self, locs, width=80, pad=35):
        it = locs["it"]
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs * self.world
        iteration_time = locs["collection_time"] + locs["learn_time"]
        self.tot_time += iteration_time
        w = self.writer
        ep_string = ""
        if locs["ep_infos"]:
            for key in locs["ep_infos"][0]:
                vals = [torch.as_tensor(e[key], dtype=torch.float).reshape(-1).to(self.device) for e in locs["ep_infos"]]
                value = torch.cat(vals).mean().item()
                w.add_scalar("Episode/" + key, value, it)
                ep_string += f"{f'Mean episode {key}:':>{pad}} {value:.4f}\n"
        fps = int(self.num_steps_per_env * self.env.num_envs * self.world / iteration_time)
        alg = self.algorithm
        w.add_scalar("Loss/value_function", locs["mean_value_loss"], it)
        w.add_scalar("Loss/surrogate", locs["mean_surrogate_loss"], it)
        w.add_scalar("Loss/learning_rate", alg.learning_rate, it)
        w.add_scalar("Loss/kl", alg.mean_kl, it)
        w.add_scalar("Perf/total_fps", fps, it)
        w.add_scalar("Perf/collection time", locs["collection_time"], it)
        w.add_scalar("Perf/learning_time", locs["learn_time"], it)
        if len(locs["rewbuffer"]) > 0:
            mr, ml = statistics.mean(locs["rewbuffer"]), statistics.mean(locs["lenbuffer"])
            w.add_scalar("Train/mean_reward", mr, it)
            w.add_scalar("Train/mean_episode_length", ml, it)
            w.add_scalar("Train/mean_reward/time", mr, self.tot_time)
            w.add_scalar("Train/mean_episode_length/time", ml, self.tot_time)
        stds = alg.actor_critic.std.detach()
        for i, s in enumerate(stds):
            w.add_scalar(f"Policy/noise_std_{i}", s.item(), it)
        w.add_scalar("Policy/mean_noise_std", stds.mean().item(), it)
        head = f" Learning iteration {it}/{self.current_learning_iteration + locs['num_learning_iterations']} "
        out = f"{'#' * width}\n{head.center(width, ' ')}\n\n"
        out += f"{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)\n"
        out += f"{'Value function loss:':>{pad}} {locs['mean_value_loss']:.4f}\n{'Surrogate loss:':>{pad}} {locs['mean_surrogate_loss']:.4f}\n"
        out += f"{'Mean action noise std:':>{pad}} {stds.mean().item():.2f}\n"
        if len(locs["rewbuffer"]) > 0:
            out += f"{'Mean reward:':>{pad}} {statistics.mean(locs['rewbuffer']):.2f}\n{'Mean episode length:':>{pad}} {statistics.mean(locs['lenbuffer']):.2f}\n"
        out += ep_string + f"{'-' * width}\n{'Total timesteps:':>{pad}} {self.tot_timesteps}\n{'Iteration time:':>{pad}} {iteration_time:.2f}s\n{'Total time:':>{pad}} {self.tot_time:.2f}s\n"
        print(out)

    def save(self, path, infos=None):
        torch.save({"model_state_dict": self.algorithm.actor_critic.state_dict(),
                    "optimizer_state_dict": self.algorithm.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        loaded = torch.load(path, map_location=self.device, weights_only=False)
        self.algorithm.actor_critic.load_state_dict(loaded["model_state_dict"])
        self.algorithm.invalidate_graphs()
        if load_optimizer:
            self.algorithm.load_optimizer_state(loaded["optimizer_state_dict"])
        self.current_learning_iteration = loaded["iter"]
        return loaded["infos"]

    def get_inference_policy(self, device=None):
        self.algorithm.actor_critic.eval()
        if device is not None:
            self.algorithm.actor_critic.to(device)
        return self.algorithm.actor_critic.act_inference
