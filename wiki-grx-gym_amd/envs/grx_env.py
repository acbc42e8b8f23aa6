"""GRx vectorised environments: the reference's VecEnv surface over the fused HIP step.

Mirrors, for the registered tasks, the class chain
``BaseTask -> LeggedRobot -> LeggedRobotFFTAI -> GR1T1 (-> GR1T2)`` of the reference
(base_task.py:39-150, legged_robot.py:53-305, legged_robot_fftai.py:12-177, gr1t1.py:7-16):
same constructor signature ``cls(cfg, sim_params, physics_engine, sim_device, headless)``
(task_registry.py:98-102), same attributes (``num_envs, num_obs, num_pri_obs, num_actions,
max_episode_length, dt, device, obs_buf, pri_obs_buf, rew_buf, reset_buf, episode_length_buf,
extras, dof_pos, dof_vel, torques, commands, base_lin_vel, base_ang_vel, root_states,
feet_indices, ...``), same methods (``step, reset, get_observations,
get_privileged_observations``).  What differs is *where the work happens*: ``step`` is one C-ABI
call (one kernel launch) instead of ~70 gym calls and ~400 torch kernels, and there is no host
synchronisation in it.

Every buffer attribute is a zero-copy view of library-owned device memory (the
``gymtorch.wrap_tensor`` model, legged_robot.py:110-135); SoA-backed ones are (N, k) views with
strides (1, N).
"""
import math

import numpy as np
import torch

from .. import _capi
from ..sim import HipSim
from . import build_config
from .config import class_to_dict


class _EpisodeInfo(dict):
    """extras["episode"] of ONE step (legged_robot.py:419-428) without a kernel or a copy in env.step(): a dict -- the reference's
    type: rsl_rl's logger ASSIGNS into it (`ep_info[key] = ep_info[key].unsqueeze(0)`, on_policy_runner.py:226-231) -- whose
    entries materialise on first access as 0-dim device-tensor views of that step's row of the library's statistics history ring
    (GRX_T_EPISODE_STATS_HISTORY).  The runner reads the rows once per iteration (on_policy_runner.py:121-133); a row stays valid
    until GRX_STATS_HISTORY further launches of the handle (steps AND resets: the guard compares the library's launch numbers,
    grx_step_args.stats_seq) have gone by."""
    __slots__ = ("_env", "_slot", "_seq", "_filled")

    def __init__(self, env, slot, seq):
        dict.__init__(self)
        self._env, self._slot, self._seq, self._filled = env, slot, seq, False

    def _fill(self):
        if self._filled:
            return
        env = self._env
        cur = env._sim.stats_seq()
        if cur - self._seq >= _capi.STATS_HISTORY:
            raise RuntimeError("extras['episode'] of a step more than GRX_STATS_HISTORY launches back has been overwritten")
        if cur == self._seq:
            env._sim.flush_stats()   # the last launch's statistics are otherwise reduced by the next one
        row = env._stats_hist[self._slot]
        for k, i in env._episode_keys.items():
            dict.setdefault(self, k, row[i])   # (a caller's own assignment wins)
        self._filled = True

    def __setitem__(self, key, value):
        dict.__setitem__(self, key, value)

    def __getitem__(self, key):
        if not dict.__contains__(self, key):
            self._fill()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._fill()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key in self._env._episode_keys or dict.__contains__(self, key)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def copy(self):
        self._fill()
        return dict(dict.items(self))

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


class GRxEnv:
    """VecEnv (rsl_rl/env/vec_env.py:7-40) implemented on libgrx_hip.so."""

    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cuda:0", headless=True,
                 env_offset=0, total_envs=None):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless
        self.viewer = None
        self.init_done = False
        self.debug_viz = False
        self.height_samples = None
        sim_dt = float(_get(sim_params, "dt", cfg.sim.dt))
        # _parse_cfg (legged_robot.py:91-104)
        if cfg.terrain.mesh_type not in ("heightfield", "trimesh"):
            cfg.terrain.curriculum = False
        self.dt = cfg.control.decimation * sim_dt
        self.obs_scales = cfg.normalization.obs_scales
        self.reward_scales = class_to_dict(cfg.rewards.scales)
        self.command_ranges = class_to_dict(cfg.commands.ranges)
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        cfg.domain_rand.push_interval = np.ceil(cfg.domain_rand.push_interval_s / self.dt)
        cfg.commands.resample_command_interval = int(cfg.commands.resampling_command_interval_s / self.dt)
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_obs
        self.num_pri_obs = cfg.env.num_pri_obs
        self.num_actions = cfg.env.num_actions
        seed = getattr(cfg, "seed", 1)
        # create_sim (legged_robot.py:507-528): terrain + simulation handle
        self.terrain = None
        if cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            from ..utils.terrain import Terrain
            self.terrain = Terrain(cfg.terrain, total_envs or self.num_envs, seed=seed)
        elif cfg.terrain.mesh_type != "plane":
            raise ValueError("Terrain mesh type not recognised. Allowed types are [plane, heightfield, trimesh]")
        c, keep, meta = build_config.build(cfg, sim_dt, self.num_envs, env_offset, total_envs, seed, self.terrain)
        self._sim = HipSim(c, sim_device, keep)
        self.device = str(self._sim.device) if self._sim.device.type == "cpu" else f"cuda:{self._sim.device.index or 0}"
        self._meta = meta
        rm = meta["model"]
        # asset facts the reference reads back from gym (legged_robot.py:966-977, 1092-1161)
        self.num_dof = self.num_dofs = rm.num_dofs
        self.num_bodies = rm.num_links
        self.body_names = list(rm.body_names)
        self.dof_names = list(rm.dof_names)
        dev = self._sim.device
        self.feet_indices = torch.tensor(meta["feet_links"], dtype=torch.long, device=dev)
        self.termination_contact_indices = torch.tensor(meta["termination_links"], dtype=torch.long, device=dev)
        self.penalised_contact_indices = torch.tensor(meta["penalised_links"], dtype=torch.long, device=dev)
        # the body index sets of gr1t1.py:18-113 (_create_envs_get_indices): links whose name contains cfg.asset.<x>_name
        for name in ("torso", "forehead", "imu", "waist", "head", "thigh", "shank", "sole", "upper_arm", "lower_arm", "hand",
                     "arm_base", "arm_end"):
            sub = getattr(cfg.asset, name + "_name", None)
            setattr(self, name + "_indices", torch.tensor(rm.links_containing(sub) if sub else [], dtype=torch.long, device=dev))
        kp, kd, q0 = build_config.resolve_gains(cfg, rm.dof_names)
        self.p_gains = torch.tensor(kp, dtype=torch.float, device=dev)
        self.d_gains = torch.tensor(kd, dtype=torch.float, device=dev)
        self.default_dof_pos = torch.tensor(q0, dtype=torch.float, device=dev).unsqueeze(0)
        self.dof_pos_limits = torch.tensor(build_config.soft_dof_pos_limits(rm, cfg.rewards.soft_dof_pos_limit), dtype=torch.float, device=dev)
        self.dof_vel_limits = torch.tensor(rm.dof_vel_limit, dtype=torch.float, device=dev)
        self.torque_limits = torch.tensor(rm.dof_effort, dtype=torch.float, device=dev)
        for name in ("knee", "hip_roll", "hip_yaw", "hip_pitch", "ankle", "ankle_pitch"):
            setattr(self, name + "_indices", rm.dofs_containing(getattr(cfg.asset, name + "_name", name)))
        # reward bookkeeping as _prepare_reward_function leaves it (legged_robot.py:840-866)
        for k in list(self.reward_scales.keys()):
            if self.reward_scales[k] == 0:
                self.reward_scales.pop(k)
            else:
                self.reward_scales[k] *= self.dt
        self.reward_names = [n for n in self.reward_scales if n != "termination"]
        self._term_index = {n: _capi.REWARD_TERMS.index(n) for n in self.reward_scales}
        # buffers (base_task.py:69-76, legged_robot.py:106-203): zero-copy views
        t = self._sim.tensor
        # obs_buf / pri_obs_buf are REBOUND to fresh tensors every step, as in the reference (torch.cat / torch.clip
        # create new tensors, gr1t1.py:282, legged_robot.py:241): rsl_rl keeps a reference to the observation it
        # acted on until after env.step() (ppo.py:160-161, 194), so handing out the live library view would alias.
        # Here: no copy either -- the step kernel writes straight into one of TWO buffers the wrapper alternates between
        # (grx_step_args.obs_out / pri_obs_out): the tensor handed out by step t stays intact until step t + 2.
        self._obs_view = t("OBS")
        self._pri_view = t("PRI_OBS") if self.num_pri_obs is not None else None
        self._obs_ring = [torch.zeros_like(self._obs_view) for _ in range(2)]
        self._pri_ring = [torch.zeros_like(self._pri_view) for _ in range(2)] if self._pri_view is not None else None
        self._ring = 0
        self.obs_buf = self._obs_ring[1]
        self.pri_obs_buf = self._pri_ring[1] if self._pri_ring is not None else None
        self.rew_buf = t("REW")
        self._reset_u8 = t("RESET")
        self._timeout_u8 = t("TIME_OUT")
        self.reset_buf = self._reset_u8.view(torch.bool)
        self.time_out_buf = self._timeout_u8.view(torch.bool)
        self._episode_length = t("EPISODE_LENGTH")
        self.root_states = t("ROOT_STATES")
        self.base_pos = self.root_states[:, 0:3]
        self.base_quat = self.root_states[:, 3:7]
        self.dof_pos, self.dof_vel = t("DOF_POS"), t("DOF_VEL")
        self.torques, self.actions = t("TORQUES"), t("ACTIONS")
        self.last_actions, self.last_dof_vel = t("LAST_ACTIONS"), t("LAST_DOF_VEL")
        self.last_last_actions = self.last_actions        # identical by construction (FF:94 after LR:299)
        self.commands = t("COMMANDS")
        self.base_lin_vel, self.base_ang_vel = t("BASE_LIN_VEL"), t("BASE_ANG_VEL")
        self.base_projected_gravity = t("PROJECTED_GRAVITY")
        self.feet_contact_forces = t("FEET_CONTACT_FORCE")
        self.feet_contact = t("FEET_CONTACT").view(torch.bool)
        self.feet_air_time, self.feet_land_time, self.feet_height = t("FEET_AIR_TIME"), t("FEET_LAND_TIME"), t("FEET_HEIGHT")
        self.avg_feet_contact_force, self.avg_feet_speed_xyz = t("AVG_FEET_FORCE"), t("AVG_FEET_SPEED")
        self._measured_heights = None
        self.base_heights_offset = t("BASE_HEIGHTS_OFFSET")
        self.env_origins = t("ENV_ORIGINS")
        self.terrain_levels, self.terrain_types = t("TERRAIN_LEVELS"), t("TERRAIN_TYPES")
        self.motor_strength_scales = t("MOTOR_STRENGTH")
        self._episode_sums = t("EPISODE_SUMS")
        self._stats_hist = t("EPISODE_STATS_HISTORY")
        self._episode_keys = {"rew_" + n: i for n, i in self._term_index.items()}
        if self.cfg.terrain.curriculum:
            self._episode_keys["terrain_level"] = _capi.NUM_REWARD_TERMS + 1
        self.episode_sums = {n: self._episode_sums[i] for n, i in self._term_index.items()}
        # (N, num_bodies, 3) net contact force per URDF link, last sub-step (LR:117): zero-copy view of the library tensor
        self.contact_forces = t("CONTACT_FORCES")[:, :self.num_bodies]
        self._rbs = None
        self.noise_scale_vec = self._noise_scale_vec()
        self.common_step_counter = 0
        self.extras = {}
        self.custom_origins = cfg.terrain.mesh_type in ("heightfield", "trimesh")
        # action latency N(5, 2) sub-steps, one draw per step shared by all envs (FF:53-54);
        # numpy's global generator like the reference, so set_seed() governs it
        self._delay_rng = np.random
        self.fixed_action_delay = None
        self.init_done = True

    # ------------------------------------------------------------------ VecEnv surface
    @property
    def episode_length_buf(self):
        return self._episode_length

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        # the runner REBINDS this attribute (on_policy_runner.py:126): copy into the library buffer
        self._episode_length.copy_(value.to(self._episode_length.dtype))

    @property
    def rigid_body_states(self):
        """(N, num_links, 13) pos / quat xyzw / lin vel / ang vel of every URDF link, world frame: the layout of
        gym.acquire_rigid_body_state_tensor (legged_robot.py:113,134).  A zero-copy view of the library tensor, brought up to date
        WHEN READ (cfg.env.publish_rigid_body_states = True: grx_refresh, one small launch per step at most -- the
        gym.refresh_rigid_body_state_tensor model, legged_robot_fftai.py:76; "every_step": the step kernel writes it after its last
        sub-step, 1.9 KB per env-step).  Either way it shows the state before reset_idx / _push_robots, as the reference's does."""
        full = self._sim.tensor("RIGID_BODY_STATES")     # (refreshes an on-demand tensor; the view object is cached)
        if self._rbs is None:
            self._rbs = full[:, :self.num_bodies]
        return self._rbs

    @property
    def measured_heights(self):
        """(N, 121) raw terrain heights under the scan points (legged_robot.py:289, 1235-1274), taken before reset_idx: a zero-copy view,
        brought up to date when read (cfg.env.publish_measured_heights = True) or written by every step ("every_step")."""
        self._measured_heights = self._sim.tensor("MEASURED_HEIGHTS")
        return self._measured_heights

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.pri_obs_buf

    def step(self, actions):
        """legged_robot.py:222-246 -- one fused kernel launch."""
        a = actions.to(device=self._sim.device, dtype=torch.float32)
        if not a.is_contiguous():
            a = a.contiguous()
        delay = self.fixed_action_delay
        if delay is None:
            delay = max(0.0, float(self._delay_rng.normal(loc=5, scale=2, size=1)[0]))   # FF:53-54
        self.common_step_counter += 1
        k = self._ring
        self._ring ^= 1
        slot = self._sim.step(a, delay, self.common_step_counter, obs_out=self._obs_ring[k],
                              pri_obs_out=self._pri_ring[k] if self._pri_ring is not None else None)
        self.obs_buf = self._obs_ring[k]
        if self._pri_ring is not None:
            self.pri_obs_buf = self._pri_ring[k]
        # extras (legged_robot.py:419-440): no kernel, no copy -- a view object over the step's statistics row
        self.extras["episode"] = _EpisodeInfo(self, slot, self._sim.last_stats_seq)
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf
        return self.obs_buf, self.pri_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self):
        """base_task.py:117-121: reset every env, then one zero-action step."""
        self._sim.reset_all()
        obs, pri, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self._sim.device))
        return obs, pri

    def reset_idx(self, env_ids):
        """legged_robot.py:377-440 for callers outside step() (inside it resets are masked in the fused kernel, no host sync):
        curriculum move, dof / root / command draws, history and timers zeroed; the envs' episode sums go to the episode
        statistics of the next extras["episode"]."""
        if len(env_ids) == 0:
            return
        self._sim.reset_idx(env_ids)

    def set_dof_state_indexed(self, env_ids, dof_pos, dof_vel):
        """gym.set_dof_state_tensor_indexed (legged_robot.py:737-740): rows env_ids of full (N, nd) tensors"""
        self._sim.set_state_indexed(env_ids, None, dof_pos.contiguous(), dof_vel.contiguous())

    def set_root_state_indexed(self, env_ids, root_states):
        """gym.set_actor_root_state_tensor_indexed (legged_robot.py:782-784): rows env_ids of a full (N, 13) tensor"""
        self._sim.set_state_indexed(env_ids, root_states.contiguous(), None, None)

    def render(self, sync_frame_time=True):
        pass  # headless only (viewer is out of scope)

    def set_camera(self, position, lookat):
        pass

    def close(self):
        self._sim.close()

    # ------------------------------------------------------------------ helpers
    def _noise_scale_vec(self):
        """gr1t1.py:315-336"""
        n, s, lv = self.cfg.noise.noise_scales, self.obs_scales, self.cfg.noise.noise_level
        nd = self.num_dof
        v = torch.zeros(self.num_obs, device=self._sim.device)
        v[3:6] = n.ang_vel * lv * s.ang_vel
        v[6:9] = n.gravity * lv * s.gravity
        v[9:9 + nd] = n.dof_pos * lv * s.dof_pos
        v[9 + nd:9 + 2 * nd] = n.dof_vel * lv * s.dof_vel
        v[9 + 2 * nd:9 + 3 * nd] = n.action * lv * s.action
        return v


def _get(obj, name, default):
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get("sim", obj).get(name, default) if isinstance(obj.get("sim", obj), dict) else default
    return getattr(obj, name, default)


# class names of the reference (envs/__init__.py:30-50)
class LeggedRobot(GRxEnv):
    pass


class LeggedRobotFFTAI(LeggedRobot):
    pass


class GR1T1(LeggedRobotFFTAI):
    pass


class GR1T2(GR1T1):
    pass
