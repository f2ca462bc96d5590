"""HalfCheetah-style running env on the HIP engine (reference: emei/envs/mujoco/half_cheetah.py)."""
import numpy as np

from .. import spaces
from .base import HipEnv


class HalfCheetahRunningEnv(HipEnv):
    ENGINE_NAME = "HalfCheetahRunning"

    def __init__(self, freq_rate: int = 4, real_time_scale: float = 0.002, integrator="euler",
                 forward_reward_weight=1.0, ctrl_cost_weight=0.1, init_noise_params=0.1, obs_noise_params=0.0, **kwargs):
        if integrator != "euler":
            raise NotImplementedError(f"integrator {integrator!r}: only 'euler' is implemented on the HIP engine")
        if obs_noise_params != 0:
            raise NotImplementedError("obs_noise_params != 0 (mujoco_env.py:98-104) is not implemented")
        if forward_reward_weight != 1.0 or ctrl_cost_weight != 0.1:
            raise NotImplementedError("only the default reward weights (half_cheetah.py:23-24) are compiled in")
        self._forward_reward_weight = forward_reward_weight
        self._ctrl_cost_weight = ctrl_cost_weight
        self.init_noise_params = init_noise_params
        self.obs_noise_params = obs_noise_params
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise=float(init_noise_params), **kwargs)
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(18,), dtype=np.float64)
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(6,), dtype=np.float32)
        self.init_qpos = np.zeros(9)
        self.init_qvel = np.zeros(9)

    def _check_single_action(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.shape != (6,):
            raise ValueError(f"Action dimension mismatch. Expected (6,), found {a.shape}")
        return a

    def _host_init_state(self, batch_size):
        sigma = float(self.init_noise_params)
        if batch_size == 1:  # same row-slicing quirk as the other MuJoCo bodies (mujoco_env.py:243-244)
            e = [np.random.randn(1, 1) for _ in range(18)]
            pos = self.init_qpos[None, :] + e[0] * sigma
            vel = self.init_qvel[None, :] + e[1] * sigma
            return np.concatenate([pos, vel], axis=1)
        return self.np_random.standard_normal((batch_size, 18)) * sigma

    def get_batch_init_state(self, batch_size):
        s = self._host_init_state(batch_size)
        return s[:, :9], s[:, 9:]

    def transform_state_to_obs(self, batch_state):
        pos, vel = batch_state
        return np.concatenate([pos, vel], axis=1)
