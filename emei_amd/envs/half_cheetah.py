"""HalfCheetah-style running env on the HIP engine (reference: emei/envs/mujoco/half_cheetah.py)."""
import numpy as np

from .. import spaces
from .base import MujocoHipEnv


class HalfCheetahRunningEnv(MujocoHipEnv):
    ENGINE_NAME = "HalfCheetahRunning"
    NQ = 9

    def __init__(self, freq_rate: int = 4, real_time_scale: float = 0.002, integrator="euler",
                 forward_reward_weight=1.0, ctrl_cost_weight=0.1, init_noise_params=0.1, obs_noise_params=0.0, **kwargs):
        self._forward_reward_weight = forward_reward_weight
        self._ctrl_cost_weight = ctrl_cost_weight
        params = {}  # only what differs from half_cheetah.py:23-24 travels to the engine
        if forward_reward_weight != 1.0:
            params["forward_reward_weight"] = forward_reward_weight
        if ctrl_cost_weight != 0.1:
            params["ctrl_cost_weight"] = ctrl_cost_weight
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise_params=init_noise_params, obs_noise_params=obs_noise_params,
                         engine_env_params=params, **kwargs)
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(18,), dtype=np.float64)
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(6,), dtype=np.float32)
