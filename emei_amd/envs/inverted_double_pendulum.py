"""InvertedDoublePendulum envs on the HIP engine (reference:
emei/envs/mujoco/inverted_double_pendulum.py on emei/envs/mujoco/mujoco_env.py).  Dynamics: MuJoCo's
3-DoF cart + two poles in closed form with emei's forward-Euler position override; parity with
libmujoco is unpinned.  Reference quirks kept: the observation "wrap"
``(theta + pi) % 2 * pi - pi`` (:59) and the declared observation_space shape (4,) although the
observation has 6 entries (:29)."""
import numpy as np

from .. import spaces
from .base import MujocoHipEnv


class BaseInvertedDoublePendulumEnv(MujocoHipEnv):
    ENGINE_NAME = "BoundaryInvertedDoublePendulumBalancing"
    NQ = 3

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator="euler",
                 init_noise_params=5e-3, obs_noise_params=0.0, **kwargs):
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise_params=init_noise_params, obs_noise_params=obs_noise_params, **kwargs)
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(4,), dtype=np.float64)  # sic, :29
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)  # ctrlrange, xml:45
        # the reference stores this under `_causal_graph`, so get_transition_graph() finds None (:42)
        self._causal_graph = np.array([[0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1], [1, 0, 0, 0, 0, 0],
                                       [0, 1, 0, 1, 1, 1], [0, 0, 1, 1, 1, 1], [0, 0, 0, 1, 1, 1]])
        self.jnt_range = np.array([[-3.0, 3.0], [0.0, 0.0], [0.0, 0.0]])

    def _state_to_obs_np(self, state):
        obs = state.copy()
        obs[:, 1:3] = (obs[:, 1:3] + np.pi) % 2 * np.pi - np.pi  # :59, operator precedence as in the reference
        return obs


class ReboundInvertedDoublePendulumBalancingEnv(BaseInvertedDoublePendulumEnv):
    ENGINE_NAME = "ReboundInvertedDoublePendulumBalancing"  # :63-90


class BoundaryInvertedDoublePendulumBalancingEnv(BaseInvertedDoublePendulumEnv):
    ENGINE_NAME = "BoundaryInvertedDoublePendulumBalancing"  # :93-123


class ReboundInvertedDoublePendulumSwingUpEnv(BaseInvertedDoublePendulumEnv):
    ENGINE_NAME = "ReboundInvertedDoublePendulumSwingUp"  # :126-153


class BoundaryInvertedDoublePendulumSwingUpEnv(BaseInvertedDoublePendulumEnv):
    ENGINE_NAME = "BoundaryInvertedDoublePendulumSwingUp"  # :156-196
