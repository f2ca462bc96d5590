"""InvertedPendulum envs on the HIP engine (reference: emei/envs/mujoco/inverted_pendulum.py on
emei/envs/mujoco/mujoco_env.py).  The dynamics are MuJoCo's 2-DoF cart/pole model in closed form
stepped with the integrator the reference selects (mujoco_env.py:70-79; default: MuJoCo's Euler
velocity update + emei's forward-Euler position override); parity with libmujoco itself is unpinned
(DESIGN.md)."""
import numpy as np

from .. import spaces
from .base import MujocoHipEnv


class BaseInvertedPendulumEnv(MujocoHipEnv):
    """inverted_pendulum.py:12-49."""

    ENGINE_NAME = "BoundaryInvertedPendulumBalancing"
    SWINGUP = False
    NQ = 2

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator="euler",
                 init_noise_params=5e-3, obs_noise_params=0.0, **kwargs):
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise_params=init_noise_params, obs_noise_params=obs_noise_params, **kwargs)
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(4,), dtype=np.float64)
        self.action_space = spaces.Box(low=-3.0, high=3.0, shape=(1,), dtype=np.float32)  # ctrlrange, xml:23
        # x theta v omega action -> x theta v omega   (inverted_pendulum.py:39-41)
        self._transition_graph = np.array([[0, 0, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0], [0, 1, 1, 1], [0, 0, 1, 1]])
        self._reward_mech_graph = None
        self._termination_graph = None
        self.jnt_range = np.array([[-2.0, 2.0], [-np.inf, np.inf] if self.SWINGUP else [-np.pi / 2, np.pi / 2]])

    def _state_to_obs_np(self, state):
        obs = state.copy()
        obs[:, 1] = (obs[:, 1] + np.pi) % (2 * np.pi) - np.pi  # inverted_pendulum.py:45-49
        return obs


class ReboundInvertedPendulumBalancingEnv(BaseInvertedPendulumEnv):
    ENGINE_NAME = "ReboundInvertedPendulumBalancing"  # inverted_pendulum.py:52-79


class BoundaryInvertedPendulumBalancingEnv(BaseInvertedPendulumEnv):
    ENGINE_NAME = "BoundaryInvertedPendulumBalancing"  # :82-111


class ReboundInvertedPendulumSwingUpEnv(BaseInvertedPendulumEnv):
    ENGINE_NAME = "ReboundInvertedPendulumSwingUp"  # :114-146
    SWINGUP = True


class BoundaryInvertedPendulumSwingUpEnv(BaseInvertedPendulumEnv):
    ENGINE_NAME = "BoundaryInvertedPendulumSwingUp"  # :149-183
    SWINGUP = True
