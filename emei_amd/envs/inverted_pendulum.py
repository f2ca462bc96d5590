"""InvertedPendulum envs on the HIP engine (reference: emei/envs/mujoco/inverted_pendulum.py on
emei/envs/mujoco/mujoco_env.py).  The dynamics are MuJoCo's 2-DoF cart/pole model in closed form
with emei's forward-Euler position override; parity with libmujoco itself is unpinned."""
import numpy as np

from .. import spaces
from .base import HipEnv


class BaseInvertedPendulumEnv(HipEnv):
    """inverted_pendulum.py:12-49."""

    ENGINE_NAME = "BoundaryInvertedPendulumBalancing"
    SWINGUP = False

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator="euler",
                 init_noise_params=5e-3, obs_noise_params=0.0, **kwargs):
        if integrator != "euler":
            # mujoco_env.py:71-79 also knows "semi_implicit_euler" and "rk4"; only the forward-Euler
            # path is in scope here
            raise NotImplementedError(f"integrator {integrator!r}: only 'euler' is implemented on the HIP engine")
        if obs_noise_params != 0:
            raise NotImplementedError("obs_noise_params != 0 (mujoco_env.py:98-104) is not implemented")
        if not isinstance(init_noise_params, (int, float)):
            raise NotImplementedError("tuple/dict init_noise_params (mujoco_env.py:218-227) are not implemented")
        self.init_noise_params = init_noise_params
        self.obs_noise_params = obs_noise_params
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator,
                         init_noise=float(init_noise_params), **kwargs)
        self.observation_space = spaces.Box(low=-np.inf, high=np.inf, shape=(4,), dtype=np.float64)
        self.action_space = spaces.Box(low=-3.0, high=3.0, shape=(1,), dtype=np.float32)  # ctrlrange, xml:23
        self.init_qpos = np.zeros(2)
        self.init_qvel = np.zeros(2)
        # x theta v omega action -> x theta v omega   (inverted_pendulum.py:39-41)
        self._transition_graph = np.array([[0, 0, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0], [0, 1, 1, 1], [0, 0, 1, 1]])
        self._reward_mech_graph = None
        self._termination_graph = None
        self.jnt_range = np.array([[-2.0, 2.0], [-np.inf, np.inf] if self.SWINGUP else [-np.pi / 2, np.pi / 2]])

    def _check_single_action(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.shape != (1,):
            raise ValueError(f"Action dimension mismatch. Expected (1,), found {a.shape}")  # gym do_simulation
        return a

    def _host_init_state(self, batch_size):
        """mujoco_env.py:137-140,197-249.  For batch_size == 1 the reference's row slicing adds ONE
        sigma*N(0,1) draw (global numpy stream) to every qpos entry and a second one to every qvel
        entry, and consumes two more draws for the second joint; that is reproduced.  For
        batch_size > 1 the reference raises ValueError (non-broadcastable); the vectorised form draws
        per-coordinate i.i.d. noise from the env's seeded generator instead."""
        sigma = float(self.init_noise_params)
        if batch_size == 1:
            e = [np.random.randn(1, 1) for _ in range(4)]  # pos j0, vel j0, pos j1 (dropped), vel j1 (dropped)
            pos = self.init_qpos[None, :] + e[0] * sigma
            vel = self.init_qvel[None, :] + e[1] * sigma
            return np.concatenate([pos, vel], axis=1)
        return np.concatenate([np.tile(self.init_qpos, (batch_size, 1)), np.tile(self.init_qvel, (batch_size, 1))], axis=1) \
            + self.np_random.standard_normal((batch_size, 4)) * sigma

    def _state_to_obs_np(self, state):
        obs = state.copy()
        obs[:, 1] = (obs[:, 1] + np.pi) % (2 * np.pi) - np.pi  # inverted_pendulum.py:45-49
        return obs

    def get_batch_init_state(self, batch_size):
        s = self._host_init_state(batch_size)
        return s[:, :2], s[:, 2:]  # (pos, vel), mujoco_env.py:137-140

    def transform_state_to_obs(self, batch_state):
        pos, vel = batch_state
        return np.concatenate([pos, vel], axis=1)  # mujoco_env.py:142-144


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
