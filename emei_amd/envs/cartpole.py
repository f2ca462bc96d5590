"""CartPole envs on the HIP engine (reference: emei/envs/classic_control/cartpole.py, base_control.py)."""
import math

import numpy as np

from .. import spaces
from .base import HipEnv


class BaseCartPoleEnv(HipEnv):
    """cartpole.py:16-60: constants, spaces.  Abstract like the reference: reset() raises
    NotImplementedError because get_batch_init_state is not defined (test_cartpole.py:4-11)."""

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator: str = "euler", ode_method: str = "euler",
                 **kwargs):
        """integrator: accepted and IGNORED, as in the reference (base_control.py:73 calls ODE_approximation without `method`,
        so every classic-control env steps with forward Euler whatever this says).
        ode_method (not a reference kwarg): the `method` argument of ODE_approximation itself — "rk4" runs the function's other
        branch (base_control.py:165-170, float32 k-stages) on the device; an explicit opt-in, "euler" = what step() does."""
        if ode_method not in ("euler", "rk4"):
            raise NotImplementedError("approximation method `{}` is not suppoerted yet.".format(ode_method))  # base_control.py:172
        self.ODE_METHOD = ode_method
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator, **kwargs)
        self.gravity = 9.8
        self.mass_cart = 1.0
        self.mass_pole = 0.1
        self.total_mass = self.mass_pole + self.mass_cart
        self.length = 0.5  # actually half the pole's length
        self.force_mag = 10.0
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        high = np.array(
            [self.x_threshold * 2, np.finfo(np.float32).max, self.theta_threshold_radians * 2, np.finfo(np.float32).max],
            dtype=np.float32,
        )
        self.action_space = spaces.Discrete(2)
        self.observation_space = spaces.Box(-high, high, dtype=np.float32)

    def _check_single_action(self, action):
        if type(action) is int and 0 <= action < 2:  # the common call; everything else takes the reference's own checks
            return action
        if isinstance(action, int):  # base_control.py:62-63
            action = np.asarray(action)
        assert self.action_space.contains(action), f"{action!r} ({type(action)}) invalid"  # base_control.py:65-66
        return np.asarray(action, dtype=np.int64)

    def _extract_action(self, action):
        return self.force_mag if action == 1 else -self.force_mag  # cartpole.py:121-122,142-143

    def _host_init_state(self, batch_size):
        raise NotImplementedError  # core.py:175-177


class CartPoleBalancingEnv(BaseCartPoleEnv):
    """cartpole.py:115-132."""

    ENGINE_NAME = "CartPoleBalancing"

    def _host_init_state(self, batch_size):
        return self.np_random.uniform(low=-0.05, high=0.05, size=(batch_size, 4))  # :131-132

    def get_batch_reward(self, obs, pre_obs=None, action=None, state=None, pre_state=None):
        import torch

        if isinstance(obs, torch.Tensor):
            return torch.ones((obs.shape[0], 1), dtype=obs.dtype, device=obs.device)
        return np.ones([obs.shape[0], 1])  # :128-129, a constant: no kernel needed


class CartPoleSwingUpEnv(BaseCartPoleEnv):
    """cartpole.py:135-156.  (The observation_space keeps the 2.4-based bound, as in the reference,
    because x_threshold is raised to 5 after the space is built, :35-46 vs :140.)"""

    ENGINE_NAME = "CartPoleSwingUp"

    def __init__(self, freq_rate: int = 1, real_time_scale: float = 0.02, integrator: str = "euler", **kwargs):
        super().__init__(freq_rate=freq_rate, real_time_scale=real_time_scale, integrator=integrator, **kwargs)  # ode_method: kwargs
        self.x_threshold = 5

    def _host_init_state(self, batch_size):
        init_state = self.np_random.uniform(low=-0.05, high=0.05, size=(batch_size, 4))  # :153-156
        init_state[:, 2] += np.pi
        return init_state
