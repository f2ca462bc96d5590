"""Name -> spec table of the env ids on the env-step path (reference: emei/envs/register_env.py).

gym's registry is not a dependency; ``make(id, **kwargs)`` builds the env with the TimeLimit the
reference registers (``max_episode_steps``), realised on the device as the `truncated` bit.
"""
from .cartpole import CartPoleBalancingEnv, CartPoleSwingUpEnv
from .half_cheetah import HalfCheetahRunningEnv
from .hopper import HopperRunningEnv
from .inverted_double_pendulum import (
    BoundaryInvertedDoublePendulumBalancingEnv,
    BoundaryInvertedDoublePendulumSwingUpEnv,
    ReboundInvertedDoublePendulumBalancingEnv,
    ReboundInvertedDoublePendulumSwingUpEnv,
)
from .inverted_pendulum import (
    BoundaryInvertedPendulumBalancingEnv,
    BoundaryInvertedPendulumSwingUpEnv,
    ReboundInvertedPendulumBalancingEnv,
    ReboundInvertedPendulumSwingUpEnv,
)

# id: (class, max_episode_steps)   register_env.py:14-23, 47-86, 102-106
REGISTRY = {
    "CartPoleBalancing-v0": (CartPoleBalancingEnv, 500),
    "CartPoleSwingUp-v0": (CartPoleSwingUpEnv, 1000),
    "ReboundInvertedPendulumSwingUp-v0": (ReboundInvertedPendulumSwingUpEnv, 1000),
    "ReboundInvertedPendulumBalancing-v0": (ReboundInvertedPendulumBalancingEnv, 1000),
    "BoundaryInvertedPendulumSwingUp-v0": (BoundaryInvertedPendulumSwingUpEnv, 1000),
    "BoundaryInvertedPendulumBalancing-v0": (BoundaryInvertedPendulumBalancingEnv, 1000),
    "ReboundInvertedDoublePendulumSwingUp-v0": (ReboundInvertedDoublePendulumSwingUpEnv, 1000),
    "ReboundInvertedDoublePendulumBalancing-v0": (ReboundInvertedDoublePendulumBalancingEnv, 1000),
    "BoundaryInvertedDoublePendulumSwingUp-v0": (BoundaryInvertedDoublePendulumSwingUpEnv, 1000),
    "BoundaryInvertedDoublePendulumBalancing-v0": (BoundaryInvertedDoublePendulumBalancingEnv, 1000),
    "HalfCheetahRunning-v0": (HalfCheetahRunningEnv, 1000),
    "HopperRunning-v0": (HopperRunningEnv, 1000),  # register_env.py:87-91
}


def spec(env_id):
    if env_id not in REGISTRY:
        raise KeyError(f"No registered env with id: {env_id}")
    cls, steps = REGISTRY[env_id]
    return {"id": env_id, "entry_point": cls, "max_episode_steps": steps}


def make(env_id, **kwargs):
    """gym.make(id, freq_rate=, real_time_scale=, integrator=, ...) (zoo/conf/task/*.yaml:9-14)."""
    s = spec(env_id)
    kwargs.setdefault("max_episode_steps", s["max_episode_steps"])
    return s["entry_point"](**kwargs)
