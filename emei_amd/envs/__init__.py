from .cartpole import BaseCartPoleEnv, CartPoleBalancingEnv, CartPoleSwingUpEnv
from .half_cheetah import HalfCheetahRunningEnv
from .hopper import HopperRunningEnv
from .inverted_double_pendulum import (
    BaseInvertedDoublePendulumEnv,
    BoundaryInvertedDoublePendulumBalancingEnv,
    BoundaryInvertedDoublePendulumSwingUpEnv,
    ReboundInvertedDoublePendulumBalancingEnv,
    ReboundInvertedDoublePendulumSwingUpEnv,
)
from .inverted_pendulum import (
    BaseInvertedPendulumEnv,
    BoundaryInvertedPendulumBalancingEnv,
    BoundaryInvertedPendulumSwingUpEnv,
    ReboundInvertedPendulumBalancingEnv,
    ReboundInvertedPendulumSwingUpEnv,
)
from .registration import REGISTRY, make, spec
