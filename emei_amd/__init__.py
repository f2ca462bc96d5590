"""emei_amd — MI355X-native vectorised env-step engine behind emei's EmeiEnv surface.

Importing the package does not load the HIP library; the first engine call does, and raises if the
library or a GPU is missing (there is no CPU path).
"""
__version__ = "0.1.0"

from .core import EmeiEnv, Freezable, OfflineEnv  # noqa: F401
from .envs import (  # noqa: F401
    BaseCartPoleEnv,
    BaseInvertedDoublePendulumEnv,
    BaseInvertedPendulumEnv,
    BoundaryInvertedDoublePendulumBalancingEnv,
    BoundaryInvertedDoublePendulumSwingUpEnv,
    ReboundInvertedDoublePendulumBalancingEnv,
    ReboundInvertedDoublePendulumSwingUpEnv,
    BoundaryInvertedPendulumBalancingEnv,
    BoundaryInvertedPendulumSwingUpEnv,
    CartPoleBalancingEnv,
    CartPoleSwingUpEnv,
    HalfCheetahRunningEnv,
    HopperRunningEnv,
    ReboundInvertedPendulumBalancingEnv,
    ReboundInvertedPendulumSwingUpEnv,
    make,
    spec,
)
