"""Minimal stand-ins for gym.spaces (gym is not a dependency): only what the env surface needs."""
import numpy as np


class Space:
    def __init__(self, shape=(), dtype=None):
        self.shape, self.dtype = tuple(shape), dtype
        self._np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)
        return [seed]


class Discrete(Space):
    """gym.spaces.Discrete(n): integers {0..n-1}; shape ()."""

    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def contains(self, x):
        if isinstance(x, (int, np.integer)) and not isinstance(x, bool):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return 0 <= v < self.n

    def sample(self):
        return int(self.np_random.integers(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class Box(Space):
    """gym.spaces.Box(low, high, shape, dtype)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
        if shape is None:
            shape = np.broadcast(low, high).shape
        super().__init__(shape, np.dtype(dtype))
        self.low = np.broadcast_to(low, self.shape).copy()
        self.high = np.broadcast_to(high, self.shape).copy()

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self.np_random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
