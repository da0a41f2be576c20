"""Four-class stand-in for gym.spaces (see package docstring)."""
import numpy as np


class Space:
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)

    def __repr__(self):
        return f"Discrete({self.n})"


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)

    def __repr__(self):
        return f"MultiDiscrete({self.nvec.tolist()})"


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def __repr__(self):
        return f"Box{self.shape}"


class Dict(Space, dict):
    def __init__(self, spaces=None):
        dict.__init__(self, spaces or {})
