"""Action / observation space descriptors.

The reference takes these from `gym.spaces`; gym is an optional dependency here
(absent in the MI355X image), so the four classes the rollout path needs are provided
locally and the real gym classes are used when importable."""
import numpy as np

try:  # pragma: no cover - gym is not part of the image
    from gym.spaces import Box, Dict, Discrete, MultiDiscrete  # noqa: F401
except Exception:

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(np.shape(low) if shape is None else shape)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def __repr__(self):
            return f"Box{self.shape}"

    class Dict(dict):
        def __init__(self, spaces=None):
            super().__init__(spaces or {})


def obs_dict_to_spaces(obs):
    """{agent_id: array | dict} -> Dict of Boxes
    (reference warp_drive/utils/recursive_obs_dict_to_spaces_dict.py:13-53)."""
    assert isinstance(obs, dict)
    out = {}
    for key, val in obs.items():
        if isinstance(val, dict):
            out[key] = obs_dict_to_spaces(val)
            continue
        arr = np.asarray(val)
        if arr.ndim == 0:
            arr = arr.reshape(1)
        bound = 1e20
        while not np.isfinite(np.array(bound, dtype=arr.dtype)):
            bound /= 2
        out[key] = Box(low=-bound, high=bound, shape=arr.shape, dtype=arr.dtype)
    return Dict(out)
