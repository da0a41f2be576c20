"""Minimal stand-in for `gym` (test infrastructure only).

`gym` is not installed in this image; the reference's CPU envs import only
`gym.spaces.{Discrete,MultiDiscrete,Box,Dict}` (reference
example_envs/tag_continuous/tag_continuous.py:11,
example_envs/tag_gridworld/tag_gridworld.py:8,
warp_drive/utils/recursive_obs_dict_to_spaces_dict.py:8).  This shim exists so
`oracle/gen_golden.py` can import the *real* reference from /root/reference in
the build container and record its outputs as golden fixtures.  It is never
imported by the product package.
"""
from . import spaces  # noqa: F401
