"""CPU oracle for the rollout hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package restates, in numpy (batched over env replicas) and in plain C
(`oracle/csrc/wd_oracle.c`), the algorithm of the reference's *CPU* `step()` for
the path BASELINE.json names: TagGridWorld, TagContinuous, the categorical /
OU-Gaussian action sampler, reset-when-done and the Cartpole Euler step.  Every
function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything from here, and only as the checker / the reported CPU baseline.
The product package (`warp_drive_amd`) never imports it: its HIP path fails
loudly when the HIP library is missing instead of falling back to this code.

Pinning (see oracle/gen_golden.py, tests/golden/*.npz, tests/test_oracle_golden.py):
  * TagGridWorld  -- pinned: reference's own known-answer vectors
    (tests/example_envs/pycuda_tests/test_tag_gridworld_step_python.py) plus
    trajectories recorded from the real reference CPU env.  Bit-exact.
  * TagContinuous -- pinned: the reference has no KATs for it, so the pin is
    trajectories recorded from the real reference CPU env in this container
    (numpy 2.2.6, x86-64 AVX2/AVX512+FMA).  Bit-exact on state, obs, rewards, done.
  * Sampler / OU  -- the reference pins statistics only (its RNG streams, curand
    XORWOW and numba xoroshiro128p, are third-party and absent); the inverse-CDF
    search is restated from random.cu:33-85 and checked for exact one-hot /
    zero-probability behaviour.  Bitwise RNG parity: unpinned by design.
  * Cartpole      -- parity unpinned: the reference's CPU step is third-party
    `gym.envs.classic_control.CartPoleEnv` (absent, unpinned version); the oracle
    restates the reference's own Numba kernel cartpole_step_numba.py:5-83.
"""
