"""Pin the CPU oracle against fixtures recorded from the REAL reference
(oracle/gen_golden.py) and against the reference's own known-answer vectors."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import build as obuild
from oracle.tag_continuous_np import TagContinuousOracle
from oracle.tag_gridworld_np import TagGridWorldOracle

f32 = np.float32


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return d, json.loads(str(d["config"])) if "config" in d.files else None


# ---------------------------------------------------------------- TagGridWorld
def test_gridworld_kat(golden_dir):
    """The reference's own KATs: tests/example_envs/pycuda_tests/
    test_tag_gridworld_step_python.py:32-463 (rewards/obs at 1e-5, done exact)."""
    d = np.load(os.path.join(golden_dir, "gw_kat.npz"))
    meta = json.loads(str(d["meta"]))
    assert len(meta) >= 2
    for ci, case in enumerate(meta):
        kw = dict(case["kwargs"])
        orc = TagGridWorldOracle(num_envs=1, **kw)
        for si in range(case["n_steps"]):
            p = f"c{ci}_s{si}_"
            obs, rew, done = orc.step(d[p + "actions"][None])
            g = kw["grid_length"]
            assert np.abs(rew[0] - d[p + "kat_rewards"]).max() < 1e-5
            assert np.abs(obs[0] * g - d[p + "kat_obs_x_grid"]).max() < 1e-5
            assert bool(done[0]) == bool(d[p + "kat_done"])
            # and bit-exact against what the reference returns
            np.testing.assert_array_equal(rew[0], d[p + "ref_rewards"])
            np.testing.assert_array_equal(obs[0], d[p + "ref_obs"])
            np.testing.assert_array_equal(orc.loc_x[0], d[p + "ref_loc_x"])
            np.testing.assert_array_equal(orc.loc_y[0], d[p + "ref_loc_y"])


@pytest.mark.parametrize("tag", ["full", "partial", "g6", "g10"])
def test_gridworld_trajectory(golden_dir, tag):
    d, cfg = _load(golden_dir, f"gw_traj_{tag}.npz")
    cfg.pop("seed", None)
    E = d["actions"].shape[1]
    orc = TagGridWorldOracle(num_envs=E, **cfg)
    np.testing.assert_array_equal(orc.obs, d["obs_at_reset"])
    for t in range(d["actions"].shape[0]):
        obs, rew, done = orc.step(d["actions"][t])
        np.testing.assert_array_equal(orc.loc_x, d["loc_x"][t])
        np.testing.assert_array_equal(orc.loc_y, d["loc_y"][t])
        np.testing.assert_array_equal(orc.timestep, d["timestep"][t])
        np.testing.assert_array_equal(done.astype(bool), d["done"][t])
        np.testing.assert_array_equal(rew, d["rewards"][t])
        np.testing.assert_array_equal(obs, d["obs"][t])
        orc.reset_done_envs()


# --------------------------------------------------------------- TagContinuous
TC_TAGS = ["test1", "test2", "test3", "test4", "tagheavy", "bench5x100", "bench5x100_full"]


@pytest.mark.parametrize("tag", TC_TAGS)
def test_tag_continuous_trajectory(golden_dir, tag):
    """Free-running (no re-sync) bit-exact replay of the reference CPU env."""
    d, cfg = _load(golden_dir, f"tc_traj_{tag}.npz")
    E = d["actions"].shape[1]
    orc = TagContinuousOracle(num_envs=E, **cfg)
    np.testing.assert_array_equal(orc.agent_types, d["agent_types"])
    np.testing.assert_array_equal(orc.start_x, d["start_x"].astype(f32))
    np.testing.assert_array_equal(orc.start_dir, d["start_dir"].astype(f32))
    np.testing.assert_array_equal(orc.acceleration_actions, d["acceleration_actions"])
    np.testing.assert_array_equal(orc.turn_actions, d["turn_actions"])
    np.testing.assert_array_equal(orc.skill_levels, d["skill_levels"])
    np.testing.assert_array_equal(orc.step_rewards, d["step_rewards"])
    assert orc.distance_margin_for_reward == d["distance_margin_for_reward"]
    np.testing.assert_array_equal(orc.obs, d["obs_at_reset"])
    for t in range(d["actions"].shape[0]):
        obs, rew, done = orc.step(d["actions"][t])
        for k, attr in (("loc_x", "loc_x"), ("loc_y", "loc_y"), ("speed", "speed"),
                        ("direction", "direction"), ("acceleration", "acceleration"),
                        ("still_in_the_game", "sig"), ("edge_hit_reward_penalty", "edge_pen"),
                        ("num_runners", "num_runners"), ("timestep", "timestep")):
            np.testing.assert_array_equal(getattr(orc, attr), d[k][t], err_msg=f"{k} t={t}")
        np.testing.assert_array_equal(done.astype(bool), d["done"][t], err_msg=f"done t={t}")
        np.testing.assert_array_equal(rew.astype(np.float64), d["rewards"][t], err_msg=f"rew t={t}")
        np.testing.assert_array_equal(obs, d["obs"][t], err_msg=f"obs t={t}")
        orc.reset_done_envs()


# ------------------------------------------------------------ C restatement
def _clib():
    lib = ctypes.CDLL(obuild.build())
    for fn in (lib.wdo_np_cosf, lib.wdo_np_sinf, lib.wdo_powf2):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        fn.restype = None
    return lib


def test_c_sincos_bit_identical_to_numpy():
    """np_sincosf() in oracle/csrc/wd_oracle.c (and the identical device routine)
    must reproduce numpy's float32 cos/sin bit for bit on [0, 2pi]."""
    lib = _clib()
    rng = np.random.RandomState(0)
    x = np.concatenate([
        (rng.rand(2_000_000) * 2 * np.pi).astype(f32),
        np.array([0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi], dtype=f32),
        np.linspace(0, 2 * np.pi, 100_001).astype(f32),
    ])
    out = np.empty_like(x)
    lib.wdo_np_cosf(x.ctypes.data, out.ctypes.data, x.size)
    np.testing.assert_array_equal(out.view(np.uint32), np.cos(x).view(np.uint32))
    lib.wdo_np_sinf(x.ctypes.data, out.ctypes.data, x.size)
    np.testing.assert_array_equal(out.view(np.uint32), np.sin(x).view(np.uint32))


def test_powf2_is_numpy_scalar_power():
    lib = _clib()
    rng = np.random.RandomState(1)
    x = (rng.rand(20000) * 40 - 20).astype(f32)
    out = np.empty_like(x)
    lib.wdo_powf2(x.ctypes.data, out.ctypes.data, x.size)
    ref = np.array([v ** 2 for v in x], dtype=f32)
    np.testing.assert_array_equal(out, ref)
