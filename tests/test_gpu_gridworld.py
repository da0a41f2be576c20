"""TagGridWorld on the MI355X vs the reference: KATs, recorded trajectories, oracle at
BASELINE config[1] size.  Integer state / done: bit-exact.  Observations: bit-exact.
Rewards: <= 1 ulp (the reference API narrows the reward scalars to float32 before they
reach any device, data_manager.py:348-351, while its CPU step adds them in float64)."""
import json
import os

import numpy as np
import pytest

from oracle.tag_gridworld_np import TagGridWorldOracle

pytestmark = pytest.mark.gpu


def _mk(cfg, E):
    from tests.hip_harness import make_wrapper, require_gpu
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld

    require_gpu()
    return make_wrapper(CUDATagGridWorld(**cfg), E)


def _check_step(w, orc, rew_ref=None, tag=""):
    from tests.hip_harness import OBS, REW, pull, ulp_diff

    np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep, err_msg=tag)
    np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32), err_msg=tag)
    assert ulp_diff(pull(w, REW), orc.rewards.astype(np.float32)).max() <= 1, tag


def test_gridworld_kat(golden_dir):
    """reference tests/example_envs/pycuda_tests/test_tag_gridworld_step_cuda.py restated:
    the KAT vectors driven through the device kernel, batched as independent replicas."""
    from tests.hip_harness import OBS, REW, pull, push_actions

    d = np.load(os.path.join(golden_dir, "gw_kat.npz"))
    meta = json.loads(str(d["meta"]))
    for ci, case in enumerate(meta):
        kw = dict(case["kwargs"])
        kw["starting_location_x"] = np.array(kw["starting_location_x"])
        kw["starting_location_y"] = np.array(kw["starting_location_y"])
        w = _mk(kw, 3)
        for si in range(case["n_steps"]):
            p = f"c{ci}_s{si}_"
            push_actions(w, np.tile(d[p + "actions"], (3, 1)))
            w.step_all_envs()
            for e in range(3):
                assert np.abs(pull(w, REW)[e] - d[p + "kat_rewards"]).max() < 1e-5
                assert np.abs(pull(w, OBS)[e] * kw["grid_length"] - d[p + "kat_obs_x_grid"]).max() < 1e-5
                assert bool(pull(w, "_done_")[e]) == bool(d[p + "kat_done"])
                np.testing.assert_array_equal(pull(w, "loc_x")[e], d[p + "ref_loc_x"])
                np.testing.assert_array_equal(pull(w, OBS)[e], d[p + "ref_obs"].astype(np.float32))


@pytest.mark.parametrize("tag", ["full", "partial", "g6", "g10"])
def test_gridworld_golden_trajectory(golden_dir, tag):
    from tests.hip_harness import OBS, REW, pull, push_actions, ulp_diff

    d = np.load(os.path.join(golden_dir, f"gw_traj_{tag}.npz"))
    cfg = json.loads(str(d["config"]))
    E = d["actions"].shape[1]
    w = _mk(cfg, E)
    np.testing.assert_array_equal(pull(w, OBS), d["obs_at_reset"].astype(np.float32))
    for t in range(d["actions"].shape[0]):
        push_actions(w, d["actions"][t])
        w.step_all_envs()
        np.testing.assert_array_equal(pull(w, "loc_x"), d["loc_x"][t])
        np.testing.assert_array_equal(pull(w, "loc_y"), d["loc_y"][t])
        np.testing.assert_array_equal(pull(w, "_done_").astype(bool), d["done"][t])
        np.testing.assert_array_equal(pull(w, OBS), d["obs"][t].astype(np.float32))
        assert ulp_diff(pull(w, REW), d["rewards"][t].astype(np.float32)).max() <= 1
        w.reset_only_done_envs()
        assert pull(w, "_done_").sum() == 0


@pytest.mark.parametrize("full_obs", [True, False])
def test_gridworld_config1_vs_oracle(full_obs):
    """BASELINE config[1]: 10x10, 5 agents, num_envs=1000, 2+ episodes incl. resets."""
    from tests.hip_harness import push_actions

    cfg = dict(num_taggers=4, grid_length=10, episode_length=100, seed=27, wall_hit_penalty=0.1,
               tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01,
               use_full_observation=full_obs)
    E = 1000
    w = _mk(cfg, E)
    ocfg = dict(cfg)
    ocfg.pop("seed")
    orc = TagGridWorldOracle(num_envs=E, **ocfg)
    rng = np.random.RandomState(1234)
    for t in range(220):
        a = rng.randint(0, 5, size=(E, 5)).astype(np.int32)
        push_actions(w, a)
        w.step_all_envs()
        orc.step(a)
        _check_step(w, orc, tag=f"t={t}")
        w.reset_only_done_envs()
        orc.reset_done_envs()
    _check_step(w, orc, tag="after final reset")


def test_gridworld_ragged_sizes():
    """replica counts that do not fill the last packed block; other agent counts"""
    from tests.hip_harness import push_actions

    for E, taggers in ((1, 4), (13, 1), (257, 7), (50, 63)):
        cfg = dict(num_taggers=taggers, grid_length=5, episode_length=7, use_full_observation=True)
        w = _mk(cfg, E)
        orc = TagGridWorldOracle(num_envs=E, **cfg)
        rng = np.random.RandomState(E)
        for t in range(16):
            a = rng.randint(0, 5, size=(E, taggers + 1)).astype(np.int32)
            push_actions(w, a)
            w.step_all_envs()
            orc.step(a)
            _check_step(w, orc, tag=f"E={E} N={taggers + 1} t={t}")
            w.reset_only_done_envs()
            orc.reset_done_envs()
