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
TC_TAGS = ["test1", "test2", "test3", "test4", "tagheavy", "bench5x100", "bench5x100_full", "bench5x100_ep", "big5x250", "big5x1000"]


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
    # (the 1005-agent fixture stores the observations as float32 -- the cast every device comparison applies to the
    # reference's float64 rows anyway -- to stay small: compare in the fixture's dtype)
    as_stored = (lambda a: a) if d["obs"].dtype == np.float64 else (lambda a: a.astype(d["obs"].dtype))
    np.testing.assert_array_equal(as_stored(orc.obs), d["obs_at_reset"])
    for t in range(d["actions"].shape[0]):
        obs, rew, done = orc.step(d["actions"][t])
        for k, attr in (("loc_x", "loc_x"), ("loc_y", "loc_y"), ("speed", "speed"),
                        ("direction", "direction"), ("acceleration", "acceleration"),
                        ("still_in_the_game", "sig"), ("edge_hit_reward_penalty", "edge_pen"),
                        ("num_runners", "num_runners"), ("timestep", "timestep")):
            np.testing.assert_array_equal(getattr(orc, attr), d[k][t], err_msg=f"{k} t={t}")
        np.testing.assert_array_equal(done.astype(bool), d["done"][t], err_msg=f"done t={t}")
        np.testing.assert_array_equal(rew.astype(np.float64), d["rewards"][t], err_msg=f"rew t={t}")
        np.testing.assert_array_equal(as_stored(obs), d["obs"][t], err_msg=f"obs t={t}")
        orc.reset_done_envs()


# ------------------------------------------------------------ C restatement
def _clib():
    lib = ctypes.CDLL(obuild.build())
    for fn in (lib.wdo_np_cosf, lib.wdo_np_sinf, lib.wdo_powf2):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        fn.restype = None
    return lib


def _tie_neighbourhoods():
    """+-200 ulps around (k + 0.5) * pi/2 (quadrant rounding ties, e.g. 5*pi/4) and k * pi/2"""
    out = []
    for k in range(9):
        for c in ((k + 0.5) * np.pi / 2, k * np.pi / 2):
            bits = np.array([c], dtype=f32).view(np.int32)[0] + np.arange(-200, 201)
            out.append(bits.astype(np.int32).view(f32))
    x = np.concatenate(out)
    return x[(x >= 0) & (x <= 7)]


def test_c_sincos_bit_identical_to_numpy():
    """np_sincosf() in oracle/csrc/wd_oracle.c (and the identical device routine)
    must reproduce numpy's float32 cos/sin bit for bit on [0, 2pi]."""
    lib = _clib()
    rng = np.random.RandomState(0)
    x = np.concatenate([
        (rng.rand(2_000_000) * 2 * np.pi).astype(f32),
        np.array([0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi], dtype=f32),
        np.linspace(0, 2 * np.pi, 100_001).astype(f32),
        _tie_neighbourhoods(),
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


@pytest.mark.parametrize("tag", ["test2", "test3", "tagheavy", "bench5x100", "bench5x100_full", "big5x250", "big5x1000"])
def test_c_step_matches_reference(golden_dir, tag):
    """The C restatement (bench.py's cpu_baseline 'port') replays the reference bit-exactly."""
    d, cfg = _load(golden_dir, f"tc_traj_{tag}.npz")
    E = d["actions"].shape[1]
    orc = TagContinuousOracle(num_envs=E, **cfg)  # supplies the seeded start + tables
    lib = ctypes.CDLL(obuild.build())
    lib.wdo_tc_step.restype = None

    class Cfg(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("E", "N", "T", "K", "full", "exits")] + \
                   [(n, ctypes.c_float) for n in ("L", "vmax", "edge", "margin", "tr", "tp", "er")] + \
                   [("na", ctypes.c_int), ("nt", ctypes.c_int)]

    c = Cfg(E, orc.N, orc.T, orc.K, int(orc.use_full_observation), int(orc.runner_exits),
            float(orc.grid_length), float(orc.max_speed), float(orc.edge_hit_penalty),
            float(orc.distance_margin_for_reward), float(orc.tag_reward_for_tagger),
            float(orc.tag_penalty_for_runner), float(orc.end_of_game_reward_for_runner),
            len(orc.acceleration_actions), len(orc.turn_actions))
    st = {k: np.ascontiguousarray(getattr(orc, k)).copy() for k in
          ("loc_x", "loc_y", "speed", "direction", "acceleration", "edge_pen", "sig", "num_runners",
           "timestep", "done")}
    obs = np.zeros((E, orc.N, orc.obs_dim), f32)
    rew = np.zeros((E, orc.N), f32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for t in range(min(d["actions"].shape[0], 120)):
        acts = np.ascontiguousarray(d["actions"][t], dtype=np.int32)
        lib.wdo_tc_step(ctypes.byref(c), P(st["loc_x"]), P(st["loc_y"]), P(st["speed"]), P(st["direction"]),
                        P(st["acceleration"]), P(orc.agent_types), P(st["edge_pen"]),
                        P(orc.acceleration_actions), P(orc.turn_actions), P(orc.skill_levels), P(st["sig"]),
                        P(obs), P(acts), P(rew), P(orc.step_rewards), P(st["num_runners"]), P(st["done"]),
                        P(st["timestep"]), ctypes.c_int(2))
        for k, g in (("loc_x", "loc_x"), ("loc_y", "loc_y"), ("speed", "speed"), ("direction", "direction"),
                     ("acceleration", "acceleration"), ("sig", "still_in_the_game"), ("num_runners", "num_runners")):
            np.testing.assert_array_equal(st[k], d[g][t], err_msg=f"{k} t={t}")
        np.testing.assert_array_equal(st["done"].astype(bool), d["done"][t])
        np.testing.assert_array_equal(rew, d["rewards"][t].astype(f32))
        np.testing.assert_array_equal(obs, d["obs"][t].astype(f32), err_msg=f"obs t={t}")
        m = st["done"] > 0  # device-style reset of finished replicas
        for k, v in (("loc_x", orc.start_x), ("loc_y", orc.start_y), ("direction", orc.start_dir)):
            st[k][m] = v
        for k, v in (("speed", 0), ("acceleration", 0), ("edge_pen", 0), ("sig", 1),
                     ("num_runners", orc.num_runners0), ("timestep", 0), ("done", 0)):
            st[k][m] = v


@pytest.mark.parametrize("tag", ["test3", "tagheavy", "bench5x100", "big5x250", "big5x1000"])
def test_c_oracle_neighbour_ids_match_the_numpy_oracle(golden_dir, tag):
    """`TagContinuousCOracle.nearest_ids` (the checker of the device's nearest_neighbor_ids output in the
    full-size fused-tick tests) against the numpy oracle's k_nearest_neighbors restatement
    (tag_continuous.py:422-444: stable (distance, id) order, -1 padding) on the reference's recorded action
    streams -- both oracles replay the reference bit-exactly, so their ids must be identical too."""
    from oracle.tag_continuous_c import TagContinuousCOracle

    d, cfg = _load(golden_dir, f"tc_traj_{tag}.npz")
    E = d["actions"].shape[1]
    onp = TagContinuousOracle(num_envs=E, **cfg)
    oc = TagContinuousCOracle(E, n_threads=2, **cfg)
    padded = 0
    for t in range(min(d["actions"].shape[0], 60)):
        onp.step(d["actions"][t])
        oc.step(d["actions"][t])
        np.testing.assert_array_equal(oc.obs, onp.obs.astype(f32))
        in_game = oc.sig_before > 0
        np.testing.assert_array_equal(oc.nearest_ids[in_game], onp.nearest_ids[in_game], err_msg=f"t={t}")
        assert (oc.nearest_ids[~in_game] == -1).all()
        padded += int((oc.nearest_ids[in_game] < 0).sum())
        onp.reset_done_envs()
        oc.reset_done_envs()
    assert padded > 0 or tag != "tagheavy"  # tagheavy: rows with fewer than K others in the game do occur


def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32-10 (kat_vectors) pin the
    generator the device sampler and its CPU restatement share."""
    from oracle.core_np import philox4x32_10, u01_open_closed

    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = philox4x32_10(*[np.uint32(c) for c in ctr], *key)
        assert tuple(int(g) for g in got) == want
    u = u01_open_closed(np.array([0, 0xFFFFFFFF, 0x80000000], dtype=np.uint32))
    assert u.dtype == np.float32 and u[0] == np.float32(2.0 ** -24) and u[1] == 1.0 and u[2] == np.float32(0.5) + np.float32(2.0 ** -24)


def test_single_head_tick_draw_is_word_epoch_mod_4_of_the_block():
    """single-head fused ticks draw word (epoch & 3) of the Philox block (row, epoch >> 2, tag, 4): four consecutive
    epochs of a row share one block (csrc/kernels/wd_common.h::wd_tick_draw)"""
    from oracle.core_np import philox4x32_10, single_head_tick_uniform, u01_open_closed

    rows, tag, k0, k1 = 37, 0x1234567, 0xDEADBEEF, 0x42
    for e0 in (0, 5, 4094, 0xFFFFFFFC):
        for d in range(4):
            ep = np.full(rows, (e0 & ~3) + d, dtype=np.uint32)
            words = philox4x32_10(np.arange(rows, dtype=np.uint32), ep >> np.uint32(2), np.uint32(tag), np.uint32(4), k0, k1)
            np.testing.assert_array_equal(single_head_tick_uniform(rows, ep, k0, k1, tag), u01_open_closed(words[d]))
    mixed = np.arange(rows, dtype=np.uint32) * 3 + 1   # different epochs per row
    u = single_head_tick_uniform(rows, mixed, k0, k1, tag)
    for r in (0, 7, 36):
        w = philox4x32_10(np.uint32(r), mixed[r] >> np.uint32(2), np.uint32(tag), np.uint32(4), k0, k1)
        assert u[r] == u01_open_closed(w[int(mixed[r]) & 3])


CP_TOL = 1e-5  # absolute, on positions / velocities / angles of O(1)


def test_cartpole_oracle_vs_reference_kernel_source(golden_dir):
    """tests/golden/cp_traj.npz = the reference's cartpole_step_numba.py:5-83 executed under a
    numba.cuda stand-in (oracle/gen_golden.py::gen_cartpole_traj).  The oracle restates Numba's dtype
    flow (cosf/sinf, float64 promotion through the 4.0/3.0 literal), the stand-in runs the same source with
    Python/numpy scalars -- so floats agree to rounding (observed 1.9e-6 free-running over whole
    episodes), and timesteps, terminations and rewards agree exactly."""
    from oracle.cartpole_np import CartPoleOracle

    g = np.load(os.path.join(golden_dir, "cp_traj.npz"))
    ticks, E = g["actions"].shape[:2]
    orc = CartPoleOracle(E, int(g["episode_length"]), initial_state=g["initial_state"])
    worst = 0.0
    for t in range(ticks):
        orc.step(g["actions"][t])
        worst = max(worst, float(np.abs(orc.state - g["state"][t][:, 0]).max()))
        np.testing.assert_allclose(orc.state, g["state"][t][:, 0], rtol=0, atol=CP_TOL, err_msg=f"t={t}")
        np.testing.assert_allclose(orc.obs, g["obs"][t][:, 0], rtol=0, atol=CP_TOL)
        np.testing.assert_array_equal(orc.done, g["done"][t])
        np.testing.assert_array_equal(orc.timestep, g["timestep"][t])
        np.testing.assert_array_equal(orc.rewards, g["rewards"][t][:, 0])
        orc.reset_done_envs()
    assert int(g["done"].sum()) > 200 and int((g["timestep"] == int(g["episode_length"])).sum()) > 0
    assert worst < CP_TOL
