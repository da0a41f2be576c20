"""TagContinuous on the MI355X vs the reference CPU step.

State (loc_x, loc_y, speed, direction, acceleration, still_in_the_game, num_runners),
rewards, done: BIT-EXACT, free-running over whole episodes incl. resets (north_star asks
1e-5; the device restates numpy's float32 cos/sin, so there is no drift to tolerate).
Observations: bit-exact, with ONE known, rare deviation that the tests classify and bound instead of hiding: the reference
squares the coordinate differences with numpy SCALAR power, i.e. libm's powf(x, 2) (tag_continuous.py:403-420), the kernels
with x * x.  powf is not correctly rounded; where it differs from x * x in the last bit AND that bit decides which of two
candidates' float32 distances is smaller (or whether they tie and the lower id goes first), the order of two neighbours in a
row can differ from the reference's.  Seen: 0 of 218 M rows at 5 x 100 (two whole episodes of 2000 replicas), 1 of 20.9 M
rows at 5 x 1000 (scripts/soak_parity.py, round 6; the row's two candidates had equal float32 distances from x * x and
distances one ulp apart from powf).  Every mismatching row must be such a near-tie (<= 2 ulp) and their number is bounded;
the suite's own inputs produce none (0 of 23.4 M rows).
"""
import json
import os

import numpy as np
import pytest

from oracle.tag_continuous_np import TagContinuousOracle

pytestmark = pytest.mark.gpu

STATE = (("loc_x", "loc_x"), ("loc_y", "loc_y"), ("speed", "speed"), ("direction", "direction"),
         ("acceleration", "acceleration"), ("still_in_the_game", "sig"),
         ("edge_hit_reward_penalty", "edge_pen"), ("num_runners", "num_runners"), ("_timestep_", "timestep"))


from tests.hip_harness import NEAR_TIE_TOTALS as _TOTALS  # noqa: E402


def _mk(cfg, E):
    from tests.hip_harness import make_wrapper, require_gpu
    from warp_drive_amd.envs.tag_continuous import TagContinuous

    require_gpu()
    return make_wrapper(TagContinuous(**cfg), E)


def _near_tie_rows(orc, bad_rows):
    """Every mismatching (env, agent) row must have two candidate distances within 2 ulp
    among its K+1 nearest -- otherwise it is a real bug."""
    from tests.hip_harness import ulp_diff

    d = orc.neighbor_dist  # [E, N, N] float32, inf = invalid (set by the oracle's knn())
    K = orc.K
    for e, i in bad_rows:
        row = np.sort(d[e, i][np.isfinite(d[e, i])])[: K + 1]
        if len(row) < 2 or ulp_diff(row[1:], row[:-1]).min() > 2:
            return False
    return True


def _compare(w, orc, tag, stats):
    from tests.hip_harness import OBS, REW, pull

    for name, attr in STATE:
        np.testing.assert_array_equal(pull(w, name), getattr(orc, attr), err_msg=f"{name} {tag}")
    np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=f"done {tag}")
    np.testing.assert_array_equal(pull(w, REW), orc.rewards, err_msg=f"rewards {tag}")
    obs_dev, obs_ref = pull(w, OBS), orc.obs.astype(np.float32)
    if not np.array_equal(obs_dev, obs_ref):
        bad = np.argwhere((obs_dev != obs_ref).any(axis=2))
        assert not orc.use_full_observation, f"full-obs mismatch {tag}: {bad[:5]}"
        assert _near_tie_rows(orc, bad), f"obs mismatch that is not a near-tie {tag}: {bad[:5]}"
        stats["near_tie_rows"] += len(bad)
        _TOTALS["near_tie_rows"] += len(bad)
    stats["rows"] += obs_ref.shape[0] * obs_ref.shape[1]
    _TOTALS["rows"] += obs_ref.shape[0] * obs_ref.shape[1]
    if not orc.use_full_observation and orc.K <= orc.N - 1:
        # nearest_neighbor_ids [E, N, K]: the reference order (distance, then id; -1 = fewer than K others in
        # the game) for every agent that was in the game when the observation was generated
        ids_dev = pull(w, "nearest_neighbor_ids").reshape(orc.nearest_ids.shape)
        m = (obs_ref[..., -1] != 0) & (obs_dev == obs_ref).all(axis=2)
        np.testing.assert_array_equal(ids_dev[m], orc.nearest_ids[m], err_msg=f"nearest_neighbor_ids {tag}")


def _check_ids_after_fused_tick(w, ids_ref, in_game, fin, obs_dev, obs_ref, tag):
    """`nearest_neighbor_ids` [E, N, K] after a fused tick, for every replica that did not finish on it (a
    finished one was reset in the launch: its array holds the reset copy): rows of agents in the game when
    the observation was generated carry the reference's (distance, id) order with -1 padding
    (tag_continuous.py:422-444), rows of agents out of it are all -1.  Rows whose observation differs (counted
    near-ties) are left to the near-tie check."""
    from tests.hip_harness import pull

    E, N, K = ids_ref.shape
    ids_dev = pull(w, "nearest_neighbor_ids").reshape(E, N, K)
    rows_ok = (obs_dev == obs_ref).all(axis=2) & ~fin[:, None]
    m = rows_ok & in_game
    np.testing.assert_array_equal(ids_dev[m], ids_ref[m], err_msg=f"nearest_neighbor_ids {tag}")
    out = rows_ok & ~in_game
    assert (ids_dev[out] == -1).all(), f"nearest_neighbor_ids of agents out of the game {tag}"
    return int(m.sum()), int((ids_ref[m] < 0).sum())


def _run_lockstep(cfg, E, ticks, seed, stats=None):
    from tests.hip_harness import OBS, pull, push_actions

    stats = stats if stats is not None else {"near_tie_rows": 0, "rows": 0}
    w = _mk(cfg, E)
    orc = TagContinuousOracle(num_envs=E, **cfg)
    np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32))
    rng = np.random.RandomState(seed)
    na, nt = len(orc.acceleration_actions), len(orc.turn_actions)
    for t in range(ticks):
        a = np.stack([rng.randint(0, na, size=(E, orc.N)), rng.randint(0, nt, size=(E, orc.N))], axis=2)
        push_actions(w, a)
        w.step_all_envs()
        orc.step(a)
        _compare(w, orc, f"t={t}", stats)
        w.reset_only_done_envs()
        orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "_done_"), 0)
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x)
        np.testing.assert_array_equal(pull(w, "still_in_the_game"), orc.sig)
        if orc.timestep.min() == 0:  # some replica was reset: its observation must be the reset one
            m = orc.timestep == 0
            np.testing.assert_array_equal(pull(w, OBS)[m], orc.obs.astype(np.float32)[m])
    assert stats["near_tie_rows"] <= max(2, stats["rows"] // 100000), stats
    return stats


TC_TAGS = ["test1", "test2", "test3", "test4", "tagheavy", "bench5x100", "bench5x100_full", "bench5x100_ep",
           "big5x250", "big5x1000"]  # replicas of 255 / 1005 agents recorded from the reference (the `_N512` / `_N1024`
                                     # entries, prefiltered search)


@pytest.mark.parametrize("tag", TC_TAGS)
def test_tag_continuous_golden_trajectory(golden_dir, tag):
    """Free-running replay of trajectories recorded from the REAL reference CPU env
    (incl. resets), compared with the recorded outputs -- no oracle in the loop."""
    from tests.hip_harness import OBS, REW, pull, push_actions

    d = np.load(os.path.join(golden_dir, f"tc_traj_{tag}.npz"))
    cfg = json.loads(str(d["config"]))
    E = d["actions"].shape[1]
    w = _mk(cfg, E)
    np.testing.assert_array_equal(pull(w, OBS), d["obs_at_reset"].astype(np.float32))
    mismatched_rows = 0
    for t in range(d["actions"].shape[0]):
        push_actions(w, d["actions"][t])
        w.step_all_envs()
        for k in ("loc_x", "loc_y", "speed", "direction", "acceleration", "still_in_the_game",
                  "edge_hit_reward_penalty", "num_runners"):
            np.testing.assert_array_equal(pull(w, k), d[k][t], err_msg=f"{k} t={t}")
        np.testing.assert_array_equal(pull(w, "_timestep_"), d["timestep"][t])
        np.testing.assert_array_equal(pull(w, "_done_").astype(bool), d["done"][t])
        np.testing.assert_array_equal(pull(w, REW), d["rewards"][t].astype(np.float32), err_msg=f"rew t={t}")
        mismatched_rows += int((pull(w, OBS) != d["obs"][t].astype(np.float32)).any(axis=2).sum())
        w.reset_only_done_envs()
    assert mismatched_rows == 0, f"{mismatched_rows} observation rows differ from the reference"


@pytest.mark.parametrize("full_obs", [False, True])
def test_small_configs_vs_oracle(full_obs):
    cfg = dict(num_taggers=3, num_runners=10, grid_length=8.0, episode_length=25, seed=11,
               max_acceleration=0.3, min_acceleration=-0.3, max_turn=1.5, min_turn=-1.5,
               num_acceleration_levels=6, num_turn_levels=6, edge_hit_penalty=-0.25,
               use_full_observation=full_obs, num_other_agents_observed=5, tagging_distance=0.12,
               tag_reward_for_tagger=3.0, tag_penalty_for_runner=-2.0, step_penalty_for_tagger=-0.05,
               step_reward_for_runner=0.05, end_of_game_reward_for_runner=1.5,
               runner_exits_game_after_tagged=True)
    _run_lockstep(cfg, E=37, ticks=80, seed=5)


@pytest.mark.parametrize("K", [1, 3, 7, 13, 20, 40])
def test_k_specialisations_and_generic_k(K):
    """register-resident top-K kernels (K <= 32) and the generic-K kernel (K = 40)"""
    cfg = dict(num_taggers=4, num_runners=44, grid_length=12.0, episode_length=15, seed=3,
               max_acceleration=0.2, min_acceleration=-0.2, num_acceleration_levels=4, num_turn_levels=4,
               use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.05,
               runner_exits_game_after_tagged=True)
    _run_lockstep(cfg, E=9, ticks=20, seed=K)


def test_few_survivors_padding():
    """most runners get tagged out: neighbour lists shorter than K are zero padded, agents
    out of the game observe zeros, num_runners == 0 ends the episode"""
    cfg = dict(num_taggers=6, num_runners=6, grid_length=3.0, episode_length=60, seed=2, max_speed=0.5,
               max_acceleration=0.3, min_acceleration=-0.3, num_acceleration_levels=4, num_turn_levels=4,
               use_full_observation=False, num_other_agents_observed=8, tagging_distance=0.3,
               runner_exits_game_after_tagged=True)
    stats = _run_lockstep(cfg, E=16, ticks=90, seed=9)
    assert stats["rows"] > 0


BENCH_CFG = dict(num_taggers=5, num_runners=100, grid_length=20.0, episode_length=500, max_acceleration=0.1,
                 min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20,
                 num_turn_levels=20, skill_level_runner=1.0, skill_level_tagger=1.0, max_speed=1.0, seed=274880,
                 use_full_observation=False, num_other_agents_observed=10, tagging_distance=0.02,
                 tag_reward_for_tagger=10.0, tag_penalty_for_runner=-10.0, step_penalty_for_tagger=-0.0,
                 step_reward_for_runner=0.0, edge_hit_penalty=-0.0, end_of_game_reward_for_runner=1.0,
                 runner_exits_game_after_tagged=True)


def test_bench_shape_vs_oracle():
    """BASELINE config[2] shape (5 taggers x 100 runners, K = 10) at a size the oracle steps in seconds"""
    _run_lockstep(dict(BENCH_CFG, episode_length=12), E=64, ticks=30, seed=77)


def test_bench_shape_soak_vs_oracle():
    """3 million observation rows at the BASELINE shape (5 x 100, K = 10), free-running through
    episode ends and restarts, every tick compared with the oracle: state, rewards, done and
    observations bit-exact (a mismatching row would have to be a verified <= 2 ulp near-tie; the
    session total is printed at the end of the run)."""
    _run_lockstep(dict(BENCH_CFG, episode_length=60), E=192, ticks=150, seed=2024)


def test_bench_full_size_properties():
    """num_envs = 2000 (BASELINE config[2]): size-independent properties.
    Replicas receive identical actions in groups of 8 => identical outputs inside a group;
    positions stay inside the arena; observation rows of agents in the game end with t/T."""
    from tests.hip_harness import OBS, REW, pull, push_actions

    E, G = 2000, 8
    w = _mk(BENCH_CFG, E)
    rng = np.random.RandomState(0)
    for t in range(1, 6):
        a = np.stack([rng.randint(0, 21, size=(E // G, 105)), rng.randint(0, 21, size=(E // G, 105))], axis=2)
        push_actions(w, np.repeat(a, G, axis=0))
        w.step_all_envs()
        for name in ("loc_x", "loc_y", "speed", REW, OBS, "still_in_the_game"):
            v = pull(w, name)
            v = v.reshape((E // G, G) + v.shape[1:])
            assert (v == v[:, :1]).all(), f"{name}: replicas with identical inputs diverged at t={t}"
        x, y = pull(w, "loc_x"), pull(w, "loc_y")
        assert x.min() >= 0 and x.max() <= 20 and y.min() >= 0 and y.max() <= 20
        obs, sig = pull(w, OBS), pull(w, "still_in_the_game")
        np.testing.assert_array_equal(pull(w, "_timestep_"), t)
        # sig is post-tag; rows of agents still in the game must carry the time feature
        assert np.all(obs[..., -1][sig == 1] == np.float32(t / 500))
    # first 64 replicas against the oracle, same action stream
    orc = TagContinuousOracle(num_envs=8, **BENCH_CFG)
    rng = np.random.RandomState(0)
    for t in range(5):
        a = np.stack([rng.randint(0, 21, size=(E // G, 105)), rng.randint(0, 21, size=(E // G, 105))], axis=2)
        orc.step(a[:8])
    np.testing.assert_array_equal(pull(w, "loc_x")[: 8 * G: G], orc.loc_x)
    np.testing.assert_array_equal(pull(w, REW)[: 8 * G: G], orc.rewards)
    np.testing.assert_array_equal(pull(w, OBS)[: 8 * G: G], orc.obs.astype(np.float32))


@pytest.mark.parametrize("full_obs,acc_levels,turn_levels,runners,K,E", [
    (False, 6, 8, 30, 6, 23), (True, 6, 8, 30, 6, 23),
    (False, 30, 40, 17, 5, 9),     # probability rows longer than the register path (31 / 41 actions)
    (False, 1, 1, 100, 10, 5),     # two actions per head, the benchmark's agent count
    (False, 3, 3, 130, 4, 3),      # more than 128 agents: three wavefronts per replica, 9-bit search keys
    (True, 2, 5, 60, 3, 4),        # full observations whose width is a multiple of four (16-byte stores)
    (False, 70, 3, 10, 3, 2),      # action table larger than its LDS copy (71 > 64 entries)
    (False, 6, 8, 400, 6, 3),      # more than 256 agents: the two heads are sampled one after the other from ONE slab
    (False, 9, 5, 300, 6, 3),      # ... heads of unequal size, the first the larger (wavefronts' rows of the two heads overlap)
    (False, 20, 20, 1000, 10, 2),  # 1004 agents: sixteen wavefronts per replica, 10-bit search keys (`_N1024` entry)
])
def test_fused_tick_kernel(full_obs, acc_levels, turn_levels, runners, K, E):
    """HipTagContinuousTick: sampling + step + in-kernel reset in ONE launch.  The actions it
    sampled are pulled back and replayed through the oracle; finished replicas must already
    be reset when the launch returns while `_done_` still reports them for the trainer."""
    import torch
    from tests.hip_harness import OBS, REW, make_wrapper, pull, require_gpu
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders
    from warp_drive_amd.env_wrapper import EnvWrapper

    require_gpu()
    n_taggers = 5 if (full_obs and runners == 60) else 4   # 65 agents -> 64 neighbour slots
    cfg = dict(num_taggers=n_taggers, num_runners=runners, grid_length=6.0, episode_length=11, seed=5, max_speed=0.6,
               max_acceleration=0.3, min_acceleration=-0.3, num_acceleration_levels=acc_levels,
               num_turn_levels=turn_levels, use_full_observation=full_obs, num_other_agents_observed=K,
               tagging_distance=0.12, edge_hit_penalty=-0.2, step_reward_for_runner=0.01,
               runner_exits_game_after_tagged=True)
    na, nt = acc_levels + 1, turn_levels + 1
    w = EnvWrapper(env_obj=TagContinuous(**cfg), num_envs=E, env_backend="hip")
    w.reset_all_envs()
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=11)
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                      push_data_batch_placeholders=False)
    rng = np.random.RandomState(0)
    N = w.n_agents
    probs = [torch.from_numpy(rng.dirichlet(np.ones(a), size=(E, N)).astype(np.float32)).cuda() for a in (na, nt)]
    engine = RolloutEngine(w, sampler, probabilities=probs)
    assert engine.fused and engine.step_kernel_name.startswith("HipTagContinuousTick")
    orc = TagContinuousOracle(num_envs=E, **cfg)
    stats = {"near_tie_rows": 0, "rows": 0}
    counts = [np.zeros(na), np.zeros(nt)]
    finished_total = 0
    from oracle.core_np import fused_tick_uniforms, sample_actions_counting
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import _stream_tag

    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    probs_host = [p.cpu().numpy() for p in probs]
    for t in range(40):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        assert (rng_words[4:] == t).all()  # one epoch per tick and agent row
        engine.run(1)
        torch.cuda.synchronize()
        a = pull(w, "sampled_actions")
        # draw-for-draw: the kernel's Philox uniforms restated on the CPU -> identical indices
        u0, u1 = fused_tick_uniforms(E * N, rng_words[4:], rng_words[0], rng_words[1], _stream_tag("tick"))
        np.testing.assert_array_equal(a[..., 0], sample_actions_counting(probs_host[0], u0.reshape(E, N)))
        np.testing.assert_array_equal(a[..., 1], sample_actions_counting(probs_host[1], u1.reshape(E, N)))
        assert a[..., 0].max() < na and a[..., 1].max() < nt and a.min() >= 0
        counts[0] += np.bincount(a[..., 0].reshape(-1), minlength=na)
        counts[1] += np.bincount(a[..., 1].reshape(-1), minlength=nt)
        orc.step(a)
        np.testing.assert_array_equal(pull(w, REW), orc.rewards, err_msg=f"rewards t={t}")
        np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=f"done t={t}")  # still set
        fin = orc.done > 0
        finished_total += int(fin.sum())
        obs_before_reset = orc.obs.astype(np.float32).copy()
        ids_ref = None if full_obs else orc.nearest_ids.copy()
        in_game = obs_before_reset[..., -1] != 0   # time column: set for agents in the game at observation time
        orc.reset_done_envs()
        for name, attr in STATE:
            np.testing.assert_array_equal(pull(w, name), getattr(orc, attr), err_msg=f"{name} t={t}")
        obs_dev = pull(w, OBS)
        np.testing.assert_array_equal(obs_dev[fin], orc.obs.astype(np.float32)[fin])      # reset observation
        if ids_ref is not None and K <= N - 1:
            _check_ids_after_fused_tick(w, ids_ref, in_game, fin, obs_dev, obs_before_reset, f"t={t}")
        live = ~fin
        if not np.array_equal(obs_dev[live], obs_before_reset[live]):
            assert not full_obs
            stats["near_tie_rows"] += int((obs_dev[live] != obs_before_reset[live]).any(axis=2).sum())
    assert finished_total >= 2 * E and stats["near_tie_rows"] <= 1
    for c, p in zip(counts, probs):
        expected = p.cpu().numpy().reshape(-1, c.size).sum(0) * 40
        assert np.abs(c - expected).max() < 6 * np.sqrt(expected.max())


def _near_tie_rows_c(orc, bad_rows):
    """as _near_tie_rows, for the C oracle (distances rebuilt for the mismatching rows only)"""
    from tests.hip_harness import ulp_diff

    for e, i in bad_rows:
        d = orc.neighbor_distances(int(e), int(i))
        row = np.sort(d[np.isfinite(d)])[: orc.K + 1]
        if len(row) < 2 or ulp_diff(row[1:], row[:-1]).min() > 2:
            return False
    return True


HEADLINE_TICK = "HipTagContinuousTick_K10_N105A21"   # BASELINE shape, sizes folded (warp_drive_amd/build.py UNITS)


@pytest.fixture
def runtime_size_entries(monkeypatch):
    """the `_K<k>` entries with runtime sizes instead of the shape-specialised one"""
    from warp_drive_amd.envs.tag_continuous import TagContinuous

    monkeypatch.setattr(TagContinuous, "SHAPE_ENTRIES", False)


def test_runtime_size_entry_at_the_headline_shape(runtime_size_entries):
    """`HipTagContinuousTick_K10` (runtime sizes; what every 65 .. 128-agent shape other than BASELINE's runs) at the
    BASELINE shape, 300 replicas x 40 ticks of a 15-tick episode against the C oracle"""
    _fused_ticks_vs_c_oracle(dict(BENCH_CFG, episode_length=15), 300, 40, 5, kernel="HipTagContinuousTick_K10")


@pytest.mark.parametrize("full_obs,E,ticks", [(False, 2000, 44), (True, 64, 36)])
def test_headline_fused_tick_full_size(full_obs, E, ticks):
    """The (kernel, shape) pair bench.py reports: the fused HipTagContinuousTick_K10_N105A21 at BASELINE
    configs[2] -- 5 taggers + 100 runners, 20 + 20 levels (21-way heads), K = 10, num_envs = 2000 --
    through RolloutEngine, EVERY replica compared with the C oracle on every tick: the sampled actions
    draw for draw (Philox restated on the host, random.cu:51-85), then state / observations / rewards /
    done and the post-reset state (reset.cu:9-75; 15-tick episodes, so every replica ends and restarts
    at least twice).  Same for the full-observation variant (generic tick kernel) at N = 105.
    Reference: example_envs/tag_continuous/tag_continuous.py:796-887."""
    import torch
    from oracle.core_np import fused_tick_uniforms, sample_actions_counting
    from oracle.tag_continuous_c import TagContinuousCOracle
    from tests.hip_harness import OBS, REW, pull, require_gpu
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    require_gpu()
    cfg = dict(BENCH_CFG, episode_length=15, use_full_observation=full_obs)
    w = EnvWrapper(env_obj=TagContinuous(**cfg), num_envs=E, env_backend="hip")
    w.reset_all_envs()
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=274880)
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                      push_data_batch_placeholders=False)
    N = w.n_agents
    assert N == 105
    rng = np.random.RandomState(3)
    probs_host = [rng.dirichlet(np.ones(21), size=(E, N)).astype(np.float32) for _ in range(2)]
    probs = [torch.from_numpy(p).cuda() for p in probs_host]
    engine = RolloutEngine(w, sampler, probabilities=probs)
    assert engine.fused
    assert engine.step_kernel_name == ("HipTagContinuousTick" if full_obs else HEADLINE_TICK)
    orc = TagContinuousCOracle(E, n_threads=min(32, os.cpu_count() or 1), **cfg)
    np.testing.assert_array_equal(pull(w, OBS), orc.obs)
    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    near_tie = rows = finished_total = id_rows = 0
    restarts = np.zeros(E, dtype=np.int64)
    for t in range(ticks):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        assert (rng_words[4:] == t).all()
        engine.run(1)
        torch.cuda.synchronize()
        a = pull(w, "sampled_actions")
        u0, u1 = fused_tick_uniforms(E * N, rng_words[4:], rng_words[0], rng_words[1], _stream_tag("tick"))
        np.testing.assert_array_equal(a[..., 0], sample_actions_counting(probs_host[0], u0.reshape(E, N)))
        np.testing.assert_array_equal(a[..., 1], sample_actions_counting(probs_host[1], u1.reshape(E, N)))
        orc.step(a)
        np.testing.assert_array_equal(pull(w, REW), orc.rewards, err_msg=f"rewards t={t}")
        np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=f"done t={t}")  # still set for the trainer
        fin = orc.done > 0
        finished_total += int(fin.sum())
        restarts += fin
        obs_before_reset = orc.obs[~fin].copy()
        obs_all_before_reset = orc.obs.copy()
        orc.reset_done_envs()
        for name, attr in STATE:
            np.testing.assert_array_equal(pull(w, name), getattr(orc, attr), err_msg=f"{name} t={t}")
        obs_dev = pull(w, OBS)
        np.testing.assert_array_equal(obs_dev[fin], orc.obs[fin], err_msg=f"reset observation t={t}")
        if not full_obs:
            n_id, n_pad = _check_ids_after_fused_tick(w, orc.nearest_ids, orc.sig_before > 0, fin, obs_dev,
                                                      obs_all_before_reset, f"t={t}")
            id_rows += n_id
        live = np.flatnonzero(~fin)
        if not np.array_equal(obs_dev[live], obs_before_reset):
            assert not full_obs, f"full-obs mismatch t={t}"
            bad = np.argwhere((obs_dev[live] != obs_before_reset).any(axis=2))
            bad[:, 0] = live[bad[:, 0]]
            assert _near_tie_rows_c(orc, bad), f"obs mismatch that is not a near-tie t={t}: {bad[:5]}"
            near_tie += len(bad)
        rows += E * N
    _TOTALS["near_tie_rows"] += near_tie
    _TOTALS["rows"] += rows
    assert restarts.min() >= 2 and finished_total >= 2 * E, (restarts.min(), finished_total)
    assert full_obs or id_rows > 0.9 * rows * 0.8, (id_rows, rows)  # nearly every in-game row's ids were compared
    assert near_tie <= max(2, rows // 100000), (near_tie, rows)


def _fused_ticks_vs_c_oracle(cfg, E, ticks, seed, kernel=HEADLINE_TICK, before_tick=None):
    """The fused tick (sample + step + reset in one launch) with the benchmark's uniform policy, every tick
    compared with the C oracle: sampled actions replayed, then state / observations / rewards / done /
    nearest_neighbor_ids and the post-reset state.  Returns (mean live agents per tick, id rows compared,
    padded id entries compared, rows)."""
    import torch
    from oracle.tag_continuous_c import TagContinuousCOracle
    from tests.hip_harness import OBS, REW, pull, require_gpu
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    require_gpu()
    w = EnvWrapper(env_obj=TagContinuous(**cfg), num_envs=E, env_backend="hip")
    w.reset_all_envs()
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=seed)
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                      push_data_batch_placeholders=False)
    engine = RolloutEngine(w, sampler)  # uniform probabilities: the benchmark's policy
    assert engine.step_kernel_name == kernel
    orc = TagContinuousCOracle(E, n_threads=min(16, os.cpu_count() or 1), **cfg)
    N = orc.N
    live_seen, near_tie, rows, id_rows, id_pads = [], 0, 0, 0, 0
    for t in range(ticks):
        if before_tick is not None:
            before_tick(t, w)
        engine.run(1)
        torch.cuda.synchronize()
        orc.step(pull(w, "sampled_actions"))
        np.testing.assert_array_equal(pull(w, REW), orc.rewards, err_msg=f"rewards t={t}")
        np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=f"done t={t}")
        fin = orc.done > 0
        obs_before_reset = orc.obs[~fin].copy()
        obs_all_before_reset = orc.obs.copy()
        live_seen.append(float(orc.sig_before.sum(axis=1).mean()))
        orc.reset_done_envs()
        for name, attr in STATE:
            np.testing.assert_array_equal(pull(w, name), getattr(orc, attr), err_msg=f"{name} t={t}")
        obs_dev = pull(w, OBS)
        np.testing.assert_array_equal(obs_dev[fin], orc.obs[fin])
        n_id, n_pad = _check_ids_after_fused_tick(w, orc.nearest_ids, orc.sig_before > 0, fin, obs_dev,
                                                  obs_all_before_reset, f"t={t}")
        id_rows += n_id
        id_pads += n_pad
        live = np.flatnonzero(~fin)
        if not np.array_equal(obs_dev[live], obs_before_reset):
            bad = np.argwhere((obs_dev[live] != obs_before_reset).any(axis=2))
            bad[:, 0] = live[bad[:, 0]]
            assert _near_tie_rows_c(orc, bad), f"obs mismatch that is not a near-tie t={t}: {bad[:5]}"
            near_tie += len(bad)
        rows += E * N
    _TOTALS["near_tie_rows"] += near_tie
    _TOTALS["rows"] += rows
    assert near_tie <= max(2, rows // 5000000), near_tie
    _fused_ticks_vs_c_oracle.last_near_tie_rows = near_tie  # (scripts/soak_parity.py reports it)
    return live_seen, id_rows, id_pads, rows


def test_whole_episode_at_the_headline_shape():
    """The neighbour search runs over the agents still in the game, packed: 105 of them at the start of an
    episode, fewer than 64 (one wavefront of searchers, the second one skips the search) from about tick 150
    on, ~27 at tick 500 under a uniform random policy -- and 105 again after the restart.  The fused tick at
    the BASELINE shape runs a whole 500-tick episode plus the first 60 ticks of the next one with 48
    replicas; every tick is compared with the C oracle (actions replayed, state / observations / rewards /
    done / nearest ids exact) and the live-agent count must really sweep the range."""
    live_seen, id_rows, _, rows = _fused_ticks_vs_c_oracle(dict(BENCH_CFG), 48, 560, 99)
    assert id_rows > 0.4 * rows, (id_rows, rows)  # the ids of every in-game row were compared (54 of 105 on average)
    # the episode did sweep the packed-search regimes: two wavefronts of searchers, then one, then 105 again
    assert max(live_seen[:20]) > 95 and min(live_seen[400:499]) < 45 and live_seen[505] > 95, (live_seen[:3], live_seen[480:510:5])


def test_arena_empties_below_K_agents_at_the_headline_shape():
    """N = 105 > 64 with FEWER THAN K + 1 agents left in the game: a tagging distance of a fifth of the arena
    empties it within ~60 ticks (the taggers never leave), so the packed search runs with fewer candidates
    than neighbour slots, rows are padded with -1 / zeros, most rows belong to agents out of the game, and
    replicas whose last runner is tagged finish early and restart inside the launch (tag_continuous.py:422-444,
    :880-883).  130 ticks of a 110-tick episode, every tick against the C oracle incl. nearest_neighbor_ids."""
    cfg = dict(BENCH_CFG, tagging_distance=0.2, episode_length=110)
    live_seen, id_rows, id_pads, rows = _fused_ticks_vs_c_oracle(cfg, 40, 130, 7)
    # (tick 110 is the first of the next episode: everybody is back; a fifth of the arena then empties it again fast)
    assert min(live_seen[60:110]) < 14 and live_seen[110] > 95, (live_seen[60:110:10], live_seen[108:113])
    assert id_pads > 500, id_pads  # rows with fewer than K others in the game were compared


@pytest.mark.parametrize("runners,taggers,tagging_distance,kernel", [(600, 10, 0.14, "HipTagContinuousTick_K10_N1024"),
                                                                     (300, 6, 0.12, "HipTagContinuousTick_K10_N512")])
def test_prefiltered_search_of_big_replicas(runners, taggers, tagging_distance, kernel):
    """Replicas of more than 128 agents search their neighbours inside a radius derived from the previous tick's
    neighbours (`knn_prev`, tc_pre_pass1 / tc_pre_pass2) while at least 200 agents are in the game.  70 ticks of a 45-tick
    episode with a tagging distance that takes the arena from full to under 200 agents (prefilter on, then off) and
    back to full at the restart, every tick against the C oracle incl. nearest_neighbor_ids.  The hint is ONLY a
    hint: it is overwritten with random bits, with zeros (everybody remembers agents 0, 0, 0 ...), with one far
    agent and with "none" along the way -- the results must not move."""
    from tests.hip_harness import pull

    cfg = dict(BENCH_CFG, num_taggers=taggers, num_runners=runners, grid_length=30.0, tagging_distance=tagging_distance,
               episode_length=45)
    rng = np.random.default_rng(runners)
    hints = []
    N = runners + taggers

    def scribble(t, w):
        shape = w.cuda_data_manager.get_shape("knn_prev")
        if t > 0:
            hints.append(pull(w, "knn_prev").copy())
        if t == 3:
            _push_state(w, knn_prev=rng.integers(-2**31, 2**31, size=shape, dtype=np.int64).astype(np.int32))
        elif t == 6:
            _push_state(w, knn_prev=np.zeros(shape, np.int32))
        elif t == 9:
            _push_state(w, knn_prev=np.full(shape, (N - 1) * 0x10001, np.int32))
        elif t == 12:
            _push_state(w, knn_prev=np.full(shape, -1, np.int32))

    live_seen, id_rows, _, rows = _fused_ticks_vs_c_oracle(cfg, 3, 70, 11, kernel=kernel, before_tick=scribble)
    # the prefilter was on (at least 200 agents in the game) through the scribbles, then off, then on again after the restart
    assert live_seen[0] == N and live_seen[13] > 200 > min(live_seen[:45]) and live_seen[46] > 200, live_seen[:50:5]
    # the kernel did remember neighbours (valid 16-bit ids) while the prefilter was on
    h = hints[1].view(np.uint16).reshape(3, N, 16)
    assert ((h[..., :13] < N).sum(axis=-1) >= 12).mean() > 0.4, "knn_prev was not written"


def _push_state(w, **arrays):
    """overwrite device state arrays in place (any registered array, torch-accessible or not)"""
    import torch
    from warp_drive_amd.managers import hip_driver as drv

    dm = w.cuda_data_manager
    for name, v in arrays.items():
        v = np.ascontiguousarray(v, dtype=dm.get_dtype(name)).reshape(dm.get_shape(name))
        if dm.is_data_on_device_via_torch(name):
            dm.data_on_device_via_torch(name).copy_(torch.from_numpy(v))
        else:
            drv.memcpy_htod(dm.device_data(name), v)
        dm.invalidate_derived(name)  # (a write that bypasses the manager: bookkeeping derived from the array is void)
    torch.cuda.synchronize()


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 7])
def test_exact_and_sqrt_ties_at_the_cut(K):
    """Crafted positions: a lattice (many exactly equal distances, so the K-th nearest is tied with
    others and the reference's stable id order decides, tag_continuous.py:435-437) plus a pair whose
    squared distances to agent 0 differ by one ulp while their float32 sqrt is identical -- there the
    reference prefers the LOWER id although its squared distance is the larger one.  Agents do not
    move (zero acceleration / turn at zero speed), so every tick repeats the ties.  No near-tie
    allowance: these must match exactly."""
    from tests.hip_harness import OBS, pull, push_actions

    cfg = dict(num_taggers=2, num_runners=10, grid_length=20.0, episode_length=9, seed=1, max_acceleration=0.1,
               min_acceleration=-0.1, num_acceleration_levels=4, num_turn_levels=4, use_full_observation=False,
               num_other_agents_observed=K, tagging_distance=1e-5, runner_exits_game_after_tagged=True)
    E = 5
    w = _mk(cfg, E)
    orc = TagContinuousOracle(num_envs=E, **cfg)
    N = orc.N
    # agent 0 at (5,5); agent 1 at (6, 5.0003): d2 = 1 + 2^-23, sqrt -> 1.0; agent 2 at (6,5): d2 = 1, sqrt = 1.0
    xs = np.array([5, 6, 6, 4, 5, 5, 7, 3, 6, 4, 9, 9], dtype=np.float32)
    ys = np.array([5, 5.0003, 5, 5, 6, 4, 5, 5, 6, 4, 9, 10], dtype=np.float32)
    assert len(xs) == N
    d2_a = np.float32((xs[1] - xs[0]) ** 2) + np.float32((ys[1] - ys[0]) * (ys[1] - ys[0]))
    assert np.float32(d2_a) > np.float32(1.0) and np.sqrt(np.float32(d2_a), dtype=np.float32) == np.float32(1.0)
    state = dict(loc_x=np.tile(xs, (E, 1)), loc_y=np.tile(ys, (E, 1)), speed=np.zeros((E, N), np.float32),
                 direction=np.zeros((E, N), np.float32), acceleration=np.zeros((E, N), np.float32))
    orc.set_state(**state)
    _push_state(w, **state)
    a0 = int(np.argmin(np.abs(orc.acceleration_actions)))
    t0 = int(np.argmin(np.abs(orc.turn_actions)))
    assert orc.acceleration_actions[a0] == 0 and orc.turn_actions[t0] == 0
    act = np.zeros((E, N, 2), dtype=np.int32)
    act[..., 0], act[..., 1] = a0, t0
    for t in range(4):
        push_actions(w, act)
        w.step_all_envs()
        orc.step(act)
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x)
        np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y)
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32), err_msg=f"K={K} t={t}")
    if K == 1:  # the lower id wins the sqrt tie although it is farther in squared distance
        ids = orc.knn()
        assert ids[0, 0, 0] == 1


@pytest.mark.parametrize("K,n_runners", [(3, 10), (6, 10), (3, 180), (6, 180), (3, 600), (6, 600)])
def test_candidates_a_few_ulps_apart_at_the_cut(K, n_runners):
    """The one-pass search orders candidates by squared distance with the low 7 bits dropped (buckets of
    128 ulps; 9 bits = 512 ulps for replicas of more than 128 agents, 10 bits = 1024 ulps beyond 512) and must rebuild the reference's
    order exactly wherever that is too coarse.  One replica per case: agent 0's K-th and (K+1)-th
    candidates sit `delta` ulps of squared distance apart -- inside one bucket, across a bucket boundary,
    just inside / outside the 383-apart (1535-apart) rule -- with either id the closer one; with `triple` a
    third candidate shares the zone (the exact fallback takes over: the two-pass search up to 128
    candidates, the K-pass scan beyond).  Agents do not move.  Exact comparison with the oracle."""
    from tests.hip_harness import OBS, pull, push_actions

    cfg = dict(num_taggers=2, num_runners=n_runners, grid_length=20.0, episode_length=9, seed=1,
               max_acceleration=0.1, min_acceleration=-0.1, num_acceleration_levels=4, num_turn_levels=4,
               use_full_observation=False, num_other_agents_observed=K, tagging_distance=1e-5,
               runner_exits_game_after_tagged=True)
    deltas = [0, 1, 2, 3, 60, 127, 128, 129, 200, 255, 256, 257, 382, 383, 384, 500, 2000]
    if n_runners > 126:  # the bucket boundaries of the 9-bit keys
        deltas += [511, 512, 513, 1023, 1024, 1025, 1534, 1535, 1536, 3000]
    if n_runners > 510:  # ... and of the 10-bit keys (replicas of 513 .. 1024 agents: buckets of 1024 ulps, 3071-apart rule)
        deltas += [2047, 2048, 2049, 3070, 3071, 3072, 4095, 4096, 4097, 6000]
    cases = [(d, swap, triple) for d in deltas for swap in (0, 1) for triple in (0, 1)]
    E = len(cases)
    w = _mk(cfg, E)
    orc = TagContinuousOracle(num_envs=E, **cfg)
    N = orc.N
    f32 = np.float32
    ulp4 = np.spacing(f32(4.0))
    xs = np.zeros((E, N), f32)
    ys = np.zeros((E, N), f32)
    achieved = []
    for e, (delta, swap, triple) in enumerate(cases):
        # agent 0 at (8, 8); K-1 candidates closer than 2 on the +x ray; the pair at squared distance 4
        # and 4 + delta ulps; the rest far away on a line
        x = [8.0] + [8.0 + 0.3 * (j + 1) for j in range(K - 1)]
        y = [8.0] * K
        b = f32(np.sqrt(np.float64(delta) * np.float64(ulp4)))  # fl(b*b) ~ delta ulps of 4
        pair = [(10.0, 8.0), (6.0, float(f32(8.0) + b))]        # squared distances 4 and ~4 + delta ulps
        if swap:
            pair = pair[::-1]
        for px, py in pair:
            x.append(px); y.append(py)
        if triple:  # a third candidate 40 ulps above squared distance 4, below agent 0
            b3 = f32(np.sqrt(40.0 * np.float64(ulp4)))
            x.append(float(f32(8.0) + b3)); y.append(6.0)
        k = len(x)
        for j in range(k, N):  # the rest far away, on a line inside the arena
            x.append(14.0 + 5.5 * (j - k) / max(N - k, 1)); y.append(15.0)
        xs[e], ys[e] = np.array(x, f32), np.array(y, f32)
        dx = (xs[e, 0] - xs[e]).astype(f32); dy = (ys[e, 0] - ys[e]).astype(f32)
        d2 = ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)
        achieved.append(int(abs(int(d2[K + 1].view(np.int32)) - int(d2[K].view(np.int32)))))
    # the construction does produce the ulp distances it aims for (to within rounding of b*b)
    for (delta, _, _), got in zip(cases, achieved):
        assert abs(got - delta) <= max(2, delta // 50), (delta, got)
    state = dict(loc_x=xs, loc_y=ys, speed=np.zeros((E, N), f32), direction=np.zeros((E, N), f32),
                 acceleration=np.zeros((E, N), f32))
    orc.set_state(**state)
    _push_state(w, **state)
    a0 = int(np.argmin(np.abs(orc.acceleration_actions)))
    t0 = int(np.argmin(np.abs(orc.turn_actions)))
    act = np.zeros((E, N, 2), dtype=np.int32)
    act[..., 0], act[..., 1] = a0, t0
    for t in range(2):
        push_actions(w, act)
        w.step_all_envs()
        orc.step(act)
        got, want = pull(w, OBS), orc.obs.astype(np.float32)
        for e in range(E):
            np.testing.assert_array_equal(got[e], want[e], err_msg=f"K={K} t={t} case (delta, swap, triple)={cases[e]}")


@pytest.mark.parametrize("runners,E,ticks,kernel", [(1000, 40, 30, "HipTagContinuousTick_K10_N1024"),
                                                    (500, 64, 40, "HipTagContinuousTick_K10_N512"),
                                                    # most of an episode: the cell grid shrinks from 8 x 8 to 4 x 4 with the
                                                    # agents in the game, then the packing goes back to id order
                                                    (1000, 6, 430, "HipTagContinuousTick_K10_N1024")])
def test_big_replicas_of_the_bench_configuration_vs_c_oracle(runners, E, ticks, kernel):
    """The configurations `bench.py --num-runners 1000 / 500` times (BASELINE's physics, 5 taggers, 21-way heads, the uniform
    policy; cell-sorted, prefiltered neighbour search with its hints alive from tick 1 on): E replicas that drift apart
    through their own sampled actions, every tick of the fused kernel against the C oracle -- actions draw for draw, state,
    rewards, done, observation rows and nearest_neighbor_ids, tolerance 0."""
    cfg = dict(BENCH_CFG, num_runners=runners)
    live_seen, id_rows, id_pads, rows = _fused_ticks_vs_c_oracle(cfg, E, ticks, 7, kernel=kernel)
    assert live_seen[0] > 0.9 * (runners + 5) and id_rows > 0.8 * E * ticks * live_seen[-1]
    if ticks > 400:
        assert live_seen[-1] < 0.4 * (runners + 5), live_seen[-1]


@pytest.mark.parametrize("side,spacing,kernel", [(32, 0.625, "HipTagContinuousStep_K10_N1024"),
                                                 (20, 1.0, "HipTagContinuousStep_K10_N512")])
def test_cell_sorted_search_on_a_lattice_of_exact_ties(side, spacing, kernel):
    """Replicas of more than 128 agents pack the agents in the game by GRID CELL for the neighbour search (tc_knn.h
    "CELL-SORTED packing": a wavefront's searchers are neighbours in space and look at the cell rows around them only),
    so the order of the candidates has nothing to do with their ids any more, while the reference breaks equal distances
    by id (tag_continuous.py:422-444).  side x side agents stand still on a square lattice, ids dealt at random: every
    agent has 4 neighbours at distance 1, 4 at sqrt(2), 4 at 2, 8 at sqrt(5) lattice steps -- K = 10 cuts the third ring,
    K-th and (K+1)-th candidate are EXACTLY as far, everywhere, and which two of the four get in is decided by id alone.
    Every fourth lattice line is a cell border (cells of 2.5 / 4.0 units), the arena's border rows have fewer neighbours.
    Four ticks (the first searches inside the cell cover alone, the later ones inside the remembered neighbours' radius),
    observations and nearest_neighbor_ids compared with tolerance 0 -- no near-tie allowance."""
    from tests.hip_harness import OBS, pull, push_actions

    n = side * side
    cfg = dict(num_taggers=4, num_runners=n - 4, grid_length=20.0, episode_length=9, seed=1,
               max_acceleration=0.1, min_acceleration=-0.1, num_acceleration_levels=4, num_turn_levels=4,
               use_full_observation=False, num_other_agents_observed=10, tagging_distance=1e-5,
               runner_exits_game_after_tagged=True)
    E = 3
    w = _mk(cfg, E)
    assert w.env.resolve_step_function_name("HipTagContinuousStep") == kernel
    orc = TagContinuousOracle(num_envs=E, **cfg)
    f32 = np.float32
    rng = np.random.default_rng(side)
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xs, ys = np.zeros((E, n), f32), np.zeros((E, n), f32)
    for e in range(E):
        perm = rng.permutation(n)
        xs[e, perm] = (gx.ravel() * spacing).astype(f32)
        ys[e, perm] = (gy.ravel() * spacing).astype(f32)
    state = dict(loc_x=xs, loc_y=ys, speed=np.zeros((E, n), f32), direction=np.zeros((E, n), f32),
                 acceleration=np.zeros((E, n), f32))
    orc.set_state(**state)
    _push_state(w, **state)
    act = np.zeros((E, n, 2), dtype=np.int32)
    act[..., 0] = int(np.argmin(np.abs(orc.acceleration_actions)))
    act[..., 1] = int(np.argmin(np.abs(orc.turn_actions)))
    for t in range(4):
        push_actions(w, act)
        w.step_all_envs()
        orc.step(act)
        np.testing.assert_array_equal(pull(w, "loc_x"), xs, err_msg="the agents were to stand still")
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32), err_msg=f"observations t={t}")
        np.testing.assert_array_equal(pull(w, "nearest_neighbor_ids").reshape(orc.nearest_ids.shape), orc.nearest_ids,
                                      err_msg=f"nearest_neighbor_ids t={t}")
    # the construction is what it claims: agent rows away from the border hold an exact tie at the cut
    d = orc.neighbor_dist[0]
    srt = np.sort(d, axis=1)
    assert (srt[:, 9] == srt[:, 10]).mean() > 0.8


@pytest.mark.parametrize("n_runners,kernel", [(996, "HipTagContinuousStep_K10_N1024"), (400, "HipTagContinuousStep_K10_N512")])
def test_cell_sorted_search_with_a_crowd_in_one_corner(n_runners, kernel):
    """The cell-sorted search sizes its cells for a uniform crowd; here 70 % of the agents stand inside a blob of radius
    ~1 in one corner of a 20 x 20 arena (one or two cells hold most of the replica, a searcher's 3 x 3 block covers
    everybody it can see) and the rest are spread thinly (their K-th neighbour lies far outside their cell block: the
    radius check fails and the wavefront repeats with the full chain).  The agents drift with their reset speeds for 6
    ticks; every tick against the oracle (state, observations, nearest_neighbor_ids)."""
    from tests.hip_harness import pull, push_actions

    cfg = dict(num_taggers=4, num_runners=n_runners, grid_length=20.0, episode_length=30, seed=3,
               max_acceleration=0.1, min_acceleration=-0.1, num_acceleration_levels=4, num_turn_levels=4, max_speed=0.3,
               use_full_observation=False, num_other_agents_observed=10, tagging_distance=1e-5,
               runner_exits_game_after_tagged=True)
    E = 2
    w = _mk(cfg, E)
    assert w.env.resolve_step_function_name("HipTagContinuousStep") == kernel
    orc = TagContinuousOracle(num_envs=E, **cfg)
    N = orc.N
    f32 = np.float32
    rng = np.random.default_rng(n_runners)
    xs, ys = np.zeros((E, N), f32), np.zeros((E, N), f32)
    for e in range(E):
        crowd = rng.random(N) < 0.7
        cx, cy = (2.0, 2.5) if e == 0 else (19.0, 18.5)  # (the second replica's crowd leans on two arena walls)
        xs[e] = np.where(crowd, np.clip(cx + 0.5 * rng.standard_normal(N), 0.0, 20.0), 20.0 * rng.random(N)).astype(f32)
        ys[e] = np.where(crowd, np.clip(cy + 0.5 * rng.standard_normal(N), 0.0, 20.0), 20.0 * rng.random(N)).astype(f32)
    state = dict(loc_x=xs, loc_y=ys, speed=(0.2 * rng.random((E, N))).astype(f32),
                 direction=(6.28 * rng.random((E, N))).astype(f32), acceleration=np.zeros((E, N), f32))
    orc.set_state(**state)
    _push_state(w, **state)
    stats = {"near_tie_rows": 0, "rows": 0}
    na, nt = len(orc.acceleration_actions), len(orc.turn_actions)
    act_rng = np.random.RandomState(5)
    for t in range(6):
        a = np.stack([act_rng.randint(0, na, size=(E, N)), act_rng.randint(0, nt, size=(E, N))], axis=2)
        push_actions(w, a)
        w.step_all_envs()
        orc.step(a)
        _compare(w, orc, f"t={t}", stats)
    assert stats["near_tie_rows"] <= 2, stats
    assert (pull(w, "still_in_the_game") == 1).all()


@pytest.mark.parametrize("n_runners,K,full_obs", [(146, 8, False), (146, 8, True), (300, 10, False), (500, 10, False),
                                                  (525, 5, False), (1020, 3, False), (1000, 10, False), (700, 16, False),
                                                  (600, 20, False)])
def test_many_agents_paths(n_runners, K, full_obs):
    """replicas of more than 128 agents: with partial observations up to 512 agents take the `_N512` entries (9 id bits
    in the search keys, blocks of up to eight wavefronts), 513 .. 1024 agents the `_N1024` entries (10 id bits, blocks
    of up to sixteen wavefronts, K <= 16); full observations and K > 16 beyond 512 agents take the generic entry
    points (tc_generic_impl: K-pass selection, one block of up to 1024 threads per replica)"""
    cfg = dict(num_taggers=4, num_runners=n_runners, grid_length=30.0, episode_length=6, seed=13,
               max_acceleration=0.2, min_acceleration=-0.2, num_acceleration_levels=5, num_turn_levels=5,
               use_full_observation=full_obs, num_other_agents_observed=K, tagging_distance=0.2,
               runner_exits_game_after_tagged=True)
    from warp_drive_amd.envs.tag_continuous import TagContinuous

    name = TagContinuous(**cfg).resolve_step_function_name("HipTagContinuousStep")
    N = 4 + n_runners
    want = ("HipTagContinuousStep" if (full_obs or (N > 512 and K > 16)) else
            f"HipTagContinuousStep_K{min(k for k in (4, 8, 10, 12, 16) if k >= K)}_N1024" if N > 512 else
            f"HipTagContinuousStep_K{min(k for k in (2, 4, 6, 8, 10, 12, 16, 24, 32) if k >= K)}_N512")
    assert want is None or name == want, (name, want)
    _run_lockstep(cfg, E=3, ticks=8, seed=n_runners)


@pytest.mark.parametrize("case", range(48))
def test_random_configurations(case):
    """seeded random shapes and reward settings: agent counts around the wavefront / block boundaries,
    K from 1 to N-1, both observation modes, both exit rules, replica counts that do not fill a block"""
    rng = np.random.RandomState(1000 + case)
    n_taggers = int(rng.randint(1, 7))
    n_runners = int(rng.choice([1, 2, 5, 30, 58, 59, 60, 63, 64, 65, 100, 123, 124, 127, 130]))
    N = n_taggers + n_runners
    full = bool(rng.randint(0, 2)) and N <= 70
    K = int(rng.randint(1, min(N - 1, 33) + 1)) if N > 2 else 1
    cfg = dict(num_taggers=n_taggers, num_runners=n_runners, grid_length=float(rng.choice([2.0, 7.5, 20.0])),
               episode_length=int(rng.randint(3, 12)), seed=int(rng.randint(1, 10 ** 6)),
               max_speed=float(rng.choice([0.5, 1.0, 2.5])), max_acceleration=0.25, min_acceleration=-0.25,
               max_turn=float(rng.choice([0.8, 2.356])), min_turn=-float(rng.choice([0.8, 2.356])),
               num_acceleration_levels=int(rng.randint(1, 9)), num_turn_levels=int(rng.randint(1, 9)),
               skill_level_runner=float(rng.choice([0.7, 1.0])), skill_level_tagger=float(rng.choice([0.9, 1.0, 1.3])),
               use_full_observation=full, num_other_agents_observed=K,
               tagging_distance=float(rng.choice([0.02, 0.3, 1.0])), tag_reward_for_tagger=float(rng.choice([1.0, 10.0])),
               tag_penalty_for_runner=-float(rng.choice([1.0, 10.0])), step_penalty_for_tagger=-float(rng.choice([0.0, 0.01])),
               step_reward_for_runner=float(rng.choice([0.0, 0.01])), edge_hit_penalty=-float(rng.choice([0.0, 0.5])),
               end_of_game_reward_for_runner=float(rng.choice([0.0, 1.0])),
               runner_exits_game_after_tagged=bool(rng.randint(0, 2)))
    _run_lockstep(cfg, E=int(rng.choice([1, 2, 3, 7, 33])), ticks=14, seed=case)


def test_rollout_falls_back_when_the_tick_does_not_fit_lds():
    """~1000 agents x 41-way heads: even ONE head's probability slab (165 KB) exceeds a workgroup's LDS, so the
    rollout engine must use the separate launches (and still run).  (With 21-way heads the tick does fit since
    round 4: the heads are sampled one after the other from one slab, test_fused_tick_kernel.)"""
    import torch
    from tests.hip_harness import pull
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine

    cfg = dict(num_taggers=4, num_runners=1000, grid_length=40.0, episode_length=5, seed=3,
               num_acceleration_levels=40, num_turn_levels=40, use_full_observation=False,
               num_other_agents_observed=3, tagging_distance=0.05)
    w = _mk(cfg, 2)
    assert not w.env.can_fuse_tick()
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=1)
    engine = RolloutEngine(w, sampler)
    assert not engine.fused and len(engine.entry_names) == 4
    engine.run(7)
    torch.cuda.synchronize()
    assert pull(w, "_timestep_").max() <= 5 and np.isfinite(pull(w, "loc_x")).all()


def test_host_restore_of_state_voids_the_cleared_row_flags():
    """`obs_rows_cleared` lets the sparse row gather skip rows of agents that left the game; it is only valid
    while kernels and reset paths are the sole writers of observations / still_in_the_game.  A host restore of the
    state (`reset_device`, the manager's host -> device refresh) in the middle of an episode -- every agent back in
    the game, stale zero rows in the observation array -- must void the flags: the next ticks equal the oracle
    restarted from the same state.  Late-episode regime forced by a large tagging distance (sparse gather)."""
    from tests.hip_harness import OBS, pull, push_actions

    cfg = dict(BENCH_CFG, tagging_distance=0.2, episode_length=200)
    E = 6
    w = _mk(cfg, E)
    orc = TagContinuousOracle(num_envs=E, **cfg)
    rng = np.random.RandomState(5)

    def tick(t):
        a = np.stack([rng.randint(0, 21, size=(E, orc.N)), rng.randint(0, 21, size=(E, orc.N))], axis=2)
        push_actions(w, a)
        w.step_all_envs()
        orc.step(a)
        np.testing.assert_array_equal(pull(w, "still_in_the_game"), orc.sig, err_msg=f"t={t}")
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32), err_msg=f"obs t={t}")

    for t in range(40):
        tick(t)
    assert orc.sig.sum(axis=1).max() < 50 and pull(w, "obs_rows_cleared").sum() > 0  # the sparse form is in use
    dm = w.cuda_data_manager
    for name in ("loc_x", "loc_y", "speed", "direction", "acceleration", "still_in_the_game", "num_runners",
                 "edge_hit_reward_penalty"):
        dm.reset_device(name)   # host copies = the start of the episode
    assert pull(w, "obs_rows_cleared").sum() == 0
    orc.reset_all()
    orc.timestep[:] = 40  # (_timestep_ was not restored)
    for t in range(40, 46):
        tick(t)


@pytest.mark.parametrize("runners,K,levels,E,entry", [
    (100, 10, 20, 257, "HipTagContinuousTickA_K10_N105A21"),   # the BASELINE shape: the entry with its sizes folded
    (40, 7, 6, 61, "HipTagContinuousTickA_K8"),                # runtime sizes, K below the specialisation's
    (200, 10, 20, 9, "HipTagContinuousTickA_K10_N512"),
])
def test_tick_on_given_actions_equals_the_sampling_tick(runners, K, levels, E, entry):
    """`TickA` entries (step + restore of finished replicas on actions that are ALREADY in `sampled_actions`: the policy
    forward's epilogue draws them, training/policy_kernel.py) against the sampling tick: the same actions give the same
    arrays, bit for bit, through a whole episode and the restarts."""
    import torch
    from tests.hip_harness import ACT, OBS, REW, pull, require_gpu
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    require_gpu()
    cfg = dict(num_taggers=5, num_runners=runners, grid_length=8.0, episode_length=30, seed=3, max_speed=0.5,
               max_acceleration=0.2, min_acceleration=-0.2, num_acceleration_levels=levels, num_turn_levels=levels,
               use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.15, edge_hit_penalty=-0.1,
               runner_exits_game_after_tagged=True)

    def make(presampled):
        w = EnvWrapper(env_obj=TagContinuous(**cfg), num_envs=E, env_backend="hip")
        w.reset_all_envs()
        sampler = HIPSampler(w.cuda_function_manager)
        sampler.init_random(seed=21)
        create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                          push_data_batch_placeholders=False)
        return w, RolloutEngine(w, sampler, probabilities=None, presampled_actions=presampled)

    (wa, ea), (wb, eb) = make(False), make(True)
    assert eb.step_kernel_name == entry and ea.step_kernel_name == entry.replace("TickA", "Tick")
    act_a = wa.cuda_data_manager.data_on_device_via_torch(ACT)
    act_b = wb.cuda_data_manager.data_on_device_via_torch(ACT)
    names = [n for n, _ in STATE] + [OBS, REW, "_done_", "_timestep_", "nearest_neighbor_ids", "num_runners"]
    finished = 0
    for t in range(70):
        ea.run(1)                 # draws its actions (uniform probabilities) and steps
        act_b.copy_(act_a)        # the same actions, given
        eb.run(1)
        torch.cuda.synchronize()
        for n in names:
            np.testing.assert_array_equal(pull(wb, n), pull(wa, n), err_msg=f"{n} t={t}")
        finished += int((pull(wa, "_done_") > 0).sum())
    assert finished >= 2 * E


def test_shape_entries_built_on_demand(monkeypatch):
    """`WD_TC_JIT_SHAPES=1`: a shape without a prebuilt specialised entry gets one compiled at start-up (one hipcc of one
    unit, kept for later runs; the reference compiles its templated source for every run, pycuda_function_manager.py:
    133-232) -- 45 agents, K = 8, 8-way heads, three replicas per 192-thread block -- and the fused tick through it
    matches the C oracle tick by tick through episode ends, like the runtime-size entry it replaces."""
    from warp_drive_amd import build as wd_build
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers import hip_driver as drv

    monkeypatch.setattr(TagContinuous, "JIT_SHAPE_ENTRIES", True)
    cfg = dict(BENCH_CFG, num_runners=40, num_other_agents_observed=8, num_acceleration_levels=7, num_turn_levels=7,
               episode_length=12, tagging_distance=0.1)
    _fused_ticks_vs_c_oracle(cfg, 77, 30, 13, kernel="HipTagContinuousTick_K8_N45A8")
    built = [f for f in wd_build.shape_units_on_disk() if f.startswith("wd_kernels_tc_k8_n45a8t")]
    assert len(built) == 1 and drv.manifest().get("HipTagContinuousTickA_K8_N45A8") == built[0]
