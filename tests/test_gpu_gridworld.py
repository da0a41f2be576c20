"""TagGridWorld on the MI355X vs the reference: KATs, recorded trajectories, oracle at
BASELINE config[1] size.  Integer state / done: bit-exact.  Observations: bit-exact.
Rewards: bit-exact too -- `float32(reward)` of the CPU step's float64 sum: the kernels take the four reward
scalars as float64 and narrow each possible sum once (csrc/kernels/tag_gridworld_rewards.h; the reference's own
CUDA kernel adds float32-narrowed scalars and is one ulp off its CPU step for the shipped run config)."""
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
    from tests.hip_harness import OBS, REW, pull

    np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=tag)
    np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep, err_msg=tag)
    np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32), err_msg=tag)
    np.testing.assert_array_equal(pull(w, REW), orc.rewards.astype(np.float32), err_msg=tag)


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
    from tests.hip_harness import OBS, REW, pull, push_actions

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
        np.testing.assert_array_equal(pull(w, REW), d["rewards"][t].astype(np.float32))
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


@pytest.mark.parametrize("full_obs,E", [(True, 1000), (False, 77)])
def test_gridworld_fused_tick(full_obs, E):
    """HipTagGridWorldTick: sampling + step + in-kernel reset in ONE launch.  The sampled actions are
    checked draw-for-draw against the CPU restatement of the kernel's Philox uniforms, then replayed
    through the oracle; finished replicas must already be reset when the launch returns while
    `_done_` still reports them."""
    import torch
    from oracle.core_np import sample_actions_counting, single_head_tick_uniform
    from tests.hip_harness import OBS, REW, pull
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    cfg = dict(num_taggers=4, grid_length=10, episode_length=23, seed=27, wall_hit_penalty=0.1,
               tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01,
               use_full_observation=full_obs)
    w = _mk(cfg, E)
    N = w.n_agents
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=5)  # (placeholders were pushed by the harness)
    rng = np.random.RandomState(3)
    probs = torch.from_numpy(rng.dirichlet(np.ones(5), size=(E, N)).astype(np.float32)).cuda()
    engine = RolloutEngine(w, sampler, probabilities=[probs])
    assert engine.fused and engine.step_kernel_name == "HipTagGridWorldTick" and len(engine.entry_names) == 1
    ocfg = dict(cfg)
    ocfg.pop("seed")
    orc = TagGridWorldOracle(num_envs=E, **ocfg)
    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    probs_host = probs.cpu().numpy()
    finished = 0
    for t in range(60):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        engine.run(1)
        torch.cuda.synchronize()
        a = pull(w, "sampled_actions")[..., 0]
        u = single_head_tick_uniform(E * N, rng_words[4:], rng_words[0], rng_words[1], _stream_tag("tick"))
        np.testing.assert_array_equal(a, sample_actions_counting(probs_host, u.reshape(E, N)), err_msg=f"t={t}")
        orc.step(a)
        np.testing.assert_array_equal(pull(w, "_done_"), orc.done, err_msg=f"done t={t}")   # still set
        np.testing.assert_array_equal(pull(w, REW), orc.rewards.astype(np.float32))
        fin = orc.done > 0
        finished += int(fin.sum())
        obs_step = orc.obs.astype(np.float32).copy()
        orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x, err_msg=f"t={t}")
        np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y, err_msg=f"t={t}")
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep, err_msg=f"t={t}")
        obs_dev = pull(w, OBS)
        np.testing.assert_array_equal(obs_dev[~fin], obs_step[~fin], err_msg=f"t={t}")
        np.testing.assert_array_equal(obs_dev[fin], orc.obs.astype(np.float32)[fin], err_msg=f"t={t}")
    assert finished >= 2 * E


@pytest.mark.parametrize("full_obs,E,ticks,general", [
    (True, 1000, 16, False), (True, 1000, 16, True), (False, 77, 9, True),
    # 51 replicas per 256-thread block (an ODD count: the tables in front of the LDS observation image are then
    # 8 bytes past a 16-byte boundary) and a last block of 36 replicas, whose slice is a multiple of 16 bytes
    # and leaves through the float4 record path
    (True, 513 * 51 + 36, 4, True),
    # the specialised kernel at a size whose row stride is NOT a multiple of 16 bytes (the record path falls back to
    # 4-byte stores on three ticks of four) with a last block of 7 replicas, several blocks per CU
    (True, 12 * 2600 + 7, 5, False)])
def test_gridworld_rollout_records_every_tick(full_obs, E, ticks, general):
    """HipTagGridWorldRollout: T ticks of a fixed-policy rollout in one launch.  Row k of the env-level batch
    tensors is tick k: the observation the actions were sampled on, the actions (draw for draw: the Philox draw of
    tick k of T single-tick launches), the rewards and the done flag, replayed through the oracle (integer moves and
    observations and rewards exact); finished replicas restart inside the launch;
    the per-tick arrays hold the state after the last tick."""
    import torch
    from oracle.core_np import sample_actions_counting, single_head_tick_uniform
    from tests.hip_harness import OBS, pull
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine

    cfg = dict(num_taggers=4, grid_length=10, episode_length=23, seed=27, wall_hit_penalty=0.1,
               tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01,
               use_full_observation=full_obs)
    w = _mk(cfg, E)
    w.env.ticks_per_launch = ticks
    if general:  # (5 agents with full observations would take HipTagGridWorldRollout_N5)
        w.env.SPECIALISED_ROLLOUT = False
    N = w.n_agents
    F = 4 * N + 1 if full_obs else 6
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=5)
    rng = np.random.RandomState(3)
    probs = torch.from_numpy(rng.dirichlet(np.ones(5), size=(E, N)).astype(np.float32)).cuda()
    batch = {"obs": torch.full((ticks, E, N, F), 7.0, device="cuda"),
             "actions": torch.full((ticks, E, N, 1), -1, dtype=torch.int32, device="cuda"),
             "rewards": torch.full((ticks, E, N), -1.0, device="cuda"),
             "done": torch.full((ticks, E), -1, dtype=torch.int32, device="cuda")}
    engine = RolloutEngine(w, sampler, probabilities=[probs], rollout_batch=batch)
    # 5 agents with full observations take the kernel specialised for that shape (blocks of one wavefront) unless told not to
    want_kernel = "HipTagGridWorldRollout" if general else "HipTagGridWorldRollout_N5"
    assert engine.fused and engine.step_kernel_name == want_kernel and engine.ticks_per_launch == ticks
    if E > 20000 and general:
        assert w.env._geometry()[0] == 51
    ocfg = dict(cfg)
    ocfg.pop("seed")
    orc = TagGridWorldOracle(num_envs=E, **ocfg)
    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    probs_host = probs.cpu().numpy()
    finished = 0
    for launch in range(6 if E < 20000 else 12):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        assert (rng_words[4:] == launch * ticks).all()
        engine.run(1)
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in batch.items()}
        for k in range(ticks):
            np.testing.assert_array_equal(b["obs"][k], orc.obs.astype(np.float32), err_msg=f"obs row {k} of launch {launch}")
            u = single_head_tick_uniform(E * N, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1], _stream_tag("tick"))
            a = sample_actions_counting(probs_host, u.reshape(E, N))
            np.testing.assert_array_equal(b["actions"][k, :, :, 0], a, err_msg=f"actions row {k}")
            orc.step(a)
            np.testing.assert_array_equal(b["rewards"][k], orc.rewards.astype(np.float32), err_msg=f"rewards row {k}")
            np.testing.assert_array_equal(b["done"][k], orc.done, err_msg=f"done row {k}")
            finished += int((orc.done > 0).sum())
            last_done = orc.done.copy()
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x)
        np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y)
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep)
        np.testing.assert_array_equal(pull(w, "_done_"), last_done)
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32))
    assert finished >= 2 * E


@pytest.mark.parametrize("hidden,E", [(32, 1000), (64, 257)])
def test_gridworld_rollout_with_the_policies_inside_the_kernel(hidden, E):
    """HipTagGridWorldRollout_N5_H<hidden>: a whole batch of ticks in one launch with the two policy networks (tagger
    for agents 0 - 3, runner for agent 4; two hidden layers, weights in LDS) evaluated by the kernel on every tick's
    observation rows (the reference's loop: trainer_base.py:383-428).  Row k of the recorded batch: the observation
    must be the oracle's; the action must be the inverse-CDF draw (Philox restated on the host, random.cu:51-85) on the
    probabilities of oracle/tag_gridworld_np.py::policy_probabilities -- the float32 restatement of the in-kernel
    forward -- except where the uniform sits within 2e-6 of a decision threshold (device expf vs numpy exp; the oracle
    then follows the device's action); rewards, done and the state after every launch exact.  And T single-tick
    launches... are covered by the fixed-policy test: the tick code is the same template."""
    import torch
    from oracle.core_np import single_head_tick_uniform
    from oracle.tag_gridworld_np import policy_probabilities, running_sums
    from tests.hip_harness import OBS, pull
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import pack_gridworld_policy, rollout_policy_width

    ticks, N, F = 20, 5, 21
    cfg = dict(num_taggers=4, grid_length=10, episode_length=23, seed=27, wall_hit_penalty=0.1,
               tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01, use_full_observation=True)
    w = _mk(cfg, E)
    w.env.ticks_per_launch = ticks
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=5)
    torch.manual_seed(hidden)
    models = [FullyConnected(F, [5], [hidden, hidden]).cuda() for _ in range(2)]  # tagger, runner
    with torch.no_grad():  # (decisive enough that the actions occur with varied probabilities)
        for m, scale in zip(models, (4.0, 7.0)):
            m.policy_head[0].weight.mul_(scale)
            m.fc["0"][0].weight.mul_(2.0)
    assert all(rollout_policy_width(m, F, w.env.ROLLOUT_POLICY_WIDTHS) == hidden for m in models)
    packed = [pack_gridworld_policy(m) for m in models]
    packed_host = [p.cpu().numpy() for p in packed]
    probs = torch.full((E, N, 5), 0.2, device="cuda")
    batch = {"obs": torch.full((ticks, E, N, F), 7.0, device="cuda"),
             "actions": torch.full((ticks, E, N, 1), -1, dtype=torch.int32, device="cuda"),
             "rewards": torch.full((ticks, E, N), -1.0, device="cuda"),
             "done": torch.full((ticks, E), -1, dtype=torch.int32, device="cuda")}
    engine = RolloutEngine(w, sampler, probabilities=[probs], rollout_batch=batch, rollout_policy=(packed, hidden))
    assert engine.fused and engine.step_kernel_name == f"HipTagGridWorldRollout_N5_H{hidden}"
    ocfg = dict(cfg)
    ocfg.pop("seed")
    orc = TagGridWorldOracle(num_envs=E, **ocfg)
    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    near = draws = finished = 0
    for launch in range(5):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        engine.run(1)
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in batch.items()}
        for k in range(ticks):
            obs = orc.obs.astype(np.float32)
            np.testing.assert_array_equal(b["obs"][k], obs, err_msg=f"obs row {k} of launch {launch}")
            p = np.empty((E, N, 5), np.float32)
            p[:, :4] = policy_probabilities(packed_host[0], hidden, obs[:, :4].reshape(-1, F)).reshape(E, 4, 5)
            p[:, 4] = policy_probabilities(packed_host[1], hidden, obs[:, 4])
            cum = running_sums(p.reshape(-1, 5)).reshape(E, N, 5)
            u = single_head_tick_uniform(E * N, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1],
                                         _stream_tag("tick")).reshape(E, N)
            want = np.minimum((cum < u[..., None]).sum(axis=-1), 4).astype(np.int32)
            got = b["actions"][k, :, :, 0]
            bad = got != want
            if bad.any():  # only where the uniform sits on a threshold
                gap = np.abs(cum[bad] - u[bad][:, None]).min(axis=1)
                assert (gap < 2e-6).all(), (launch, k, gap.max(), np.argwhere(bad)[:5])
            near += int(bad.sum())
            draws += E * N
            orc.step(got)
            np.testing.assert_array_equal(b["rewards"][k], orc.rewards.astype(np.float32), err_msg=f"rewards row {k}")
            np.testing.assert_array_equal(b["done"][k], orc.done, err_msg=f"done row {k}")
            finished += int((orc.done > 0).sum())
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x)
        np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y)
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep)
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32))
    hist = np.bincount(b["actions"].ravel(), minlength=5) / b["actions"].size
    assert finished >= E and near <= 2 + draws // 50000 and hist.max() < 0.95, (finished, near, hist)
