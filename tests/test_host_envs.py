"""Host-side (CPU) environments of the package vs fixtures recorded from the reference.
Covers BASELINE config[0]: TagGridWorld 6x6, 5 agents, num_envs=2, pure CPU step()."""
import json
import os

import numpy as np
import pytest

from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.envs.tag_continuous import TagContinuous
from warp_drive_amd.envs.tag_gridworld import TagGridWorld


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return d, json.loads(str(d["config"]))


def _obs(o, n):
    return np.stack([np.asarray(o[a], dtype=np.float64) for a in range(n)])


@pytest.mark.parametrize("tag", ["full", "partial", "g6", "g10"])
def test_gridworld_cpu_backend(golden_dir, tag):
    d, cfg = _load(golden_dir, f"gw_traj_{tag}.npz")
    E = d["actions"].shape[1]
    envs = [EnvWrapper(env_obj=TagGridWorld(**cfg), env_backend="cpu") for _ in range(E)]
    n = envs[0].n_agents
    for i, e in enumerate(envs):
        np.testing.assert_array_equal(_obs(e.reset(), n), d["obs_at_reset"][i])
    for t in range(d["actions"].shape[0]):
        for i, e in enumerate(envs):
            obs, rew, done, _ = e.step({a: int(d["actions"][t, i, a]) for a in range(n)})
            np.testing.assert_array_equal(_obs(obs, n), d["obs"][t, i])
            np.testing.assert_array_equal(np.array([rew[a] for a in range(n)]), d["rewards"][t, i])
            assert bool(done["__all__"]) == bool(d["done"][t, i])
            np.testing.assert_array_equal(e.env.global_state["loc_x"][e.env.timestep], d["loc_x"][t, i])
            if done["__all__"]:
                e.reset()


@pytest.mark.parametrize("tag", ["test1", "test2", "test3", "test4", "tagheavy", "bench5x100", "bench5x100_ep", "big5x250"])
def test_tag_continuous_cpu_backend(golden_dir, tag):
    d, cfg = _load(golden_dir, f"tc_traj_{tag}.npz")
    E = d["actions"].shape[1]
    envs = [EnvWrapper(env_obj=TagContinuous(**cfg), env_backend="cpu") for _ in range(E)]
    n = envs[0].n_agents
    for i, e in enumerate(envs):
        np.testing.assert_array_equal(_obs(e.reset(), n), d["obs_at_reset"][i])
    T = min(d["actions"].shape[0], 60)
    for t in range(T):
        for i, e in enumerate(envs):
            obs, rew, done, _ = e.step({a: d["actions"][t, i, a] for a in range(n)})
            ts = e.env.timestep
            for k in ("loc_x", "loc_y", "speed", "direction", "acceleration"):
                np.testing.assert_array_equal(e.env.global_state[k][ts], d[k][t, i], err_msg=f"{k} t={t}")
            np.testing.assert_array_equal(e.env.still_in_the_game, d["still_in_the_game"][t, i])
            np.testing.assert_array_equal(np.array([float(rew[a]) for a in range(n)]), d["rewards"][t, i])
            np.testing.assert_array_equal(_obs(obs, n), d["obs"][t, i], err_msg=f"obs t={t}")
            assert bool(done["__all__"]) == bool(d["done"][t, i])
            if done["__all__"]:
                e.reset()


def test_tag_continuous_picks_the_entry_point_by_agent_count_and_k():
    """the fast-path specialisations: K <= 32 up to 128 agents and (`_N512`) up to 512, the `_N1024` entries (K <= 16) for 513 .. 1024 agents,
    the generic entry for full observations and for what no specialisation covers
    (csrc/kernels/tag_continuous.hip: WD_TC_SPECIALISE / WD_TC_SPECIALISE_BIG)"""
    from warp_drive_amd.envs.tag_continuous import TagContinuous

    def name(runners, K, full=False):
        env = TagContinuous(num_taggers=5, num_runners=runners, use_full_observation=full, num_other_agents_observed=K,
                            seed=1)
        return env.resolve_step_function_name("HipTagContinuousStep")

    assert name(100, 10) == "HipTagContinuousStep_K10"
    assert name(100, 9) == "HipTagContinuousStep_K10" and name(100, 11) == "HipTagContinuousStep_K12"
    assert name(123, 10) == "HipTagContinuousStep_K10" and name(124, 10) == "HipTagContinuousStep_K10_N512"
    assert name(500, 32) == "HipTagContinuousStep_K32_N512" and name(100, 33) == "HipTagContinuousStep"
    assert name(600, 10) == "HipTagContinuousStep_K10_N1024" and name(1019, 3) == "HipTagContinuousStep_K4_N1024"
    assert name(1000, 16) == "HipTagContinuousStep_K16_N1024" and name(1000, 17) == "HipTagContinuousStep"
    assert name(100, 10, full=True) == "HipTagContinuousStep"

    # with a function manager that has it, the entry whose sizes are folded at compile time wins (BASELINE shape only)
    from warp_drive_amd.managers.function_manager import HIPFunctionManager

    class Manager:
        _num_agents, _num_envs = 105, 4
        packed_geometry = HIPFunctionManager.packed_geometry  # (the entry is compiled for this geometry's 128 threads)

        def has_function(self, fname):
            return fname == "HipTagContinuousStep_K10_N105A21"

    env = TagContinuous(num_taggers=5, num_runners=100, num_other_agents_observed=10, use_full_observation=False, num_acceleration_levels=20, num_turn_levels=20, seed=1)
    env.cuda_function_manager = Manager()
    assert env.resolve_step_function_name("HipTagContinuousStep") == "HipTagContinuousStep_K10_N105A21"
    env = TagContinuous(num_taggers=5, num_runners=100, num_other_agents_observed=9, use_full_observation=False, num_acceleration_levels=20, num_turn_levels=20, seed=1)
    env.cuda_function_manager = Manager()
    assert env.resolve_step_function_name("HipTagContinuousStep") == "HipTagContinuousStep_K10"


def test_tag_continuous_lds_fits_a_workgroup_up_to_1024_agents():
    """the fused tick's LDS image: both probability slabs up to 256 agents, ONE slab beyond (the heads are sampled one
    after the other): 1005 agents with 21-way heads fit a workgroup's 160 KB, 41-way heads do not"""
    from warp_drive_amd.envs.tag_continuous import TagContinuous

    def lds(runners, levels):
        env = TagContinuous(num_taggers=5, num_runners=runners, use_full_observation=False, num_other_agents_observed=10,
                            num_acceleration_levels=levels, num_turn_levels=levels, seed=1)
        n = env.num_agents
        return env.lds_bytes(1, fused=True, threads=(n + 63) // 64 * 64)

    assert lds(100, 20) <= 20480              # eight blocks per CU at the BASELINE shape
    assert lds(500, 20) <= 80 * 1024          # two blocks per CU at ~510 agents (one slab: 42 KB, not 85)
    assert lds(1000, 20) <= 160 * 1024 < lds(1000, 40)


def test_observation_placeholders_one_reset_equals_a_reset_per_replica(monkeypatch):
    """envs whose reset draws nothing (RESET_IS_DETERMINISTIC) are reset once on the host and the observation copied
    to every replica: the arrays must be what the reference's reset-per-replica loop produces (data_loader.py:348)"""
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.envs.tag_gridworld import TagGridWorld
    from warp_drive_amd.training import data_loader

    class Wrapper:
        def __init__(self, env, E):
            self.env, self.n_envs, self.resets = env, E, 0

        def obs_at_reset(self):
            self.resets += 1
            return self.env.reset()

    feeds = []
    monkeypatch.setattr(data_loader, "_push", lambda w, feed: feeds.append(feed))
    for env in (TagContinuous(num_taggers=2, num_runners=9, num_other_agents_observed=4, seed=3),
                TagGridWorld(num_taggers=4, grid_length=7, episode_length=9, seed=5, use_full_observation=True)):
        ids = sorted(env.reset().keys())
        w = Wrapper(env, 6)
        data_loader._observation_placeholders(w, ids, len(ids))
        assert w.resets == 1
        env.RESET_IS_DETERMINISTIC = False  # (instance attribute: the reference's way)
        w = Wrapper(env, 6)
        data_loader._observation_placeholders(w, ids, len(ids))
        assert w.resets == 6
        one, per_replica = feeds[-2], feeds[-1]
        assert list(one.keys()) == list(per_replica.keys())
        for key in one:
            a, b = np.asarray(one[key]["data"]), np.asarray(per_replica[key]["data"])
            assert a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes(), key


def test_gridworld_rollout_kernel_choice():
    """`HipTagGridWorldRollout_N5` serves exactly the shape it is written for (5 agents, full observations, the
    positions and the observations as the only registered reset arrays, the kernel present); anything else -- and
    `SPECIALISED_ROLLOUT = False` -- keeps the general kernel (envs/tag_gridworld.py::_specialised_rollout_shape)"""
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld

    class FakeDM:
        reset_data_list = ["loc_x", "loc_y", "observations"]

    class FakeFM:
        present = True

        def has_function(self, name):
            return self.present and name == "HipTagGridWorldRollout_N5"

    def env(**kw):
        cfg = dict(num_taggers=4, grid_length=10, episode_length=100, seed=1, use_full_observation=True)
        cfg.update(kw)
        e = CUDATagGridWorld(**cfg)
        e.cuda_data_manager, e.cuda_function_manager = FakeDM(), FakeFM()
        return e

    e = env()
    assert e._specialised_rollout_shape()
    assert e._specialised_rollout("HipTagGridWorldRollout", (64, 1, 1), 115) == "HipTagGridWorldRollout_N5"
    assert e._specialised_rollout("HipTagGridWorldRollout", (256, 1, 1), 115) is None      # blocks of several wavefronts
    assert e._specialised_rollout("HipTagGridWorldRollout", (64, 1, 1), 0) is None         # no room for the restore cache
    assert e._specialised_rollout("HipTagGridWorldTick", (64, 1, 1), 115) is None
    assert not env(num_taggers=5)._specialised_rollout_shape()                             # 6 agents
    assert not env(use_full_observation=False)._specialised_rollout_shape()
    assert not env(grid_length=64)._specialised_rollout_shape()                            # cells beyond the quotient table
    assert not env(episode_length=5000)._specialised_rollout_shape()
    e = env()
    e.cuda_data_manager.reset_data_list = ["loc_x", "loc_y", "observations", "something_else"]
    assert not e._specialised_rollout_shape()   # an array the kernel would not restore
    e = env()
    e.cuda_function_manager.present = False
    assert not e._specialised_rollout_shape()   # the extra code object is not there
    e = env()
    e.SPECIALISED_ROLLOUT = False
    assert not e._specialised_rollout_shape()


@pytest.mark.parametrize("hidden", [32, 64])
def test_gridworld_policy_packing_matches_the_framework_model(hidden):
    """the live-policy TagGridWorld rollout (csrc/kernels/tag_gridworld_n5.hip): the packed layout
    (training/policy_kernel.py::pack_gridworld_policy) and the oracle's float32 restatement of the in-kernel forward
    (oracle/tag_gridworld_np.py::policy_probabilities) reproduce training.models.FullyConnected to rounding"""
    import torch

    from oracle.tag_gridworld_np import policy_probabilities, running_sums
    from warp_drive_amd.envs.tag_gridworld import gridworld_policy_floats
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import pack_gridworld_policy

    torch.manual_seed(hidden)
    model = FullyConnected(21, [5], [hidden, hidden])
    packed = pack_gridworld_policy(model).numpy()
    assert packed.size == gridworld_policy_floats(hidden) and packed.size % 4 == 0
    again = torch.full((packed.size,), 7.0)
    assert np.array_equal(pack_gridworld_policy(model, out=again).numpy(), packed)  # refill in place
    x = np.random.RandomState(1).rand(40, 21).astype(np.float32)
    p = policy_probabilities(packed, hidden, x)
    with torch.no_grad():
        q = model.forward_inference(torch.from_numpy(x))[0]
    q = (q[0] if isinstance(q, (list, tuple)) else q).numpy()
    assert np.abs(p - q).max() < 1e-6
    c = running_sums(p)
    assert (np.diff(c, axis=1) >= 0).all() and np.abs(c[:, -1] - 1).max() < 1e-6


def test_deterministic_reset_shortcut_is_not_inherited_past_a_reset_override():
    """data_loader.reset_is_deterministic: the one-reset-for-all-replicas shortcut holds for the classes that declare
    it, and is dropped for a subclass that overrides reset() without declaring it again"""
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld, CUDATagGridWorldWithResetPool, TagGridWorld
    from warp_drive_amd.training.data_loader import reset_is_deterministic

    assert reset_is_deterministic(TagContinuous(num_taggers=2, num_runners=3, seed=1))
    assert reset_is_deterministic(TagGridWorld(num_taggers=4)) and reset_is_deterministic(CUDATagGridWorld(num_taggers=4))
    assert not reset_is_deterministic(CUDATagGridWorldWithResetPool(num_taggers=4))

    class RandomStarts(TagGridWorld):
        def reset(self):
            return super().reset()

    class RandomStartsDeclared(RandomStarts):
        RESET_IS_DETERMINISTIC = True

    assert not reset_is_deterministic(RandomStarts(num_taggers=4))
    assert reset_is_deterministic(RandomStartsDeclared(num_taggers=4))
