"""Device placeholders for a rollout: observations, sampled actions, rewards and the
optional training batches.  This fixes the (env, agent, feature) layout the kernels and
the PyTorch trainer share in place.

Mirror of reference warp_drive/training/utils/data_loader.py:30-709 (same names, shapes
and dtypes):
    observations[_<policy>][_<key>]   float32 [E, n, *obs]      reset-registered, torch
    sampled_actions[_<k>][_<policy>]  int32   [E, n, 1] per head (+ [E, n, heads] combined)
    rewards[_<policy>]                float32 [E, n]
    *_batch_<policy>                  [T_batch, E, n, ...]       (optional)
"""
import logging

import numpy as np

from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.data_feed import DataFeed
from warp_drive_amd.utils.spaces import Box, Dict, Discrete, MultiDiscrete

_OBSERVATIONS = Constants.OBSERVATIONS
_PROCESSED_OBSERVATIONS = Constants.PROCESSED_OBSERVATIONS
_ACTIONS = Constants.ACTIONS
_ACTION_MASK = Constants.ACTION_MASK
_REWARDS = Constants.REWARDS
_DONE_FLAGS = Constants.DONE_FLAGS


def all_equal(iterable):
    return len(set(iterable)) <= 1


def get_obs(obs, agent_ids, obs_dim_corresponding_to_num_agents="first", obs_key=None):
    rows = [obs[a] if obs_key is None else obs[a][obs_key] for a in agent_ids]
    arr = np.array(rows)
    if obs_dim_corresponding_to_num_agents == "last" and len(agent_ids) > 1:
        return np.swapaxes(arr, 0, -1)
    return arr


def get_flattened_obs_size(observation_space):
    if isinstance(observation_space, Box):
        return int(np.prod(observation_space.shape))
    if isinstance(observation_space, Dict):
        return int(sum(np.prod(v.shape) for k, v in observation_space.items() if k != _ACTION_MASK))
    raise NotImplementedError("Observation space must be of Box or Dict type")


def _action_layout(action_space):
    """-> (dtype, head_sizes or None for a single head, n_heads)."""
    if isinstance(action_space, Discrete):
        return np.int32, [int(action_space.n)], 1
    if isinstance(action_space, MultiDiscrete):
        sizes = [int(v) for v in action_space.nvec]
        assert len(sizes) > 1
        return np.int32, sizes, len(sizes)
    if isinstance(action_space, Box):
        return np.float32, None, int(action_space.shape[0])
    raise NotImplementedError("Only 'Discrete', 'MultiDiscrete' or 'Box' type action spaces are supported!")


def _validate_policy_map(env_wrapper, policy_tag_to_agent_id_map):
    obs = env_wrapper.obs_at_reset()
    if policy_tag_to_agent_id_map is None:
        policy_tag_to_agent_id_map = {"shared": sorted(obs.keys())}
    assert isinstance(policy_tag_to_agent_id_map, dict) and len(policy_tag_to_agent_id_map) > 0
    owner = {}
    for tag, ids in policy_tag_to_agent_id_map.items():
        assert isinstance(ids, (list, tuple, np.ndarray))
        for a in ids:
            assert a not in owner, f"{a} is mapped to multiple policies!"
            owner[a] = tag
    for a in obs:
        assert a in owner, f"{a} is not mapped to any policy!"
    return policy_tag_to_agent_id_map


def _validate_spaces(agent_ids, env_wrapper):
    obs_spaces = [env_wrapper.env.observation_space[a] for a in agent_ids]
    assert all_equal([type(s) for s in obs_spaces])
    if isinstance(obs_spaces[0], Box):
        assert all_equal([tuple(s.shape) for s in obs_spaces])
    elif isinstance(obs_spaces[0], Dict):
        assert all_equal([tuple(s.keys()) for s in obs_spaces])
        assert all_equal([tuple(tuple(v.shape) for v in s.values()) for s in obs_spaces])
    else:
        raise NotImplementedError("Only 'Box' or 'Dict' type observation spaces are supported!")
    act_spaces = [env_wrapper.env.action_space[a] for a in agent_ids]
    assert all_equal([type(s) for s in act_spaces])
    layouts = []
    for s in act_spaces:
        _, sizes, heads = _action_layout(s)
        layouts.append((tuple(sizes) if sizes else None, heads))
    assert all_equal(layouts)


def _push(env_wrapper, feed):
    if len(feed):
        env_wrapper.cuda_data_manager.push_data_to_device(feed, torch_accessible=True)


def reset_is_deterministic(env):
    """True when `env.reset()` provably returns the same observation every time: some class of its MRO sets
    RESET_IS_DETERMINISTIC = True AND no class below that one overrides reset().  A subclass whose reset() draws random
    starts (or wraps a reset pool) therefore falls back to the reference's one-reset-per-replica behaviour
    (data_loader.py:348) unless it sets the flag itself -- inheriting the flag silently would replicate ONE
    observation over all replicas."""
    if "RESET_IS_DETERMINISTIC" in vars(env):  # an instance says so itself
        return bool(vars(env)["RESET_IS_DETERMINISTIC"])
    mro = type(env).__mro__
    flag_owner = next((c for c in mro if "RESET_IS_DETERMINISTIC" in c.__dict__), None)
    if flag_owner is None or not flag_owner.__dict__["RESET_IS_DETERMINISTIC"]:
        return False
    reset_owner = next((c for c in mro if "reset" in c.__dict__), None)
    return reset_owner is None or issubclass(flag_owner, reset_owner)


def _observation_placeholders(env_wrapper, agent_ids, obs_dim, suffix=""):
    E = env_wrapper.n_envs
    # The reference resets the host env once per replica (data_loader.py:348).  An env whose reset draws nothing
    # (RESET_IS_DETERMINISTIC: TagContinuous and TagGridWorld restart from the positions drawn in the constructor)
    # returns the same observation every time, and a host reset of a 1005-agent replica takes 70 ms: one reset,
    # E copies -- the same arrays, 2000 x sooner.
    same = reset_is_deterministic(env_wrapper.env)
    obs = [env_wrapper.obs_at_reset()] if same else [env_wrapper.obs_at_reset() for _ in range(E)]

    def stack(rows):
        return np.repeat(rows[0][None], E, axis=0) if same else np.stack(rows, axis=0)

    feed = DataFeed()
    first = obs[0][agent_ids[0]]
    if isinstance(first, (list, np.ndarray)):
        stacked = stack([get_obs(o, agent_ids, obs_dim) for o in obs])
        feed.add_data(name=_OBSERVATIONS + suffix, data=stacked, save_copy_and_apply_at_reset=True)
    elif isinstance(first, dict):
        for key in first:
            stacked = stack([get_obs(o, agent_ids, obs_dim, obs_key=key) for o in obs])
            feed.add_data(name=f"{_OBSERVATIONS}{suffix}_{key}", data=stacked, save_copy_and_apply_at_reset=True)
    else:
        raise NotImplementedError("Only array or dict type observations are supported!")
    _push(env_wrapper, feed)


def _action_placeholders(env_wrapper, agent_ids, suffix=""):
    E, n = env_wrapper.n_envs, len(agent_ids)
    dtype, _, heads = _action_layout(env_wrapper.env.action_space[agent_ids[0]])
    feed = DataFeed()
    if heads == 1:
        feed.add_data(name=_ACTIONS + suffix, data=np.zeros((E, n, 1), dtype=dtype))
    else:
        # one [E, n, 1] placeholder per head (the sampler is invoked per head) + the combined tensor
        for k in range(heads):
            feed.add_data(name=f"{_ACTIONS}_{k}{suffix}", data=np.zeros((E, n, 1), dtype=dtype))
        feed.add_data(name=_ACTIONS + suffix, data=np.zeros((E, n, heads), dtype=dtype))
    _push(env_wrapper, feed)


def _reward_placeholders(env_wrapper, agent_ids, suffix=""):
    feed = DataFeed()
    feed.add_data(name=_REWARDS + suffix, data=np.zeros((env_wrapper.n_envs, len(agent_ids)), dtype=np.float32))
    _push(env_wrapper, feed)


def _register_sampler(env_wrapper, action_sampler, agent_ids, suffix=""):
    dtype, sizes, heads = _action_layout(env_wrapper.env.action_space[agent_ids[0]])
    dm = env_wrapper.cuda_data_manager
    deterministic = sizes is None
    if heads == 1:
        action_sampler.register_actions(dm, action_name=_ACTIONS + suffix,
                                        num_actions=1 if deterministic else sizes[0], is_deterministic=deterministic)
    else:
        for k in range(heads):
            action_sampler.register_actions(dm, action_name=f"{_ACTIONS}_{k}{suffix}",
                                            num_actions=1 if deterministic else sizes[k],
                                            is_deterministic=deterministic)


def _batch_placeholders(env_wrapper, agent_ids, T_batch, tag):
    E, n, dm = env_wrapper.n_envs, len(agent_ids), env_wrapper.cuda_data_manager
    dtype, _, heads = _action_layout(env_wrapper.env.action_space[agent_ids[0]])
    feed = DataFeed()
    if not dm.is_data_on_device(f"{_ACTIONS}_batch_{tag}"):
        feed.add_data(name=f"{_ACTIONS}_batch_{tag}", data=np.zeros((T_batch, E, n, heads), dtype=dtype))
    if not dm.is_data_on_device(f"{_REWARDS}_batch_{tag}"):
        feed.add_data(name=f"{_REWARDS}_batch_{tag}", data=np.zeros((T_batch, E, n), dtype=np.float32))
    _push(env_wrapper, feed)


def create_and_push_data_placeholders(env_wrapper=None, action_sampler=None, policy_tag_to_agent_id_map=None,
                                      create_separate_placeholders_for_each_policy=False,
                                      obs_dim_corresponding_to_num_agents="first",
                                      training_batch_size_per_env=None, push_data_batch_placeholders=True):
    assert env_wrapper is not None and env_wrapper.env_backend != "cpu"
    policy_map = _validate_policy_map(env_wrapper, policy_tag_to_agent_id_map)
    if push_data_batch_placeholders:
        assert training_batch_size_per_env is not None and training_batch_size_per_env > 0, (
            "push_data_batch_placeholders is True, but training_batch_size_per_env is not defined")

    if create_separate_placeholders_for_each_policy:
        assert len(policy_map) > 1
        groups = [(f"_{tag}", list(ids)) for tag, ids in policy_map.items()]
    else:
        groups = [("", list(range(env_wrapper.n_agents)))]
    for suffix, ids in groups:
        if len(ids) > 1:
            _validate_spaces(ids, env_wrapper)
        _observation_placeholders(env_wrapper, ids, obs_dim_corresponding_to_num_agents, suffix)
        _action_placeholders(env_wrapper, ids, suffix)
        _reward_placeholders(env_wrapper, ids, suffix)
        if action_sampler:
            _register_sampler(env_wrapper, action_sampler, ids, suffix)
    for tag, ids in policy_map.items():
        logging.info(f"policy {tag}: obs {env_wrapper.env.observation_space[ids[0]]}, "
                     f"actions {env_wrapper.env.action_space[ids[0]]}")

    dm = env_wrapper.cuda_data_manager
    if training_batch_size_per_env is not None and training_batch_size_per_env > 1:
        for tag, ids in policy_map.items():
            name = f"{_PROCESSED_OBSERVATIONS}_batch_{tag}"
            if not dm.is_data_on_device_via_torch(name):
                size = get_flattened_obs_size(env_wrapper.env.observation_space[ids[0]])
                feed = DataFeed()
                feed.add_data(name=name, data=np.zeros((training_batch_size_per_env, env_wrapper.n_envs,
                                                        len(ids), size), dtype=np.float32))
                _push(env_wrapper, feed)
    if push_data_batch_placeholders:
        for tag, ids in policy_map.items():
            _batch_placeholders(env_wrapper, list(ids), training_batch_size_per_env, tag)
        name = f"{_DONE_FLAGS}_batch"
        if not dm.is_data_on_device(name):
            feed = DataFeed()
            feed.add_data(name=name, data=np.zeros((training_batch_size_per_env,) + tuple(dm.get_shape("_done_")),
                                                   dtype=np.int32))
            _push(env_wrapper, feed)
