"""CPU-vs-device consistency harness (mirror of reference
warp_drive/env_cpu_gpu_consistency_checker.py:72-579): N independent host environments against
one batched HIP environment, random actions, observations / rewards / done compared every
tick, finished replicas reset on both sides, for `num_episodes` episodes.

Differences from the reference: the default tolerance is 1e-5 absolute (the reference accepts
1 %); pass `consistency_threshold_pct` to loosen it; per-replica action streams are not
re-seeded every tick (the reference's generator re-seeds, :41-42, which repeats one action)."""
import logging

import numpy as np
import torch

from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders, get_obs
from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.spaces import Box, Discrete, MultiDiscrete

_OBSERVATIONS, _ACTIONS, _REWARDS = Constants.OBSERVATIONS, Constants.ACTIONS, Constants.REWARDS


def generate_random_actions(env, num_envs, rng):
    def one(space):
        if isinstance(space, Discrete):
            return np.int32(rng.randint(0, int(space.n)))
        if isinstance(space, MultiDiscrete):
            return rng.randint(low=[0] * len(space.nvec), high=space.nvec).astype(np.int32)
        if isinstance(space, Box):
            return rng.uniform(low=space.low, high=space.high)
        raise NotImplementedError("Only 'Discrete', 'MultiDiscrete' or 'Box' type action spaces are supported")

    return [{a: one(env.action_space[a]) for a in env.action_space} for _ in range(num_envs)]


class EnvironmentCPUvsGPU:
    def __init__(self, cpu_env_class=None, cuda_env_class=None, dual_mode_env_class=None, env_configs=None,
                 num_envs=3, num_episodes=2, env_wrapper=EnvWrapper, gpu_env_backend="hip", **_ignored):
        if dual_mode_env_class is not None:
            cpu_env_class = cuda_env_class = dual_mode_env_class
        assert cpu_env_class is not None and cuda_env_class is not None and env_configs
        self.cpu_env_class, self.cuda_env_class = cpu_env_class, cuda_env_class
        self.env_configs = env_configs
        self.num_envs, self.num_episodes = num_envs, num_episodes
        self.env_wrapper = env_wrapper

    def test_env_reset_and_step(self, consistency_threshold_pct=None, seed=None, atol=1e-5):
        for scenario, cfg in self.env_configs.items():
            E = self.num_envs
            cpu = [self.env_wrapper(env_obj=self.cpu_env_class(**cfg), env_backend="cpu") for _ in range(E)]
            obs_cpu = [e.reset() for e in cpu]
            gpu = self.env_wrapper(env_obj=self.cuda_env_class(**cfg), num_envs=E, env_backend="hip")
            gpu.reset_all_envs()
            create_and_push_data_placeholders(env_wrapper=gpu, action_sampler=None,
                                              training_batch_size_per_env=None, push_data_batch_placeholders=False)
            if hasattr(gpu.env, "step_actions"):  # action-index -> move table (checker :256-264)
                gpu.cuda_data_manager.add_shared_constants({"kIndexToActionArr": gpu.env.step_actions})
                gpu.cuda_function_manager.initialize_shared_constants(gpu.cuda_data_manager, ["kIndexToActionArr"])
            dm = gpu.cuda_data_manager
            agents = sorted(obs_cpu[0].keys())
            self._compare(np.stack([get_obs(o, agents) for o in obs_cpu]), dm.pull_data_from_device(_OBSERVATIONS),
                          consistency_threshold_pct, atol, f"{scenario}: observation at reset")
            rng = np.random.RandomState(seed)
            for t in range(1, self.num_episodes * gpu.episode_length + 1):
                actions = generate_random_actions(gpu.env, E, rng)
                stacked = np.atleast_3d(np.stack([np.stack([a[i] for i in agents]) for a in actions]))
                dm.data_on_device_via_torch(_ACTIONS)[:] = torch.from_numpy(stacked)
                outs = [cpu[e].step(actions[e]) for e in range(E)]
                gpu.step_all_envs()
                self._compare(np.stack([get_obs(o[0], agents) for o in outs]), dm.pull_data_from_device(_OBSERVATIONS),
                              consistency_threshold_pct, atol, f"{scenario}: observation t={t}")
                rew = np.array([[float(o[1][i]) for i in agents] for o in outs])
                self._compare(rew, dm.pull_data_from_device(_REWARDS), consistency_threshold_pct, atol,
                              f"{scenario}: reward t={t}")
                done_cpu = np.array([bool(o[2]["__all__"]) for o in outs])
                assert np.array_equal(done_cpu, dm.pull_data_from_device("_done_") > 0), f"{scenario}: done t={t}"
                gpu.reset_only_done_envs()
                assert dm.pull_data_from_device("_done_").sum() == 0
                if done_cpu.any():
                    new_obs = [cpu[e].reset() if done_cpu[e] else outs[e][0] for e in range(E)]
                    self._compare(np.stack([get_obs(o, agents) for o in new_obs]),
                                  dm.pull_data_from_device(_OBSERVATIONS), consistency_threshold_pct, atol,
                                  f"{scenario}: observation after reset t={t}")
            logging.info(f"scenario {scenario}: CPU and HIP outputs are consistent")

    @staticmethod
    def _compare(cpu_value, gpu_value, threshold_pct, atol, what):
        cpu_value = np.asarray(cpu_value, dtype=np.float64).reshape(np.shape(gpu_value))
        diff = np.abs(cpu_value - gpu_value)
        if threshold_pct is None:
            ok = diff <= atol
        else:
            rel = np.abs(diff / (1e-10 + cpu_value)) * 100.0
            ok = (diff < threshold_pct / 100.0) | (rel < threshold_pct)
        if not ok.all():
            idx = np.argwhere(~ok)[:5]
            raise AssertionError(f"{what}: CPU and HIP differ at {idx.tolist()} (max abs diff {diff.max():.3e})")
