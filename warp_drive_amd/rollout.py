"""Device-resident rollout tick: sample every action head -> env step -> reset finished
replicas, with no host round trip between the launches.

The reference drives the same sequence from Python with one driver call per kernel, a
device->host sync on the done flags and three synchronisations per tick
(warp_drive/training/trainers/trainer_base.py:392-426; per reset it issues one launch
per registered array, pycuda_function_manager.py:686-753).  Here the whole tick is a
fixed LaunchPlan replayed from C (or captured once into a hipGraph):

    sample_actions(head 0) ... sample_actions(head H-1)   writes [E, n, H] actions directly
    Hip<Env>Step                                          obs / rewards / done in place
    reset_when_done_fused                                 every array + done/timestep, 1 launch

The probability tensors the sampler reads are whatever the policy wrote last (torch
tensors aliased in place); for kernel-only throughput they are constant uniform tensors.
"""
import numpy as np
import torch

from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.spaces import Discrete, MultiDiscrete

_ACTIONS = Constants.ACTIONS


class UnsupportedRolloutShape(RuntimeError):
    """an env's tick entry was asked for a variant (live in-kernel policy, presampled actions, ...) that does not exist
    for this env shape -- a capability answer, raised explicitly (never an `assert`: it must survive `python -O` and must
    not be confused with an assertion that caught a bug)"""


class RolloutEngine:
    def __init__(self, env_wrapper, sampler: HIPSampler, probabilities=None, reset_done=True, fused=True,
                 rollout_batch=None, rollout_policy=None, ticks_per_launch=None, presampled_actions=False):
        """probabilities: list (one per action head) of contiguous float32 CUDA tensors
        [n_envs, n_agents, n_actions_of_head]; None = uniform.  rollout_batch: the trainer's [T, E, ...] batch
        tensors for envs whose tick kernel fuses T ticks per launch and records every tick itself
        (CUDAClassicControlCartPoleEnv.tick_launch); rollout_policy: (packed weights, hidden width) of a small policy
        that such a kernel evaluates itself on every tick.  ticks_per_launch: env ticks per launch of THIS engine
        (None = the env object's own `ticks_per_launch` attribute); the env object is left as it was, so another
        engine on the same wrapper is not affected.  presampled_actions: the tick does NOT draw the actions -- whoever
        runs before it (the policy forward's epilogue, training/policy_kernel.py) has written `sampled_actions` -- and
        is the env's step + reset entry (`env.has_presampled_tick()`)."""
        env = env_wrapper.env
        self.presampled = bool(presampled_actions)
        saved = getattr(env, "ticks_per_launch", 1)
        if ticks_per_launch is not None:
            env.ticks_per_launch = int(ticks_per_launch)
        try:
            self._build(env_wrapper, sampler, probabilities, reset_done, fused, rollout_batch, rollout_policy)
        finally:
            if ticks_per_launch is not None:
                env.ticks_per_launch = saved

    def _build(self, env_wrapper, sampler, probabilities, reset_done, fused, rollout_batch, rollout_policy):
        assert env_wrapper.env_backend == "hip"
        self.w = env_wrapper
        self.sampler = sampler
        dm = env_wrapper.cuda_data_manager
        E, N = env_wrapper.n_envs, env_wrapper.n_agents
        space = env_wrapper.env.action_space[0]
        if isinstance(space, MultiDiscrete):
            head_sizes = [int(v) for v in space.nvec]
        elif isinstance(space, Discrete):
            head_sizes = [int(space.n)]
        else:
            raise NotImplementedError("RolloutEngine drives discrete action spaces")
        dev = dm.data_on_device_via_torch(_ACTIONS).device
        if probabilities is None:
            probabilities = [torch.full((E, N, a), 1.0 / a, dtype=torch.float32, device=dev) for a in head_sizes]
        assert len(probabilities) == len(head_sizes)
        for p, a in zip(probabilities, head_sizes):
            assert p.is_contiguous() and p.dtype == torch.float32 and tuple(p.shape) == (E, N, a)
        self.probabilities = probabilities
        self.head_sizes = head_sizes
        self.plan = drv.LaunchPlan()
        actions = dm.device_data(_ACTIONS)  # [E, N, H] int32 (H = 1 for Discrete)
        H = len(head_sizes)
        self.entry_names = []
        self._graph_ticks = 0
        # an env class that offers tick_launch() fuses sampling, step and reset in its own kernel
        # (restarts from a reset pool draw random members: that stays with the pool reset kernel)
        self.fused = bool(fused and reset_done and hasattr(env_wrapper.env, "tick_launch")
                          and H == getattr(env_wrapper.env, "TICK_HEADS", 2)
                          and len(dm.reset_target_to_pool) == 0
                          and getattr(env_wrapper.env, "can_fuse_tick", lambda: True)())
        # env ticks per launch (> 1 only for envs whose fused kernel loops over ticks, fixed policy)
        self.ticks_per_launch = int(getattr(env_wrapper.env, "ticks_per_launch", 1)) if self.fused else 1
        if self.fused:
            # whole tick = ONE launch: sampling, step and reset fused in the env's tick kernel
            extra = {"batch": rollout_batch} if rollout_batch is not None else {}
            if rollout_policy is not None:
                extra["policy"] = rollout_policy
            if self.presampled and not env_wrapper.env.has_presampled_tick():
                raise UnsupportedRolloutShape("this env / shape has no step + reset entry for given actions")
            if rollout_policy is not None:
                has = getattr(env_wrapper.env, "has_live_policy_rollout", None)
                if has is None or not has(int(rollout_policy[1]), int(head_sizes[0])):
                    raise UnsupportedRolloutShape(
                        f"{type(env_wrapper.env).__name__} has no rollout kernel that evaluates a policy of hidden width "
                        f"{rollout_policy[1]} for this shape")
            fn, args, block, grid, shared = env_wrapper.env.tick_launch(sampler, None if self.presampled else probabilities,
                                                                        env_wrapper.env_resetter, **extra)
            self.plan.add(fn, args, block, grid, shared)
            self.step_entry = 0
            self.step_kernel_name = fn.name
            self.entry_names.append(fn.name)
            return
        assert not self.presampled, "presampled_actions needs the env's fused tick entry"
        for k, (p, a) in enumerate(zip(probabilities, head_sizes)):
            fn, args, block, grid, shared = sampler.categorical_launch(
                p, actions, E * N, a, False, _stream_tag(f"{_ACTIONS}_{k}"), out_stride=H, out_offset=k)
            self.plan.add(fn, args, block, grid, shared)
            self.entry_names.append(f"sample_actions[{k}]")
        fn, args, block, grid, shared = env_wrapper.env.step_launch()
        self.plan.add(fn, args, block, grid, shared)
        self.step_entry = len(self.entry_names)
        self.step_kernel_name = fn.name
        self.entry_names.append(fn.name)
        if reset_done:
            fn, args, block, grid = env_wrapper.env_resetter.fused_launch(dm, np.int32(0), 1)
            self.plan.add(fn, args, block, grid, 0)
            self.entry_names.append(fn.name)

    def run(self, ticks, stream=None):
        """Enqueue `ticks` rollout ticks (asynchronous)."""
        self.plan.run(ticks, stream)

    def run_graph(self, ticks, ticks_per_graph=10, stream=None):
        """Same, replaying a hipGraph that holds `ticks_per_graph` ticks."""
        assert ticks % ticks_per_graph == 0
        if self._graph_ticks != ticks_per_graph:
            self.plan.instantiate_graph(ticks_per_graph, stream)
            self._graph_ticks = ticks_per_graph
        self.plan.run_graph(ticks // ticks_per_graph, stream)
