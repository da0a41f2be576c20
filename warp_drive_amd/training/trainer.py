"""Device-resident A2C / PPO trainer for the HIP rollout engine.

Mirror of reference warp_drive/training/trainers/{trainer_base,trainer_a2c}.py (config keys,
metric names, checkpoint naming), restructured so the rollout loop never synchronises with the
host: the reference pulls `done_flags.any()` to the host and synchronises three times per tick
(trainer_base.py:398-426); here a tick is

    policy forward (torch, obs read in place)  ->  probabilities written in place
    fused tick kernel (sample all heads + env step + reset of finished replicas; 1 launch)
    rewards / done / actions copied into the [T, E, n] batch tensors (torch, same stream)

and the only host round trips are the optional metric reads every `metrics_log_freq`
iterations.  Multi-GPU: one process per GPU, each with its own replicas and seed + rank; the
gradients of ALL policies live in one flat bucket (training/grad_bucket.py) that is averaged over the
ranks with a single all-reduce per training iteration (backend "nccl" = RCCL over xGMI) -- the
reference wraps every policy in its own DistributedDataParallel (trainer_a2c.py:137-146).
"""
import json
import logging
import os
import time

import numpy as np
import torch
import yaml

from warp_drive_amd import distributed as wdd
from warp_drive_amd.managers.function_manager import HIPSampler
from warp_drive_amd.rollout import RolloutEngine, UnsupportedRolloutShape
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders
from warp_drive_amd.training.grad_bucket import GradientBucket
from warp_drive_amd.training.losses import A2C, PPO
from warp_drive_amd.training import update_kernels
from warp_drive_amd.training.models import FullyConnected, action_head_sizes, flattened_obs_size
from warp_drive_amd.training.policy_kernel import (FusedPolicyForward, FusedRolloutTick, pack_gridworld_policy, pack_rollout_policy,
                                                    parameter_versions, rollout_policy_width)
from warp_drive_amd.utils.constants import Constants

_ACTIONS, _REWARDS, _OBSERVATIONS = Constants.ACTIONS, Constants.REWARDS, Constants.OBSERVATIONS
_DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_configs", "default_configs.yaml")


def recursive_merge_config_dicts(config, default_config):
    """fill `config` with every key of `default_config` it lacks (trainer_base.py:46-60)"""
    assert isinstance(config, dict) and isinstance(default_config, dict)
    for key, default in default_config.items():
        if key not in config:
            config[key] = default
        elif isinstance(default, dict) and isinstance(config[key], dict):
            recursive_merge_config_dicts(config[key], default)
    return config


class PerfStats:
    """wall-time split of the training loop (trainer_base.py:849-887)"""

    def __init__(self):
        self.iters = 0
        self.steps = 0
        self.rollout_time = 0.0
        self.training_time = 0.0

    def get_perf_stats(self):
        total = self.rollout_time + self.training_time
        return {
            "Mean rollout time per iter (ms)": 1e3 * self.rollout_time / max(self.iters, 1),
            "Mean training time per iter (ms)": 1e3 * self.training_time / max(self.iters, 1),
            "Mean steps per sec (rollout)": self.steps / max(self.rollout_time, 1e-9),
            "Mean steps per sec (training time)": self.steps / max(self.training_time, 1e-9),
            "Mean steps per sec (total)": self.steps / max(total, 1e-9),
        }


class Trainer:
    def __init__(self, env_wrapper=None, config=None, policy_tag_to_agent_id_map=None, device_id=0,
                 results_dir=None, verbose=True):
        assert env_wrapper is not None and env_wrapper.env_backend == "hip"
        assert config is not None and "trainer" in config and "policy" in config
        self.w = env_wrapper
        self.verbose = verbose
        self.rank, _, self.world = wdd.rank_info()
        self.device = torch.device("cuda", device_id)
        defaults = yaml.safe_load(open(_DEFAULT_CONFIG))
        for key, default in defaults.items():
            if key == "policy":
                for pol in config["policy"]:
                    recursive_merge_config_dicts(config["policy"][pol], default)
            else:
                config[key] = recursive_merge_config_dicts(config.get(key, {}), default)
        self.config = config
        E = env_wrapper.n_envs
        if policy_tag_to_agent_id_map is None:
            policy_tag_to_agent_id_map = {"shared": list(range(env_wrapper.n_agents))}
        self.policy_map = {k: list(v) for k, v in policy_tag_to_agent_id_map.items()}
        self.policies = list(self.policy_map)
        assert set(self.policies) == set(config["policy"]), "every policy needs a config entry"
        tcfg = config["trainer"]
        self.num_envs = E
        self.batch_len = max(1, int(tcfg["train_batch_size"]) // E)  # ticks per training iteration
        self.train_batch_size = self.batch_len * E
        # total env steps (all replicas) // steps per iteration, trainer_base.py:268-275
        self.num_iters = int(tcfg["num_episodes"]) * env_wrapper.episode_length // self.train_batch_size
        if self.num_iters == 0:
            raise ValueError("Not enough steps to even perform a single training iteration!. Please increase the "
                             "number of episodes or reduce the training batch size.")
        self.save_dir = results_dir or os.path.join(config["saving"]["basedir"], config["saving"]["name"],
                                                    config["saving"]["tag"], str(int(time.time())))
        if self.rank == 0:
            os.makedirs(self.save_dir, exist_ok=True)
            json.dump(config, open(os.path.join(self.save_dir, "run_config.json"), "w"), indent=2, default=str)

        # ---- device data: first reset pushes the env arrays, then the placeholders
        env_wrapper.reset_all_envs()
        self.sampler = HIPSampler(env_wrapper.cuda_function_manager)
        self.sampler.init_random(seed=wdd.rank_seed(tcfg.get("seed", 0) or 0, self.rank) + 1)
        create_and_push_data_placeholders(env_wrapper=env_wrapper, action_sampler=self.sampler,
                                          policy_tag_to_agent_id_map=self.policy_map,
                                          training_batch_size_per_env=self.batch_len,
                                          push_data_batch_placeholders=True)
        dm = env_wrapper.cuda_data_manager
        self.obs = dm.data_on_device_via_torch(_OBSERVATIONS)
        self.actions = dm.data_on_device_via_torch(_ACTIONS)
        self.rewards = dm.data_on_device_via_torch(_REWARDS)
        self.done = dm.data_on_device_via_torch("_done_")
        self.done_batch = dm.data_on_device_via_torch(f"{Constants.DONE_FLAGS}_batch")
        self.head_sizes = action_head_sizes(env_wrapper.env.action_space[0])
        N = env_wrapper.n_agents
        # policy output lives in fixed tensors the tick kernel reads in place
        self.probs = [torch.full((E, N, a), 1.0 / a, dtype=torch.float32, device=self.device) for a in self.head_sizes]
        self.engine = RolloutEngine(env_wrapper, self.sampler, probabilities=self.probs, reset_done=True,
                                    fused=bool(self.config["trainer"].get("fused_rollout", True)))
        if not self.engine.fused:
            # envs without a fused tick kernel: the reset launch clears `_done_`, so it runs after the
            # flags were copied into the batch (still device-side, no host sync)
            self.engine = RolloutEngine(env_wrapper, self.sampler, probabilities=self.probs, reset_done=False)

        # ---- models, optimisers, objectives
        self.models, self.optimizers, self.trainers, self.ids = {}, {}, {}, {}
        self.current_timestep = {}
        for pol in self.policies:
            pcfg = config["policy"][pol]
            ids = self.policy_map[pol]
            obs_size = flattened_obs_size(env_wrapper.env.observation_space[ids[0]])
            model = FullyConnected(obs_size, self.head_sizes, pcfg["model"]["fc_dims"]).to(self.device)
            self.current_timestep[pol] = 0
            ckpt = pcfg["model"].get("model_ckpt_filepath", "")
            if ckpt:
                self.load_model_checkpoint({pol: ckpt}, models={pol: model})
            self.models[pol] = model
            self.ids[pol] = torch.tensor(ids, dtype=torch.long, device=self.device)
            self.optimizers[pol] = torch.optim.Adam(model.parameters(), lr=pcfg["lr"])
            common = dict(discount_factor_gamma=pcfg["gamma"], normalize_advantage=pcfg["normalize_advantage"],
                          normalize_return=pcfg["normalize_return"], vf_loss_coeff=pcfg["vf_loss_coeff"],
                          entropy_coeff=pcfg["entropy_coeff"])
            algo = pcfg["algorithm"].upper()
            if algo == "A2C":
                self.trainers[pol] = A2C(**common)
            elif algo == "PPO":
                self.trainers[pol] = PPO(clip_param=pcfg["clip_param"], **common)
            else:
                raise NotImplementedError(f"algorithm {algo}: only A2C and PPO are supported")
        # one flat gradient bucket over the trainable policies: one all-reduce per iteration at N > 1
        trainable = [self.models[p] for p in self.policies if config["policy"][p]["to_train"]]
        self.grad_bucket = GradientBucket(trainable, self.device) if trainable else None
        if self.grad_bucket is not None:
            self.grad_bucket.broadcast_parameters(src=0)
        # positive / negative replica down-sampling of the objectives (trainer_base.py:210, a2c.py:196-220)
        self.neg_pos_env_ratio = tcfg.get("neg_pos_env_ratio", -1)
        # precision of the UPDATE's GEMMs: "float32" (reference semantics) or "bfloat16" (autocast: bf16
        # matrix cores, float32 master weights, float32 softmax / loss / optimizer)
        self._update_dtype = {"float32": None, "bfloat16": torch.bfloat16}[str(tcfg.get("update_dtype", "float32"))]
        self.batch = {
            pol: {
                "obs": dm.data_on_device_via_torch(f"{Constants.PROCESSED_OBSERVATIONS}_batch_{pol}")
                if self.batch_len > 1 else torch.zeros((1, E, len(self.policy_map[pol]),
                                                        flattened_obs_size(env_wrapper.env.observation_space[
                                                            self.policy_map[pol][0]])), device=self.device),
                "actions": dm.data_on_device_via_torch(f"{_ACTIONS}_batch_{pol}"),
                "rewards": dm.data_on_device_via_torch(f"{_REWARDS}_batch_{pol}"),
            } for pol in self.policies
        }
        # episodic reward bookkeeping, all on the device
        self._ep_reward = {p: torch.zeros((E, len(self.policy_map[p])), device=self.device) for p in self.policies}
        # (per replica: summed over the replicas when a metric is read; the record kernel accumulates them in place)
        self._ep_sum = {p: torch.zeros(E, device=self.device) for p in self.policies}
        self._ep_cnt = torch.zeros(E, device=self.device)
        self.perf_stats = PerfStats()
        self.metrics = {}
        # rollout tick as a hipGraph: needs a single-launch env tick (no Python-side branching)
        # batch row of the current tick, one copy per replica (element 0 is "the" counter of the framework-op path; the
        # three-launch tick's record kernel advances every replica's own)
        self._b_rows = torch.zeros(E, dtype=torch.long, device=self.device)
        self._b_idx = self._b_rows[:1]
        self._tick_graph = None
        # "auto": where the tick is bound by the host's ~70 framework dispatches rather than by the GPU -- few observation
        # rows per tick (TagGridWorld at configs[1]: 5 000 rows, 386 -> 132 us per tick when replayed from a graph;
        # TagContinuous at configs[2], 210 000 rows, is GPU-bound: no difference)
        want = tcfg.get("graph_rollout", "auto")
        if isinstance(want, str):
            want = want.lower() == "true" or (want.lower() == "auto" and E * env_wrapper.n_agents <= 50000)
        self._want_graph = bool(want) and self.engine.fused
        # precision of the policy forward in the ROLLOUT (the update always runs in float32, as the
        # reference): "float32" (default, reference semantics) or "bfloat16" (the MLP's GEMMs on the
        # bf16 matrix cores; the sampler still reads float32 probabilities)
        self._rollout_dtype = {"float32": None, "bfloat16": torch.bfloat16}[str(tcfg.get("rollout_dtype", "float32"))]
        # float32 rollouts of a supported policy shape run the forward as ONE kernel that reads the
        # env's observation rows in place and writes the sampler's probability tensors and the batch
        # copy of the rows (training/policy_kernel.py); `fused_policy_forward: False` keeps the
        # framework path
        self._fused_forward = {pol: None for pol in self.policies}
        if self._rollout_dtype is None and bool(tcfg.get("fused_policy_forward", True)) and self.device.type == "cuda":
            for pol in self.policies:
                m = self._inference_model(pol)
                obs_size = flattened_obs_size(env_wrapper.env.observation_space[self.policy_map[pol][0]])
                rows = E * len(self.policy_map[pol])
                if (FusedPolicyForward.supports(m, obs_size) and self.obs.dtype == torch.float32
                        and rows >= int(tcfg.get("fused_policy_forward_min_rows", 0))):
                    self._fused_forward[pol] = FusedPolicyForward(env_wrapper.cuda_function_manager, m, obs_size,
                                                                  arithmetic=str(tcfg.get("policy_arithmetic", "bf16x3")))
        self._ids32 = {pol: self.ids[pol].to(torch.int32) for pol in self.policies}
        # ---- the update's non-GEMM work as hand-written kernels (`trainer.fused_update`, default on; one or two heads)
        # The handle lives on THIS trainer and on its models (no process-wide switch: another trainer in the process with
        # `fused_update: False`, or on another device, is unaffected).
        self._fused_update, self._update_kernels, self.update_plan = False, None, {}
        if bool(tcfg.get("fused_update", True)) and self.device.type == "cuda" and len(self.head_sizes) <= 2 \
                and sum(self.head_sizes) + 1 <= 64:
            self._update_kernels = update_kernels.UpdateKernels(env_wrapper.cuda_function_manager)
            self._fused_update = True
            for pol in self.policies:
                self.models[pol].update_kernels = self._update_kernels
                if config["policy"][pol]["to_train"]:
                    # on record, once: which steps of this policy's update run on the hand-written kernels and which fall
                    # back to the framework (the kernels cover float32 networks of two 256-wide hidden layers fully)
                    self.update_plan[pol] = self._update_kernels.update_plan(
                        self.models[pol], self.batch_len * E * len(self.policy_map[pol]), autocast=self._update_dtype is not None)
                    if self.rank == 0:
                        update_kernels.UpdateKernels.log_update_plan(pol, self.update_plan[pol])
        # ---- the whole tick in THREE launches (`trainer.fused_tick`, default on): every policy's forward in one launch
        # with the actions drawn in its epilogue, the env's step + reset on those actions, the bookkeeping
        # (training/policy_kernel.py::FusedRolloutTick).  Needs: every policy on the fused forward with one network
        # shape, a two-head action space, and an env with a step + reset entry for given actions.
        self._fast_tick = None
        # `_stored_for`: per policy, the parameter versions the stored activations of the LAST rollout were computed with
        # (= what the forward kernel's packed weights were a copy of), or absent: nothing stored
        self._stored, self._stored_for = None, {}
        fw = [self._fused_forward[pol] for pol in self.policies]
        if (bool(tcfg.get("fused_tick", True)) and all(f is not None for f in fw) and len(fw) <= 2 and self.engine.fused
                and len(self.head_sizes) == 2 and self.batch_len > 1 and self.actions.dtype == torch.int32
                and getattr(env_wrapper.env, "has_presampled_tick", lambda: False)()
                and all((f.H, f.kt1, f.heads) == (fw[0].H, fw[0].kt1, fw[0].heads) for f in fw)):
            from warp_drive_amd.managers.function_manager import _stream_tag

            # `trainer.reuse_rollout_activations` (default on; float32 update, bf16x3 forward): the forward launch also
            # stores the hidden activations and outputs of every batch row -- 2 x [T, rows, 256] + [T, rows, 43] float32
            # per policy, 22 GB at configs[2] of the GPU's 288 -- and the update's forward pass becomes a read of them
            # (on-policy: the weights have not changed in between).  Off when the buffers would take more than half of
            # the free memory.
            self._stored = None
            if (bool(tcfg.get("reuse_rollout_activations", True)) and self._update_dtype is None and self._fused_update
                    and all(f.bx3 for f in fw) and all(len(self.models[pol].fc) == 2 for pol in self.policies)):
                W = sum(self.head_sizes) + 1
                need = sum(4 * self.batch_len * E * len(self.policy_map[pol]) * (2 * fw[0].H + W) for pol in self.policies)
                free = torch.cuda.mem_get_info(self.device)[0]
                if need <= free // 2:
                    try:
                        self._stored = {}
                        for pol in self.policies:
                            n = len(self.policy_map[pol])
                            self._stored[pol] = tuple(torch.empty((self.batch_len, E, n, c), dtype=torch.float32, device=self.device)
                                                      for c in (fw[0].H, fw[0].H, W)) if config["policy"][pol]["to_train"] else None
                    except torch.cuda.OutOfMemoryError:
                        # somebody else took the memory between the check and the allocation (another rank on the same
                        # device, another process): the buffers are an optimisation, not a requirement
                        self._stored = None
                        torch.cuda.empty_cache()
                        logging.warning(f"reuse_rollout_activations: {need / 2**30:.1f} GiB of buffers could not be allocated; the "
                                        "update recomputes its forward pass")
                else:
                    logging.warning(f"reuse_rollout_activations: {need / 2**30:.1f} GiB of buffers do not fit half of the free "
                                    f"memory ({free / 2**30:.1f} GiB); the update recomputes its forward pass")
            self._fast_tick = FusedRolloutTick(
                env_wrapper.cuda_function_manager, fw, [self.ids[pol] for pol in self.policies],
                self.obs.reshape(E, N, -1), self.actions, self.rewards, self.done, self.sampler.rng_state,
                _stream_tag("tick"), self._b_rows, [self.batch[pol]["obs"] for pol in self.policies],
                [self.batch[pol]["actions"] for pol in self.policies], [self.batch[pol]["rewards"] for pol in self.policies],
                self.done_batch, [self._ep_reward[pol] for pol in self.policies],
                [self._ep_sum[pol] for pol in self.policies], self._ep_cnt,
                stored=None if self._stored is None else [self._stored[pol] for pol in self.policies])
            self._presampled_engine = RolloutEngine(env_wrapper, self.sampler, probabilities=self.probs, reset_done=True,
                                                    presampled_actions=True)
        # ---- whole-batch rollout in ONE launch: envs whose tick kernel can evaluate small policies itself (Cartpole:
        # csrc/kernels/cartpole.hip; TagGridWorld with 5 agents and full observations: tag_gridworld_n5.hip -- two hidden
        # layers of 32 / 64 units, one head) run all `batch_len` ticks of a training batch -- policy forward, sampling,
        # step, reset, recording of the batch rows -- in a single launch.  `trainer.fused_rollout_policy: False` keeps
        # the per-tick path.
        self._batch_rollout = None
        self._setup_batch_rollout(env_wrapper, tcfg)

    def _setup_batch_rollout(self, env_wrapper, tcfg):
        env = env_wrapper.env
        if not (bool(tcfg.get("fused_rollout_policy", True)) and self.engine.fused and self._rollout_dtype is None
                and hasattr(env, "ROLLOUT_POLICY_WIDTHS") and self.batch_len > 1 and len(self.head_sizes) == 1
                and self.head_sizes[0] <= 8):
            return
        # the agent groups the kernel evaluates one network for, in its argument order; every group must be exactly
        # covered by ONE policy of the trainer (one policy may serve several groups)
        groups = getattr(env, "rollout_policy_groups", lambda: [list(range(env_wrapper.n_agents))])()
        owners = []
        for g in groups:
            owner = [p for p in self.policies if set(g) <= set(self.policy_map[p])]
            if len(owner) != 1:
                return
            owners.append(owner[0])
        if set(owners) != set(self.policies) or sum(len(g) for g in groups) != env_wrapper.n_agents:
            return
        widths = set()
        for pol in self.policies:
            obs_size = flattened_obs_size(env.observation_space[self.policy_map[pol][0]])
            widths.add(rollout_policy_width(self.models[pol], obs_size, env.ROLLOUT_POLICY_WIDTHS))
        if len(widths) != 1 or None in widths:
            return
        width = widths.pop()
        pack = pack_gridworld_policy if getattr(env, "ROLLOUT_POLICY_PACKING", "") == "gridworld" else pack_rollout_policy
        packed = {pol: pack(self.models[pol]).to(self.device) for pol in self.policies}
        E, N, T = self.num_envs, env_wrapper.n_agents, self.batch_len
        if len(self.policies) == 1:
            pol = self.policies[0]  # the per-policy batch tensors ARE the env-level ones
            env_batch = {"obs": self.batch[pol]["obs"], "actions": self.batch[pol]["actions"],
                         "rewards": self.batch[pol]["rewards"], "done": self.done_batch}
            split = None
        else:  # the kernel records env-level rows; they are scattered into the per-policy batches after the launch
            F = int(self.obs.reshape(E, N, -1).shape[-1])
            env_batch = {"obs": torch.zeros((T, E, N, F), dtype=torch.float32, device=self.device),
                         "actions": torch.zeros((T, E, N, 1), dtype=torch.int32, device=self.device),
                         "rewards": torch.zeros((T, E, N), dtype=torch.float32, device=self.device),
                         "done": self.done_batch}
            split = env_batch
        arg = packed[owners[0]] if len(groups) == 1 else [packed[o] for o in owners]
        try:
            engine = RolloutEngine(env_wrapper, self.sampler, probabilities=self.probs, reset_done=True,
                                   rollout_batch=env_batch, rollout_policy=(arg, width),
                                   ticks_per_launch=self.batch_len)  # (the env object keeps its own setting)
        except UnsupportedRolloutShape as err:  # e.g. TagGridWorld with another shape: the kernel does not exist for it
            logging.info(f"whole-batch rollout not available for this shape ({err}); using the per-tick path")
            return
        self.engine = engine
        self._batch_rollout = {"packed": packed, "pack": pack, "split": split}
        self._want_graph = False

    # --------------------------------------------------------------------------- rollout
    def _inference_model(self, pol):
        return self.models[pol]

    def _rollout_forward(self, pol, obs_p):
        """policy forward of the rollout: probabilities per head, float32"""
        return self._inference_model(pol).forward_inference(obs_p, dtype=self._rollout_dtype)[0]

    @torch.no_grad()
    def _tick(self):
        """One rollout tick, entirely on the device and free of host-side indices: policy forward ->
        fused env tick -> batch bookkeeping at row `self._b_idx` (a device counter), so the same
        sequence of launches can be replayed from a hipGraph."""
        if self._fast_tick is not None:
            self._fast_tick.forward()         # all policies: forward + the actions of both heads + batch rows (obs, actions)
            self._presampled_engine.run(1)    # the env's step + reset of finished replicas on those actions
            self._fast_tick.record()          # rewards / done rows, episodic sums, batch row += 1
            return
        b = self._b_idx
        flat_obs = self.obs.reshape(self.num_envs, self.w.n_agents, -1)
        for pol in self.policies:
            ids = self.ids[pol]
            if self._fused_forward[pol] is not None:
                self._fused_forward[pol](flat_obs, self._ids32[pol], self.probs, obs_out=self.batch[pol]["obs"],
                                         batch_row=b)
                continue
            obs_p = flat_obs if len(self.policies) == 1 else flat_obs.index_select(1, ids)
            self.batch[pol]["obs"].index_copy_(0, b, obs_p.unsqueeze(0))
            probs = self._rollout_forward(pol, obs_p)
            for h, p in enumerate(probs):
                if len(self.policies) == 1:
                    self.probs[h].copy_(p)
                else:
                    self.probs[h].index_copy_(1, ids, p)
        self.engine.run(1)  # sample + step (+ reset when fused), asynchronous on torch's stream
        self.done_batch.index_copy_(0, b, self.done.unsqueeze(0))
        finished = (self.done > 0).to(torch.float32)  # before the reset launch clears `_done_`
        if not self.engine.fused:
            self.w.reset_only_done_envs()
        for pol in self.policies:
            ids = self.ids[pol]
            a = self.actions if len(self.policies) == 1 else self.actions.index_select(1, ids)
            r = self.rewards if len(self.policies) == 1 else self.rewards.index_select(1, ids)
            self.batch[pol]["actions"].index_copy_(0, b, a.unsqueeze(0))
            self.batch[pol]["rewards"].index_copy_(0, b, r.unsqueeze(0))
            self._ep_reward[pol] += r
            self._ep_sum[pol] += self._ep_reward[pol].mean(dim=1) * finished
            self._ep_reward[pol] *= (1.0 - finished)[:, None]
        self._ep_cnt += finished
        b += 1

    def _rollout_state_arrays(self):
        """(address, bytes) of everything a rollout tick changes besides the batch rows it records: every device array of
        the env's data manager except the [T, ...] batch placeholders, the sampler's generator state and the episodic
        counters"""
        dm = self.w.cuda_data_manager
        arrays = [(int(p), int(p.nbytes)) for name, p in dm._device_data_pointer.items()
                  if "_batch" not in name and int(p.nbytes) > 0]
        rng = self.sampler.rng_state
        arrays.append((int(rng), int(rng.nbytes)))
        for t in [*self._ep_reward.values(), *self._ep_sum.values(), self._ep_cnt]:
            arrays.append((int(t.data_ptr()), t.numel() * t.element_size()))
        return arrays

    def _capture_tick_graph(self):
        """hipGraph of one tick (torch.cuda.CUDAGraph; the env kernel is launched through the C-ABI on
        the capturing stream and is recorded like any other node).  A rollout is then `batch_len`
        graph replays: one host call per tick instead of ~70 framework dispatches.

        Capture needs a few REAL warm-up ticks (allocator, library handles).  They step the env, advance the generator
        and accumulate into the episodic counters, so everything a tick changes is copied aside first and put back
        afterwards: a run that replays the graph starts from exactly the state an eager run starts from (same seed, same
        trajectory, same first logged "Mean episodic reward")."""
        from warp_drive_amd.managers import hip_driver as drv

        arrays = self._rollout_state_arrays()
        offsets = np.cumsum([0] + [(n + 255) // 256 * 256 for _, n in arrays])
        aside = torch.empty(int(offsets[-1]), dtype=torch.uint8, device=self.device)
        for (ptr, n), off in zip(arrays, offsets):
            drv.memcpy_dtod(aside.data_ptr() + int(off), ptr, n)
        graph = None
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # allocator / library warm-up outside the capture
                for _ in range(3):
                    self._b_rows.zero_()
                    self._tick()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._b_rows.zero_()
            with torch.cuda.graph(graph):
                self._tick()
            torch.cuda.synchronize()
        except Exception as err:  # capture is an optimisation: fall back to eager ticks, loudly
            logging.warning(f"rollout tick could not be captured in a hipGraph ({err}); running it eagerly")
            torch.cuda.synchronize()
            graph = None
        for (ptr, n), off in zip(arrays, offsets):   # the warm-up ticks never happened
            drv.memcpy_dtod(ptr, aside.data_ptr() + int(off), n)
        self._b_rows.zero_()
        torch.cuda.synchronize()
        return graph

    @torch.no_grad()
    def _generate_rollout_batch_in_one_launch(self):
        """the whole batch of ticks as ONE launch (the kernel evaluates the policies itself), then the episodic
        reward bookkeeping of `_tick`, vectorised over the recorded rows"""
        br = self._batch_rollout
        for pol in self.policies:
            br["pack"](self.models[pol], out=br["packed"][pol])  # the weights of this iteration
        self.engine.run(1)
        T = self.batch_len
        d = self.done_batch[:T] > 0                              # [T, E]
        idx = torch.arange(T, device=self.device)[:, None].expand(T, self.num_envs)
        last = torch.where(d, idx, torch.full_like(idx, -1)).cummax(dim=0).values   # latest finished tick <= t
        prev = torch.cat([torch.full_like(last[:1], -1), last[:-1]], dim=0)         # ... < t
        end = last[-1]                                           # [E]
        for pol in self.policies:
            if br["split"] is not None:  # env-level rows -> this policy's batch tensors
                ids = self.ids[pol]
                for key in ("obs", "actions", "rewards"):
                    self.batch[pol][key][:T].copy_(br["split"][key][:T].index_select(2, ids))
            r = self.batch[pol]["rewards"][:T]                       # [T, E, n]
            total = torch.cumsum(r, dim=0) + self._ep_reward[pol][None]  # reward since the last start carried in
            base = torch.gather(total, 0, prev.clamp(min=0)[..., None].expand_as(total))
            base = torch.where((prev >= 0)[..., None], base, torch.zeros_like(base))
            episode = total - base                                   # reward of the running episode up to tick t
            self._ep_sum[pol] += (episode.mean(dim=2) * d).sum(dim=0)
            carried = torch.gather(total, 0, end.clamp(min=0)[None, :, None].expand(1, *total.shape[1:]))[0]
            self._ep_reward[pol] = torch.where((end >= 0)[:, None], total[-1] - carried, total[-1])
        self._ep_cnt += d.sum(dim=0)

    def _generate_rollout_batch(self):
        if self._batch_rollout is not None:
            return self._generate_rollout_batch_in_one_launch()
        if self._tick_graph is None and self._want_graph:
            self._tick_graph = self._capture_tick_graph()
            self._want_graph = self._tick_graph is not None
        self._b_rows.zero_()
        for _ in range(self.batch_len):
            if self._tick_graph is not None:
                self._tick_graph.replay()
            else:
                self._tick()
        # the stored activations belong to THIS batch and to the weights the forward kernel read: the packed copy, made from
        # the parameters at the versions recorded here.  The update compares them with the parameters' versions of its own
        # moment: any change in between (load_state_dict, a manual edit, an optimizer step somebody else took) and it
        # recomputes its forward pass instead of differentiating stale activations
        self._stored_for = {}
        if self._fast_tick is not None and self._stored is not None:
            self._stored_for = {pol: self._fused_forward[pol].packed_versions for pol in self.policies
                                if self._stored.get(pol) is not None}

    @property
    def _rollout_filled_stored(self):
        """True while the last rollout's stored activations exist (whether they are still VALID is decided per policy at
        update time: `_stored_activations_valid`)"""
        return bool(self._stored_for)

    def _stored_activations_valid(self, pol):
        want = self._stored_for.get(pol)
        if want is None or self._stored is None or self._stored.get(pol) is None:
            return False
        if want != parameter_versions(self.models[pol]):
            if not getattr(self, "_warned_stale", False):
                logging.warning(f"policy '{pol}': its parameters changed between the rollout and the update (outside the "
                                "trainer's own optimizer step); the update recomputes its forward pass instead of reading "
                                "the activations the rollout stored")
                self._warned_stale = True
            return False
        return True

    # ---------------------------------------------------------------------------- update
    def _update_model_params(self, iteration, log):
        metrics = {}
        done = self.done_batch
        trained = [pol for pol in self.policies if self.config["policy"][pol]["to_train"]]
        if not trained:
            return metrics
        self.grad_bucket.zero()  # (the .grad views stay attached to the bucket: no zero_grad(set_to_none))
        for pol in trained:
            batch = self.batch[pol]
            if self._fused_update and self.neg_pos_env_ratio <= 0:
                # the objective and its gradient with respect to the network's output as ONE kernel, the ReLU masks and
                # bias gradients of the backward as one pass each (training/update_kernels.py), and for two 256-wide float32
                # hidden layers the matrix products of the backward as well
                if self._stored_activations_valid(pol):
                    # the forward pass is a read: the rollout's forward kernel stored these rows' activations and outputs
                    out = self.models[pol].forward_logits_stored(batch["obs"][: self.batch_len], *self._stored[pol])
                else:
                    with torch.autocast(device_type=self.device.type, dtype=self._update_dtype or torch.bfloat16,
                                        enabled=self._update_dtype is not None):
                        out = self.models[pol].forward_logits(batch["obs"][: self.batch_len])
                loss, m = self.trainers[pol].compute_loss_and_metrics_from_logits(
                    self.current_timestep[pol], out.float(), batch["actions"][: self.batch_len],
                    batch["rewards"][: self.batch_len], done[: self.batch_len], self.head_sizes, log,
                    kernels=self._update_kernels)
                torch.autograd.backward(loss, grad_tensors=update_kernels.unit_gradient(loss.device))  # = loss.backward()
                if log:
                    metrics[pol] = m
                continue
            with torch.autocast(device_type=self.device.type, dtype=self._update_dtype or torch.bfloat16,
                                enabled=self._update_dtype is not None):
                probs, values = self.models[pol](batch["obs"][: self.batch_len])
            probs, values = [p.float() for p in probs], values.float()
            loss, m = self.trainers[pol].compute_loss_and_metrics(
                timestep=self.current_timestep[pol], actions_batch=batch["actions"].long(),
                rewards_batch=batch["rewards"], done_flags_batch=done, action_probabilities_batch=probs,
                value_functions_batch=values, perform_logging=log, negative_positive_ratio=self.neg_pos_env_ratio)
            loss.backward()  # accumulates into this policy's slice of the flat bucket
            if log:
                metrics[pol] = m
        # the stored activations belonged to the weights that are about to change: a second update on the same batch (or
        # anything else before the next rollout) recomputes its forward pass
        self._stored_for = {}
        self.grad_bucket.all_reduce_mean()  # ONE collective for all policies (RCCL over xGMI at N > 1)
        for pol in trained:
            pcfg = self.config["policy"][pol]
            if pcfg["clip_grad_norm"]:
                torch.nn.utils.clip_grad_norm_(self.models[pol].parameters(), pcfg["max_grad_norm"])
            self.optimizers[pol].step()
            if self._fused_forward[pol] is not None:
                self._fused_forward[pol].pack()  # the rollout kernel reads re-packed weights
            self.models[pol].refresh_inference_cache()
            self.current_timestep[pol] += self.train_batch_size
            if log:
                m = metrics[pol]
                m["Current timestep"] = self.current_timestep[pol]
                m["Learning rate"] = pcfg["lr"]
                cnt = float(self._ep_cnt.sum().item())
                m["Mean episodic reward"] = float(self._ep_sum[pol].sum().item()) / cnt if cnt > 0 else float("nan")
        return metrics

    # ----------------------------------------------------------------------------- train
    def train(self, num_iters=None):
        iters = self.num_iters if num_iters is None else int(num_iters)
        log_freq = int(self.config["saving"]["metrics_log_freq"])
        save_freq = int(self.config["saving"]["model_params_save_freq"])
        for it in range(iters):
            log = (it % log_freq == 0) or (it == iters - 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self._generate_rollout_batch()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            metrics = self._update_model_params(it, log)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            self.perf_stats.iters += 1
            self.perf_stats.steps += self.train_batch_size
            self.perf_stats.rollout_time += t1 - t0
            self.perf_stats.training_time += t2 - t1
            if log:
                self.metrics = metrics
                self._log_metrics(it, metrics)
                for pol in self.policies:
                    self._ep_sum[pol].zero_()
                self._ep_cnt.zero_()
            if save_freq > 0 and ((it + 1) % save_freq == 0 or it == iters - 1):
                self.save_model_checkpoint()
        return self.metrics

    def _log_metrics(self, iteration, metrics):
        record = {"Iterations Completed": iteration + 1, **{p: m for p, m in metrics.items()},
                  "perf": self.perf_stats.get_perf_stats()}
        name = "results.json" if self.world == 1 else f"results_device_{self.rank}.json"
        if self.rank == 0 or self.world > 1:
            os.makedirs(self.save_dir, exist_ok=True)
            with open(os.path.join(self.save_dir, name), "a") as f:
                f.write(json.dumps(record) + "\n")
        if self.verbose and self.rank == 0:
            perf = record["perf"]
            head = ", ".join(f"{p}: loss {m['Total loss']:.4f} ep.rew {m['Mean episodic reward']:.3f}"
                             for p, m in metrics.items())
            logging.warning(f"[iter {iteration + 1}] {head} | rollout {perf['Mean steps per sec (rollout)']:.3e} "
                            f"steps/s, total {perf['Mean steps per sec (total)']:.3e} steps/s")

    # ------------------------------------------------------------------------ checkpoints
    def _unwrap(self, model):
        return model.module if hasattr(model, "module") else model

    def save_model_checkpoint(self):
        """`{policy}_{timestep}.state_dict`, rank 0 only (trainer_a2c.py:361-384)"""
        if self.rank != 0:
            return
        for pol in self.policies:
            path = os.path.join(self.save_dir, f"{pol}_{self.current_timestep[pol]}.state_dict")
            torch.save(self._unwrap(self.models[pol]).state_dict(), path)

    def load_model_checkpoint(self, ckpts_dict, models=None):
        """resume from `{policy}_{timestep}.state_dict`; the timestep is parsed from the name
        (trainer_a2c.py:341-359)"""
        models = models or {p: self._unwrap(m) for p, m in self.models.items()}
        for pol, path in ckpts_dict.items():
            assert os.path.isfile(path), f"invalid model checkpoint path {path}"
            models[pol].load_state_dict(torch.load(path, map_location=self.device))
            self._stored_for = {}  # (activations stored by an earlier rollout belong to the old weights)
            models[pol].refresh_inference_cache()
            stem = os.path.basename(path).split(".state_dict")[0]
            try:
                self.current_timestep[pol] = int(stem.split("_")[-1])
            except ValueError:
                pass
            if getattr(self, "_fused_forward", {}).get(pol) is not None:
                self._fused_forward[pol].pack()

    # ------------------------------------------------------------------- evaluation (f4)
    @torch.no_grad()
    def _policy_probabilities(self):
        """policy forward on the current observations; fills the [E, N, A_h] tensors the sampler reads
        and returns {policy: [per-head probabilities of its agents]}"""
        flat_obs = self.obs.reshape(self.num_envs, self.w.n_agents, -1)
        out = {}
        for pol in self.policies:
            ids = self.ids[pol]
            if self._fused_forward[pol] is not None:
                # the same kernel as the training rollout, so that evaluation (use_argmax at near-ties
                # included) picks what the rollout would pick
                self._fused_forward[pol](flat_obs, self._ids32[pol], self.probs)
                out[pol] = [p if len(self.policies) == 1 else p.index_select(1, ids) for p in self.probs]
                continue
            obs_p = flat_obs if len(self.policies) == 1 else flat_obs.index_select(1, ids)
            probs = self._rollout_forward(pol, obs_p)
            out[pol] = probs
            for h, p in enumerate(probs):
                if len(self.policies) == 1:
                    self.probs[h].copy_(p)
                else:
                    self.probs[h].index_copy_(1, ids, p)
        return out

    def fetch_episode_states(self, list_of_states=None, env_id=0, include_rewards_actions=False,
                             include_probabilities=False, policy="", **sample_params):
        """Step through one episode with the trained policies and return the requested device arrays of
        replica `env_id` for every tick (reference trainer_base.py:689-792: same arguments and return
        values; arrays are float64 [T + 1, ...] with NaN after the episode's end).

        The reference pulls every requested array to the host on every tick; here the replica's slices
        are copied device-to-device into [T + 1, ...] buffers by the episode logger
        (HIPLogController.attach_states) and pulled once at the end.  The tick is sample -> step WITHOUT
        the reset of finished replicas, so the terminal state is observable, as in the reference."""
        from warp_drive_amd.managers.function_manager import HIPLogController, _stream_tag

        assert 0 <= env_id < self.num_envs
        list_of_states = [] if list_of_states is None else list_of_states
        assert isinstance(list_of_states, list)
        dm, w = self.w.cuda_data_manager, self.w
        T, H = w.episode_length, len(self.head_sizes)
        w.reset_all_envs()  # every replica restarts; done flags are cleared
        logger = HIPLogController(w.cuda_function_manager)
        for state in list_of_states:
            assert dm.is_data_on_device(state), f"{state} is not a valid array name on the GPU!"
        extra = [n for n in ((_ACTIONS, _REWARDS) if include_rewards_actions else ()) if n not in list_of_states]
        logger.attach_states(dm, list_of_states + ["_done_"] + extra)
        logger.reset_log(dm, env_id=env_id)  # logs s_0
        actions_ptr = dm.device_data(_ACTIONS)
        E, N = self.num_envs, w.n_agents
        episode_probabilities, end = {}, T
        # rewards / actions of tick t live in row t + 1 of the log (they are written by the step)
        for t in range(T):
            probabilities = self._policy_probabilities()
            for h, (p, a) in enumerate(zip(self.probs, self.head_sizes)):
                fn, args, block, grid, shared = self.sampler.categorical_launch(
                    p, actions_ptr, E * N, a, bool(sample_params.get("use_argmax", False)),
                    _stream_tag(f"{_ACTIONS}_{h}"), out_stride=H, out_offset=h)
                fn(*args, block=block, grid=grid, shared=shared)
            w.step_all_envs()
            logger.update_log(dm, t + 1)  # s_{t+1} (and a_t, r_{t+1}, done)
            if include_probabilities:
                sel = {policy: probabilities[policy]} if len(policy) > 0 else probabilities
                episode_probabilities[t] = {pol: [p[env_id].detach().cpu().numpy() for p in ps]
                                            for pol, ps in sel.items()}
        log = logger.fetch_log(dm, check_last_valid_step=False)
        done = log["_done__for_log"]
        finished = np.flatnonzero(done[1:] > 0)
        if len(finished):
            end = int(finished[0]) + 1  # the tick whose step finished the episode
        episode_states = {}
        for state in list_of_states:
            full = np.full((T + 1, *dm.get_shape(state)[1:]), np.nan, dtype=np.float64)
            full[: end + 1] = log[f"{state}_for_log"][: end + 1]
            episode_states[state] = full
        if include_probabilities:
            episode_probabilities = {t: p for t, p in episode_probabilities.items() if t < end}
        if not include_rewards_actions:
            return episode_states
        ids = self.policy_map[policy] if len(policy) > 0 else list(range(N))
        acts = np.zeros((T, len(ids), *dm.get_shape(_ACTIONS)[2:]), dtype=dm.get_dtype(_ACTIONS))
        rews = np.zeros((T, len(ids)), dtype=np.float32)
        acts[:end] = log[f"{_ACTIONS}_for_log"][1: end + 1][:, ids]
        rews[:end] = log[f"{_REWARDS}_for_log"][1: end + 1][:, ids]
        if include_probabilities:
            return episode_states, acts, rews, episode_probabilities
        return episode_states, acts, rews

    def graceful_close(self):
        torch.cuda.synchronize()
        wdd.barrier()
