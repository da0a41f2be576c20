"""TagGridWorld: N taggers chase one runner on an integer grid.

Host-side mirror of reference example_envs/tag_gridworld/tag_gridworld.py
(`TagGridWorld` :22-317, `CUDATagGridWorld` :320-380,
`CUDATagGridWorldWithResetPool` :383-475): same constructor, same reset()/step()
contract, same DataFeed registration; the device step launches the gfx950 kernel
`HipTagGridWorldStep` (csrc/kernels/tag_gridworld.hip) instead of the CUDA/Numba ones.
"""
import os

import numpy as np

from warp_drive_amd.utils import spaces
from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.data_feed import DataFeed
from warp_drive_amd.utils.gpu_environment_context import CUDAEnvironmentContext

_OBSERVATIONS = Constants.OBSERVATIONS
_ACTIONS = Constants.ACTIONS
_REWARDS = Constants.REWARDS
_LOC_X = "loc_x"
_LOC_Y = "loc_y"


class TagGridWorld:
    """CPU environment (one replica)."""

    name = "TagGridWorld"
    RESET_IS_DETERMINISTIC = True  # reset() restarts from the constructor's starting locations: no random draw

    def __init__(self, num_taggers=10, grid_length=10, episode_length=100, starting_location_x=None,
                 starting_location_y=None, seed=None, wall_hit_penalty=0.1, tag_reward_for_tagger=10.0,
                 tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01, use_full_observation=True,
                 env_backend="cpu"):
        assert num_taggers > 0 and episode_length > 0
        self.num_taggers = num_taggers
        self.num_agents = num_taggers + 1  # the last agent is the only runner (:65-66)
        self.episode_length = episode_length
        self.grid_length = grid_length
        self.np_random = np.random
        if seed is not None:
            self.seed(seed)
        # type 0 = tagger, 1 = runner (:81-87)
        self.agent_type = {a: int(a >= num_taggers) for a in range(self.num_agents)}
        self.taggers = {a: True for a in range(num_taggers)}
        self.runners = {self.num_agents - 1: True}
        if starting_location_x is None:
            assert starting_location_y is None
            centre = int(0.5 * grid_length)  # taggers start in the centre, the runner at (0, 0)
            starting_location_x = centre * np.ones(self.num_agents)
            starting_location_y = centre * np.ones(self.num_agents)
            starting_location_x[-1] = 0
            starting_location_y[-1] = 0
        else:
            assert len(starting_location_x) == self.num_agents
            assert len(starting_location_y) == self.num_agents
        self.starting_location_x = starting_location_x
        self.starting_location_y = starting_location_y
        # no-op, right, left, up, down
        self.step_actions = np.array([[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1]])
        self.observation_space = None  # filled in by EnvWrapper
        self.action_space = {a: spaces.Discrete(len(self.step_actions)) for a in range(self.num_agents)}
        self.timestep = None
        self.global_state = None
        self.wall_hit_penalty = wall_hit_penalty
        self.tag_reward_for_tagger = tag_reward_for_tagger
        self.tag_penalty_for_runner = tag_penalty_for_runner
        self.step_cost_for_tagger = step_cost_for_tagger
        self.use_full_observation = use_full_observation
        self.env_backend = env_backend

    def seed(self, seed=None):
        self.np_random.seed(seed)
        return [seed]

    # ---------------------------------------------------------------- host reset / step
    def reset(self):
        self.timestep = 0
        T1 = self.episode_length + 1
        self.global_state = {
            _LOC_X: np.zeros((T1, self.num_agents), dtype=np.int32),
            _LOC_Y: np.zeros((T1, self.num_agents), dtype=np.int32),
        }
        self.global_state[_LOC_X][0] = self.starting_location_x
        self.global_state[_LOC_Y][0] = self.starting_location_y
        return self.generate_observation()

    def generate_observation(self):
        n, L, t = self.num_agents, self.grid_length, self.timestep
        x = self.global_state[_LOC_X][t]
        y = self.global_state[_LOC_Y][t]
        time = float(t) / self.episode_length
        types = np.array([self.agent_type[a] for a in range(n)])
        obs = {}
        if self.use_full_observation:
            shared = np.concatenate([x / L, y / L, types])
            for a in range(n):
                me = np.zeros(n)
                me[a] = 1
                obs[a] = np.concatenate([shared, me, [time]])
            return obs
        # partial: own cell, the other side's cell (runner for taggers, closest tagger for the runner)
        d2 = np.square(x[:-1] - x[-1]) + np.square(y[:-1] - y[-1])
        closest = int(np.argmin(d2))
        for a in range(n):
            o = n - 1 if a < n - 1 else closest
            obs[a] = np.array([x[a] / L, y[a] / L, x[o] / L, y[o] / L, self.agent_type[a], time])
        return obs

    def step(self, actions=None):
        self.timestep += 1
        assert isinstance(actions, dict) and len(actions) == self.num_agents
        n, L, t = self.num_agents, self.grid_length, self.timestep
        moves = self.step_actions[[actions[a] for a in range(n)]]
        x = self.global_state[_LOC_X][t - 1] + moves[:, 0]
        y = self.global_state[_LOC_Y][t - 1] + moves[:, 1]
        cx, cy = np.clip(x, 0, L), np.clip(y, 0, L)
        hit_wall = (x != cx) | (y != cy)
        self.global_state[_LOC_X][t] = cx
        self.global_state[_LOC_Y][t] = cy
        tag = bool(((cx[:-1] == cx[-1]) & (cy[:-1] == cy[-1])).any())
        reward = np.empty(n)
        reward[:-1] = self.tag_reward_for_tagger if tag else -1.0 * self.step_cost_for_tagger
        reward[-1] = -1.0 * self.tag_penalty_for_runner if tag else 1.0 * self.step_cost_for_tagger
        reward = reward + (-1.0 * self.wall_hit_penalty * hit_wall)
        rew = {a: reward[a] for a in range(n)}
        obs = self.generate_observation()
        done = {"__all__": t >= self.episode_length or tag}
        return obs, rew, done, {}


def gridworld_policy_floats(hidden):
    """floats of one packed policy of the live-policy rollout kernel (gw5_policy_floats in tag_gridworld_n5.hip):
    W0 [H][24], b0 [H], W1 [H][H], b1 [H], Wp [5][H], bp [5], rounded up to whole 16-byte vectors"""
    H = int(hidden)
    return (H * 24 + H + H * H + H + 5 * H + 5 + 3) // 4 * 4


_STEP_ARGS = [
    _LOC_X, _LOC_Y, _ACTIONS, "_done_", _REWARDS, _OBSERVATIONS, "wall_hit_penalty",
    "tag_reward_for_tagger", "tag_penalty_for_runner", "step_cost_for_tagger", "use_full_observation",
    "world_boundary", "_timestep_", ("episode_length", "meta"), ("n_agents", "meta"), ("n_envs", "meta"),
]


class _DeviceStepMixin(CUDAEnvironmentContext):
    """Device step: one launch of HipTagGridWorldStep over all replicas."""

    TICK_HEADS = 1  # action heads the fused tick kernel samples (RolloutEngine)

    def _scalar_feed(self):
        return [
            ("wall_hit_penalty", self.wall_hit_penalty),
            ("tag_reward_for_tagger", self.tag_reward_for_tagger),
            ("tag_penalty_for_runner", self.tag_penalty_for_runner),
            ("step_cost_for_tagger", self.step_cost_for_tagger),
            ("use_full_observation", self.use_full_observation),
            ("world_boundary", self.grid_length),
        ]

    def _geometry(self):
        """Blocks pack whole replicas (the kernels derive replicas-per-block from blockDim).  Small
        batches get small blocks so that the launch still spreads over the chip (1000 replicas of 5
        agents: 84 single-wavefront blocks instead of 20 blocks of 256 threads)."""
        fm = self.cuda_function_manager
        choice = None
        for max_threads in (256, 128, 64):
            choice = fm.packed_geometry(self.num_agents, max_threads=max(max_threads, self.num_agents))
            if choice[2][0] >= 512:
                break
        return choice

    def lds_bytes(self, epb):
        """dynamic LDS of HipTagGridWorldStep / Tick: positions (int + normalised float), per-replica
        timestep / done, and the [epb * N, F] observation image"""
        A = epb * self.num_agents
        F = 4 * self.num_agents + 1 if self.use_full_observation else 6
        tables = (4 * (4 * A + 2 * epb + 2) + 15) // 16 * 16   # positions, time steps, done + 2 vote flags; the image starts 16-byte aligned (read as float4)
        with_image = tables + 4 * A * F
        return with_image if with_image <= 60000 else tables  # WD_GW_IMAGE_MAX_BYTES

    def _step_args(self):
        """the step kernel's positional arguments.  The four reward scalars go in as FLOAT64 -- the env's own Python
        floats, not the float32 copies the data manager holds -- so that the kernel can form `reward_tag +
        reward_penalty` the way the CPU step does (float64, narrowed once: csrc/kernels/tag_gridworld_rewards.h)."""
        args = list(self.cuda_step_function_feed(_STEP_ARGS))
        first = _STEP_ARGS.index("wall_hit_penalty")
        args[first:first + 4] = [np.float64(self.wall_hit_penalty), np.float64(self.tag_reward_for_tagger),
                                 np.float64(self.tag_penalty_for_runner), np.float64(self.step_cost_for_tagger)]
        return args

    def step_launch(self):
        """(function, args, block, grid, shared_bytes) of one device step."""
        epb, block, grid = self._geometry()
        return self.cuda_step, self._step_args(), block, grid, self.lds_bytes(epb)

    ticks_per_launch = 1    # > 1 (with batch tensors): fixed-policy rollout, T ticks fused per launch

    ROLLOUT_POLICY_WIDTHS = (32, 64)   # HipTagGridWorldRollout_N5_H<width>
    ROLLOUT_POLICY_PACKING = "gridworld"   # training.policy_kernel.pack_gridworld_policy

    def rollout_policy_groups(self):
        """agents the live-policy rollout kernel evaluates ONE network for, in its argument order: the taggers, the
        runner (run_configs/tag_gridworld.yaml maps them to the "tagger" / "runner" policies)"""
        return [list(range(self.num_agents - 1)), [self.num_agents - 1]]

    def has_live_policy_rollout(self, width, n_actions):
        """does a rollout kernel exist that evaluates the two policy networks itself (HipTagGridWorldRollout_N5_H<width>)
        for this env shape?  5 agents, full observations, 5 actions, hidden width 32 / 64 (RolloutEngine asks before it
        calls `tick_launch(policy=...)`)"""
        return (int(width) in self.ROLLOUT_POLICY_WIDTHS and int(n_actions) == 5 and len(self.step_actions) == 5
                and self._specialised_rollout_shape()
                and self.cuda_function_manager.has_function(f"HipTagGridWorldRollout_N5_H{int(width)}"))

    def tick_launch(self, sampler, probabilities, resetter, env_range=None, batch=None, policy=None):
        """Fused rollout tick: sample the action + step + reset finished replicas in ONE launch
        (HipTagGridWorldTick).  probabilities = [float32 CUDA tensor [E, N, n_actions]].  `_done_`
        stays set for replicas that finished on the tick (already reset); the next tick clears it.
        With `ticks_per_launch` > 1 and `batch` = {"obs": [T, E, N, F] float32, "actions": [T, E, N, 1] int32,
        "rewards": [T, E, N] float32, "done": [T, E] int32} (env-level batch tensors, T >= ticks_per_launch) the
        launch is HipTagGridWorldRollout: T ticks of a fixed-policy rollout, tick k recorded in row k.
        `policy` (optional, 5 agents / full observations only) = ((packed tagger policy, packed runner policy), hidden
        width) -- float32 CUDA tensors from training.policy_kernel.pack_gridworld_policy: the launch evaluates the two
        policy networks itself on every tick's observation rows (HipTagGridWorldRollout_N5_H<width>) instead of reading
        `probabilities`."""
        from warp_drive_amd.managers.function_manager import _stream_tag

        assert env_range is None, "replica ranges are a TagContinuous experiment"
        assert len(probabilities) == 1
        fm, dm = self.cuda_function_manager, self.cuda_data_manager
        rollout = batch is not None and int(self.ticks_per_launch) > 1
        assert rollout or int(self.ticks_per_launch) == 1, "ticks_per_launch > 1 needs the batch tensors"
        name = self.cuda_step.name.replace("Step", "Rollout" if rollout else "Tick")
        fm.initialize_functions([name])
        _, reset_args, _, _ = resetter.fused_launch(dm, 0, 0)  # builds / refreshes the descriptor table
        epb, block, grid = self._geometry()
        if rollout and self._specialised_rollout_shape():
            # the specialised rollout kernel runs blocks of ONE wavefront (12 replicas) at every batch size
            epb, block, grid = 12, (64, 1, 1), ((int(dm.meta_info("n_envs")) + 11) // 12, 1)
        args = self._step_args() + [
            sampler.rng_state, probabilities[0], np.int32(probabilities[0].shape[-1]), reset_args[0], reset_args[1],
            _stream_tag("tick")]
        if rollout:
            import torch

            E, N, T = int(dm.meta_info("n_envs")), self.num_agents, int(self.ticks_per_launch)
            F = 4 * N + 1 if self.use_full_observation else 6
            assert int(probabilities[0].shape[-1]) <= 8 and self.lds_bytes(epb) > 4 * epb * N * F and len(self.step_actions) == 5, \
                "the rollout kernel needs the LDS observation image and at most 8 actions"
            want = {"obs": ((E, N, F), torch.float32), "actions": ((E, N, 1), torch.int32),
                    "rewards": ((E, N), torch.float32), "done": ((E,), torch.int32)}
            for key, (shape, dtype) in want.items():
                t = batch[key]
                assert t.is_cuda and t.is_contiguous() and t.dtype == dtype and t.shape[0] >= T and \
                    tuple(t.shape[1:]) == shape, (key, tuple(t.shape), t.dtype)
            # the rows finished replicas are restored from are kept in LDS for the whole launch (no loads inside the
            # tick loop): the sum of the registered arrays' row lengths per replica, 0 = no room
            cache_dwords = sum(int(np.prod(dm.get_shape(k)[1:])) for k in dm.reset_data_list)
            lds = self.lds_bytes(epb)
            if lds + 4 * epb * cache_dwords <= 60000:
                lds += 4 * epb * cache_dwords
            else:
                cache_dwords = 0
            args += [np.int32(T), batch["obs"], batch["actions"], batch["rewards"], batch["done"], np.int32(cache_dwords)]
            special = self._specialised_rollout(name, block, cache_dwords)
            if policy is not None:
                from warp_drive_amd.rollout import UnsupportedRolloutShape

                (tagger, runner), width = policy
                if special is None or not self.has_live_policy_rollout(width, int(probabilities[0].shape[-1])):
                    raise UnsupportedRolloutShape("the live-policy rollout exists for 5 agents with full observations, "
                                                  "5 actions and hidden widths 32 / 64 only")
                n_w = gridworld_policy_floats(width)
                for t in (tagger, runner):
                    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n_w
                special = f"{special}_H{width}"
                fm.initialize_functions([special])
                # the fixed-policy kernel's LDS with the time table rounded up to 16 bytes, then the two policies
                lds5 = 4 * (epb * N * F + epb * cache_dwords + 64 + (int(self.episode_length) + 1 + 3) // 4 * 4 + 2 * n_w)
                return (fm.get_function(special), args + [fm.global_address("kIndexToActionArr"), tagger, runner], block,
                        grid, (lds5 + 15) // 16 * 16)
            if special is not None:
                # the kernel specialised for this shape (csrc/kernels/tag_gridworld_n5.hip, its own code object):
                # same arguments + the device address of the action table the host uploads into the main code object
                fm.initialize_functions([special])
                lds5 = 4 * (epb * N * F + epb * cache_dwords + 64 + int(self.episode_length) + 1)
                return (fm.get_function(special), args + [fm.global_address("kIndexToActionArr")], block, grid,
                        (lds5 + 15) // 16 * 16)
            return fm.get_function(name), args, block, grid, lds
        if policy is not None:
            from warp_drive_amd.rollout import UnsupportedRolloutShape

            raise UnsupportedRolloutShape("the live-policy rollout needs ticks_per_launch > 1 and the batch tensors")
        return fm.get_function(name), args, block, grid, self.lds_bytes(epb)

    SPECIALISED_ROLLOUT = os.environ.get("WD_GW_ROLLOUT_N5", "1") != "0"  # False: always the general rollout kernel

    def _specialised_rollout_shape(self):
        """the shape `HipTagGridWorldRollout_N5` is written for: 5 agents, full observations, and the registered reset
        arrays exactly the positions and the observations (the kernel restores those, and only those, in registers /
        LDS and writes them out after the last tick)"""
        dm, fm = self.cuda_data_manager, self.cuda_function_manager
        return (self.SPECIALISED_ROLLOUT and self.num_agents == 5 and bool(self.use_full_observation)
                and int(self.grid_length) <= 63 and int(self.episode_length) <= 4095
                and sorted(dm.reset_data_list) == sorted([_LOC_X, _LOC_Y, _OBSERVATIONS])
                and fm.has_function("HipTagGridWorldRollout_N5"))

    def _specialised_rollout(self, name, block, cache_dwords):
        """`HipTagGridWorldRollout_N5` when the launch has its shape, blocks of one wavefront and the restore rows
        cached in LDS, else None: the general kernel."""
        ok = (name == "HipTagGridWorldRollout" and int(block[0]) == 64 and cache_dwords > 0
              and self._specialised_rollout_shape())
        return name + "_N5" if ok else None

    def step(self, actions=None):
        self.timestep += 1
        if self.env_backend != "hip":
            raise Exception(f"{type(self).__name__} expects env_backend = 'hip'")
        fn, args, block, grid, shared = self.step_launch()
        fn(*args, block=block, grid=grid, shared=shared)


class CUDATagGridWorld(_DeviceStepMixin, TagGridWorld):
    """Device version (reference :320-380); the class name is kept for drop-in use."""

    def __init__(self, *args, **kwargs):
        TagGridWorld.__init__(self, *args, **kwargs)
        CUDAEnvironmentContext.__init__(self)

    def get_data_dictionary(self):
        feed = DataFeed()
        for key in (_LOC_X, _LOC_Y):
            feed.add_data(name=key, data=self.global_state[key][0], save_copy_and_apply_at_reset=True,
                          log_data_across_episode=True)
        feed.add_data_list(self._scalar_feed())
        return feed


class CUDATagGridWorldWithResetPool(_DeviceStepMixin, TagGridWorld):
    """Device version whose replicas restart from a random member of a pool (reference :383-475)."""

    POOL_SIZE = 5  # hard-coded in the reference too (:429)
    RESET_IS_DETERMINISTIC = False  # (kept per-replica: the observation placeholders are built the reference's way)

    def __init__(self, *args, **kwargs):
        TagGridWorld.__init__(self, *args, **kwargs)
        CUDAEnvironmentContext.__init__(self)

    def get_data_dictionary(self):
        feed = DataFeed()
        for key in (_LOC_X, _LOC_Y):
            feed.add_data(name=key, data=self.global_state[key][0], save_copy_and_apply_at_reset=False,
                          log_data_across_episode=False)
        feed.add_data_list(self._scalar_feed())
        return feed

    def get_reset_pool_dictionary(self):
        L = int(self.grid_length)
        cells = np.linspace(1, L - 1, L - 1)

        def draw():
            v = self.np_random.choice(cells, self.num_agents).astype(np.int32)
            v[-1] = 0  # the runner always restarts in the corner
            return v

        xs, ys = [], []
        for _ in range(self.POOL_SIZE):
            xs.append(draw())
            ys.append(draw())
        pool = DataFeed()
        pool.add_pool_for_reset(name=f"{_LOC_X}_reset_pool", data=np.stack(xs), reset_target=_LOC_X)
        pool.add_pool_for_reset(name=f"{_LOC_Y}_reset_pool", data=np.stack(ys), reset_target=_LOC_Y)
        return pool
