"""TagContinuous: taggers chase runners on a continuous square.

Host-side mirror of reference example_envs/tag_continuous/tag_continuous.py:28-887
(same constructor, reset()/step() contract and DataFeed registration).  On the
"hip" backend step() is one launch of the gfx950 kernel `HipTagContinuousStep[_K<k>]`
(csrc/kernels/tag_continuous.hip) with the reference kernel's argument list
(tag_continuous.py:806-840) plus n_envs.

The host (CPU) step is the env's own implementation of the same rules -- it serves the
"cpu" backend and the initial reset on the host (env_wrapper.py) and is vectorised over
agents with numpy instead of the reference's per-agent Python loops.  One deliberate
difference from the reference CPU code: neighbour distances square with x*x (IEEE),
not with libm powf(x, 2) that `np.float32 ** 2` happens to call (tag_continuous.py:414);
the two differ by one ulp for ~0.07 % of inputs and only matter for exact near-ties.
The device kernel squares the same way, so host and device agree with each other.
"""
import copy
import os

import numpy as np

from warp_drive_amd.utils import spaces
from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.data_feed import DataFeed
from warp_drive_amd.utils.gpu_environment_context import CUDAEnvironmentContext

_OBSERVATIONS = Constants.OBSERVATIONS
_ACTIONS = Constants.ACTIONS
_REWARDS = Constants.REWARDS
_LOC_X, _LOC_Y, _SP, _DIR, _ACC, _SIG = "loc_x", "loc_y", "speed", "direction", "acceleration", "still_in_the_game"

# register-resident top-K specialisations compiled into the code object
_K_SPECIALISATIONS = (2, 4, 6, 8, 10, 12, 16, 24, 32)

f32 = np.float32


class TagContinuous(CUDAEnvironmentContext):
    name = "TagContinuous"
    RESET_IS_DETERMINISTIC = True  # reset() restarts from the positions drawn in the constructor: no random draw

    def __init__(self, num_taggers=1, num_runners=10, grid_length=10.0, episode_length=100,
                 starting_location_x=None, starting_location_y=None, starting_directions=None, seed=None,
                 max_speed=1.0, skill_level_runner=1.0, skill_level_tagger=1.0, max_acceleration=1.0,
                 min_acceleration=-1.0, max_turn=np.pi / 2, min_turn=-np.pi / 2, num_acceleration_levels=10,
                 num_turn_levels=10, edge_hit_penalty=-0.0, use_full_observation=True,
                 num_other_agents_observed=2, tagging_distance=0.01, tag_reward_for_tagger=1.0,
                 step_penalty_for_tagger=-0.0, tag_penalty_for_runner=-1.0, step_reward_for_runner=0.0,
                 end_of_game_reward_for_runner=1.0, runner_exits_game_after_tagged=True, env_backend="cpu"):
        super().__init__()
        self.float_dtype, self.int_dtype = np.float32, np.int32
        self.eps = f32(1e-10)
        assert num_taggers > 0 and num_runners > 0 and episode_length > 0 and grid_length > 0
        self.num_taggers, self.num_runners = num_taggers, num_runners
        self.num_agents = n = num_taggers + num_runners
        self.episode_length = episode_length
        self.grid_length = f32(grid_length)
        self.grid_diagonal = self.grid_length * np.sqrt(2)  # float32 * float64 -> float64
        assert edge_hit_penalty <= 0
        self.edge_hit_penalty = f32(edge_hit_penalty)

        # -- seeded start: same np.random call order as the reference (:152-195)
        self.np_random = np.random
        if seed is not None:
            self.seed(seed)
        tagger_ids = set(self.np_random.choice(np.arange(n), num_taggers, replace=False))
        self.agent_type = {a: int(a in tagger_ids) for a in range(n)}  # 1 = tagger, 0 = runner
        self.taggers = {a: True for a in range(n) if self.agent_type[a] == 1}
        self.runners = {a: True for a in range(n) if self.agent_type[a] == 0}
        if starting_location_x is None:
            assert starting_location_y is None
            starting_location_x = self.grid_length * self.np_random.rand(n)
            starting_location_y = self.grid_length * self.np_random.rand(n)
        else:
            assert len(starting_location_x) == n and len(starting_location_y) == n
        if starting_directions is None:
            starting_directions = self.np_random.choice([0, np.pi / 2, np.pi, np.pi * 3 / 2], n, replace=True)
        else:
            assert len(starting_directions) == n
        self.starting_location_x = starting_location_x
        self.starting_location_y = starting_location_y
        self.starting_directions = starting_directions
        self.starting_speeds = np.zeros(n, dtype=f32)
        self.starting_accelerations = np.zeros(n, dtype=f32)

        # -- action tables: index 0 is the no-op, then `levels` values from min to max (:219-232)
        self.max_speed = f32(max_speed)
        assert num_acceleration_levels >= 0 and num_turn_levels >= 0
        self.num_acceleration_levels, self.num_turn_levels = num_acceleration_levels, num_turn_levels
        self.max_acceleration, self.min_acceleration = f32(max_acceleration), f32(min_acceleration)
        self.max_turn, self.min_turn = f32(max_turn), f32(min_turn)
        self.acceleration_actions = np.insert(
            np.linspace(self.min_acceleration, self.max_acceleration, num_acceleration_levels), 0, 0).astype(f32)
        self.turn_actions = np.insert(np.linspace(self.min_turn, self.max_turn, num_turn_levels), 0, 0).astype(f32)

        types = np.array([self.agent_type[a] for a in range(n)])
        self._types = types.astype(np.int32)
        self.skill_levels = list((types * f32(skill_level_tagger) + (1 - types) * f32(skill_level_runner)).astype(f32))
        self.runner_exits_game_after_tagged = runner_exits_game_after_tagged

        self.timestep = None
        self.global_state = None
        self.observation_space = None  # set by EnvWrapper
        self.action_space = {a: spaces.MultiDiscrete((len(self.acceleration_actions), len(self.turn_actions)))
                             for a in range(n)}
        self.use_full_observation = use_full_observation
        assert num_other_agents_observed <= n
        self.num_other_agents_observed = num_other_agents_observed
        assert 0 <= tagging_distance <= 1
        self.distance_margin_for_reward = f32(tagging_distance * self.grid_length)
        assert tag_reward_for_tagger >= 0 and step_penalty_for_tagger <= 0
        assert tag_penalty_for_runner <= 0 and step_reward_for_runner >= 0 and end_of_game_reward_for_runner >= 0
        self.tag_reward_for_tagger = f32(tag_reward_for_tagger)
        self.step_penalty_for_tagger = f32(step_penalty_for_tagger)
        self.tag_penalty_for_runner = f32(tag_penalty_for_runner)
        self.step_reward_for_runner = f32(step_reward_for_runner)
        self.step_rewards = list((types * self.step_penalty_for_tagger + (1 - types) * self.step_reward_for_runner).astype(f32))
        self.end_of_game_reward_for_runner = f32(end_of_game_reward_for_runner)
        self.edge_hit_reward_penalty = None
        self.still_in_the_game = None
        self.env_backend = env_backend
        self.runners_at_reset = copy.deepcopy(self.runners)

    def seed(self, seed=None):
        self.np_random.seed(seed)
        return [seed]

    # ---------------------------------------------------------------------- host reset
    def reset(self):
        self.timestep = 0
        n, T1 = self.num_agents, self.episode_length + 1
        gs = {k: np.zeros((T1, n), dtype=f32) for k in (_LOC_X, _LOC_Y, _SP, _DIR, _ACC)}
        gs[_LOC_X][0] = self.starting_location_x
        gs[_LOC_Y][0] = self.starting_location_y
        gs[_SP][0] = self.starting_speeds
        gs[_DIR][0] = self.starting_directions
        gs[_ACC][0] = self.starting_accelerations
        gs[_SIG] = np.ones((T1, n), dtype=np.int32)
        self.global_state = gs
        self.still_in_the_game = np.ones(n, dtype=np.int32)
        self.edge_hit_reward_penalty = np.zeros(n, dtype=f32)
        self.runners = copy.deepcopy(self.runners_at_reset)
        self.num_runners = len(self.runners)
        return self.generate_observation()

    # ------------------------------------------------------------------- host kinematics
    def update_state(self, delta_accelerations, delta_turns):
        t, gs, sig = self.timestep, self.global_state, self.still_in_the_game
        two_pi = f32(2 * np.pi)
        direction = (((gs[_DIR][t - 1] + delta_turns) % two_pi) * sig).astype(f32)
        acc = gs[_ACC][t - 1] + delta_accelerations
        vmax = (self.max_speed * np.array(self.skill_levels)).astype(f32)
        speed = (np.clip(gs[_SP][t - 1] + acc, f32(0.0), vmax) * sig).astype(f32)
        acc = (acc * (speed > 0) * (speed < vmax)).astype(f32)
        x = (gs[_LOC_X][t - 1] + speed * np.cos(direction)).astype(f32)
        y = (gs[_LOC_Y][t - 1] + speed * np.sin(direction)).astype(f32)
        L = self.grid_length
        outside = ~((x >= 0) & (x <= L) & (y >= 0) & (y <= L))
        self.edge_hit_reward_penalty = (self.edge_hit_penalty * outside).astype(f32)
        gs[_LOC_X][t] = np.clip(x, f32(0.0), L)
        gs[_LOC_Y][t] = np.clip(y, f32(0.0), L)
        gs[_SP][t], gs[_DIR][t], gs[_ACC][t] = speed, direction, acc

    # ------------------------------------------------------------------- host observation
    def _neighbour_table(self):
        """[n, K] ids of the K nearest other agents still in the game, ordered by
        (distance, id); -1 pads rows with fewer candidates."""
        t, n, K = self.timestep, self.num_agents, self.num_other_agents_observed
        x, y = self.global_state[_LOC_X][t], self.global_state[_LOC_Y][t]
        dx, dy = x[:, None] - x[None, :], y[:, None] - y[None, :]
        dist = np.sqrt(dx * dx + dy * dy).astype(f32)
        dist[:, self.still_in_the_game == 0] = np.inf
        dist[np.arange(n), np.arange(n)] = np.inf
        if K > n:
            dist = np.concatenate([dist, np.full((n, K - n), np.inf, dtype=f32)], axis=1)
        order = np.argsort(dist, axis=1, kind="stable")[:, :K]
        return np.where(np.isinf(np.take_along_axis(dist, order, axis=1)), -1, order)

    def generate_observation(self):
        t, n, gs = self.timestep, self.num_agents, self.global_state
        sig = self.still_in_the_game
        div = f32(self.max_speed + self.eps)
        feats = np.stack([
            gs[_LOC_X][t].astype(np.float64) / self.grid_diagonal,
            gs[_LOC_Y][t].astype(np.float64) / self.grid_diagonal,
            (gs[_SP][t] / div).astype(np.float64),
            (gs[_ACC][t] / div).astype(np.float64),
            (gs[_DIR][t] / f32(2 * np.pi)).astype(np.float64),
        ])  # [5, n]
        types = self._types.astype(np.float64)
        time = float(t) / self.episode_length
        obs = {}
        if self.use_full_observation:
            for a in range(n):
                others = np.array([j for j in range(n) if j != a], dtype=np.int64)
                rel = (feats[:, others] - feats[:, [a]]) if sig[a] else np.zeros((5, n - 1))
                block = np.vstack([rel, types[others], sig[others].astype(np.float64)])
                obs[a] = np.concatenate([block.reshape(-1), [time if sig[a] else 0.0]])
            return obs
        K = self.num_other_agents_observed
        nbr = self._neighbour_table()
        for a in range(n):
            row = np.zeros((7, K))
            if sig[a]:
                ids = nbr[a][nbr[a] >= 0]
                m = len(ids)
                row[:5, :m] = feats[:, ids] - feats[:, [a]]
                row[5, :m] = types[ids]
                row[6, :m] = sig[ids]
            obs[a] = np.concatenate([row.reshape(-1), [time if sig[a] else 0.0]])
        return obs

    # ------------------------------------------------------------------------ host rewards
    def compute_reward(self):
        t, n, gs = self.timestep, self.num_agents, self.global_state
        rew = {a: 0.0 for a in range(n)}
        for a in range(n):
            if self.still_in_the_game[a]:
                rew[a] += self.edge_hit_reward_penalty[a]
                rew[a] += self.step_rewards[a]
        taggers = sorted(self.taggers)
        runners = sorted(self.runners)
        if runners:
            x, y = gs[_LOC_X][t], gs[_LOC_Y][t]
            dx = x[runners][:, None] - x[taggers][None, :]
            dy = y[runners][:, None] - y[taggers][None, :]
            dist = np.sqrt(dx * dx + dy * dy)
            closest = dist.argmin(axis=1)
            for idx, runner in enumerate(runners):
                if dist[idx, closest[idx]] < self.distance_margin_for_reward:
                    rew[runner] += self.tag_penalty_for_runner
                    rew[taggers[closest[idx]]] += self.tag_reward_for_tagger
                    if self.runner_exits_game_after_tagged:
                        self.still_in_the_game[runner] = 0
                        del self.runners[runner]
                        self.num_runners -= 1
                        gs[_SIG][t:, runner] = 0
        if t == self.episode_length:
            for runner in self.runners:
                rew[runner] += self.end_of_game_reward_for_runner
        return rew

    # --------------------------------------------------------------------- device data
    def get_data_dictionary(self):
        n, K = self.num_agents, self.num_other_agents_observed
        feed = DataFeed()
        for key in (_LOC_X, _LOC_Y, _SP, _DIR, _ACC):
            feed.add_data(name=key, data=self.global_state[key][0], save_copy_and_apply_at_reset=True)
        feed.add_data(name="agent_types", data=[self.agent_type[a] for a in range(n)])
        feed.add_data(name="num_runners", data=self.num_runners, save_copy_and_apply_at_reset=True)
        feed.add_data(name="num_other_agents_observed", data=K)
        feed.add_data(name="grid_length", data=self.grid_length)
        feed.add_data(name="edge_hit_reward_penalty", data=self.edge_hit_reward_penalty,
                      save_copy_and_apply_at_reset=True)
        feed.add_data(name="step_rewards", data=self.step_rewards)
        feed.add_data(name="edge_hit_penalty", data=self.edge_hit_penalty)
        feed.add_data(name="max_speed", data=self.max_speed)
        feed.add_data(name="acceleration_actions", data=self.acceleration_actions)
        feed.add_data(name="turn_actions", data=self.turn_actions)
        feed.add_data(name="num_acceleration_actions", data=len(self.acceleration_actions))
        feed.add_data(name="num_turn_actions", data=len(self.turn_actions))
        feed.add_data(name="skill_levels", data=self.skill_levels)
        feed.add_data(name="use_full_observation", data=self.use_full_observation)
        feed.add_data(name="distance_margin_for_reward", data=self.distance_margin_for_reward)
        feed.add_data(name="tag_reward_for_tagger", data=self.tag_reward_for_tagger)
        feed.add_data(name="tag_penalty_for_runner", data=self.tag_penalty_for_runner)
        feed.add_data(name="end_of_game_reward_for_runner", data=self.end_of_game_reward_for_runner)
        # The reference registers two [N, N-1] scratch arrays the CUDA kernel sorts in HBM
        # (and resets: 2 x 43.7 KB per replica at N = 105).  The HIP kernel selects neighbours
        # in registers, so they shrink to one-element placeholders that keep the names valid.
        feed.add_data(name="neighbor_distances", data=np.zeros((1,), dtype=np.float32))
        feed.add_data(name="neighbor_ids_sorted_by_distance", data=np.zeros((1,), dtype=np.int32))
        feed.add_data(name="nearest_neighbor_ids", data=np.zeros((n, K), dtype=np.int32),
                      save_copy_and_apply_at_reset=True)
        # 1 = the agent's observation row in HBM is all zeros already (it left the game on an earlier tick):
        # rows of agents out of the game stay zero until the episode restarts, so the sparse form of the row
        # gather clears them once instead of rewriting them every tick (device-side bookkeeping; restored with
        # the other arrays at a reset)
        feed.add_data(name="obs_rows_cleared", data=np.zeros((n,), dtype=np.int32), save_copy_and_apply_at_reset=True)
        # Replicas of more than 128 agents: the ids (16 bits each, 0xffff = none) of every agent's K + 3 nearest
        # others of the previous tick, 32 bytes per agent -- the hint the prefiltered neighbour search starts from
        # (tc_pre_pass1 / tc_pre_pass2 in csrc/kernels/tc_knn.h).  Only a hint: the kernel checks the radius it derives from it, so
        # the content never changes a result.  (Registered like per-replica state, which is what makes the wrapper
        # allocate one copy per replica; a reset then restores "none", and the first tick of the new episode searches
        # without a radius.)
        big = self._fast_path() and n > 128
        feed.add_data(name="knn_prev", data=np.full((n, 8) if big else (1, 8), -1, dtype=np.int32),
                      save_copy_and_apply_at_reset=True)
        feed.add_data(name="runner_exits_game_after_tagged", data=self.runner_exits_game_after_tagged)
        feed.add_data(name="still_in_the_game", data=self.still_in_the_game, save_copy_and_apply_at_reset=True)
        return feed

    def derived_device_state(self):
        """`obs_rows_cleared` is valid only while the kernels and the reset paths are the sole writers of the
        observations and of still_in_the_game: a host write to either zero-fills it (EnvWrapper registers this with
        the data manager; after a direct write through a torch tensor call `dm.invalidate_derived(name)`)."""
        return {"obs_rows_cleared": (_OBSERVATIONS, _SIG)}

    _STEP_ARGS = [
        _LOC_X, _LOC_Y, _SP, _DIR, _ACC, "agent_types", "edge_hit_reward_penalty", "edge_hit_penalty",
        "grid_length", "acceleration_actions", "turn_actions", "max_speed", "num_other_agents_observed",
        "skill_levels", "runner_exits_game_after_tagged", "still_in_the_game", "use_full_observation",
        _OBSERVATIONS, _ACTIONS, "neighbor_distances", "neighbor_ids_sorted_by_distance",
        "nearest_neighbor_ids", _REWARDS, "step_rewards", "num_runners", "distance_margin_for_reward",
        "tag_reward_for_tagger", "tag_penalty_for_runner", "end_of_game_reward_for_runner", "_done_",
        "_timestep_", ("n_agents", "meta"), ("episode_length", "meta"), ("n_envs", "meta"),
        "num_acceleration_actions", "num_turn_actions", "obs_rows_cleared", "knn_prev",
    ]  # + kEnvBegin appended by step_launch / tick_launch

    FAST_PATH_MAX_AGENTS = 1024  # tc_fast_impl: 7 id bits in the search keys up to 128 agents, 9 up to 512 (blocks of up to
                                 # 512 threads), 10 up to 1024 (the `_N1024` entries: blocks of 1024 threads)
    _K_SPECIALISATIONS_N1024 = (4, 8, 10, 12, 16)
    STAGE_TARGET_BYTES = 5400    # WD_TC_STAGE_TARGET in tag_continuous.hip

    def _fast_path(self):
        ks = _K_SPECIALISATIONS if self.num_agents <= 512 else self._K_SPECIALISATIONS_N1024
        return (not self.use_full_observation and self.num_agents <= self.FAST_PATH_MAX_AGENTS
                and 1 <= self.num_other_agents_observed <= ks[-1])

    def resolve_step_function_name(self, default_name):
        """The register-resident top-K specialisation that covers K (N <= 512, partial obs), else the
        generic kernel."""
        if not self._fast_path():
            return default_name
        big = self.num_agents > 512
        K = self.num_other_agents_observed
        # the shape with its sizes folded at compile time, when the build has one (warp_drive_amd/build.py UNITS: the
        # BASELINE shape, 105 agents / K = 10 / 21-way heads); WD_TC_SHAPE_ENTRIES=0 keeps the runtime-size entries
        shaped = f"{default_name}_K{K}_N{self.num_agents}A{len(self.acceleration_actions)}"
        fm = getattr(self, "cuda_function_manager", None)
        if self.SHAPE_ENTRIES and len(self.acceleration_actions) == len(self.turn_actions) and fm is not None:
            threads = self._geometry()[1][0]
            if fm.has_function(shaped) and self._shape_entry_threads(fm, shaped) == threads:
                return shaped
            if (self.JIT_SHAPE_ENTRIES and self.num_agents <= 128 and K in _K_SPECIALISATIONS and not fm.has_function(shaped)):
                # opt-in: compile this shape's entries now (one ~15 s hipcc, kept for later runs), as the reference
                # compiles its templated source for every run (pycuda_function_manager.py:133-232)
                from warp_drive_amd import build as wd_build
                from warp_drive_amd.managers import hip_driver as drv

                wd_build.build_shape_unit(K, self.num_agents, len(self.acceleration_actions), threads)
                drv.reload_manifest()
                if fm.has_function(shaped) and self._shape_entry_threads(fm, shaped) == threads:
                    return shaped
        for k in (self._K_SPECIALISATIONS_N1024 if big else _K_SPECIALISATIONS):
            if k >= K:
                return f"{default_name}_K{k}" + ("_N1024" if big else "_N512" if self.num_agents > 128 else "")
        return default_name

    SHAPE_ENTRIES = os.environ.get("WD_TC_SHAPE_ENTRIES", "1") != "0"
    # threads per block the prebuilt shape-specialised entries are compiled for (-DWD_TC_SHAPE_THREADS, warp_drive_amd/build.py);
    # objects built on demand carry theirs in the file name (wd_kernels_tc_k<k>_n<n>a<a>t<threads>.hsaco)
    SHAPE_ENTRY_THREADS = {"HipTagContinuousStep_K10_N105A21": 128}
    JIT_SHAPE_ENTRIES = os.environ.get("WD_TC_JIT_SHAPES", "0") == "1"

    def _shape_entry_threads(self, fm, entry):
        if entry in self.SHAPE_ENTRY_THREADS:
            return self.SHAPE_ENTRY_THREADS[entry]
        from warp_drive_amd import build as wd_build

        m = wd_build.SHAPE_UNIT_PATTERN.match(os.path.basename(fm.code_object_path_of(entry) or ""))
        return int(m.group(4)) if m else -1

    def lds_bytes(self, epb, fused=False, threads=None):
        """dynamic LDS of HipTagContinuousStep / Tick for `epb` packed replicas (tc_carve_fast /
        tc_carve_generic in the kernel file)"""
        N = self.num_agents
        A = epb * N
        K = 0 if self.use_full_observation else self.num_other_agents_observed

        def align16(v):
            return (v + 15) // 16 * 16

        if self._fast_path():
            F = 7 * K + 1
            n_waves = ((A if threads is None else threads) + 63) // 64
            stage_rows = max(1, min(64, (self.STAGE_TARGET_BYTES // 2 if n_waves > 4 else self.STAGE_TARGET_BYTES) // (4 * F)))
            stage_dwords = align16(4 * stage_rows * F) // 4 + 4 + 16   # row images + the list of live rows
            if epb == 1 and N > 128:  # + room for the prefiltered search's candidate lists (WD_TC_LIST_DWORDS, tc_knn.h)
                stage_dwords = max(stage_dwords, 1152)
            area = 32 * A + 8 * epb * ((N + 3) // 4 * 4 + 8) + 4 * A + 4 * A   # features, padded positions, 2 flag arrays
            if epb == 1:  # packed positions of the agents in the game + the packed-index -> id table
                area = align16(area) + 8 * ((N + 3) // 4 * 4 + 8) + align16(2 * (N + 1))
            area = align16(area + 2 * A * K) + 4 * stage_dwords * n_waves  # 16-bit neighbour ids, staging
        else:
            area = 32 * A + align16(4 * A * max(K, 1)) + 4 * 4 * A
        if fused:  # the work area doubles as the two probability slabs (global_load_lds targets); replicas of more
            # than 256 agents sample the heads one after the other from ONE slab (tc_one_slab in the kernel file)
            slabs = (align16(4 * A * len(self.acceleration_actions)), align16(4 * A * len(self.turn_actions)))
            area = max(area, max(slabs) if N > 256 else sum(slabs))
        tables = 4 * N + 4 * (2 * 64 + 32) + 4 * 4 * epb + 16   # tagger list, action tables, per-wavefront counts, per-replica scalars
        if self._fast_path() and N > 128:  # + the 64 cell counters of the cell-sorted neighbour search (TcTables::cell_cnt)
            tables += 4 * 64
        return align16(area) + tables

    def _geometry(self):
        """(replicas per block, block, grid): whole replicas packed into blocks of at most 256 threads
        with the fewest idle lanes -- 105 agents: one replica per 128-thread block (two wavefronts),
        5 agents: 51 replicas per 256-thread block.  ONE trip per block: grid = ceil(replicas / epb)
        (tc_fast_impl relies on it).  Full observations prefer the largest such block: that phase is
        bound by the store path and 4-wave blocks keep twice the stores in flight
        (scripts/write_pattern_probe.py)."""
        return self.cuda_function_manager.packed_geometry(
            self.num_agents, max_threads=256, prefer_large=bool(self.use_full_observation))

    def _range_args(self, env_range):
        """step-kernel arguments for replicas [begin, end): kNumEnvs carries `end`, kEnvBegin `begin`"""
        args = list(self.cuda_step_function_feed(self._STEP_ARGS))
        epb, block, grid = self._geometry()
        if env_range is None:
            return args + [np.int32(0)], epb, block, grid
        begin, end = int(env_range[0]), int(env_range[1])
        n_envs_pos = self._STEP_ARGS.index(("n_envs", "meta"))
        args[n_envs_pos] = np.int32(end)
        grid = (max(1, min(grid[0], (end - begin + epb - 1) // epb)), 1)
        return args + [np.int32(begin)], epb, block, grid

    def step_launch(self, env_range=None):
        """(function, args, block, grid, shared_bytes) of one device tick (optionally of a replica range)."""
        args, epb, block, grid = self._range_args(env_range)
        return self.cuda_step, args, block, grid, self.lds_bytes(epb, threads=block[0])

    LDS_PER_WORKGROUP = 160 * 1024  # gfx950

    def can_fuse_tick(self):
        """False when the fused tick's LDS image (work area or the two probability slabs, whichever is
        larger) does not fit a workgroup -- e.g. ~1000 agents with 21-way action heads; the rollout then
        uses the separate sampler / step / reset launches."""
        epb, block, _ = self._geometry()
        return self.lds_bytes(epb, fused=True, threads=block[0]) <= self.LDS_PER_WORKGROUP

    def has_presampled_tick(self):
        """True when this shape has a `TickA` entry: step + reset of finished replicas on actions that are already in
        `sampled_actions` (drawn by the policy forward's epilogue, training/policy_kernel.py)"""
        name = self.cuda_step.name.replace("Step", "TickA")
        return self._fast_path() and self.cuda_function_manager.has_function(name)

    def tick_launch(self, sampler, probabilities, resetter, env_range=None):
        """Fused rollout tick: sample both action heads + step + reset finished replicas in ONE
        launch (HipTagContinuousTick[_K<k>]).  probabilities = [acceleration, turn] float32 CUDA
        tensors [E, N, n_actions].  `_done_` stays set for replicas that finished on the tick
        (already reset); the next tick clears it.
        probabilities = None: the actions are NOT drawn here -- they are in `sampled_actions` already (the policy
        forward's epilogue drew them, same counters, same search) -- and the launch is the `TickA` entry: step +
        reset, nothing fetched or sampled."""
        from warp_drive_amd.managers.function_manager import _stream_tag

        fm, dm = self.cuda_function_manager, self.cuda_data_manager
        presampled = probabilities is None
        name = self.cuda_step.name.replace("Step", "TickA" if presampled else "Tick")
        fm.initialize_functions([name])
        fn = fm.get_function(name)
        _, reset_args, _, _ = resetter.fused_launch(dm, 0, 0)  # builds / refreshes the descriptor table
        table, n_arrays = reset_args[0], reset_args[1]
        args, epb, block, grid = self._range_args(env_range)
        if presampled:
            null = np.uint64(0)
            args = args + [null, null, null, table, n_arrays, _stream_tag("tick")]
            return fn, args, block, grid, self.lds_bytes(epb, fused=False, threads=block[0])
        assert len(probabilities) == 2
        args = args + [sampler.rng_state, probabilities[0], probabilities[1], table, n_arrays, _stream_tag("tick")]
        return fn, args, block, grid, self.lds_bytes(epb, fused=True, threads=block[0])

    # ------------------------------------------------------------------------------ step
    def step(self, actions=None):
        self.timestep += 1
        if self.env_backend != "cpu":
            fn, args, block, grid, shared = self.step_launch()
            fn(*args, block=block, grid=grid, shared=shared)
            return None
        assert isinstance(actions, dict) and len(actions) == self.num_agents
        acc_ids = [actions[a][0] for a in range(self.num_agents)]
        turn_ids = [actions[a][1] for a in range(self.num_agents)]
        assert all(0 <= i <= self.num_acceleration_levels for i in acc_ids)
        assert all(0 <= i <= self.num_turn_levels for i in turn_ids)
        self.update_state(self.acceleration_actions[acc_ids], self.turn_actions[turn_ids])
        obs = self.generate_observation()  # before tagging updates still_in_the_game
        rew = self.compute_reward()
        done = {"__all__": (self.timestep >= self.episode_length) or (self.num_runners == 0)}
        return obs, rew, done, {}
