"""numpy oracle for TagContinuous, batched over env replicas (test infrastructure).

Restates reference example_envs/tag_continuous/tag_continuous.py:
  __init__ (seeded start)  :152-305
  update_state             :339-401
  compute_distance         :403-420
  k_nearest_neighbors      :422-444
  generate_observation     :446-610
  compute_reward           :612-678
  reset                    :758-794
  step (CPU branch) / done :853-887

Dtype discipline follows what the reference does under numpy >= 2 (NEP 50), made
explicit with astype so the oracle does not depend on numpy's promotion rules:
  * kinematics are float32 end to end (:355-374), cos/sin are numpy's float32 ufuncs;
  * loc_x/loc_y are normalised in float64 (float32 / np.float64 grid_diagonal, :146,:454);
    speed/acceleration/direction are normalised in float32 (:456-458) and widened;
    the neighbour difference is taken in float64 (:560) and narrowed to float32 when
    pushed to the device (data_manager.py:263-269);
  * compute_distance squares np.float32 *scalars* with `** 2`, which is libm
    powf(x, 2) (NOT x*x: 0.07 % of inputs differ by 1 ulp) -- taken from libm
    through oracle/csrc/wd_oracle.c:wdo_powf2;  compute_reward squares *arrays*
    (`** 2` -> np.square == x*x, :630-641).
"""
import ctypes

import numpy as np

from . import build as _build

f32 = np.float32
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(_build.build())
        _LIB.wdo_powf2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        _LIB.wdo_powf2.restype = None
    return _LIB


def powf2(a):
    a = np.ascontiguousarray(a, dtype=f32)
    out = np.empty_like(a)
    _lib().wdo_powf2(a.ctypes.data, out.ctypes.data, a.size)
    return out


class TagContinuousOracle:
    def __init__(
        self,
        num_envs,
        num_taggers=1,
        num_runners=10,
        grid_length=10.0,
        episode_length=100,
        starting_location_x=None,
        starting_location_y=None,
        starting_directions=None,
        seed=None,
        max_speed=1.0,
        skill_level_runner=1.0,
        skill_level_tagger=1.0,
        max_acceleration=1.0,
        min_acceleration=-1.0,
        max_turn=np.pi / 2,
        min_turn=-np.pi / 2,
        num_acceleration_levels=10,
        num_turn_levels=10,
        edge_hit_penalty=-0.0,
        use_full_observation=True,
        num_other_agents_observed=2,
        tagging_distance=0.01,
        tag_reward_for_tagger=1.0,
        step_penalty_for_tagger=-0.0,
        tag_penalty_for_runner=-1.0,
        step_reward_for_runner=0.0,
        end_of_game_reward_for_runner=1.0,
        runner_exits_game_after_tagged=True,
    ):
        self.E = int(num_envs)
        self.num_taggers = int(num_taggers)
        self.num_runners0 = int(num_runners)
        self.N = N = self.num_taggers + self.num_runners0
        self.T = int(episode_length)
        self.grid_length = f32(grid_length)
        self.grid_diagonal = np.float64(self.grid_length) * np.sqrt(2)  # :146
        self.edge_hit_penalty = f32(edge_hit_penalty)
        # ---- seeded start, same np.random call order as :152-195
        rs = np.random.RandomState(seed) if seed is not None else np.random
        taggers = rs.choice(np.arange(N), self.num_taggers, replace=False)
        self.agent_types = np.zeros(N, dtype=np.int32)
        self.agent_types[np.asarray(taggers)] = 1  # 1 = tagger, 0 = runner  (:165-171)
        if starting_location_x is None:
            starting_location_x = self.grid_length * rs.rand(N)
            starting_location_y = self.grid_length * rs.rand(N)
        if starting_directions is None:
            starting_directions = rs.choice([0, np.pi / 2, np.pi, np.pi * 3 / 2], N, replace=True)
        self.start_x = np.asarray(starting_location_x).astype(f32)  # stored into a f32 array (:337)
        self.start_y = np.asarray(starting_location_y).astype(f32)
        self.start_dir = np.asarray(starting_directions).astype(f32)
        self.max_speed = f32(max_speed)
        # :219-232
        acc = np.linspace(f32(min_acceleration), f32(max_acceleration), num_acceleration_levels)
        self.acceleration_actions = np.insert(acc, 0, 0).astype(f32)
        trn = np.linspace(f32(min_turn), f32(max_turn), num_turn_levels)
        self.turn_actions = np.insert(trn, 0, 0).astype(f32)
        t = self.agent_types
        self.skill_levels = (t * f32(skill_level_tagger) + (1 - t) * f32(skill_level_runner)).astype(f32)
        self.runner_exits = bool(runner_exits_game_after_tagged)
        self.use_full_observation = bool(use_full_observation)
        self.K = int(num_other_agents_observed)
        self.distance_margin_for_reward = f32(f32(tagging_distance) * self.grid_length)  # :271
        self.tag_reward_for_tagger = f32(tag_reward_for_tagger)
        self.tag_penalty_for_runner = f32(tag_penalty_for_runner)
        self.end_of_game_reward_for_runner = f32(end_of_game_reward_for_runner)
        self.step_rewards = (t * f32(step_penalty_for_tagger) + (1 - t) * f32(step_reward_for_runner)).astype(f32)
        self.eps = f32(1e-10)
        self.reset_all()

    @property
    def obs_dim(self):
        return 7 * (self.N - 1) + 1 if self.use_full_observation else 7 * self.K + 1

    # ------------------------------------------------------------------ reset
    def reset_all(self):
        E, N = self.E, self.N
        tile = lambda v: np.tile(v, (E, 1)).copy()
        self.loc_x = tile(self.start_x)
        self.loc_y = tile(self.start_y)
        self.direction = tile(self.start_dir)
        self.speed = np.zeros((E, N), dtype=f32)
        self.acceleration = np.zeros((E, N), dtype=f32)
        self.sig = np.ones((E, N), dtype=np.int32)
        self.edge_pen = np.zeros((E, N), dtype=f32)
        self.num_runners = np.full(E, self.num_runners0, dtype=np.int32)
        self.timestep = np.zeros(E, dtype=np.int32)
        self.done = np.zeros(E, dtype=np.int32)
        self.rewards = np.zeros((E, N), dtype=f32)
        self.obs = self.generate_observation()
        self.obs_at_reset = self.obs.copy()
        return self.obs

    def reset_done_envs(self):
        """Device-side reset semantics (reset.cu:9-75 over every array registered
        with save_copy_and_apply_at_reset, tag_continuous.py:685-755, plus the
        observations placeholder), then undo done / timestep."""
        m = self.done > 0
        if not m.any():
            return
        self.loc_x[m] = self.start_x
        self.loc_y[m] = self.start_y
        self.direction[m] = self.start_dir
        self.speed[m] = 0
        self.acceleration[m] = 0
        self.sig[m] = 1
        self.edge_pen[m] = 0
        self.num_runners[m] = self.num_runners0
        self.obs[m] = self.obs_at_reset[m]
        self.timestep[m] = 0
        self.done[m] = 0

    def set_state(self, **arrays):
        """Overwrite state arrays (used by lock-step re-sync tests)."""
        for k, v in arrays.items():
            cur = getattr(self, k)
            setattr(self, k, np.ascontiguousarray(v, dtype=cur.dtype).reshape(cur.shape).copy())

    # ------------------------------------------------------------ observation
    def _normalised(self):
        nx = self.loc_x.astype(np.float64) / self.grid_diagonal
        ny = self.loc_y.astype(np.float64) / self.grid_diagonal
        div = f32(self.max_speed + self.eps)
        nsp = (self.speed / div).astype(f32).astype(np.float64)
        nac = (self.acceleration / div).astype(f32).astype(np.float64)
        ndir = (self.direction / f32(2 * np.pi)).astype(f32).astype(np.float64)
        return np.stack([nx, ny, nsp, nac, ndir], axis=1)  # [E, 5, N] float64

    def knn(self):
        """ids [E, N, K] (-1 = padding) of the K nearest other agents still in the game,
        ordered by (distance, id); :403-444."""
        E, N, K = self.E, self.N, self.K
        dx = self.loc_x[:, :, None] - self.loc_x[:, None, :]  # x[agent] - x[other], float32
        dy = self.loc_y[:, :, None] - self.loc_y[:, None, :]
        d = np.sqrt(powf2(dx) + powf2(dy)).astype(f32)
        invalid = (self.sig[:, None, :] == 0) | np.eye(N, dtype=bool)[None]
        d = np.where(invalid, np.inf, d)
        if K > N:  # only K == N is legal (assert K <= N, :264): pad so argsort has K columns
            d = np.concatenate([d, np.full((E, N, K - N), np.inf, dtype=d.dtype)], axis=-1)
        order = np.argsort(d, axis=-1, kind="stable")[:, :, :K]
        dk = np.take_along_axis(d, order, axis=-1)
        ids = np.where(np.isinf(dk), -1, order).astype(np.int32)
        self.neighbor_dist = d
        return ids

    def generate_observation(self):
        E, N = self.E, self.N
        feat = self._normalised()  # [E,5,N]
        types = self.agent_types.astype(np.float64)
        sig = self.sig.astype(np.float64)
        tfrac = self.timestep.astype(np.float64) / self.T  # :474
        in_game = self.sig > 0
        if self.use_full_observation:  # :476-519
            M = N - 1
            obs = np.zeros((E, N, 7 * M + 1), dtype=np.float64)
            others = np.array([[j for j in range(N) if j != i] for i in range(N)])  # [N, M]
            for c in range(5):
                diff = feat[:, c, :][:, None, :] - feat[:, c, :][:, :, None]  # [E, i, j] = f[j]-f[i]
                g = np.take_along_axis(diff, np.broadcast_to(others[None], (E, N, M)), axis=2)
                obs[:, :, c * M:(c + 1) * M] = np.where(in_game[:, :, None], g, 0.0)
            obs[:, :, 5 * M:6 * M] = types[others][None]
            obs[:, :, 6 * M:7 * M] = sig[:, others]
            obs[:, :, 7 * M] = np.where(in_game, tfrac[:, None], 0.0)
            self.nearest_ids = None
            return obs
        # partial :520-608
        K = self.K
        ids = self.knn()  # [E,N,K], -1 padding
        self.nearest_ids = ids
        valid = (ids >= 0) & in_game[:, :, None]
        safe = np.where(ids >= 0, ids, 0)
        obs = np.zeros((E, N, 7 * K + 1), dtype=np.float64)
        for c in range(5):
            f = feat[:, c, :]  # [E,N]
            nb = np.take_along_axis(np.broadcast_to(f[:, None, :], (E, N, N)), safe, axis=2)
            obs[:, :, c * K:(c + 1) * K] = np.where(valid, nb - f[:, :, None], 0.0)
        obs[:, :, 5 * K:6 * K] = np.where(valid, types[safe], 0.0)
        sg = np.take_along_axis(np.broadcast_to(sig[:, None, :], (E, N, N)), safe, axis=2)
        obs[:, :, 6 * K:7 * K] = np.where(valid, sg, 0.0)
        obs[:, :, 7 * K] = np.where(in_game, tfrac[:, None], 0.0)
        return obs

    # ------------------------------------------------------------------- step
    def step(self, actions):
        """actions int [E, N, 2] = (acceleration index, turn index)."""
        a = np.asarray(actions).reshape(self.E, self.N, 2)
        self.timestep = self.timestep + 1
        da = self.acceleration_actions[a[..., 0]]
        dt = self.turn_actions[a[..., 1]]
        sig = self.sig
        # ---- update_state :339-401
        direction = ((self.direction + dt) % f32(2 * np.pi) * sig).astype(f32)
        acc = (self.acceleration + da).astype(f32)
        vmax = (self.max_speed * self.skill_levels).astype(f32)[None, :]
        speed = (np.clip(self.speed + acc, f32(0.0), vmax) * sig).astype(f32)
        acc = (acc * (speed > 0) * (speed < vmax)).astype(f32)
        x = (self.loc_x + speed * np.cos(direction)).astype(f32)
        y = (self.loc_y + speed * np.sin(direction)).astype(f32)
        L = self.grid_length
        crossed = ~((x >= 0) & (x <= L) & (y >= 0) & (y <= L))
        self.loc_x = np.clip(x, f32(0.0), L).astype(f32)
        self.loc_y = np.clip(y, f32(0.0), L).astype(f32)
        self.edge_pen = (self.edge_hit_penalty * crossed).astype(f32)
        self.speed, self.direction, self.acceleration = speed, direction, acc
        # ---- observation (before tagging changes still_in_the_game) :876
        self.obs = self.generate_observation()
        # ---- compute_reward :612-678
        self._compute_reward()
        self.done = ((self.timestep >= self.T) | (self.num_runners == 0)).astype(np.int32)
        return self.obs, self.rewards, self.done

    def _compute_reward(self):
        E, N = self.E, self.N
        in_game = self.sig > 0
        rew = np.zeros((E, N), dtype=f32)
        rew = np.where(in_game, (rew + self.edge_pen).astype(f32), rew)
        rew = np.where(in_game, (rew + self.step_rewards[None, :]).astype(f32), rew)
        tag_ids = np.nonzero(self.agent_types == 1)[0]
        is_runner = (self.agent_types == 0)[None, :] & in_game  # self.runners
        rx = self.loc_x[:, :, None] - self.loc_x[:, tag_ids][:, None, :]
        ry = self.loc_y[:, :, None] - self.loc_y[:, tag_ids][:, None, :]
        d = np.sqrt(rx * rx + ry * ry).astype(f32)  # [E, N, T]
        dmin = d.min(axis=2)
        nearest = tag_ids[d.argmin(axis=2)]  # first min wins
        tagged = is_runner & (dmin < self.distance_margin_for_reward)
        rew = np.where(tagged, (rew + self.tag_penalty_for_runner).astype(f32), rew)
        # taggers collect one tag_reward per tagged runner, added sequentially (:660-665)
        cnt = np.zeros((E, N), dtype=np.int64)
        e_idx, r_idx = np.nonzero(tagged)
        np.add.at(cnt, (e_idx, nearest[e_idx, r_idx]), 1)
        for k in range(int(cnt.max()) if cnt.size else 0):
            rew = np.where(cnt > k, (rew + self.tag_reward_for_tagger).astype(f32), rew)
        self.tagged = tagged
        if self.runner_exits:
            self.sig = np.where(tagged, 0, self.sig).astype(np.int32)
            self.num_runners = (self.num_runners - tagged.sum(axis=1)).astype(np.int32)
            still_runner = is_runner & ~tagged
        else:
            still_runner = is_runner
        end = (self.timestep == self.T)[:, None] & still_runner
        rew = np.where(end, (rew + self.end_of_game_reward_for_runner).astype(f32), rew)
        self.rewards = rew.astype(f32)
