"""TagContinuous oracle on the plain-C restatement (oracle/csrc/wd_oracle.c), batched over replicas.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): the same
attributes and methods as oracle/tag_continuous_np.TagContinuousOracle -- which supplies the seeded
start state and the tables (reference tag_continuous.py:152-305) -- but `step` is wdo_tc_step
(tag_continuous.py:339-678, :853-887), fast enough for full-size lock-step runs (2000 replicas x 105
agents per tick in well under a second on one core).  The C step is pinned bit-exactly to the
reference's recorded trajectories by tests/test_oracle_golden.py::test_c_step_matches_reference.
"""
import ctypes

import numpy as np

from . import build as _build
from .tag_continuous_np import TagContinuousOracle, powf2

f32 = np.float32


class _Cfg(ctypes.Structure):  # wdo_tc_cfg
    _fields_ = [("n_envs", ctypes.c_int), ("n_agents", ctypes.c_int), ("episode_length", ctypes.c_int),
                ("K", ctypes.c_int), ("use_full_observation", ctypes.c_int), ("runner_exits", ctypes.c_int),
                ("grid_length", ctypes.c_float), ("max_speed", ctypes.c_float),
                ("edge_hit_penalty", ctypes.c_float), ("margin", ctypes.c_float),
                ("tag_reward", ctypes.c_float), ("tag_penalty", ctypes.c_float), ("end_reward", ctypes.c_float),
                ("n_acc", ctypes.c_int), ("n_turn", ctypes.c_int)]


_STATE = ("loc_x", "loc_y", "speed", "direction", "acceleration", "edge_pen", "sig", "num_runners", "timestep",
          "done")


class TagContinuousCOracle:
    def __init__(self, num_envs, n_threads=1, **cfg):
        # one replica from the numpy oracle gives the seeded start, the tables and the reset observation
        # (every replica starts from the same state, env_wrapper.py:288-304)
        one = TagContinuousOracle(num_envs=1, **cfg)
        self._one = one
        self.E, self.N, self.T, self.K = int(num_envs), one.N, one.T, one.K
        self.n_threads = int(n_threads)
        for k in ("agent_types", "acceleration_actions", "turn_actions", "skill_levels", "step_rewards",
                  "start_x", "start_y", "start_dir", "num_runners0", "use_full_observation", "runner_exits",
                  "grid_length", "max_speed", "edge_hit_penalty", "distance_margin_for_reward",
                  "tag_reward_for_tagger", "tag_penalty_for_runner", "end_of_game_reward_for_runner", "obs_dim"):
            setattr(self, k, getattr(one, k))
        self._lib = ctypes.CDLL(_build.build())
        self._lib.wdo_tc_step.restype = None
        self._lib.wdo_tc_step_ids.restype = None
        self._cfg = _Cfg(self.E, self.N, self.T, self.K, int(self.use_full_observation), int(self.runner_exits),
                         float(self.grid_length), float(self.max_speed), float(self.edge_hit_penalty),
                         float(self.distance_margin_for_reward), float(self.tag_reward_for_tagger),
                         float(self.tag_penalty_for_runner), float(self.end_of_game_reward_for_runner),
                         len(self.acceleration_actions), len(self.turn_actions))
        self.reset_all()

    def reset_all(self):
        E, one = self.E, self._one
        for k in _STATE:
            v = getattr(one, k)
            setattr(self, k, np.ascontiguousarray(np.repeat(v, E, axis=0)))
        self.rewards = np.zeros((E, self.N), f32)
        self.obs_at_reset = np.ascontiguousarray(one.obs_at_reset.astype(f32)[0])
        # (a fresh writable array: for one replica the broadcast is already contiguous and ascontiguousarray would
        # hand back the read-only VIEW of obs_at_reset, which the C step then writes through)
        self.obs = np.array(np.broadcast_to(self.obs_at_reset, (E,) + self.obs_at_reset.shape), dtype=np.float32, order="C")
        return self.obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N, 2)
        P = lambda v: v.ctypes.data_as(ctypes.c_void_p)
        self.sig_before = self.sig.copy()  # observations are generated before this tick's tagging (:876)
        # nearest_ids [E, N, K]: the ids the observation rows were built from, in the reference's (distance, id)
        # order (:422-444); -1 = fewer than K others in the game, or the agent itself is out of it
        if self.use_full_observation:
            self.nearest_ids, ids_p = None, ctypes.c_void_p(None)
        else:
            self.nearest_ids = np.empty((self.E, self.N, self.K), np.int32)
            ids_p = P(self.nearest_ids)
        self._lib.wdo_tc_step_ids(ctypes.byref(self._cfg), P(self.loc_x), P(self.loc_y), P(self.speed),
                                  P(self.direction), P(self.acceleration), P(self.agent_types), P(self.edge_pen),
                                  P(self.acceleration_actions), P(self.turn_actions), P(self.skill_levels),
                                  P(self.sig), P(self.obs), P(a), P(self.rewards), P(self.step_rewards),
                                  P(self.num_runners), P(self.done), P(self.timestep), ids_p,
                                  ctypes.c_int(self.n_threads))

    def run_ticks(self, actions, n_ticks, n_threads=None, chunk=4):
        """`n_ticks` ticks of every replica with the SAME actions each tick, without resets and without the
        neighbour-id output, entirely inside one C call (bench.py's cpu_baseline leg: no per-tick numpy work, replicas
        handed to the threads in chunks of `chunk` with a dynamic schedule, scratch allocated once per thread)."""
        assert self.timestep.max() + n_ticks <= self.T, "run_ticks does not reset: stay inside the episode"
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.N, 2)
        P = lambda v: v.ctypes.data_as(ctypes.c_void_p)
        self._lib.wdo_tc_run_ticks.restype = None
        self._lib.wdo_tc_run_ticks(ctypes.byref(self._cfg), P(self.loc_x), P(self.loc_y), P(self.speed),
                                   P(self.direction), P(self.acceleration), P(self.agent_types), P(self.edge_pen),
                                   P(self.acceleration_actions), P(self.turn_actions), P(self.skill_levels),
                                   P(self.sig), P(self.obs), P(a), P(self.rewards), P(self.step_rewards),
                                   P(self.num_runners), P(self.done), P(self.timestep), ctypes.c_int(int(n_ticks)),
                                   ctypes.c_int(int(chunk)),
                                   ctypes.c_int(int(self.n_threads if n_threads is None else n_threads)))

    def reset_done_envs(self):
        """device-side reset semantics, as TagContinuousOracle.reset_done_envs"""
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
        self.obs[m] = self.obs_at_reset
        self.timestep[m] = 0
        self.done[m] = 0

    def neighbor_distances(self, env, agent):
        """float32 distances of `agent` to every other agent in the game when the last step's observation
        was generated (inf otherwise), as the
        reference's compute_distance evaluates them (numpy scalar ** 2 = libm powf, :403-420)."""
        dx = self.loc_x[env, agent] - self.loc_x[env]
        dy = self.loc_y[env, agent] - self.loc_y[env]
        d = np.sqrt(powf2(dx) + powf2(dy)).astype(f32)
        d[self.sig_before[env] == 0] = np.inf
        d[agent] = np.inf
        return d
