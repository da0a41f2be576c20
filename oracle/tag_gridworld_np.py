"""numpy oracle for TagGridWorld, batched over env replicas (test infrastructure).

Restates reference example_envs/tag_gridworld/tag_gridworld.py:
  update_state           :152-192
  generate_observation   :194-275
  reset                  :277-289
  step                   :291-317
State is int32 exactly like the reference's global_state (:283-288).  Rewards
and observations are computed in float64 (as the reference does, since its
scalars are Python floats) and exposed both as float64 and as the float32 view
the device holds (DataManager narrows to 32 bit, data_manager.py:263-269).
"""
import numpy as np

# reference tag_gridworld.py:104
STEP_ACTIONS = np.array([[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1]], dtype=np.int64)


class TagGridWorldOracle:
    def __init__(
        self,
        num_envs,
        num_taggers=10,
        grid_length=10,
        episode_length=100,
        starting_location_x=None,
        starting_location_y=None,
        wall_hit_penalty=0.1,
        tag_reward_for_tagger=10.0,
        tag_penalty_for_runner=2.0,
        step_cost_for_tagger=0.01,
        use_full_observation=True,
        seed=None,
    ):
        self.E = int(num_envs)
        self.num_taggers = int(num_taggers)
        self.N = self.num_taggers + 1  # :66 exactly one runner, the last agent
        self.L = grid_length
        self.T = int(episode_length)
        if starting_location_x is None:  # :89-96
            starting_location_x = int(0.5 * grid_length) * np.ones(self.N)
            starting_location_x[-1] = 0
            starting_location_y = int(0.5 * grid_length) * np.ones(self.N)
            starting_location_y[-1] = 0
        self.start_x = np.asarray(starting_location_x).astype(np.int32)
        self.start_y = np.asarray(starting_location_y).astype(np.int32)
        self.wall_hit_penalty = wall_hit_penalty
        self.tag_reward_for_tagger = tag_reward_for_tagger
        self.tag_penalty_for_runner = tag_penalty_for_runner
        self.step_cost_for_tagger = step_cost_for_tagger
        self.use_full_observation = bool(use_full_observation)
        # :81-87  taggers are type 0, the runner is type 1
        self.agent_types = np.array([0] * self.num_taggers + [1], dtype=np.int64)
        self.reset_all()

    # ------------------------------------------------------------------ reset
    def reset_all(self):
        self.loc_x = np.tile(self.start_x, (self.E, 1))
        self.loc_y = np.tile(self.start_y, (self.E, 1))
        self.timestep = np.zeros(self.E, dtype=np.int32)
        self.done = np.zeros(self.E, dtype=np.int32)
        self.obs = self.generate_observation()
        self.obs_at_reset = self.obs.copy()
        self.rewards = np.zeros((self.E, self.N), dtype=np.float64)
        return self.obs

    def reset_done_envs(self):
        """Device-side reset semantics: reset.cu:9-75 applied to loc_x, loc_y and
        the observations placeholder (data_loader.py:360-364), then undo done."""
        m = self.done > 0
        self.loc_x[m] = self.start_x
        self.loc_y[m] = self.start_y
        self.obs[m] = self.obs_at_reset[m]
        self.timestep[m] = 0
        self.done[m] = 0

    # ------------------------------------------------------------ observation
    def generate_observation(self):
        E, N, L = self.E, self.N, self.L
        t_frac = self.timestep.astype(np.float64) / self.T  # :222  float(t)/T
        if self.use_full_observation:  # :196-224
            nx = self.loc_x / L  # int32 / python number -> float64
            ny = self.loc_y / L
            obs = np.empty((E, N, 4 * N + 1), dtype=np.float64)
            obs[:, :, 0:N] = nx[:, None, :]
            obs[:, :, N:2 * N] = ny[:, None, :]
            obs[:, :, 2 * N:3 * N] = self.agent_types[None, None, :]
            obs[:, :, 3 * N:4 * N] = np.eye(N)[None]
            obs[:, :, 4 * N] = t_frac[:, None]
            return obs
        # partial observation :225-274 -> 6 features
        obs = np.empty((E, N, 6), dtype=np.float64)
        obs[:, :, 0] = self.loc_x / L
        obs[:, :, 1] = self.loc_y / L
        # taggers see the runner (last agent)
        obs[:, : N - 1, 2] = (self.loc_x[:, -1] / L)[:, None]
        obs[:, : N - 1, 3] = (self.loc_y[:, -1] / L)[:, None]
        # the runner sees the closest tagger: integer squared distance, first argmin
        d = np.square(self.loc_x[:, :-1] - self.loc_x[:, -1:]) + np.square(
            self.loc_y[:, :-1] - self.loc_y[:, -1:]
        )
        j = np.argmin(d, axis=1)
        ar = np.arange(E)
        obs[:, N - 1, 2] = self.loc_x[ar, j] / L
        obs[:, N - 1, 3] = self.loc_y[ar, j] / L
        obs[:, :, 4] = self.agent_types[None, :]
        obs[:, :, 5] = t_frac[:, None]
        return obs

    # ------------------------------------------------------------------- step
    def step(self, actions):
        """actions: int [E, N] (or [E, N, 1]) in {0..4}."""
        a = np.asarray(actions).reshape(self.E, self.N)
        self.timestep = self.timestep + 1  # :295
        ax = STEP_ACTIONS[a, 0]
        ay = STEP_ACTIONS[a, 1]
        x = self.loc_x + ax  # int32 + int64 -> int64  (:156-157)
        y = self.loc_y + ay
        cx = np.clip(x, 0, self.L)
        cy = np.clip(y, 0, self.L)
        penalty = -1.0 * self.wall_hit_penalty * ((x != cx) | (y != cy))  # :163-170
        self.loc_x = cx.astype(np.int32)
        self.loc_y = cy.astype(np.int32)
        T_ = self.num_taggers
        tag = ((cx[:, :T_] == cx[:, -1:]) & (cy[:, :T_] == cy[:, -1:])).any(axis=1)  # :175-178
        reward_tag = np.empty((self.E, self.N), dtype=np.float64)
        reward_tag[:, :T_] = np.where(
            tag[:, None], self.tag_reward_for_tagger, -1.0 * self.step_cost_for_tagger
        )
        reward_tag[:, -1] = np.where(
            tag, -1.0 * self.tag_penalty_for_runner, 1.0 * self.step_cost_for_tagger
        )
        self.rewards = reward_tag + penalty  # :187
        self.obs = self.generate_observation()
        self.done = ((self.timestep >= self.T) | tag).astype(np.int32)  # :314
        return self.obs, self.rewards, self.done


# ---------------------------------------------------------------------------------------------------------------
# The live-policy rollout kernel (csrc/kernels/tag_gridworld_n5.hip, HipTagGridWorldRollout_N5_H32 / _H64) evaluates a
# small two-hidden-layer policy per agent and tick.  Float32 restatement of that forward on the PACKED weights
# (training/policy_kernel.py::pack_gridworld_policy), test infrastructure like cartpole_np.py::policy_probabilities.
POLICY_IN, POLICY_IN_STRIDE, POLICY_ACTIONS = 21, 24, 5


def policy_probabilities(packed, hidden, obs):
    """obs [R, 21] float32 -> probabilities [R, 5] float32: acc = bias, one fused multiply-add per input in index
    order (emulated in float64: the product of two float32 is exact there), ReLU, softmax with the maximum
    subtracted, float32 running sum of the exponentials, one division per action -- what gw5_policy_cum computes"""
    f32, H = np.float32, int(hidden)
    w = np.asarray(packed, dtype=f32)
    o = 0
    W0 = w[o:o + H * POLICY_IN_STRIDE].reshape(H, POLICY_IN_STRIDE)[:, :POLICY_IN]; o += H * POLICY_IN_STRIDE
    b0 = w[o:o + H]; o += H
    W1 = w[o:o + H * H].reshape(H, H); o += H * H
    b1 = w[o:o + H]; o += H
    Wp = w[o:o + POLICY_ACTIONS * H].reshape(POLICY_ACTIONS, H); o += POLICY_ACTIONS * H
    bp = w[o:o + POLICY_ACTIONS]

    def layer(x, W, b):
        acc = np.broadcast_to(b, (x.shape[0], W.shape[0])).astype(f32).copy()
        for j in range(W.shape[1]):
            acc = (W[None, :, j].astype(np.float64) * x[:, j:j + 1].astype(np.float64) + acc.astype(np.float64)).astype(f32)
        return acc

    x = np.asarray(obs, dtype=f32)
    h1 = np.maximum(layer(x, W0, b0), f32(0))
    h2 = np.maximum(layer(h1, W1, b1), f32(0))
    logits = layer(h2, Wp, bp)
    e = np.exp((logits - logits.max(axis=1, keepdims=True)).astype(f32)).astype(f32)
    total = np.zeros(e.shape[0], f32)
    for a in range(POLICY_ACTIONS):
        total = (total + e[:, a]).astype(f32)
    return (e / total[:, None]).astype(f32)


def running_sums(p):
    """the float32 running sums the inverse-CDF sampler compares the uniform with (random.cu:51-85)"""
    c = np.zeros_like(p)
    acc = np.zeros(p.shape[0], np.float32)
    for a in range(p.shape[1]):
        acc = p[:, a] if a == 0 else (acc + p[:, a]).astype(np.float32)
        c[:, a] = acc
    return c
