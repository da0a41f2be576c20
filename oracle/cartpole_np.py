"""numpy oracle for the Cartpole Euler step, batched (test infrastructure).

The reference's CPU step is third-party gym.envs.classic_control.CartPoleEnv
(example_envs/single_agent/classic_control/cartpole/cartpole.py:7,21,29; gym is absent and only
pinned as gym>=0.26).  This restates the reference's own device kernel,
cartpole_step_numba.py:5-83, with Numba's dtype flow made explicit: float32 state and
scalars; the Python literal 4.0/3.0 widens the pole-acceleration denominator, thetaacc and
xacc to float64.  cos/sin are numpy's float32 kernels.
PINNED against tests/golden/cp_traj.npz: that kernel's source executed under a numba.cuda
stand-in in the build container (oracle/gen_golden.py::gen_cartpole_traj); floats agree to
2e-6 abs free-running over whole episodes (the stand-in evaluates cos/sin in float64), discrete
outputs exactly (tests/test_oracle_golden.py::test_cartpole_oracle_vs_reference_kernel_source).
"""
import numpy as np

f32 = np.float32


class CartPoleOracle:
    gravity, masscart, masspole, length = 9.8, 1.0, 0.1, 0.5
    force_mag, tau = 10.0, 0.02
    theta_threshold_radians = 12 * 2 * np.pi / 360
    x_threshold = 2.4

    def __init__(self, num_envs, episode_length=500, initial_state=None):
        self.E, self.T = int(num_envs), int(episode_length)
        self.initial_state = np.zeros(4, f32) if initial_state is None else np.asarray(initial_state, f32)
        self.reset_all()

    def reset_all(self):
        self.state = np.tile(self.initial_state, (self.E, 1)).astype(f32)
        self.timestep = np.zeros(self.E, np.int32)
        self.done = np.zeros(self.E, np.int32)
        self.obs = self.state.copy()
        self.rewards = np.zeros(self.E, f32)

    def reset_done_envs(self):
        m = self.done > 0
        self.state[m] = self.initial_state
        self.obs[m] = self.initial_state
        self.timestep[m] = 0
        self.done[m] = 0

    def step(self, actions):
        a = np.asarray(actions).reshape(self.E)
        self.timestep = self.timestep + 1
        x, x_dot, theta, theta_dot = (self.state[:, i].astype(f32) for i in range(4))
        force = np.where(a > 0.5, f32(self.force_mag), f32(-self.force_mag)).astype(f32)
        cos, sin = np.cos(theta), np.sin(theta)
        total_mass = f32(self.masspole + self.masscart)
        pml = f32(self.masspole * self.length)
        temp = ((force + ((pml * (theta_dot * theta_dot)).astype(f32) * sin).astype(f32)).astype(f32)
                / total_mass).astype(f32)
        frac = ((f32(self.masspole) * (cos * cos).astype(f32)).astype(f32) / total_mass).astype(f32)
        den = np.float64(f32(self.length)) * (4.0 / 3.0 - frac.astype(np.float64))
        num = ((f32(self.gravity) * sin).astype(f32) - (cos * temp).astype(f32)).astype(f32)
        thetaacc = num.astype(np.float64) / den
        xacc = temp.astype(np.float64) - np.float64(pml) * thetaacc * cos.astype(np.float64) / np.float64(total_mass)
        tau = f32(self.tau)
        nx = (x + (tau * x_dot).astype(f32)).astype(f32)
        nx_dot = (x_dot.astype(np.float64) + np.float64(tau) * xacc).astype(f32)
        ntheta = (theta + (tau * theta_dot).astype(f32)).astype(f32)
        ntheta_dot = (theta_dot.astype(np.float64) + np.float64(tau) * thetaacc).astype(f32)
        self.state = np.stack([nx, nx_dot, ntheta, ntheta_dot], axis=1).astype(f32)
        self.obs = self.state.copy()
        xt, tt = f32(self.x_threshold), f32(self.theta_threshold_radians)
        terminated = (nx < -xt) | (nx > xt) | (ntheta < -tt) | (ntheta > tt)
        self.rewards = np.ones(self.E, f32)
        self.done = np.where((self.timestep == self.T) | terminated, 1, self.done).astype(np.int32)
        return self.obs, self.rewards, self.done


def policy_probabilities(packed, hidden, obs, n_actions=2):
    """float32 restatement of the in-kernel rollout policy (csrc/kernels/cartpole.hip::cp_policy_cum): two hidden
    layers of `hidden` ReLU units and one softmax head over `packed` = [W0 [H][4], b0, W1 [H][H], b1, Wp [A][H], bp];
    acc = bias, then one fused multiply-add per input in index order (emulated in float64: the product of two
    float32 is exact there; the single rounding of the sum to float32 can differ from a hardware fma in the
    last bit about once in 2^29 operations), softmax with the maximum subtracted.  obs [E, 4] -> probs [E, A]."""
    f32, H, A = np.float32, int(hidden), int(n_actions)
    w = np.asarray(packed, dtype=f32)
    o = 0
    W0 = w[o:o + 4 * H].reshape(H, 4); o += 4 * H
    b0 = w[o:o + H]; o += H
    W1 = w[o:o + H * H].reshape(H, H); o += H * H
    b1 = w[o:o + H]; o += H
    Wp = w[o:o + A * H].reshape(A, H); o += A * H
    bp = w[o:o + A]

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
    s = np.zeros(e.shape[0], f32)
    for a in range(A):
        s = (s + e[:, a]).astype(f32)
    return (e / s[:, None]).astype(f32)
