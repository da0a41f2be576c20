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
