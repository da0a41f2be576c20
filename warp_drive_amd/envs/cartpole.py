"""ClassicControl CartPole, single agent per replica (BASELINE config 5).

Host-side mirror of reference example_envs/single_agent/classic_control/cartpole/
cartpole.py:19-141 and single_agent/base.py.  The reference's CPU step delegates to
third-party `gym.envs.classic_control.CartPoleEnv` (absent here); this class carries the
classic constants itself and its CPU step is the same Euler update the reference's device
kernel implements (cartpole_step_numba.py:29-83).  Parity is pinned by a trajectory recorded from
that kernel source run under a numba.cuda stand-in (tests/golden/cp_traj.npz).
The device step launches `HipClassicControlCartPoleEnvStep`.
"""
import math

import numpy as np

from warp_drive_amd.utils import spaces
from warp_drive_amd.utils.constants import Constants
from warp_drive_amd.utils.data_feed import DataFeed
from warp_drive_amd.utils.gpu_environment_context import CUDAEnvironmentContext

_OBSERVATIONS = Constants.OBSERVATIONS
_ACTIONS = Constants.ACTIONS
_REWARDS = Constants.REWARDS


class CartPolePhysics:
    """The constants of gym's CartPoleEnv (classic_control/cartpole.py in gym >= 0.26)."""
    gravity = 9.8
    masscart = 1.0
    masspole = 0.1
    length = 0.5  # half the pole's length
    force_mag = 10.0
    tau = 0.02
    theta_threshold_radians = 12 * 2 * math.pi / 360
    x_threshold = 2.4


def euler_step(state, action, p=CartPolePhysics):
    """One Euler tick with the dtype flow of the reference's Numba kernel
    (float32 state/scalars; the 4.0/3.0 literal widens thetaacc/xacc to float64)."""
    f32 = np.float32
    x, x_dot, theta, theta_dot = (f32(v) for v in state)
    force = f32(p.force_mag) if action > 0.5 else f32(-p.force_mag)
    costheta, sintheta = np.cos(theta), np.sin(theta)  # numpy float32 kernels
    total_mass = f32(p.masspole + p.masscart)
    polemass_length = f32(p.masspole * p.length)
    temp = f32(f32(force + f32(f32(polemass_length * f32(theta_dot * theta_dot)) * sintheta)) / total_mass)
    den = np.float64(f32(p.length)) * (4.0 / 3.0 - np.float64(f32(f32(f32(p.masspole) * f32(costheta * costheta)) / total_mass)))
    thetaacc = np.float64(f32(f32(f32(p.gravity) * sintheta) - f32(costheta * temp))) / den
    xacc = np.float64(temp) - np.float64(polemass_length) * thetaacc * np.float64(costheta) / np.float64(total_mass)
    tau = f32(p.tau)
    nx = f32(x + f32(tau * x_dot))
    nx_dot = f32(np.float64(x_dot) + np.float64(tau) * xacc)
    ntheta = f32(theta + f32(tau * theta_dot))
    ntheta_dot = f32(np.float64(theta_dot) + np.float64(tau) * thetaacc)
    return np.array([nx, nx_dot, ntheta, ntheta_dot], dtype=f32)


class ClassicControlCartPoleEnv:
    name = "ClassicControlCartPoleEnv"

    def __init__(self, episode_length=500, env_backend="cpu", reset_pool_size=0, seed=None, initial_state=None):
        """`initial_state` (optional, beyond the reference's signature): a fixed start state instead of
        the seeded draw -- the golden-trajectory tests start from the fixture's."""
        self.initial_state = None if initial_state is None else np.asarray(initial_state, dtype=np.float32)
        self.num_agents = 1
        self.agents = {0: True}
        assert episode_length > 0
        self.episode_length = episode_length
        self.env_backend = env_backend
        self.reset_pool_size = reset_pool_size
        self.seed = seed
        self.timestep = None
        self.physics = CartPolePhysics
        self._rng = np.random.default_rng(seed)
        high = np.array([self.physics.x_threshold * 2, np.finfo(np.float32).max,
                         self.physics.theta_threshold_radians * 2, np.finfo(np.float32).max], dtype=np.float32)
        self.action_space = {0: spaces.Discrete(2)}
        self.observation_space = {0: spaces.Box(-high, high, dtype=np.float32)}
        self.state = None

    def _draw_initial_state(self, fixed):
        if fixed and self.initial_state is not None:
            return np.asarray(self.initial_state, dtype=np.float32).copy()
        rng = np.random.default_rng(self.seed) if fixed else self._rng
        return rng.uniform(low=-0.05, high=0.05, size=(4,)).astype(np.float32)

    def reset(self):
        self.timestep = 0
        self.state = self._draw_initial_state(fixed=self.reset_pool_size < 2)
        return {0: self.state.copy()}

    def step(self, action=None):
        self.timestep += 1
        assert isinstance(action, dict) and len(action) == 1
        self.state = euler_step(self.state, action[0], self.physics)
        x, theta = self.state[0], self.state[2]
        p = self.physics
        terminated = bool(x < -p.x_threshold or x > p.x_threshold or theta < -p.theta_threshold_radians
                          or theta > p.theta_threshold_radians)
        done = {"__all__": self.timestep >= self.episode_length or terminated}
        return {0: self.state.copy()}, {0: 1.0}, done, {}


class CUDAClassicControlCartPoleEnv(CUDAEnvironmentContext, ClassicControlCartPoleEnv):
    def __init__(self, *args, **kwargs):
        ClassicControlCartPoleEnv.__init__(self, *args, **kwargs)
        CUDAEnvironmentContext.__init__(self)

    def get_data_dictionary(self):
        p = self.physics
        feed = DataFeed()
        initial_state = self._draw_initial_state(fixed=True)
        feed.add_data(name="state", data=np.atleast_2d(initial_state),
                      save_copy_and_apply_at_reset=self.reset_pool_size < 2)
        feed.add_data_list([
            ("gravity", p.gravity), ("masspole", p.masspole), ("total_mass", p.masspole + p.masscart),
            ("length", p.length), ("polemass_length", p.masspole * p.length), ("force_mag", p.force_mag),
            ("tau", p.tau), ("theta_threshold_radians", p.theta_threshold_radians),
            ("x_threshold", p.x_threshold),
        ])
        return feed

    def get_reset_pool_dictionary(self):
        pool = DataFeed()
        if self.reset_pool_size >= 2:
            states = np.stack([np.atleast_2d(self._draw_initial_state(fixed=False))
                               for _ in range(self.reset_pool_size)], axis=0)
            assert states.ndim == 3 and states.shape[2] == 4
            pool.add_pool_for_reset(name="state_reset_pool", data=states, reset_target="state")
        return pool

    _STEP_ARGS = ["state", _ACTIONS, "_done_", _REWARDS, _OBSERVATIONS, "gravity", "masspole", "total_mass",
                  "length", "polemass_length", "force_mag", "tau", "theta_threshold_radians", "x_threshold",
                  "_timestep_", ("episode_length", "meta"), ("n_envs", "meta")]

    def step_launch(self):
        n_envs = int(self.cuda_data_manager.meta_info("n_envs"))
        block = (256, 1, 1)
        grid = (max(1, min(4096, (n_envs + 255) // 256)), 1)
        return self.cuda_step, self.cuda_step_function_feed(self._STEP_ARGS), block, grid, 0

    TICK_HEADS = 1          # action heads the fused tick kernel samples (RolloutEngine)
    ticks_per_launch = 1    # > 1: fixed-policy rollout, T ticks fused per launch (the HBM-ceiling run)

    ROLLOUT_POLICY_WIDTHS = (32, 64)   # HipClassicControlCartPoleEnvRollout_H<width>

    def has_live_policy_rollout(self, width, n_actions):
        """does a rollout kernel exist that evaluates the policy network itself (…Rollout_H<width>)?  (RolloutEngine asks
        before it calls `tick_launch(policy=...)`)"""
        name = self.cuda_step.name.replace("Step", f"Rollout_H{int(width)}")
        return int(width) in self.ROLLOUT_POLICY_WIDTHS and int(n_actions) <= 8 and self.cuda_function_manager.has_function(name)

    def tick_launch(self, sampler, probabilities, resetter, env_range=None, batch=None, policy=None):
        """Fused rollout tick(s): sample + step + reset of a finished replica, `ticks_per_launch`
        times in ONE launch (HipClassicControlCartPoleEnvTick).  probabilities = [float32 CUDA tensor
        [E, 1, n_actions]]; `_done_` reports the last tick of the launch.  `batch` (optional) = the
        trainer's batch tensors {"obs": [T, E, 1, 4] float32, "actions": [T, E, 1, 1] int32, "rewards":
        [T, E, 1] float32, "done": [T, E] int32} with T >= ticks_per_launch: tick k of the launch writes
        their row k (what trainer_base.py:392-426 records per tick).  `policy` (optional) = (packed float32
        CUDA tensor from training.policy_kernel.pack_rollout_policy, hidden width): the launch evaluates the
        policy network on every tick's observation itself (HipClassicControlCartPoleEnvRollout_H<width>)
        instead of reading `probabilities`."""
        from warp_drive_amd.managers.function_manager import _stream_tag

        assert env_range is None and len(probabilities) == 1
        fm, dm = self.cuda_function_manager, self.cuda_data_manager
        name = self.cuda_step.name.replace("Step", "Tick")
        shared, pol_args = 0, [np.uint64(0), np.int32(0)]
        if policy is not None:
            packed, width = policy
            n_act = int(probabilities[0].shape[-1])
            if not self.has_live_policy_rollout(width, n_act):
                from warp_drive_amd.rollout import UnsupportedRolloutShape

                raise UnsupportedRolloutShape(f"no in-kernel policy of width {width} with {n_act} actions")
            n_w = 4 * width + width + width * width + width + n_act * width + n_act
            assert packed.is_cuda and packed.dtype.is_floating_point and packed.numel() == n_w and packed.is_contiguous()
            name = self.cuda_step.name.replace("Step", f"Rollout_H{width}")
            shared, pol_args = 4 * n_w, [packed, np.int32(width)]
        fm.initialize_functions([name])
        _, reset_args, _, _ = resetter.fused_launch(dm, 0, 0)  # builds / refreshes the descriptor table
        _, args, block, grid, _ = self.step_launch()
        null = np.uint64(0)
        if batch is not None:
            import torch

            E, T = int(dm.meta_info("n_envs")), int(self.ticks_per_launch)
            want = {"obs": ((E, 1, 4), torch.float32), "actions": ((E, 1, 1), torch.int32),
                    "rewards": ((E, 1), torch.float32), "done": ((E,), torch.int32)}
            for key, (shape, dtype) in want.items():
                t = batch[key]
                assert t.is_cuda and t.is_contiguous() and t.dtype == dtype and t.shape[0] >= T and \
                    tuple(t.shape[1:]) == shape, (key, tuple(t.shape), t.dtype)
            batch_args = [batch["obs"], batch["actions"], batch["rewards"], batch["done"]]
        else:
            batch_args = [null, null, null, null]
        args = list(args) + [sampler.rng_state, probabilities[0], np.int32(probabilities[0].shape[-1]), reset_args[0],
                             reset_args[1], _stream_tag("tick"), np.int32(self.ticks_per_launch)] + batch_args + pol_args + \
            [np.int32(self.invariant_divide_ok())]
        return fm.get_function(name), args, block, grid, shared

    def invariant_divide_ok(self):
        """1 when the device PROVED, exhaustively, that the tick kernels' three-instruction division by this env's total mass
        (csrc/kernels/cartpole.hip::cp_div_invariant) equals the correctly rounded division for every float32 dividend of two
        binades, both signs -- one launch of HipCartPoleVerifyInvariantDivide, once per env object; 0: the kernels divide.
        (True for the classic 1.1 and for every other mass tried so far.)"""
        if getattr(self, "_inv_div_ok", None) is None:
            import torch

            fm = self.cuda_function_manager
            fm.initialize_functions(["HipCartPoleVerifyInvariantDivide"])
            ok = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
            total_mass = np.float32(self.physics.masspole + self.physics.masscart)
            fm.get_function("HipCartPoleVerifyInvariantDivide")(total_mass, np.float32(1.0) / total_mass, ok,
                                                                block=(256, 1, 1), grid=(4096, 1), shared=0)
            self._inv_div_ok = int(ok.item())
        return self._inv_div_ok

    def step(self, actions=None):
        self.timestep += 1
        if self.env_backend != "hip":
            raise Exception("CUDAClassicControlCartPoleEnv expects env_backend = 'hip'")
        fn, args, block, grid, shared = self.step_launch()
        fn(*args, block=block, grid=grid, shared=shared)
