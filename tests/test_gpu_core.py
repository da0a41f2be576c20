"""Sampler, reset, data/function managers and logger on the MI355X.

Restates the reference's own tests for these components (they all need a device):
  tests/warp_drive/pycuda_tests/test_action_sampler.py:42-257
  tests/warp_drive/numba_tests/test_ou_sampler.py:41-82
  tests/warp_drive/pycuda_tests/test_env_reset.py:38-245
  tests/warp_drive/numba_tests/test_pool_reset.py:38-141
  tests/warp_drive/pycuda_tests/test_function_manager.py:71-230
  tests/warp_drive/pycuda_tests/test_data_manager.py:24-88
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT = "sampled_actions"


def _managers(num_agents=5, num_envs=2, episode_length=1):
    from tests.hip_harness import require_gpu
    from warp_drive_amd.managers.data_manager import HIPDataManager
    from warp_drive_amd.managers.function_manager import HIPFunctionManager

    require_gpu()
    dm = HIPDataManager(num_agents=num_agents, episode_length=episode_length, num_envs=num_envs)
    fm = HIPFunctionManager(num_agents=int(dm.meta_info("n_agents")), num_envs=int(dm.meta_info("n_envs")))
    fm.load_hip_from_binary_file()
    return dm, fm


def _feed(**kw):
    from warp_drive_amd.utils.data_feed import DataFeed

    f = DataFeed()
    for k, v in kw.items():
        f.add_data(name=k, data=v)
    return f


# ------------------------------------------------------------------------------ sampler
def test_action_sampler_statistics():
    from warp_drive_amd.managers.function_manager import HIPSampler

    dm, fm = _managers()
    sampler = HIPSampler(fm)
    sampler.init_random(seed=None)
    dm.push_data_to_device(_feed(**{f"{ACT}_a": np.zeros((2, 5, 1), dtype=np.int32)}), torch_accessible=True)
    sampler.register_actions(dm, f"{ACT}_a", 3)
    assert dm.get_shape(f"{ACT}_a_cum_distr") == (2, 5, 3)
    p = np.array([[[0.333, 0.333, 0.333], [0.2, 0.5, 0.3], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]],
                  [[0.1, 0.7, 0.2], [0.7, 0.2, 0.1], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]]])
    dist = torch.from_numpy(p).float().cuda()
    n = 10000
    out = torch.empty((n, 2, 5), dtype=torch.int32, device="cuda")
    for i in range(n):
        sampler.sample(dm, dist, action_name=f"{ACT}_a")
        out[i] = dm.data_on_device_via_torch(f"{ACT}_a")[:, :, 0]
    a = out.cpu().numpy()
    for e in range(2):
        for g in range(5):
            for k in range(3):
                freq = (a[:, e, g] == k).mean()
                if p[e, g, k] in (0.0, 1.0):
                    assert freq == p[e, g, k]  # one-hot exact, zero-probability never drawn
                else:
                    assert abs(freq - p[e, g, k]) <= 0.1 * p[e, g, k] + 1e-9
    # independence across threads / replicas / iterations (test_action_sampler.py:159-257)
    dm.push_data_to_device(_feed(**{f"{ACT}_b": np.zeros((2, 5, 1), dtype=np.int32)}), torch_accessible=True)
    sampler.register_actions(dm, f"{ACT}_b", 4)
    dist = torch.full((2, 5, 4), 0.25, device="cuda")
    out = torch.empty((n, 2, 5), dtype=torch.int32, device="cuda")
    for i in range(n):
        sampler.sample(dm, dist, action_name=f"{ACT}_b")
        out[i] = dm.data_on_device_via_torch(f"{ACT}_b")[:, :, 0]
    b = out.cpu().numpy().astype(np.float64)
    assert b.reshape(n, -1).std(axis=1).mean() > 0.9  # threads of one draw differ
    assert b.std(axis=0).mean() > 0.9                # successive draws differ
    assert abs(b.mean() - 1.5) < 0.02
    assert abs(np.corrcoef(b[:-1].reshape(-1), b[1:].reshape(-1))[0, 1]) < 0.02


def test_sampler_given_draws_matches_reference_search():
    """same uniforms -> same indices as the reference's prefix-sum + binary search"""
    from oracle.core_np import sample_actions, sample_actions_counting

    rng = np.random.RandomState(0)
    p = rng.dirichlet(np.ones(21), size=4000).astype(np.float32)
    u = ((rng.randint(0, 1 << 24, size=4000) + 1) * 2.0 ** -24).astype(np.float32)
    np.testing.assert_array_equal(sample_actions(p, u), sample_actions_counting(p, u))


def test_sampler_argmax_and_large_shapes():
    from warp_drive_amd.managers.function_manager import HIPSampler

    E, N, A = 300, 105, 21
    dm, fm = _managers(num_agents=N, num_envs=E)
    sampler = HIPSampler(fm)
    sampler.init_random(seed=7)
    dm.push_data_to_device(_feed(**{ACT: np.zeros((E, N, 1), dtype=np.int32)}), torch_accessible=True)
    sampler.register_actions(dm, ACT, A)
    rng = np.random.RandomState(1)
    p = rng.dirichlet(np.ones(A), size=(E, N)).astype(np.float32)
    dist = torch.from_numpy(p).cuda()
    sampler.sample(dm, dist, ACT, use_argmax=True)
    np.testing.assert_array_equal(dm.pull_data_from_device(ACT)[..., 0], p.argmax(-1))
    counts = np.zeros(A)
    reps = 60
    for _ in range(reps):
        sampler.sample(dm, dist, ACT)
        got = dm.pull_data_from_device(ACT)[..., 0]
        assert got.min() >= 0 and got.max() < A
        counts += np.bincount(got.reshape(-1), minlength=A)
    expected = p.reshape(-1, A).sum(0) * reps
    assert np.abs(counts - expected).max() < 6 * np.sqrt(expected.max())
    # one-hot rows are exact at scale
    hot = rng.randint(0, A, size=(E, N))
    dist = torch.from_numpy(np.eye(A, dtype=np.float32)[hot]).cuda()
    for _ in range(20):
        sampler.sample(dm, dist, ACT)
        np.testing.assert_array_equal(dm.pull_data_from_device(ACT)[..., 0], hot)


def test_ou_sampler_statistics():
    """stationary std = sigma / sqrt(1 - (1 - theta)^2) and the lag-k covariance of the OU
    process (numba_tests/test_ou_sampler.py:68-82)"""
    from warp_drive_amd.managers.function_manager import HIPSampler

    E, N = 1000, 5
    dm, fm = _managers(num_agents=N, num_envs=E)
    sampler = HIPSampler(fm)
    sampler.init_random(seed=3)
    dm.push_data_to_device(_feed(**{ACT: np.zeros((E, N, 1), dtype=np.float32)}), torch_accessible=True)
    sampler.register_actions(dm, ACT, 1, is_deterministic=True)
    assert dm.get_shape(f"{ACT}_ou_state") == (E, N, 1)
    mu = torch.zeros((E, N, 1), device="cuda")
    damping, stddev, steps = 0.15, 0.2, 4000
    out = torch.empty((steps, E, N), device="cuda")
    for i in range(steps):
        sampler.sample(dm, mu, ACT, damping=damping, stddev=stddev, scale=1.0)
        out[i] = dm.data_on_device_via_torch(ACT)[:, :, 0]
    x = out[500:].cpu().numpy().astype(np.float64)
    std = stddev / np.sqrt(1 - (1 - damping) ** 2)
    assert abs(x.std() - std) < 2e-3
    lag = 10
    cov = (x[:-lag] * x[lag:]).mean()
    assert abs(cov - std ** 2 * (1 - damping) ** lag) < 2e-3
    assert abs(x.mean()) < 2e-3
    # scale ~ 0 passes the deterministic action through
    mu = torch.rand((E, N, 1), device="cuda")
    sampler.sample(dm, mu, ACT, scale=0.0)
    np.testing.assert_array_equal(dm.pull_data_from_device(ACT), mu.cpu().numpy())


# -------------------------------------------------------------------------------- reset
@pytest.mark.parametrize("torch_accessible", [False, True])
def test_reset_when_done(torch_accessible):
    from warp_drive_amd.managers.function_manager import HIPEnvironmentReset
    from warp_drive_amd.utils.data_feed import DataFeed

    dm, fm = _managers()
    resetter = HIPEnvironmentReset(fm)
    f = DataFeed()
    init = {
        "f1": np.array([1.5, -2.5], dtype=np.float32),                               # 1-D float
        "i1": np.array([7, 9], dtype=np.int32),                                      # 1-D int
        "f2": np.arange(10, dtype=np.float32).reshape(2, 5) / 10,                    # 2-D float
        "i2": np.arange(10, dtype=np.int32).reshape(2, 5),                           # 2-D int
        "f3": np.arange(30, dtype=np.float32).reshape(2, 5, 3) / 7,                  # 3-D float
        "i3": np.arange(60, dtype=np.int32).reshape(2, 5, 2, 3),                     # 4-D int
    }
    for k, v in init.items():
        f.add_data(name=k, data=v, save_copy_and_apply_at_reset=True)
    dm.push_data_to_device(f, torch_accessible=torch_accessible)
    assert dm.reset_data_list == list(init)

    def scramble():
        for k, v in init.items():
            dm._host_data[k] = (v * 0 - 3).astype(v.dtype)
            dm.reset_device(k)
            dm._host_data[k] = v

    def set_done(vals):
        dm.data_on_device_via_torch("_done_")[:] = torch.tensor(vals, dtype=torch.int32)

    scramble()
    set_done([1, 0])
    resetter.reset_when_done(dm, mode="if_done")
    for k, v in init.items():
        got = dm.pull_data_from_device(k)
        np.testing.assert_array_equal(got[0], v[0], err_msg=k)      # env 0 restored
        np.testing.assert_array_equal(got[1], v[1] * 0 - 3, err_msg=k)  # env 1 untouched
    np.testing.assert_array_equal(dm.pull_data_from_device("_done_"), [0, 0])
    np.testing.assert_array_equal(dm.pull_data_from_device("_timestep_"), [0, 0])
    scramble()
    set_done([0, 0])
    resetter.reset_when_done(dm, mode="force_reset")
    for k, v in init.items():
        np.testing.assert_array_equal(dm.pull_data_from_device(k), v, err_msg=k)
    # undo_done_after_reset=False keeps the flags
    set_done([0, 1])
    resetter.reset_when_done(dm, mode="if_done", undo_done_after_reset=False)
    np.testing.assert_array_equal(dm.pull_data_from_device("_done_"), [0, 1])


def test_reset_from_pool():
    from warp_drive_amd.managers.function_manager import HIPEnvironmentReset
    from warp_drive_amd.utils.data_feed import DataFeed

    E = 4000
    dm, fm = _managers(num_agents=3, num_envs=E)
    resetter = HIPEnvironmentReset(fm)
    f = DataFeed()
    f.add_data(name="a", data=np.zeros((E, 3), dtype=np.float32))
    f.add_data(name="b", data=np.zeros((E, 3, 2), dtype=np.int32))
    pool_a = np.arange(5 * 3, dtype=np.float32).reshape(5, 3)
    pool_b = np.arange(5 * 6, dtype=np.int32).reshape(5, 3, 2) * 10
    f.add_pool_for_reset(name="a_reset_pool", data=pool_a, reset_target="a")
    f.add_pool_for_reset(name="b_reset_pool", data=pool_b, reset_target="b")
    dm.push_data_to_device(f)
    assert dm.reset_target_to_pool == {"a": "a_reset_pool", "b": "b_reset_pool"}
    resetter.init_reset_pool(dm, seed=5)
    dm.data_on_device_via_torch("_done_")[:] = 1
    resetter.reset_when_done(dm, mode="if_done")
    a, b = dm.pull_data_from_device("a"), dm.pull_data_from_device("b")
    row_a = (a[:, 0] / 3).astype(int)
    row_b = b[:, 0, 0] // 60
    np.testing.assert_array_equal(a, pool_a[row_a])
    np.testing.assert_array_equal(b, pool_b[row_b])
    np.testing.assert_array_equal(row_a, row_b)  # one draw per replica per reset call
    counts = np.bincount(row_a, minlength=5) / E
    assert np.abs(counts - 0.2).max() < 0.03      # uniform over the pool (test_pool_reset.py:118-141)
    np.testing.assert_array_equal(dm.pull_data_from_device("_done_"), 0)
    a0 = a.copy()
    dm.data_on_device_via_torch("_done_")[:] = 1
    resetter.reset_when_done(dm, mode="if_done")
    assert (dm.pull_data_from_device("a") != a0).any(axis=1).mean() > 0.6  # a fresh draw


# --------------------------------------------------------------- data / function manager
def test_data_manager_roundtrip_and_casting():
    dm, _ = _managers()
    from warp_drive_amd.utils.data_feed import DataFeed

    f = DataFeed()
    f.add_data(name="X", data=np.array([[1, 2, 3, 4, 5], [0, 0, 0, 0, 0]]))          # int64 -> int32
    f.add_data(name="Y", data=[[0.1, 0.2, 0.3, 0.4, 0.5], [0.0, 0.0, 0.0, 0.0, 0.0]])  # float64 -> float32
    f.add_data(name="a", data=100)
    f.add_data(name="b", data=0.5)
    f.add_data(name="flag", data=True)
    dm.push_data_to_device(f)
    assert dm.get_dtype("X") == "int32" and dm.get_dtype("Y") == "float32"
    assert dm.device_data("a") == 100 and dm.device_data("a").dtype == np.int32
    assert dm.device_data("b").dtype == np.float32 and dm.device_data("flag").dtype == np.int32
    np.testing.assert_array_equal(dm.pull_data_from_device("X"), [[1, 2, 3, 4, 5], [0, 0, 0, 0, 0]])
    assert dm.pull_data_from_device("Y").dtype == np.float32
    assert not dm.is_data_on_device_via_torch("X")
    f2 = DataFeed()
    f2.add_data(name="Z", data=np.ones((2, 5), dtype=np.float32))
    dm.push_data_to_device(f2, torch_accessible=True)
    assert dm.is_data_on_device_via_torch("Z")
    z = dm.data_on_device_via_torch("Z")
    z += 1  # torch writes in place; the manager sees the same memory
    np.testing.assert_array_equal(dm.pull_data_from_device("Z"), 2)
    assert int(dm.device_data("Z")) == z.data_ptr()
    with pytest.raises(AssertionError):
        dm.push_data_to_device(f2)  # duplicate registration


def test_function_manager_testkernel_and_log():
    """test_function_manager.py:71-230: the dummy `testkernel`, the episode logger and a
    per-replica reset driven by the done flag it sets."""
    from warp_drive_amd.managers.function_manager import HIPEnvironmentReset, HIPLogController
    from warp_drive_amd.utils.data_feed import DataFeed

    T = 5
    dm, fm = _managers(episode_length=T)
    fm.initialize_functions(["testkernel"])
    kernel = fm.get_function("testkernel")
    f = DataFeed()
    f.add_data(name="X", data=np.array([[0.1, 0.2, 0.3, 0.4, 0.5], [0.6, 0.7, 0.8, 0.9, 1.0]]),
               save_copy_and_apply_at_reset=True, log_data_across_episode=True)
    f.add_data(name="Y", data=np.array([[1, 2, 3, 4, 5], [0, 0, 0, 0, 0]]),
               save_copy_and_apply_at_reset=True, log_data_across_episode=True)
    f.add_data(name="multiplier", data=2.0)
    f.add_data(name="target", data=40)
    dm.push_data_to_device(f)
    t = DataFeed()
    t.add_data(name="actions", data=np.zeros((2, 5, 3), dtype=np.int32))
    dm.push_data_to_device(t, torch_accessible=True)
    logger, resetter = HIPLogController(fm), HIPEnvironmentReset(fm)
    logger.reset_log(dm, env_id=0)
    x0, y0 = dm.pull_data_from_device("X").copy(), dm.pull_data_from_device("Y").copy()
    for step in range(1, 4):
        # both calling conventions of the reference's env classes
        args = (dm.device_data("X"), dm.device_data("Y"), dm.device_data("_done_"), dm.device_data("actions"),
                dm.device_data("multiplier"), dm.device_data("target"), np.int32(step), dm.meta_info("episode_length"),
                dm.meta_info("n_agents"), dm.meta_info("n_envs"))
        if step % 2:
            kernel(*args, block=fm.block, grid=fm.grid)
        else:
            kernel[fm.grid, fm.block](*args)
        logger.update_log(dm, step)
    np.testing.assert_allclose(dm.pull_data_from_device("X"), x0 / 8, rtol=1e-6)
    np.testing.assert_array_equal(dm.pull_data_from_device("Y"), y0 * 8)
    np.testing.assert_array_equal(dm.pull_data_from_device("actions"), np.tile([0, 1, 2], (2, 5, 1)))
    np.testing.assert_array_equal(dm.pull_data_from_device("_done_"), [1, 0])  # env 0 reached y >= 40
    log = logger.fetch_log(dm)
    assert log["X_for_log"].shape == (4, 5)
    np.testing.assert_allclose(log["X_for_log"][:, 0], x0[0, 0] / 2.0 ** np.arange(4), rtol=1e-6)
    np.testing.assert_array_equal(log["Y_for_log"][3], y0[0] * 8)
    resetter.reset_when_done(dm)
    np.testing.assert_array_equal(dm.pull_data_from_device("X")[0], x0[0])
    np.testing.assert_allclose(dm.pull_data_from_device("X")[1], x0[1] / 8, rtol=1e-6)
    with pytest.raises(AssertionError):
        fm.get_function("no_such_kernel")
    with pytest.raises(Exception):
        fm.initialize_functions(["no_such_kernel"])


@pytest.mark.parametrize("E,n_ticks", [(5000, 150), (100000, 70)])
def test_cartpole_vs_oracle(E, n_ticks):
    """BASELINE configs[4] kernel against the numpy restatement of cartpole_step_numba.py, bit-exact
    at E = 5000 and at the config's own E = 100 000 (through terminations, the time-out at T = 60 and the
    restarts; the restatement itself is pinned by tests/golden/cp_traj.npz)."""
    from oracle.cartpole_np import CartPoleOracle
    from tests.hip_harness import OBS, REW, make_wrapper, pull, push_actions, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv

    require_gpu()
    T = 60
    env = CUDAClassicControlCartPoleEnv(episode_length=T, seed=32145)
    w = make_wrapper(env, E)
    orc = CartPoleOracle(E, T, initial_state=pull(w, "state")[0, 0])
    rng = np.random.RandomState(0)
    for t in range(n_ticks):
        a = rng.randint(0, 2, size=(E, 1, 1)).astype(np.int32)
        push_actions(w, a)
        w.step_all_envs()
        orc.step(a)
        np.testing.assert_array_equal(pull(w, "state")[:, 0], orc.state, err_msg=f"t={t}")
        np.testing.assert_array_equal(pull(w, OBS)[:, 0], orc.obs)
        np.testing.assert_array_equal(pull(w, "_done_"), orc.done)
        np.testing.assert_array_equal(pull(w, REW)[:, 0], orc.rewards)
        w.reset_only_done_envs()
        orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "state")[:, 0], orc.state)


def test_cartpole_vs_reference_kernel_source():
    """The device kernel replays tests/golden/cp_traj.npz -- the reference's own Numba kernel source
    (cartpole_step_numba.py:5-83) run under a numba.cuda stand-in in the build container
    (oracle/gen_golden.py::gen_cartpole_traj): same initial state, same action stream, free-running
    through terminations, time-outs and restarts.  Floats within 1e-5 abs (the stand-in evaluates
    cos/sin in float64; observed ~2e-6), everything discrete exact."""
    import os

    from tests.hip_harness import OBS, REW, make_wrapper, pull, push_actions, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv

    require_gpu()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cp_traj.npz"))
    ticks, E = g["actions"].shape[:2]
    env = CUDAClassicControlCartPoleEnv(episode_length=int(g["episode_length"]), seed=1,
                                        initial_state=g["initial_state"])  # the fixture's start state
    w = make_wrapper(env, E)
    np.testing.assert_array_equal(pull(w, "state")[0, 0], g["initial_state"])
    worst = 0.0
    for t in range(ticks):
        push_actions(w, g["actions"][t].astype(np.int32))
        w.step_all_envs()
        st = pull(w, "state")
        worst = max(worst, float(np.abs(st - g["state"][t]).max()))
        np.testing.assert_allclose(st, g["state"][t], rtol=0, atol=1e-5, err_msg=f"t={t}")
        np.testing.assert_allclose(pull(w, OBS), g["obs"][t], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(pull(w, "_done_"), g["done"][t])
        np.testing.assert_array_equal(pull(w, "_timestep_"), g["timestep"][t])
        np.testing.assert_array_equal(pull(w, REW), g["rewards"][t])
        w.reset_only_done_envs()
    assert worst < 1e-5


@pytest.mark.parametrize("ticks", [1, 4])
def test_cartpole_fused_tick(ticks):
    """HipClassicControlCartPoleEnvTick: sampling + step + in-kernel reset, `ticks` ticks per launch.
    The CPU side replays the kernel's Philox draws tick by tick through the oracle."""
    import torch
    from oracle.cartpole_np import CartPoleOracle
    from oracle.core_np import sample_actions_counting, single_head_tick_uniform
    from tests.hip_harness import OBS, REW, make_wrapper, pull, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    require_gpu()
    E, T = 3001, 40
    env = CUDAClassicControlCartPoleEnv(episode_length=T, seed=32145)
    env.ticks_per_launch = ticks
    w = make_wrapper(env, E)
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=9)  # (placeholders were pushed by the harness)
    rng = np.random.RandomState(1)
    probs = torch.from_numpy(rng.dirichlet(np.ones(2), size=(E, 1)).astype(np.float32)).cuda()
    engine = RolloutEngine(w, sampler, probabilities=[probs])
    assert engine.fused and engine.ticks_per_launch == ticks and len(engine.entry_names) == 1
    orc = CartPoleOracle(E, T, initial_state=pull(w, "state")[0, 0])
    rng_words = np.zeros(4 + E, dtype=np.uint32)
    probs_host = probs.cpu().numpy()
    finished = 0
    for launch in range(40):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        assert (rng_words[4:] == launch * ticks).all()
        engine.run(1)
        torch.cuda.synchronize()
        for k in range(ticks):
            u = single_head_tick_uniform(E, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1], _stream_tag("tick"))
            a = sample_actions_counting(probs_host, u.reshape(E, 1))
            orc.step(a.reshape(E, 1, 1))
            last = k == ticks - 1
            if last:  # what the launch leaves in HBM is its last tick
                np.testing.assert_array_equal(pull(w, "sampled_actions")[:, 0, 0], a[:, 0])
                np.testing.assert_array_equal(pull(w, "_done_"), orc.done)
                np.testing.assert_array_equal(pull(w, REW)[:, 0], orc.rewards)
                fin = orc.done > 0
                np.testing.assert_array_equal(pull(w, OBS)[~fin, 0], orc.obs[~fin])
            finished += int((orc.done > 0).sum())
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "state")[:, 0], orc.state, err_msg=f"launch {launch}")
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep)
    assert finished >= E


@pytest.mark.parametrize("E,n_launches", [(2003, 6), (100000, 3)])
def test_cartpole_rollout_records_every_tick(E, n_launches):
    """The T-tick launch with the trainer's batch tensors: row k of the observation / action / reward / done
    batches is tick k of the launch -- the observation the action was sampled on, the action (draw for draw),
    the reward and the done flag -- replayed through the oracle; the per-tick arrays hold the last tick.
    (trainer_base.py:392-426 records exactly these per tick.)"""
    assert _cartpole_recorded_rollout(E, n_launches) == 1  # (the classic physics: the middle ticks run cp_euler<true>)


def test_cartpole_fast_middle_ticks_rest_on_checked_claims():
    """The middle ticks of a recorded Cartpole launch use a quadrant-free sin / cos and a three-instruction division by the
    total mass (csrc/kernels/cartpole.hip: cp_euler<true>).  Both are claims about float32 arithmetic, checked exhaustively on
    the device: cp_sincos_small == the numpy-exact sin / cos for every float32 of [-0.75, 0.75]; the division shortcut ==
    the correctly rounded division for the env's total mass (2^25 dividends) -- and the checker does notice a wrong claim: handed
    a reciprocal that is one ulp off it reports failure (then the env would report 0 and the kernels divide)."""
    import torch
    from tests.hip_harness import make_wrapper, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv

    require_gpu()
    env = CUDAClassicControlCartPoleEnv(episode_length=20, seed=1)
    w = make_wrapper(env, 8)
    fm = w.cuda_function_manager
    fm.initialize_functions(["HipCartPoleVerifySmallAngle", "HipCartPoleVerifyInvariantDivide"])
    ok = torch.ones(1, dtype=torch.int32, device="cuda")
    fm.get_function("HipCartPoleVerifySmallAngle")(ok, block=(256, 1, 1), grid=(4096, 1), shared=0)
    assert int(ok.item()) == 1
    assert env.invariant_divide_ok() == 1
    verify = fm.get_function("HipCartPoleVerifyInvariantDivide")
    for c in (np.float32(1.1), np.float32(2.0 - 2.0 ** -23), np.float32(3.0)):
        y = np.float32(1.0) / c
        for recip, want in ((y, 1), (np.nextafter(y, np.float32(2.0)), 0), (np.nextafter(y, np.float32(0.0)), 0)):
            ok.fill_(1)
            verify(c, recip, ok, block=(256, 1, 1), grid=(4096, 1), shared=0)
            assert int(ok.item()) == want, (c, recip)


@pytest.mark.parametrize("physics,fast", [(dict(masscart=1.3, division_proof_failed=True), 0),  # (forced) the kernels divide
                                          (dict(theta_threshold_radians=0.9), 1),            # angles beyond 0.75: general sin / cos
                                          (dict(masspole=0.25, length=0.8, force_mag=7.0), 1)])
def test_cartpole_recorded_rollout_with_other_physics(physics, fast):
    """the recorded rollout stays bit-exact where the fast middle ticks' preconditions do NOT hold (the kernel decides once
    per launch and wavefront) and for other constants where they do"""
    assert _cartpole_recorded_rollout(1500, 5, physics) == fast


def _cartpole_recorded_rollout(E, n_launches, physics=None):
    import torch
    from oracle.cartpole_np import CartPoleOracle
    from oracle.core_np import sample_actions_counting, single_head_tick_uniform
    from tests.hip_harness import OBS, make_wrapper, pull, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine

    require_gpu()
    T, ticks = 23, 16  # (E = 100 000: BASELINE configs[4]'s own size)
    env = CUDAClassicControlCartPoleEnv(episode_length=T, seed=32145)
    oracle_cls = CartPoleOracle
    if physics:  # the same constants for the env (class CartPolePhysics) and the oracle (class attributes)
        physics = dict(physics)
        if physics.pop("division_proof_failed", False):
            env._inv_div_ok = 0  # what invariant_divide_ok() would cache had the exhaustive check failed
        env.physics = type("Physics", (env.physics,), physics)
        oracle_cls = type("Oracle", (CartPoleOracle,), physics)
    env.ticks_per_launch = ticks
    w = make_wrapper(env, E)
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=4)
    rng = np.random.RandomState(2)
    probs = torch.from_numpy(rng.dirichlet(np.ones(2), size=(E, 1)).astype(np.float32)).cuda()
    batch = {"obs": torch.full((ticks, E, 1, 4), 7.0, device="cuda"),
             "actions": torch.full((ticks, E, 1, 1), -1, dtype=torch.int32, device="cuda"),
             "rewards": torch.full((ticks, E, 1), -1.0, device="cuda"),
             "done": torch.full((ticks, E), -1, dtype=torch.int32, device="cuda")}
    engine = RolloutEngine(w, sampler, probabilities=[probs], rollout_batch=batch)
    assert engine.fused and engine.ticks_per_launch == ticks
    orc = oracle_cls(E, T, initial_state=pull(w, "state")[0, 0])
    rng_words = np.zeros(4 + E, dtype=np.uint32)
    probs_host = probs.cpu().numpy()
    finished = 0
    for launch in range(n_launches):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        engine.run(1)
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in batch.items()}
        for k in range(ticks):
            np.testing.assert_array_equal(b["obs"][k, :, 0], orc.obs, err_msg=f"obs row {k} of launch {launch}")
            u = single_head_tick_uniform(E, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1], _stream_tag("tick"))
            a = sample_actions_counting(probs_host, u.reshape(E, 1))
            orc.step(a.reshape(E, 1, 1))
            np.testing.assert_array_equal(b["actions"][k, :, 0, 0], a[:, 0])
            np.testing.assert_array_equal(b["rewards"][k, :, 0], orc.rewards)
            np.testing.assert_array_equal(b["done"][k], orc.done)
            finished += int((orc.done > 0).sum())
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "state")[:, 0], orc.state)
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep)
        np.testing.assert_array_equal(pull(w, OBS)[:, 0], orc.obs)  # finished replicas already hold the reset observation
    assert finished >= n_launches // 2 * E
    return env.invariant_divide_ok()


@pytest.mark.parametrize("hidden", [32, 64])
def test_cartpole_rollout_with_the_policy_inside_the_kernel(hidden):
    """HipClassicControlCartPoleEnvRollout_H<hidden>: a whole batch of ticks in one launch, the policy network
    (two hidden layers, one head; weights in LDS) evaluated by the kernel on every tick's observation.
    Row k of the recorded batch: the observation must be the oracle's; the action must be the inverse-CDF
    draw (Philox restated on the host, random.cu:51-85) on the probabilities of oracle/cartpole_np.py::
    policy_probabilities -- the float32 restatement of the in-kernel forward -- except where the uniform sits
    within 2e-6 of the decision threshold (device expf vs numpy exp); the oracle then follows the device's
    action.  And T single-tick launches of the same kernel record the same rows as one T-tick launch."""
    import torch
    from oracle.cartpole_np import CartPoleOracle, policy_probabilities
    from oracle.core_np import single_head_tick_uniform
    from tests.hip_harness import make_wrapper, pull, require_gpu
    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.models import FullyConnected
    from warp_drive_amd.training.policy_kernel import pack_rollout_policy, rollout_policy_width

    require_gpu()
    E, T, ticks = 1501, 23, 12
    torch.manual_seed(5)
    model = FullyConnected(4, [2], [hidden, hidden])
    with torch.no_grad():  # (make the policy decisive enough that both actions occur with varied probabilities)
        model.policy_head[0].weight.mul_(6.0)
    assert rollout_policy_width(model, 4) == hidden
    packed = pack_rollout_policy(model).cuda()

    def make(tpl):
        env = CUDAClassicControlCartPoleEnv(episode_length=T, seed=32145)
        env.ticks_per_launch = tpl
        w = make_wrapper(env, E)
        sampler = HIPSampler(w.cuda_function_manager)
        sampler.init_random(seed=4)
        probs = torch.full((E, 1, 2), 0.5, device="cuda")
        rows = max(tpl, 1)
        batch = {"obs": torch.full((rows, E, 1, 4), 7.0, device="cuda"),
                 "actions": torch.full((rows, E, 1, 1), -1, dtype=torch.int32, device="cuda"),
                 "rewards": torch.full((rows, E, 1), -1.0, device="cuda"),
                 "done": torch.full((rows, E), -1, dtype=torch.int32, device="cuda")}
        eng = RolloutEngine(w, sampler, probabilities=[probs], rollout_batch=batch, rollout_policy=(packed, hidden))
        assert eng.step_kernel_name == f"HipClassicControlCartPoleEnvRollout_H{hidden}"
        return w, sampler, batch, eng

    w, sampler, batch, eng = make(ticks)
    w1, _, batch1, eng1 = make(1)
    orc = CartPoleOracle(E, T, initial_state=pull(w, "state")[0, 0])
    rng_words = np.zeros(4 + E, dtype=np.uint32)
    near = draws = finished = 0
    for launch in range(5):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        eng.run(1)
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in batch.items()}
        for k in range(ticks):
            eng1.run(1)  # the same tick as its own launch
            torch.cuda.synchronize()
            for key in b:
                np.testing.assert_array_equal(batch1[key][0].cpu().numpy(), b[key][k], err_msg=f"{key} row {k}")
            np.testing.assert_array_equal(b["obs"][k, :, 0], orc.obs, err_msg=f"obs row {k} of launch {launch}")
            p = policy_probabilities(packed.cpu().numpy(), hidden, orc.obs)
            u = single_head_tick_uniform(E, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1], _stream_tag("tick"))
            want = (p[:, 0] < u).astype(np.int32)  # number of running sums below u, clamped to the last action
            got = b["actions"][k, :, 0, 0]
            bad = got != want
            assert (np.abs(p[bad, 0] - u[bad]) < 2e-6).all(), (launch, k, p[bad, 0], u[bad])
            near += int(bad.sum())
            draws += E
            orc.step(got.reshape(E, 1, 1))
            np.testing.assert_array_equal(b["rewards"][k, :, 0], orc.rewards)
            np.testing.assert_array_equal(b["done"][k], orc.done)
            finished += int((orc.done > 0).sum())
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "state")[:, 0], orc.state)
        np.testing.assert_array_equal(pull(w1, "state")[:, 0], orc.state)
    frac1 = float((b["actions"] == 1).mean())
    assert finished >= 2 * E and near <= 2 + draws // 100000 and 0.1 < frac1 < 0.9, (finished, near, frac1)


def test_consistency_checker_api():
    """The reference's own parity harness, at 1e-5 instead of 1 % (its scenarios:
    tests/example_envs/pycuda_tests/test_tag_continuous.py:15-80, test_tag_gridworld.py:13-38)."""
    from tests.hip_harness import require_gpu
    from warp_drive_amd.env_cpu_gpu_consistency_checker import EnvironmentCPUvsGPU
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld, TagGridWorld

    require_gpu()
    tc = {"test2": dict(num_taggers=4, num_runners=1, max_acceleration=0.05, max_turn=np.pi / 4,
                        num_acceleration_levels=3, num_turn_levels=3, grid_length=10, episode_length=30,
                        step_penalty_for_tagger=-0.1, seed=428096, skill_level_runner=1, skill_level_tagger=2,
                        use_full_observation=False, runner_exits_game_after_tagged=False, tagging_distance=0.25),
          "test3": dict(num_taggers=1, num_runners=4, max_acceleration=2, max_turn=np.pi / 2,
                        num_acceleration_levels=3, num_turn_levels=3, grid_length=10, episode_length=30,
                        step_reward_for_runner=0.1, seed=654208, skill_level_runner=1, skill_level_tagger=0.5,
                        use_full_observation=False, runner_exits_game_after_tagged=True)}
    EnvironmentCPUvsGPU(dual_mode_env_class=TagContinuous, env_configs=tc, num_envs=2,
                        num_episodes=2).test_env_reset_and_step(seed=274880)
    gw = {"full": dict(num_taggers=4, grid_length=4, episode_length=20, seed=27, use_full_observation=True),
          "partial": dict(num_taggers=4, grid_length=4, episode_length=20, seed=27, use_full_observation=False)}
    EnvironmentCPUvsGPU(cpu_env_class=TagGridWorld, cuda_env_class=CUDATagGridWorld, env_configs=gw, num_envs=2,
                        num_episodes=2).test_env_reset_and_step(seed=3)


def test_integration_md_stub_verbatim():
    """INTEGRATION.md section 2 -- the ctypes binding a reference maintainer would add -- executed VERBATIM (raw
    ctypes on libwdhip.so: no managers, no hip_driver; only its `CSRC = ...` line points at this checkout), then
    TagGridWorld (5 agents, full observations) stepped through the stub's own `launch()` with device memory from
    `wd_malloc` / `wd_memcpy_*`: positions, done, observations and rewards bit-exact against the oracle."""
    import ctypes
    import os
    import re

    import torch
    from oracle.tag_gridworld_np import TagGridWorldOracle
    from tests.hip_harness import require_gpu

    require_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## 2. The ctypes stub"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    csrc_line = [l for l in code.splitlines() if l.startswith("CSRC = ")]
    assert len(csrc_line) == 1
    code = code.replace(csrc_line[0], f"CSRC = {os.path.join(root, 'warp_drive_amd', 'csrc')!r}")
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)  # noqa: S102 -- the documented stub IS what is under test
    lib, vp, launch, check, fn = ns["lib"], ns["vp"], ns["launch"], ns["check"], ns["fn"]

    E, taggers, L, T = 37, 4, 10, 25
    N, F = taggers + 1, 4 * (taggers + 1) + 1
    orc = TagGridWorldOracle(E, num_taggers=taggers, grid_length=L, episode_length=T, wall_hit_penalty=0.1,
                             tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01)

    class Dev:  # a device block from wd_malloc with the .data_ptr() the stub's launch() packs
        def __init__(self, host):
            self.host = np.ascontiguousarray(host)
            self.ptr = vp()
            check(lib.wd_malloc(self.host.nbytes, ctypes.byref(self.ptr)), "malloc")
            self.push(self.host)

        def data_ptr(self):
            return self.ptr.value

        def push(self, host):
            host = np.ascontiguousarray(host, dtype=self.host.dtype)
            check(lib.wd_memcpy_htod(self.ptr, host.ctypes.data, host.nbytes, None), "htod")

        def pull(self):
            check(lib.wd_memcpy_dtoh(self.host.ctypes.data, self.ptr, self.host.nbytes, None), "dtoh")
            return self.host

    x, y = Dev(orc.loc_x.astype(np.int32)), Dev(orc.loc_y.astype(np.int32))
    act, done, tstep = Dev(np.zeros((E, N, 1), np.int32)), Dev(np.zeros(E, np.int32)), Dev(np.zeros(E, np.int32))
    rew, obs = Dev(np.zeros((E, N), np.float32)), Dev(orc.obs.astype(np.float32))
    threads = 64
    epb = threads // N
    lds = (4 * (4 * epb * N + 2 * epb + 2) + 15) // 16 * 16 + 4 * epb * N * F  # TagGridWorld.lds_bytes(epb)
    f64, i32 = np.float64, np.int32  # (the four reward scalars are float64 kernel arguments: tag_gridworld_rewards.h)
    rng = np.random.RandomState(3)
    for t in range(T + 3):
        a = rng.randint(0, 5, size=(E, N, 1)).astype(np.int32)
        act.push(a)
        launch(fn, [x, y, act, done, rew, obs, f64(0.1), f64(10.0), f64(2.0), f64(0.01), i32(1), i32(L), tstep,
                    i32(T), i32(N), i32(E)], grid=((E + epb - 1) // epb,), block=(threads,), shared=lds)
        check(lib.wd_sync(torch.cuda.current_stream().cuda_stream), "sync")
        orc.step(a)
        np.testing.assert_array_equal(x.pull(), orc.loc_x, err_msg=f"t={t}")
        np.testing.assert_array_equal(y.pull(), orc.loc_y)
        np.testing.assert_array_equal(done.pull(), orc.done)
        np.testing.assert_array_equal(obs.pull(), orc.obs.astype(np.float32))
        np.testing.assert_array_equal(rew.pull(), orc.rewards.astype(np.float32))
        if orc.done.any():  # (a tagged runner or the time-out ends a replica; the flags were just compared)
            break
    assert t >= 4 and orc.done.any(), t
    for d in (x, y, act, done, tstep, rew, obs):
        check(lib.wd_free(d.ptr), "free")

