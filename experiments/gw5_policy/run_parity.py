#!/usr/bin/env python3
"""Parity + timing of the PREPARED live-policy TagGridWorld rollout kernel (tag_gridworld_n5_policy.hip); run on the
GPU box after `python experiments/gw5_policy/run_parity.py build` here (hipcc cross-compiles):

    python experiments/gw5_policy/run_parity.py [hidden=32] [num_envs=1000] [ticks=20]

Row k of the recorded batch: the observation must be the oracle's; the action must be the inverse-CDF draw (Philox
restated on the host) on the probabilities of policy_oracle.probabilities -- the float32 restatement of the in-kernel
forward, tagger policy for agents 0 - 3, runner policy for agent 4 -- except where the uniform sits within 2e-6 of a
decision threshold (device expf vs numpy exp; the oracle then follows the device's action); rewards <= 1 ulp, done
and the state after every launch exact.  Same structure as tests/test_gpu_core.py::
test_cartpole_rollout_with_the_policy_inside_the_kernel; becomes a test when the kernel moves into the product."""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
HSACO = os.path.join(ROOT, "build", "variants", "gw5_policy.hsaco")


def build():
    from warp_drive_amd import build as wb

    os.makedirs(os.path.dirname(HSACO), exist_ok=True)
    cmd = [wb._hipcc(), *wb.KERNEL_FLAGS, f"-I{wb.KDIR}", os.path.join(HERE, "tag_gridworld_n5_policy.hip"), "-o", HSACO]
    print(" ".join(cmd))
    subprocess.run(cmd, check=True)


def main(hidden, E, ticks):
    import torch

    import policy_oracle as po
    from oracle.core_np import single_head_tick_uniform
    from oracle.tag_gridworld_np import TagGridWorldOracle
    from tests.hip_harness import OBS, make_wrapper, pull, ulp_diff
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld
    from warp_drive_amd.managers import hip_driver as drv
    from warp_drive_amd.managers.function_manager import HIPSampler, _stream_tag
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    cfg = dict(num_taggers=4, grid_length=10, episode_length=23, seed=27, wall_hit_penalty=0.1,
               tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01, use_full_observation=True)
    env = CUDATagGridWorld(**cfg)
    env.ticks_per_launch = ticks
    w = make_wrapper(env, E)
    N, F = 5, 21
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=5)
    rng = np.random.RandomState(3)

    def random_policy(scale):
        H = hidden
        return po.pack((rng.randn(H, 21) * 0.6).astype(np.float32), (rng.randn(H) * 0.1).astype(np.float32),
                       (rng.randn(H, H) / np.sqrt(H)).astype(np.float32), (rng.randn(H) * 0.1).astype(np.float32),
                       (rng.randn(5, H) * scale / np.sqrt(H)).astype(np.float32), (rng.randn(5) * 0.1).astype(np.float32))

    packed = [random_policy(3.0), random_policy(5.0)]  # tagger, runner
    packed_dev = [torch.from_numpy(p).cuda() for p in packed]
    probs = torch.full((E, N, 5), 0.2, device="cuda")
    batch = {"obs": torch.full((ticks, E, N, F), 7.0, device="cuda"),
             "actions": torch.full((ticks, E, N, 1), -1, dtype=torch.int32, device="cuda"),
             "rewards": torch.full((ticks, E, N), -1.0, device="cuda"),
             "done": torch.full((ticks, E), -1, dtype=torch.int32, device="cuda")}
    # the launch of the fixed-policy N5 kernel, then the same arguments + the two weight blocks for the live one
    fn, args, block, grid, lds = env.tick_launch(sampler, [probs], w.env_resetter, batch=batch)
    assert fn.name == "HipTagGridWorldRollout_N5", fn.name
    fm = w.cuda_function_manager
    fm._load_extra_modules().append(drv.Module(HSACO))
    name = f"HipTagGridWorldRollout_N5_H{hidden}"
    fm.initialize_functions([name])
    live = fm.get_function(name)
    # LDS: the N5 kernel's (image, restore cache, quotient tables) + the time table rounded to 16 bytes + two policies
    cache_dwords = int(args[-2])
    lds_live = 4 * (12 * N * F + 12 * cache_dwords + 64 + (cfg["episode_length"] + 1 + 3) // 4 * 4 + 2 * po.policy_floats(hidden))
    largs = list(args) + packed_dev

    ocfg = dict(cfg)
    ocfg.pop("seed")
    orc = TagGridWorldOracle(num_envs=E, **ocfg)
    rng_words = np.zeros(4 + E * N, dtype=np.uint32)
    near = draws = finished = 0
    for launch in range(5):
        drv.memcpy_dtoh(rng_words, sampler.rng_state)
        torch.cuda.synchronize()
        live(*largs, block=block, grid=grid, shared=(lds_live + 15) // 16 * 16)
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in batch.items()}
        for k in range(ticks):
            obs = orc.obs.astype(np.float32)
            np.testing.assert_array_equal(b["obs"][k], obs, err_msg=f"obs row {k} of launch {launch}")
            p = np.empty((E, N, 5), np.float32)
            p[:, :4] = po.probabilities(packed[0], hidden, obs[:, :4].reshape(-1, F)).reshape(E, 4, 5)
            p[:, 4] = po.probabilities(packed[1], hidden, obs[:, 4])
            cum = po.running_sums(p.reshape(-1, 5)).reshape(E, N, 5)
            u = single_head_tick_uniform(E * N, rng_words[4:] + np.uint32(k), rng_words[0], rng_words[1],
                                         _stream_tag("tick")).reshape(E, N)
            want = np.minimum((cum < u[..., None]).sum(axis=-1), 4).astype(np.int32)
            got = b["actions"][k, :, :, 0]
            bad = got != want
            if bad.any():  # only where the uniform sits on a threshold
                gap = np.abs(cum[bad] - u[bad][:, None]).min(axis=1)
                assert (gap < 2e-6).all(), (launch, k, gap.max(), np.argwhere(bad)[:5])
            near += int(bad.sum())
            draws += E * N
            orc.step(got)
            assert ulp_diff(b["rewards"][k], orc.rewards.astype(np.float32)).max() <= 1
            np.testing.assert_array_equal(b["done"][k], orc.done, err_msg=f"done row {k}")
            finished += int((orc.done > 0).sum())
            orc.reset_done_envs()
        np.testing.assert_array_equal(pull(w, "loc_x"), orc.loc_x)
        np.testing.assert_array_equal(pull(w, "loc_y"), orc.loc_y)
        np.testing.assert_array_equal(pull(w, "_timestep_"), orc.timestep)
        np.testing.assert_array_equal(pull(w, OBS), orc.obs.astype(np.float32))
    hist = np.bincount(b["actions"].ravel(), minlength=5) / b["actions"].size
    print(f"parity ok: {draws} draws, {near} on a threshold, {finished} episode ends, action shares {np.round(hist, 3)}")
    assert finished >= E and near <= 2 + draws // 50000 and hist.max() < 0.9
    # ---- timing
    for _ in range(3):
        live(*largs, block=block, grid=grid, shared=(lds_live + 15) // 16 * 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        live(*largs, block=block, grid=grid, shared=(lds_live + 15) // 16 * 16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {ticks} ticks x {E} replicas per launch = {dt * 1e6:.1f} us -> {dt * 1e6 / ticks:.2f} us per tick, "
          f"{E * ticks / dt:.3e} env-steps/s (trainer per-tick path: 390 - 490 us per tick, profiles/r04_gridworld_rollout_timing.txt)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
             int(sys.argv[3]) if len(sys.argv) > 3 else 20)
