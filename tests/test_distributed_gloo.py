"""world_size = 2 on CPU (gloo): the N > 1 path of bench.py / the rollout -- replica
sharding with no data-path collective, per-rank seeds, barrier + max-over-ranks timing."""
import json
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from warp_drive_amd import distributed as wdd
    from warp_drive_amd.envs.tag_gridworld import TagGridWorld

    r, _, w = wdd.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    first, count = wdd.shard_replicas(total_envs, w, r)
    # every rank steps its own replicas (host env here; the device path is identical per replica)
    envs = [TagGridWorld(num_taggers=4, grid_length=6, episode_length=9) for _ in range(count)]
    for e in envs:
        e.reset()
    xs = []
    for t in range(12):
        for i, e in enumerate(envs):
            rng = np.random.RandomState(1000 * (first + i) + t)  # action stream keyed by GLOBAL replica id
            _, _, done, _ = e.step({a: int(rng.randint(5)) for a in range(5)})
            if done["__all__"]:
                e.reset()
        xs.append(np.stack([e.global_state["loc_x"][e.timestep] for e in envs]))
    wdd.barrier()
    elapsed = 1.0 + rank  # rank 1 is "slower": the job's time must be the max
    agg = wdd.aggregate_throughput(count * 12, elapsed)
    np.save(os.path.join(out_dir, f"x_{rank}.npy"), np.stack(xs))
    with open(os.path.join(out_dir, f"r_{rank}.json"), "w") as f:
        json.dump({"first": first, "count": count, "agg": agg, "seed": wdd.rank_seed(274880, r),
                   "max_t": wdd.max_over_ranks(elapsed), "per_rank": wdd.gather_floats(elapsed),
                   "allreduce_us": wdd.time_allreduce(190550, 5)}, f)
    wdd.shutdown()


def _check_against_unsharded(tmp_path, total, world):
    # sharded result == unsharded result: replicas are independent, no collective needed
    from warp_drive_amd.envs.tag_gridworld import TagGridWorld

    envs = [TagGridWorld(num_taggers=4, grid_length=6, episode_length=9) for _ in range(total)]
    for e in envs:
        e.reset()
    ref = []
    for t in range(12):
        for i, e in enumerate(envs):
            rng = np.random.RandomState(1000 * i + t)
            _, _, done, _ = e.step({a: int(rng.randint(5)) for a in range(5)})
            if done["__all__"]:
                e.reset()
        ref.append(np.stack([e.global_state["loc_x"][e.timestep] for e in envs]))
    got = np.concatenate([np.load(tmp_path / f"x_{r}.npy") for r in range(world)], axis=1)
    np.testing.assert_array_equal(got, np.stack(ref))


def test_two_rank_sharding(tmp_path):
    total, world = 7, 2  # ragged on purpose
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    recs = [json.load(open(tmp_path / f"r_{r}.json")) for r in range(world)]
    assert [(r["first"], r["count"]) for r in recs] == [(0, 4), (4, 3)]
    assert [r["seed"] for r in recs] == [274880, 274881]
    for r in recs:
        assert r["max_t"] == 2.0 and abs(r["agg"] - 7 * 12 / 2.0) < 1e-9
        # what bench.py adds to its N > 1 line: every rank's own time, and the gradient bucket's all-reduce time
        assert r["per_rank"] == [1.0, 2.0] and r["allreduce_us"] > 0
    _check_against_unsharded(tmp_path, total, world)


def test_eight_rank_sharding(tmp_path):
    """the driver's 8-GPU scaling run is the first time this path sees 8 ranks on hardware: the same worker with
    WORLD_SIZE = 8 over gloo -- contiguous ragged shards that cover every replica once, seeds base + rank, the job's
    time = the slowest rank's, every rank's own time gathered in rank order, a collective of the gradient bucket's size"""
    total, world = 21, 8
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    recs = [json.load(open(tmp_path / f"r_{r}.json")) for r in range(world)]
    assert [r["count"] for r in recs] == [3, 3, 3, 3, 3, 2, 2, 2]
    assert [r["first"] for r in recs] == [0, 3, 6, 9, 12, 15, 17, 19]
    assert [r["seed"] for r in recs] == [274880 + r for r in range(world)]
    xs = np.concatenate([np.load(tmp_path / f"x_{r}.npy") for r in range(world)], axis=1)
    assert xs.shape[1] == total
    for r in recs:
        assert r["max_t"] == 8.0 and abs(r["agg"] - total * 12 / 8.0) < 1e-9
        assert r["per_rank"] == [float(k + 1) for k in range(world)] and r["allreduce_us"] > 0
    _check_against_unsharded(tmp_path, total, world)


def test_rank_cpu_slices():
    """`pin_rank_to_cpus`: every rank of a node gets its own, non-empty, disjoint share of the allowed CPUs; with the
    GPUs' NUMA nodes known, a share of the CPUs next to its GPU"""
    from warp_drive_amd.distributed import _parse_cpulist, rank_cpu_slice

    allowed = set(range(4, 132))  # 128 CPUs allowed, ids not starting at 0
    shares = [rank_cpu_slice(r, 8, allowed) for r in range(8)]
    assert all(len(s) == 16 for s in shares) and sorted(sum(shares, [])) == sorted(allowed)
    assert rank_cpu_slice(2, 8, {7}) == [7]                       # fewer CPUs than ranks: never empty
    assert sorted(sum([rank_cpu_slice(r, 3, set(range(10))) for r in range(3)], [])) == list(range(10))  # ragged
    node1 = _parse_cpulist("64-127,192-255\n")
    assert len(node1) == 128 and 200 in node1
    # ranks 4..7 drive GPUs of NUMA node 1: rank 5 is the second of four on it
    share = rank_cpu_slice(5, 8, set(range(256)), numa_cpus=node1, ranks_on_node=(1, 4))
    assert len(share) == 32 and set(share) <= node1 and share[0] == 96


def test_shard_replicas_covers_everything():
    from warp_drive_amd.distributed import shard_replicas

    for total in (1, 7, 2000, 16000):
        for world in (1, 2, 3, 8):
            spans = [shard_replicas(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
    assert shard_replicas(16000, 8, 3) == (6000, 2000)


def _lock_worker(rank, out_dir):
    from warp_drive_amd import build as wd_build

    path = wd_build.build_kernels_locked()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write(path)


def test_concurrent_ranks_build_once(tmp_path):
    """ranks started by torch.distributed.run share no Event: every rank calls the locked build
    (what HIPFunctionManager.compile_and_load_hip does without an event messenger)"""
    from warp_drive_amd import build as wd_build

    mp.spawn(_lock_worker, args=(str(tmp_path),), nprocs=8, join=True)  # (8: the ranks of one MI355X node)
    for r in range(8):
        assert open(tmp_path / f"ok_{r}").read() == wd_build.HSACO
    assert os.path.exists(wd_build.HSACO)


def test_device_index_and_backend_override(monkeypatch):
    from warp_drive_amd import distributed as wdd

    monkeypatch.delenv("WD_FORCE_DEVICE", raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert wdd.device_index() == 3 and wdd.device_index(5) == 5
    monkeypatch.setenv("WD_FORCE_DEVICE", "0")  # N > 1 tests on a 1-GPU box
    assert wdd.device_index() == 0 and wdd.device_index(5) == 0
    assert wdd.gather_ints(7) == [7]


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_bench_starts_its_own_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the driver's N = 1
    shape of the command; N = 8 = one MI355X node); without a GPU the ranks stop at the loud no-GPU assertion, not before"""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("covered by tests/test_gpu_multirank.py on a GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_ranks), "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert "bench.py needs an MI355X" in out.stderr and "but WORLD_SIZE" not in out.stderr


def test_valu_roofline_from_a_pmc_file(tmp_path, monkeypatch):
    """bench.py's second roofline: the VALU issue time of a launch from per-class instruction counts (PMC file keyed by
    the code object's sha256) and the per-class issue rates; None unless kernel, shape and code object match."""
    import hashlib
    import json as js
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from warp_drive_amd.managers import hip_driver

    hsaco = tmp_path / "k.hsaco"
    hsaco.write_bytes(b"code object")
    monkeypatch.setattr(hip_driver, "code_object_of", lambda kernel: str(hsaco))  # the object that holds the kernel
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    counters = {"SQ_WAVES": 4000.0, "SQ_INSTS_VALU": 4000.0 * 100, "SQ_INSTS_VALU_INT32": 4000.0 * 40,
                "SQ_INSTS_VALU_ADD_F32": 4000.0 * 20, "SQ_INSTS_VALU_FMA_F32": 4000.0 * 10, "SQ_THREAD_CYCLES_VALU": 4000.0 * 100 * 48}
    rec = {"kernel": "HipTagContinuousTick_K10", "num_envs": 2000, "full_obs": False,
           "hsaco_sha256": hashlib.sha256(b"code object").hexdigest(), "counters_per_launch": counters, "shader_clock_ghz": 2.0}
    js.dump(rec, open(tmp_path / "profiles" / "pmc_mix.json", "w"))
    r = bench.valu_roofline("HipTagContinuousTick_K10", 2000, False, 10.0)
    cycles = 4000 * (40 * 2.9 + 20 * 1.2 + 10 * 2.0 + 30 * 2.5)  # the unclassified 30 are "OTHER"
    assert abs(r["issue_bound_us"] - cycles / 1024 / 2000.0) < 1e-9 and abs(r["frac"] - r["issue_bound_us"] / 10.0) < 1e-12
    assert r["wave_insts"] == 100 and r["active_lanes_per_inst"] == 48
    assert bench.valu_roofline("HipTagContinuousTick_K10", 8000, False, 10.0) is None   # other shape
    hsaco.write_bytes(b"another code object")
    assert bench.valu_roofline("HipTagContinuousTick_K10", 2000, False, 10.0) is None   # other code object


class _StandInTrainer:
    """what `bench_iteration.measure_training_iteration` touches of a Trainer, on CPU: two policies (the networks of the
    TagContinuous run, shrunk), the REAL one-bucket GradientBucket over the job's process group, a rollout that only costs
    time and an update that differentiates a rank-dependent batch, averages the gradients with the bucket's one
    collective and steps -- the N > 1 control flow of training/trainer.py::_update_model_params without a GPU"""

    def __init__(self, rank, num_envs=6, ticks=5):
        import torch

        from warp_drive_amd.training.grad_bucket import GradientBucket
        from warp_drive_amd.training.losses import PPO
        from warp_drive_amd.training.models import FullyConnected

        torch.manual_seed(100 + rank)   # ranks start from DIFFERENT weights: the bucket's broadcast must align them
        self.device = torch.device("cpu")
        self.models = {"runner": FullyConnected(9, [4, 3], (16, 16)), "tagger": FullyConnected(9, [4, 3], (16, 16))}
        self.trainers = {p: PPO(clip_param=0.1) for p in self.models}
        self.optimizers = {p: torch.optim.Adam(m.parameters(), lr=1e-2) for p, m in self.models.items()}
        self.grad_bucket = GradientBucket(list(self.models.values()), self.device)
        self.grad_bucket.broadcast_parameters(src=0)
        self.num_envs, self.batch_len = num_envs, ticks
        self.train_batch_size = num_envs * ticks
        self.update_plan = {p: {"objective": "framework"} for p in self.models}
        self.rank, self.rollouts = rank, 0
        self.gen = torch.Generator().manual_seed(7 + rank)   # own replicas: own data

    def _generate_rollout_batch(self):
        self.rollouts += 1

    def _update_model_params(self, iteration, log):
        import torch

        self.grad_bucket.zero()
        for p, m in self.models.items():
            x = torch.randn(self.batch_len, self.num_envs, 2, 9, generator=self.gen)
            probs, values = m(x)
            (sum(pr.log().mean() for pr in probs) + values.square().mean()).backward()
        self.grad_bucket.all_reduce_mean()
        for p in self.models:
            self.optimizers[p].step()

    def graceful_close(self):
        pass


def _iteration_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from warp_drive_amd import distributed as wdd
    from warp_drive_amd.training.bench_iteration import measure_training_iteration

    wdd.init_process_group(backend="gloo")
    assert wdd.init_process_group(backend="gloo")[2] == world   # (a second call inside a live group is a no-op)
    tr = _StandInTrainer(rank)
    rec = measure_training_iteration(tr, warmup_iterations=1)
    rec["rollouts"] = tr.rollouts
    with open(os.path.join(out_dir, f"it_{rank}.json"), "w") as f:
        json.dump(rec, f)
    wdd.shutdown()


@pytest.mark.parametrize("world", [2, 8])
def test_trainer_object_of_the_bench_line_over_gloo(tmp_path, world):
    """bench.py --gpus N's `trainer` object (BASELINE configs[3]: PPO + the gradient all-reduce), schema and invariants,
    with world_size 2 and 8 over gloo: one warm-up + ONE timed iteration, exactly one collective in it -- timed inside the
    update on the real bucket --, whole-job env-steps / the slowest rank's iteration time, and identical parameters on every
    rank afterwards although the ranks started from different weights and saw different batches."""
    mp.spawn(_iteration_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    recs = [json.load(open(tmp_path / f"it_{r}.json")) for r in range(world)]
    keys = {"algorithm", "num_envs_per_rank", "ticks_per_iteration", "env_steps_per_iteration_all_ranks", "rollout_ms", "update_ms",
            "iteration_ms", "env_steps_per_s_end_to_end", "allreduce_us", "allreduce_us_per_rank", "gradient_bucket_bytes",
            "collectives_per_iteration", "parameter_checksum_per_rank", "parameters_identical_across_ranks", "update_plan",
            "warmup_iterations", "rollouts"}
    for r in recs:
        assert set(r) == keys, set(r) ^ keys
        assert r["algorithm"] == ["PPO"] and r["collectives_per_iteration"] == 1 and r["rollouts"] == 2
        assert r["env_steps_per_iteration_all_ranks"] == world * 30
        assert r["parameters_identical_across_ranks"] and len(set(r["parameter_checksum_per_rank"])) == 1
        assert len(r["allreduce_us_per_rank"]) == world and all(v > 0 for v in r["allreduce_us_per_rank"])
        assert r["allreduce_us"] == max(r["allreduce_us_per_rank"])
        assert r["iteration_ms"] >= max(r["rollout_ms"], r["update_ms"]) > 0
        assert abs(r["env_steps_per_s_end_to_end"] - world * 30 / (r["iteration_ms"] * 1e-3)) < 1e-6 * r["env_steps_per_s_end_to_end"]
        assert r["gradient_bucket_bytes"] == 4 * 2 * sum(p.numel() for p in _StandInTrainer(0).models["runner"].parameters())
    # every rank reports the same (reduced) record
    for k in keys - {"rollouts"}:
        assert all(r[k] == recs[0][k] for r in recs), k


def test_configs3_overrides_are_what_baseline_names():
    """the trainer leg's configuration = BASELINE configs[3] per rank: 2000 replicas, 250 ticks, PPO for both policies"""
    from warp_drive_amd.training import bench_iteration as bi

    ov = bi.configs3_overrides()
    assert ov["trainer"]["num_envs"] == 2000 and ov["trainer"]["train_batch_size"] == 500000
    import inspect

    src = inspect.getsource(bi.run_configs3_iteration)
    assert 'algorithm="PPO"' in src and '"tag_continuous"' in src
