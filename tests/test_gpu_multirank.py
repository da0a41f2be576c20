"""The N > 1 launch path on a 1-GPU box: two ranks, both pinned to device 0 (WD_FORCE_DEVICE) and
rendezvousing over gloo (RCCL refuses two ranks on one GPU).  Exercises what round 1 never started:
a rank > 0 constructing EnvWrapper without an event messenger (reference protocol:
warp_drive/training/utils/device_child_process/child_process_base.py:36-85,
pycuda_function_manager.py:170-181), bench.py launching its own ranks, and the 2-rank trainer (one
gradient bucket, one all-reduce per iteration).  The last test runs the same over RCCL when the box has
two GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.update(WD_FORCE_DEVICE="0", WD_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _bench(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                          "--num-envs", "500", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_started_plainly():
    from tests.hip_harness import require_gpu

    require_gpu()
    one = _bench([])
    # plain `python bench.py --gpus 2`: bench.py starts its own ranks.  (The configs[3] trainer iteration that N > 1 adds to
    # the line is the next test's, at a size two ranks can share one GPU with: 2 x 116 GB of stored activations do not fit.)
    two = _bench(["--gpus", "2", "--trainer-leg", "off"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["num_envs_per_gpu"] == 500
    assert two["config"]["sampler_seeds"] == [274880, 274881]  # seed + rank
    # whole-job value = all ranks' env-steps / slowest rank's time
    assert abs(two["value"] - 2 * 500 * 20 / (two["ms_per_step"] * 20 * 1e-3)) / two["value"] < 1e-6
    # both ranks share ONE device here, so the aggregate cannot double; it must still be the same order
    assert 0.3 * one["value"] < two["value"] < 2.5 * one["value"]
    assert two["roofline"]["samples"] > 0 and two["roofline"]["achieved"] > 0


def test_bench_line_carries_the_configs3_trainer_iteration():
    """`bench.py --gpus 2` (and `--trainer-leg on` at N = 1): the line carries a `trainer` object -- ONE timed PPO training
    iteration per rank after a warm-up, at a small size here -- with exactly one collective per iteration at N = 2, timed
    inside the update on the real bucket (762 200 bytes: both [256, 256] policies), identical parameters on both ranks, and
    end-to-end env-steps/s = all ranks' steps / the slowest rank's iteration.  (Two ranks on ONE device over gloo: the
    control flow of configs[3], not its speed.)"""
    from tests.hip_harness import require_gpu

    require_gpu()
    small = ["--trainer-num-envs", "60", "--trainer-ticks", "12"]
    one = _bench(["--trainer-leg", "on"] + small)["trainer"]
    assert "failed" not in one, one
    assert one["algorithm"] == ["PPO"] and one["collectives_per_iteration"] == 0 and one["allreduce_us"] is None
    assert one["env_steps_per_iteration_all_ranks"] == 60 * 12 and one["gradient_bucket_bytes"] == 762200
    two = _bench(["--gpus", "2"] + small)   # auto: on at N > 1
    t = two["trainer"]
    assert "failed" not in t, t
    assert t["collectives_per_iteration"] == 1 and t["allreduce_us"] > 0 and len(t["allreduce_us_per_rank"]) == 2
    assert t["parameters_identical_across_ranks"] and len(set(t["parameter_checksum_per_rank"])) == 1
    assert t["env_steps_per_iteration_all_ranks"] == 2 * 60 * 12 and t["num_envs_per_rank"] == 60
    assert abs(t["env_steps_per_s_end_to_end"] - 2 * 60 * 12 / (t["iteration_ms"] * 1e-3)) < 1e-6 * t["env_steps_per_s_end_to_end"]
    assert t["iteration_ms"] >= max(t["rollout_ms"], t["update_ms"]) > 0
    assert "gloo" in t["hardware_note"]
    assert set(t["update_plan"]) == {"runner", "tagger"}


def test_bench_line_survives_a_trainer_leg_that_does_not_come_back():
    """The kernel line is the driver's contract; the configs[3] trainer iteration rides along at N > 1.  If that leg does
    not return -- a rank died inside a collective and the others wait for it for ever -- a watchdog on every rank prints
    the line (rank 0) and ends the job with exit code 0.  Forced here with a timeout no iteration can meet."""
    from tests.hip_harness import require_gpu

    require_gpu()
    for extra in (["--trainer-leg", "on"], ["--gpus", "2"]):
        line = _bench(extra + ["--trainer-timeout", "0.5", "--trainer-num-envs", "60", "--trainer-ticks", "12"])
        assert line["value"] > 0 and line["roofline"]["achieved"] > 0
        assert "failed" in line["trainer"] and "0.5 s" in line["trainer"]["failed"], line["trainer"]


def test_train_two_ranks(tmp_path):
    from tests.hip_harness import require_gpu

    require_gpu()
    from warp_drive_amd.training.scripts import launch

    cmd = launch.build_command("tag_gridworld", 2, launch.free_port(),
                               ["--iters", "2", "--num_envs", "40", "--train_batch_size", "400",
                                "--results_dir", str(tmp_path)])
    out = subprocess.run(cmd, capture_output=True, text=True, env=launch.child_environment(_env()), cwd=ROOT,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    recs = []
    for rank in range(2):  # per-device result files, trainer_base.py:627-630
        path = tmp_path / f"results_device_{rank}.json"
        assert path.exists(), os.listdir(tmp_path)
        recs.append([json.loads(l) for l in open(path)])
    assert recs[0][-1]["Iterations Completed"] == 2 and recs[1][-1]["Iterations Completed"] == 2
    # the shared gradient bucket keeps the replicas' models identical: rank 0 saves, the ranks' losses differ (own replicas)
    assert any(f.endswith(".state_dict") for f in os.listdir(tmp_path))


def _nccl_worker(port, out):
    import torch
    import torch.distributed as dist

    from warp_drive_amd.training.models import FullyConnected

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    model = torch.nn.parallel.DistributedDataParallel(FullyConnected(7, [3], [8]).cuda(), device_ids=[0])
    probs, vals = model(torch.randn(4, 5, 7, device="cuda"))
    (probs[0].sum() + vals.sum()).backward()  # bucketed gradient all-reduce through RCCL
    ok = all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    open(out, "w").write(f"{dist.get_backend()} {int(ok)} {t.sum().item()}")
    dist.destroy_process_group()


def test_rccl_backend_loads_and_reduces(tmp_path):
    """backend "nccl" IS RCCL on ROCm: one rank on the one GPU of this box initialises it, runs an
    all-reduce, a barrier and a DistributedDataParallel backward (the collectives bench.py and the trainer
    use at N > 1).  Run in a child process so that the process group does not outlive the test."""
    import multiprocessing as mp
    import socket

    from tests.hip_harness import require_gpu

    require_gpu()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "nccl.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_worker, args=(port, out))
    p.start()
    p.join(300)
    assert p.exitcode == 0, f"RCCL worker exit code {p.exitcode}"
    backend, ok, total = open(out).read().split()
    assert backend == "nccl" and ok == "1" and float(total) == 28.0


def test_two_ranks_over_rccl_when_the_box_has_two_gpus(tmp_path):
    """The real N > 1 path: one rank per GPU over RCCL (WD_DIST_BACKEND / WD_FORCE_DEVICE unset).  Needs two
    visible GPUs -- RCCL refuses two ranks on one device -- and skips with the reason otherwise, so that a
    multi-GPU lease exercises it inside the GPU test tier."""
    import torch

    from tests.hip_harness import require_gpu

    require_gpu()
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: RCCL with world_size 2 needs two devices (the 1-GPU variants above use gloo)")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "WD_FORCE_DEVICE", "WD_DIST_BACKEND"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["allreduce_us"] > 0
    assert d["config"]["sampler_seeds"] == [274880, 274881]
    from warp_drive_amd.training.scripts import launch

    cmd = launch.build_command("tag_gridworld", 2, launch.free_port(),
                               ["--iters", "2", "--num_envs", "40", "--train_batch_size", "400",
                                "--results_dir", str(tmp_path)])
    out = subprocess.run(cmd, capture_output=True, text=True, env=launch.child_environment(env), cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert (tmp_path / "results_device_0.json").exists() and (tmp_path / "results_device_1.json").exists()
