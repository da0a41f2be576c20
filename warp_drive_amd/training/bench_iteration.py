"""One timed training iteration of the multi-GPU configuration (BASELINE configs[3]: TagContinuous 5 x 100, 2000 replicas per
rank, PPO, gradients of all policies averaged over the ranks by ONE all-reduce of the flat bucket -- RCCL over xGMI on a GPU
node), for `bench.py --gpus N`'s `trainer` object.

What the reference does per iteration at N > 1 (training/utils/distributed_train/distributed_trainer_pycuda.py:12-47 starts
one process per device; trainer_a2c.py:137-146 wraps every policy in its own DistributedDataParallel, i.e. one bucketed
all-reduce per policy inside `loss.backward()`): rollout of `train_batch_size / num_envs` ticks on the rank's own replicas
(no communication), loss, backward + gradient all-reduce, optimizer step.  Here the same iteration is timed from the inside:
the rollout and the update by wall clock around device synchronisations, the collective by events around the ONE
`all_reduce` the gradient bucket issues, and the replicas' models are checked to be identical afterwards."""
import time

import torch

from warp_drive_amd import distributed as wdd


def parameter_checksum(models):
    """(sum, sum of absolute values) of every parameter of every model, float64: equal on every rank iff the ranks applied
    the same averaged gradients to the same initial weights"""
    s = a = 0.0
    for name in sorted(models):
        for p in models[name].parameters():
            d = p.detach().double()
            s += float(d.sum())
            a += float(d.abs().sum())
    return s, a


def measure_training_iteration(trainer, warmup_iterations=1):
    """`warmup_iterations` untimed iterations (graph capture, code-object loads, allocator), then ONE timed iteration
    bracketed by barriers.  Returns the same dict on every rank (times = the slowest rank's)."""
    on_gpu = trainer.device.type == "cuda" and torch.cuda.is_available()

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    bucket = trainer.grad_bucket
    for it in range(int(warmup_iterations)):
        # the rollout has no collective in it: a rank that cannot run it (memory) says so BEFORE anybody enters the update's
        # all-reduce, and every rank leaves together
        err = None
        try:
            trainer._generate_rollout_batch()
            sync()
        except Exception as e:  # noqa: BLE001 -- reported, and agreed on by all ranks
            err = e
        if not wdd.all_ranks_ok(err is None):
            raise RuntimeError(f"warm-up rollout failed on {'this rank: ' + type(err).__name__ + ': ' + str(err) if err else 'another rank'}")
        trainer._update_model_params(it, False)
    sync()
    bucket.time_collectives = True
    collectives_before = bucket.collectives
    wdd.barrier()
    t0 = time.perf_counter()
    trainer._generate_rollout_batch()
    sync()
    t1 = time.perf_counter()
    trainer._update_model_params(int(warmup_iterations), False)
    sync()
    t2 = time.perf_counter()
    wdd.barrier()
    bucket.time_collectives = False
    rollout_s, update_s, iteration_s = (wdd.max_over_ranks(v) for v in (t1 - t0, t2 - t1, t2 - t0))
    steps_all_ranks = wdd.sum_over_ranks(trainer.train_batch_size)
    allreduce_us = bucket.read_allreduce_us()
    per_rank_allreduce = wdd.gather_floats(-1.0 if allreduce_us is None else allreduce_us)
    sums = wdd.gather_floats(parameter_checksum(trainer.models)[0])
    abs_sums = wdd.gather_floats(parameter_checksum(trainer.models)[1])
    world = len(sums)
    return {
        "algorithm": sorted({type(t).__name__ for t in trainer.trainers.values()}),
        "num_envs_per_rank": int(trainer.num_envs), "ticks_per_iteration": int(trainer.batch_len),
        "env_steps_per_iteration_all_ranks": int(steps_all_ranks),
        "rollout_ms": 1e3 * rollout_s, "update_ms": 1e3 * update_s, "iteration_ms": 1e3 * iteration_s,
        "env_steps_per_s_end_to_end": steps_all_ranks / iteration_s,
        # the ONE collective of the iteration, timed inside `_update_model_params` on the real gradient bucket
        "allreduce_us": None if world == 1 or allreduce_us is None else max(per_rank_allreduce),
        "allreduce_us_per_rank": None if world == 1 else per_rank_allreduce,
        "gradient_bucket_bytes": int(4 * bucket.flat.numel()),
        "collectives_per_iteration": int(bucket.collectives - collectives_before),
        "parameter_checksum_per_rank": sums,
        "parameters_identical_across_ranks": bool(all(s == sums[0] for s in sums) and all(a == abs_sums[0] for a in abs_sums)),
        "update_plan": {p: dict(v) for p, v in getattr(trainer, "update_plan", {}).items()},
        "warmup_iterations": int(warmup_iterations),
    }


def configs3_overrides(num_envs=2000, ticks=250):
    """BASELINE configs[3] per rank: the tag_continuous run config with PPO for both policies (`clip_param` from the
    defaults), `num_envs` replicas and `ticks` ticks per iteration on every rank; seed + rank is the trainer's own"""
    return {"trainer": {"num_envs": int(num_envs), "train_batch_size": int(num_envs) * int(ticks), "num_episodes": 10 ** 6},
            "saving": {"metrics_log_freq": 10 ** 6, "model_params_save_freq": 0}}


def run_configs3_iteration(num_envs=2000, ticks=250, warmup_iterations=1, results_dir=None):
    """build the configs[3] trainer on this rank's device (process group already initialised by the caller, or a single
    rank) and time one iteration"""
    import tempfile

    import yaml

    from warp_drive_amd.training.scripts import train as train_script

    ov = configs3_overrides(num_envs, ticks)
    base = yaml.safe_load(open(train_script.os.path.join(train_script._CONFIGS, "tag_continuous.yaml")))
    ov["policy"] = {p: dict(cfg, algorithm="PPO") for p, cfg in base["policy"].items()}
    with tempfile.TemporaryDirectory() as tmp:
        trainer, err = None, None
        try:
            trainer = train_script.setup_trainer("tag_continuous", ov, results_dir=results_dir or tmp, verbose=False)
        except Exception as e:  # noqa: BLE001 -- every rank must learn of it before the first collective of the iteration
            err = e
        if not wdd.all_ranks_ok(err is None):
            if trainer is not None:
                trainer.graceful_close()
            raise RuntimeError(f"trainer set-up failed on {'this rank: ' + type(err).__name__ + ': ' + str(err) if err else 'another rank'}")
        try:
            out = measure_training_iteration(trainer, warmup_iterations)
        finally:
            trainer.graceful_close()
    return out
