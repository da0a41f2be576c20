#!/usr/bin/env python3
"""Single-node multi-GPU launcher (mirror of the reference's
warp_drive/training/utils/distributed_train/distributed_trainer_pycuda.py:1-130 and
warp_drive/utils/device_child_process/child_process_base.py: one child process per device,
rank == device id, an address/port rendezvous on localhost, per-rank result files).

The reference spawns `torch.multiprocessing` children itself and hands them the device id and
an event messenger; here the rendezvous, the child processes and their restart-free lifetime are
`torch.distributed.run`'s, one rank per MI355X, gradients all-reduced by DDP over RCCL/xGMI:

    python -m warp_drive_amd.training.scripts.launch --env tag_continuous            # every visible GPU
    python -m warp_drive_amd.training.scripts.launch --env tag_continuous --num_gpus 4 --iters 20

`trainer.num_envs` and `trainer.train_batch_size` in the run config are PER GPU (replicas never
interact, so scaling out is weak scaling: 8 ranks x 2000 replicas = configs[3])."""
import argparse
import os
import socket
import subprocess
import sys

import yaml

_CONFIGS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "run_configs")


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    import torch

    return torch.cuda.device_count()


def build_command(env, num_gpus, port, passthrough):
    """argv of the torch.distributed.run invocation (one rank per GPU, localhost rendezvous)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={num_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port),
            "-m", "warp_drive_amd.training.scripts.train", "--env", env] + list(passthrough)


def child_environment(base=None):
    env = dict(os.environ if base is None else base)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver stack
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (as bench.py: read at HIP init)
    return env


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--env", "-e", default="tag_continuous")
    ap.add_argument("--num_gpus", "-n", type=int, default=None,
                    help="ranks to start (default: trainer.num_gpus of the run config, capped by the visible GPUs)")
    ap.add_argument("--dry_run", action="store_true", help="print the command instead of running it")
    args, passthrough = ap.parse_known_args(argv)
    n = args.num_gpus
    if n is None:
        cfg = yaml.safe_load(open(os.path.join(_CONFIGS, f"{args.env}.yaml")))
        n = int(cfg.get("trainer", {}).get("num_gpus", 1))
        if not args.dry_run:
            n = max(1, min(n, visible_gpus()))
    cmd = build_command(args.env, n, free_port(), passthrough)
    if args.dry_run:
        print(" ".join(cmd))
        return 0
    if n == 1:  # no rendezvous needed
        cmd = [sys.executable, "-m", "warp_drive_amd.training.scripts.train", "--env", args.env] + list(passthrough)
    return subprocess.call(cmd, env=child_environment())


if __name__ == "__main__":
    sys.exit(main())
