"""Kernel-level view of the trainer's update at configs[2]: run under
   rocprofv3 --kernel-trace --stats -- python scripts/update_profile.py [float32|bfloat16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer
dt = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
tr = setup_trainer("tag_continuous", {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "update_dtype": dt}},
                   results_dir="/tmp/up", verbose=False)
for it in range(3):   # (a rollout before every update, as in training: the update reads the activations it stored)
    tr._generate_rollout_batch()
    tr._update_model_params(it, False)
torch.cuda.synchronize()
tr.graceful_close()
