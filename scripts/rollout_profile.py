"""Three rollouts of the trainer at configs[2] (for rocprofv3 --kernel-trace --stats): python scripts/rollout_profile.py float32|bfloat16"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer
dtype = sys.argv[1]
ov = {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "graph_rollout": False, "rollout_dtype": dtype}}
tr = setup_trainer("tag_continuous", ov, results_dir=f"/tmp/rp{dtype}", verbose=False)
for _ in range(3): tr._generate_rollout_batch()
torch.cuda.synchronize()
tr.graceful_close()
