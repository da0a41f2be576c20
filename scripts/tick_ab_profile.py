import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from warp_drive_amd.training.scripts.train import setup_trainer
fast = sys.argv[1] == "1"
tr = setup_trainer("tag_continuous", {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "fused_tick": fast, "reuse_rollout_activations": False}}, results_dir="/tmp/tab", verbose=False)
for _ in range(3):
    tr._generate_rollout_batch()
torch.cuda.synchronize()
tr.graceful_close()
