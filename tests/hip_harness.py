"""Helpers shared by the GPU parity tests (everything here goes through the product's
public API: EnvWrapper -> HIP managers -> ctypes C-ABI -> gfx950 kernels)."""
import numpy as np
import torch

from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders
from warp_drive_amd.utils.constants import Constants

OBS, ACT, REW = Constants.OBSERVATIONS, Constants.ACTIONS, Constants.REWARDS


def require_gpu():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    drv.ensure_init(0)


def make_wrapper(env_obj, num_envs, sampler=None, **placeholder_kwargs):
    w = EnvWrapper(env_obj=env_obj, num_envs=num_envs, env_backend="hip")
    w.reset_all_envs()
    kw = dict(training_batch_size_per_env=None, push_data_batch_placeholders=False)
    kw.update(placeholder_kwargs)
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, **kw)
    return w


def push_actions(w, actions):
    t = w.cuda_data_manager.data_on_device_via_torch(ACT)
    a = np.asarray(actions, dtype=np.int32).reshape(tuple(t.shape))
    t.copy_(torch.from_numpy(a))


def pull(w, name):
    return w.cuda_data_manager.pull_data_from_device(name)


def ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


# session-wide totals of the TagContinuous parity comparisons (see tests/conftest.py)
NEAR_TIE_TOTALS = {"near_tie_rows": 0, "rows": 0}
