"""Named-array registry for the device (the DataManager of the drop-in boundary).

API mirror of reference warp_drive/managers/data_manager.py:17-485 (abstract
`CUDADataManager`) and pycuda_managers/pycuda_data_manager.py:29-126 (the concrete
backend), re-implemented for a single HIP backend:

  * non-torch arrays live in wd_malloc blocks (pycuda mem_alloc, :118-120);
  * torch-accessible arrays are torch tensors on the HIP device and the kernels get
    `tensor.data_ptr()` -- trainer and kernels alias the same HBM, zero host copy
    (pycuda_data_manager.py:121-126);
  * everything is narrowed to 32 bit on the way in (data_manager.py:263-269,:348-351);
  * device pointers are never re-allocated after registration, because
    CUDAFunctionFeed caches them (function_manager.py:116-134).
"""
import logging
from typing import Dict, Optional

import numpy as np
import torch

from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.utils.data_feed import DataFeed

_SCALAR_TYPES = (int, np.integer, float, np.floating)


def _as_32bit_array(name, value, warn):
    """list/ndarray -> C-contiguous ndarray with 64-bit types narrowed to 32 bit."""
    if isinstance(value, list):
        arr = np.array(value, order="C")
    elif isinstance(value, np.ndarray):
        arr = value
        if not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)
            warn(name, "F_CONTIGUOUS", "C_CONTIGUOUS")
    else:
        raise ValueError(f"the data '{name}' needs to be cast to a list or an array")
    if arr.dtype == np.float64:
        arr = arr.astype(np.float32)
        warn(name, "float64", "float32")
    elif arr.dtype == np.int64:
        arr = arr.astype(np.int32)
        warn(name, "int64", "int32")
    elif arr.dtype == np.bool_:
        arr = arr.astype(np.int32)
    return arr


def _as_32bit_scalar(value):
    return np.int32(value) if isinstance(value, (int, np.integer)) else np.float32(value)


class CUDADataManager:
    """Backend-independent bookkeeping.  Concrete backends implement `_to_device`,
    `pull_data_from_device` and `reset_device`."""

    def __init__(self, num_agents: int = None, num_envs: int = None, blocks_per_env: int = 1,
                 episode_length: int = None):
        assert num_agents is not None and num_envs is not None
        assert blocks_per_env is not None and episode_length is not None
        self._meta_info = {}
        self._host_data = {}
        self._device_data_pointer = {}
        self._device_data_via_torch = {}
        self._shared_constants = {}
        self._scalar_data_list = []
        self._reset_data_list = []
        self._reset_target_to_pool = {}
        self._log_data_list = []
        self._shape = {}
        self._dtype = {}
        self._derived = {}  # source array -> arrays derived from it on the device (register_derived_state)
        self.add_meta_info({"n_agents": num_agents, "episode_length": episode_length,
                            "n_envs": num_envs, "blocks_per_env": blocks_per_env})
        self._register_builtin_arrays()

    # -- built-ins: _log_mask_, _done_, _timestep_ (data_manager.py:75-105)
    def _register_builtin_arrays(self):
        T, E = int(self._meta_info["episode_length"]), int(self._meta_info["n_envs"])
        feed = DataFeed()
        feed.add_data(name="_log_mask_", data=np.zeros(T + 1, dtype=np.int32))
        self.push_data_to_device(feed)
        feed = DataFeed()
        feed.add_data(name="_done_", data=np.zeros(E, dtype=np.int32))
        self.push_data_to_device(feed, torch_accessible=True)
        feed = DataFeed()
        feed.add_data(name="_timestep_", data=np.zeros(E, dtype=np.int32))
        self.push_data_to_device(feed, torch_accessible=False)

    # -- scalars that never go to device memory
    def add_meta_info(self, meta: Dict):
        assert isinstance(meta, dict)
        for key, value in meta.items():
            assert key not in self._meta_info, f"the meta info with name: {key} has already been registered"
            assert isinstance(value, _SCALAR_TYPES), "the meta needs to be casted to a float or an int"
            self._meta_info[key] = _as_32bit_scalar(value)

    def add_shared_constants(self, constants: Dict):
        """Values destined for __constant__ symbols of the code object (:130-191)."""
        for key, value in constants.items():
            assert key not in self._shared_constants, (
                f"the data with name: {key} has already been added as the shared constant")
            if isinstance(value, (np.ndarray, list)):
                arr = _as_32bit_array(key, value, self._type_warning_helper)
                self._shared_constants[key] = arr
                self._shape[key], self._dtype[key] = arr.shape, arr.dtype.name
            elif isinstance(value, _SCALAR_TYPES):
                self._shared_constants[key] = _as_32bit_scalar(value)
                self._shape[key], self._dtype[key] = (), self._shared_constants[key].dtype.name
            else:
                raise ValueError(f"the shared constant '{key}' needs to be cast to a float, int, list or array")

    # -- the main entry point (:193-364)
    def push_data_to_device(self, data: Dict, torch_accessible: bool = False):
        assert isinstance(data, dict)
        for key, content in data.items():
            assert key not in self._host_data, f"the data with name: {key} has already been registered at the host"
            value, attrs = content["data"], content["attributes"]
            is_pool = bool(attrs.get("is_reset_pool", False))
            keep_reset_copy = bool(attrs["save_copy_and_apply_at_reset"]) and not is_pool
            keep_log = bool(attrs["log_data_across_episode"]) and not is_pool

            if isinstance(value, (np.ndarray, list)):
                assert key not in self._device_data_pointer, f"the data with name: {key} has already been pushed to device"
                if is_pool:
                    target = attrs["reset_target"]
                    assert target not in self._reset_target_to_pool, (
                        f"the data with name: {key} has already been registered at the reset_target_to_pool")
                    assert target not in self._reset_data_list, (
                        f"the data with name: {target} has already been registered at the reset_data_list")
                    self._reset_target_to_pool[target] = key
                arr = _as_32bit_array(key, value, self._type_warning_helper)
                self._host_data[key] = arr
                self._to_device(name=key, name_on_device=None, torch_accessible=torch_accessible)
                self._remember(key, arr)
                if keep_reset_copy:
                    assert key not in self._reset_data_list, f"the data with name: {key} has already been registered at the reset_data_list"
                    assert key not in self._reset_target_to_pool, f"the data with name: {key} has already been registered at the reset_target_to_pool"
                    self._remember(f"{key}_at_reset", arr)
                    self._to_device(key, name_on_device=f"{key}_at_reset", torch_accessible=False)
                    self._reset_data_list.append(key)
                if keep_log:
                    assert not torch_accessible, "log_data_across_episode=True is not supported for the data that have torch_accessible=True"
                    assert key not in self._log_data_list, f"the data with name: {key} has already been registered at the log_data_list"
                    assert arr.shape[0] == self.meta_info("n_envs") and arr.shape[1] == self.meta_info("n_agents")
                    log_key = f"{key}_for_log"
                    log_arr = np.zeros((int(self.meta_info("episode_length")) + 1, *arr[0].shape), dtype=arr.dtype)
                    self._host_data[log_key] = log_arr
                    self._remember(log_key, log_arr)
                    self._to_device(log_key, name_on_device=log_key)
                    self._log_data_list.append(key)
            elif isinstance(value, _SCALAR_TYPES):
                # scalars are passed by value at launch time; no device memory (:341-359)
                assert key not in self._scalar_data_list, f"the data with name: {key} has already been pushed to device"
                self._host_data[key] = _as_32bit_scalar(value)
                self._shape[key], self._dtype[key] = (), self._host_data[key].dtype.name
                self._scalar_data_list.append(key)
            else:
                raise ValueError(f"the data '{key}' needs to be casted to a float, int, list or array")

    # -- device-side state DERIVED from other arrays (no reference counterpart: the reference keeps none)
    def register_derived_state(self, target: str, sources):
        """`target` is bookkeeping a kernel derives from the arrays `sources` (e.g. TagContinuous'
        `obs_rows_cleared`: which observation rows are all zeros already).  Whenever a source is rewritten from
        the host (`reset_device`, `invalidate_derived`) the target is zero-filled, which for such arrays means
        "nothing is known": the kernels then recompute."""
        assert target in self._device_data_pointer, f"{target} is not on the device"
        for src in sources:
            self._derived.setdefault(src, [])
            if target not in self._derived[src]:
                self._derived[src].append(target)

    def invalidate_derived(self, source: str):
        """call after writing `source` on the device by other means than this manager (e.g. through the tensor of
        `data_on_device_via_torch`): zero-fills every array derived from it"""
        for target in self._derived.get(source, []):
            self._zero_fill(target)

    def _zero_fill(self, name: str):
        raise NotImplementedError

    def _remember(self, key, arr):
        self._shape[key] = arr.shape
        self._dtype[key] = arr.dtype.name
        logging.info(f"- {key:<80}: dtype={arr.dtype.name:<10}, shape={arr.shape}")

    # -- backend hooks
    def _to_device(self, name: str, name_on_device: Optional[str] = None, torch_accessible: bool = False):
        raise NotImplementedError

    def pull_data_from_device(self, name: str):
        raise NotImplementedError

    def reset_device(self, name: Optional[str] = None):
        raise NotImplementedError

    # -- lookups
    def data_on_device_via_torch(self, name: str) -> torch.Tensor:
        assert name in self._device_data_via_torch
        return self._device_data_via_torch[name]

    def meta_info(self, name: str):
        assert name in self._meta_info
        return self._meta_info[name]

    def shared_constant(self, name: str):
        assert name in self._shared_constants
        return self._shared_constants[name]

    def device_data(self, name: str):
        """Device pointer for arrays, the 32-bit host value for scalars (:409-418)."""
        if name in self._scalar_data_list:
            return self._host_data[name]
        assert name in self._device_data_pointer, f"{name} is not on the device"
        return self._device_data_pointer[name]

    def is_data_on_device(self, name: str) -> bool:
        return name in self._device_data_pointer

    def is_data_on_device_via_torch(self, name: str) -> bool:
        return self.is_data_on_device(name) and name in self._device_data_via_torch

    def get_shape(self, name: str):
        assert name in self._shape
        return self._shape[name]

    def get_dtype(self, name: str):
        assert name in self._dtype
        return self._dtype[name]

    def get_reset_pool(self, name: str):
        assert name in self._reset_target_to_pool
        return self._reset_target_to_pool[name]

    def _type_warning_helper(self, key, old, new, comment=None):
        logging.info(f"{self.__class__.__name__} casts the data '{key}' from type {old} to {new}")

    host_data = property(lambda self: self._host_data)
    scalar_data_list = property(lambda self: self._scalar_data_list)
    reset_data_list = property(lambda self: self._reset_data_list)
    reset_target_to_pool = property(lambda self: self._reset_target_to_pool)
    log_data_list = property(lambda self: self._log_data_list)
    device_data_via_torch = property(lambda self: self._device_data_via_torch)


class HIPDataManager(CUDADataManager):
    """The single concrete backend: HIP device memory through libwdhip.so.

    `device_id` selects the GPU of this process (one process per GPU, like the
    reference's process_id, env_wrapper.py:58,193)."""

    def __init__(self, num_agents: int = None, num_envs: int = None, blocks_per_env: int = 1,
                 episode_length: int = None, device_id: int = 0):
        self._device_id = int(device_id)
        drv.ensure_init(self._device_id)
        if not torch.cuda.is_available():
            raise drv.HipDriverError("HIPDataManager needs a visible MI355X (torch.cuda.is_available() is False)")
        self._torch_device = torch.device("cuda", self._device_id)
        self._owned = []
        super().__init__(num_agents=num_agents, num_envs=num_envs, blocks_per_env=blocks_per_env,
                         episode_length=episode_length)

    def _to_device(self, name, name_on_device=None, torch_accessible=False):
        assert name in self._host_data
        host = self._host_data[name]
        dst = name if name_on_device is None else name_on_device
        assert dst not in self._device_data_pointer
        if torch_accessible:
            t = torch.from_numpy(np.ascontiguousarray(host)).to(self._torch_device)
            self._device_data_via_torch[dst] = t
            self._device_data_pointer[dst] = drv.DevicePtr(t.data_ptr(), host.nbytes, owner=t)
        else:
            p = drv.mem_alloc(max(host.nbytes, 4))
            if host.nbytes:
                drv.memcpy_htod(p, host)
            self._owned.append(p)
            self._device_data_pointer[dst] = p

    def pull_data_from_device(self, name: str):
        if name in self._scalar_data_list:
            return self._host_data[name]
        if name in self._device_data_via_torch:
            return self._device_data_via_torch[name].cpu().numpy()
        assert name in self._device_data_pointer, f"{name} is not on the device"
        out = np.empty(self._shape[name], dtype=self._dtype[name])
        drv.synchronize()
        if out.nbytes:
            drv.memcpy_dtoh(out, self._device_data_pointer[name])
        return out

    def reset_device(self, name: Optional[str] = None):
        """Host -> device refresh of registered arrays, in place (pointers stay valid)."""
        names = [name] if name is not None else [k for k in self._host_data if k in self._device_data_pointer]
        for key in names:
            assert key in self._device_data_pointer and key in self._host_data
            host = self._host_data[key]
            if key in self._device_data_via_torch:
                self._device_data_via_torch[key].copy_(torch.from_numpy(host))
            elif host.nbytes:
                drv.memcpy_htod(self._device_data_pointer[key], host)
            self.invalidate_derived(key)

    def _zero_fill(self, name: str):
        if name in self._device_data_via_torch:
            self._device_data_via_torch[name].zero_()
            return
        zeros = np.zeros(self._shape[name], dtype=self._dtype[name])
        if zeros.nbytes:
            drv.memcpy_htod(self._device_data_pointer[name], zeros)

    def __del__(self):
        for p in getattr(self, "_owned", []):
            try:
                p.free()
            except Exception:
                pass
