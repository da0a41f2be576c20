"""Kernel registry, launch geometry and the launch wrappers of the rollout path.

API mirror of reference warp_drive/managers/function_manager.py:21-422 (abstract
CUDAFunctionManager / CUDAFunctionFeed / CUDASampler / CUDAEnvironmentReset /
CUDALogController) plus ONE concrete backend that replaces both
pycuda_managers/pycuda_function_manager.py and numba_managers/numba_function_manager.py:

  HIPFunctionManager   loads the prebuilt gfx950 code object through libwdhip.so
  HIPSampler           categorical (`sample_actions`) and OU/Gaussian (`sample_ou_process`)
  HIPEnvironmentReset  reset-when-done: ONE fused launch for all registered arrays
  HIPLogController     episode logger

Kernel sizes are runtime arguments, so there is no per-run templating/compilation
(pycuda_function_manager.py:133-232): `compile_and_load_hip` only (re)builds the
in-tree code object when its sources are newer, rank 0 only, with the same
multiprocessing-Event hand-shake as the reference (:170-181,:227-228).
"""
import logging
import os
import time
import zlib
from typing import Optional

import numpy as np
import torch

from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.managers.data_manager import CUDADataManager
from warp_drive_amd.utils.data_feed import DataFeed

_WAVE = 64


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


class CUDAFunctionManager:
    """Launch geometry + kernel lookup (function_manager.py:21-94)."""

    def __init__(self, num_agents: int = 1, num_envs: int = 1, blocks_per_env: int = 1, process_id: int = 0):
        if num_agents % blocks_per_env != 0:
            logging.warning("`num_agents` is not divisible by `blocks_per_env`; kernels must bound-check agent ids")
        self._num_agents = int(num_agents)
        self._num_envs = int(num_envs)
        self._blocks_per_env = int(blocks_per_env)
        self._process_id = process_id
        # the reference's default geometry: ceil(agents / blocks_per_env) threads, envs*blocks_per_env blocks
        self._block = (int((self._num_agents - 1) // self._blocks_per_env + 1), 1, 1)
        self._grid = (int(self._num_envs * self._blocks_per_env), 1)
        self._default_functions_initialized = False

    def initialize_default_functions(self):
        raise NotImplementedError

    def initialize_functions(self, func_names: Optional[list] = None):
        raise NotImplementedError

    def _get_function(self, fname):
        raise NotImplementedError

    @property
    def get_function(self):
        return self._get_function

    block = property(lambda self: self._block)
    grid = property(lambda self: self._grid)
    blocks_per_env = property(lambda self: self._blocks_per_env)


class CUDAFunctionFeed:
    """Names -> positional kernel arguments, resolved once and cached
    (function_manager.py:96-134).  `(name, "meta")`, `(name, "shared")` and
    `(name, "device")` tuples select the source."""

    def __init__(self, data_manager: CUDADataManager):
        self.data_manager = data_manager
        self._function_feeds = None

    def __call__(self, arguments: list) -> list:
        if self._function_feeds is None:
            dm, resolved = self.data_manager, []
            for arg in arguments:
                if isinstance(arg, str):
                    resolved.append(dm.device_data(arg))
                elif isinstance(arg, tuple):
                    key, source = arg[0], arg[1].lower()
                    if source in ("d", "device"):
                        resolved.append(dm.device_data(key))
                    elif source in ("m", "meta"):
                        resolved.append(dm.meta_info(key))
                    elif source in ("s", "shared"):
                        resolved.append(dm.shared_constant(key))
                    else:
                        raise Exception(f"Unknown definition of CUDA function feed: {arg}")
                else:
                    raise Exception(f"Unknown definition of CUDA function feed: {arg}")
            self._function_feeds = resolved
        return self._function_feeds


class _NeedsDefaultFunctions:
    def __init__(self, function_manager, who):
        assert function_manager._default_functions_initialized, (
            f"Default functions must be initialized before {who} can work; "
            f"call function_manager.initialize_default_functions()")
        self._function_manager = function_manager
        self._block = function_manager.block
        self._grid = function_manager.grid
        self._blocks_per_env = function_manager.blocks_per_env
        self._num_envs = function_manager._num_envs


class CUDASampler(_NeedsDefaultFunctions):
    """function_manager.py:137-208"""

    def __init__(self, function_manager: CUDAFunctionManager):
        super().__init__(function_manager, "the sampler")
        self._random_initialized = False

    def init_random(self, seed: Optional[int] = None):
        raise NotImplementedError

    def register_actions(self, data_manager: CUDADataManager, action_name: str, num_actions: int,
                         is_deterministic=False):
        """Registers the per-action scratch the reference keeps (:170-199): `<name>_cum_distr`
        (categorical) or `<name>_ou_state` (deterministic / continuous).  The HIP categorical
        kernel keeps the prefix sum in a register, so `_cum_distr` is registered for shape
        compatibility only."""
        n_agents = data_manager.get_shape(action_name)[1]
        if is_deterministic:
            num_actions = 1
        scratch = np.zeros((self._grid[0], n_agents, int(num_actions)), dtype=np.float32)
        feed = DataFeed()
        feed.add_data(name=f"{action_name}_ou_state" if is_deterministic else f"{action_name}_cum_distr",
                      data=scratch)
        data_manager.push_data_to_device(feed)

    def sample(self, data_manager, distribution: torch.Tensor, action_name: str, **sample_params):
        raise NotImplementedError


class CUDAEnvironmentReset(_NeedsDefaultFunctions):
    """function_manager.py:211-292"""

    def __init__(self, function_manager: CUDAFunctionManager):
        super().__init__(function_manager, "the environment resetter")
        self._cuda_custom_reset = None
        self._cuda_reset_feed = None
        self._random_initialized = False

    def register_custom_reset_function(self, data_manager, reset_function_name=None):
        raise NotImplementedError

    def custom_reset(self, args: Optional[list] = None, block=None, grid=None):
        raise NotImplementedError

    def init_reset_pool(self, data_manager, seed: Optional[int] = None):
        raise NotImplementedError

    @staticmethod
    def _force_flag(mode):
        if mode == "if_done":
            return np.int32(0)
        if mode == "force_reset":
            return np.int32(1)
        raise Exception(f"unknown reset mode: {mode}, only accept 'if_done' and 'force_reset' ")

    def reset_when_done(self, data_manager, mode: str = "if_done", undo_done_after_reset: bool = True):
        force_reset = self._force_flag(mode)
        self.reset_when_done_deterministic(data_manager, force_reset)
        self.reset_when_done_from_pool(data_manager, force_reset)
        if undo_done_after_reset:
            self._undo_done_flag_and_reset_timestep(data_manager, force_reset)

    def reset_when_done_deterministic(self, data_manager, force_reset):
        raise NotImplementedError

    def reset_when_done_from_pool(self, data_manager, force_reset):
        raise NotImplementedError

    def _undo_done_flag_and_reset_timestep(self, data_manager, force_reset):
        raise NotImplementedError


class CUDALogController(_NeedsDefaultFunctions):
    """function_manager.py:295-422"""

    def __init__(self, function_manager: CUDAFunctionManager):
        super().__init__(function_manager, "the log controller")
        self.last_valid_step = -1
        self._env_id = None

    def update_log(self, data_manager, step: int):
        assert step > self.last_valid_step, "update_log is trying to update the existing timestep"
        self._log_one_step(data_manager, step, self._env_id)
        self._update_log_mask(data_manager, step)

    def reset_log(self, data_manager, env_id: int = 0):
        self._env_id = env_id
        self.last_valid_step = -1
        self._reset_log_mask(data_manager)
        self.update_log(data_manager, step=0)

    def fetch_log(self, data_manager, names=None, last_step=None, check_last_valid_step=True):
        if check_last_valid_step:
            self._cuda_check_last_valid_step(data_manager)
        upto = last_step if (last_step is not None and last_step <= self.last_valid_step) else self.last_valid_step
        out = {}
        for name in (data_manager.log_data_list if names is None else names):
            key = f"{name}_for_log"
            buf = data_manager.pull_data_from_device(key)
            assert len(buf) == int(data_manager.meta_info("episode_length")) + 1
            out[key] = buf[: upto + 1]
        return out

    def _cuda_check_last_valid_step(self, data_manager):
        mask = data_manager.pull_data_from_device("_log_mask_")
        ones, zeros = np.flatnonzero(mask == 1), np.flatnonzero(mask == 0)
        if len(ones) and len(zeros) and zeros[0] < ones[-1]:
            raise Exception("there is invalid log data in the middle")
        found = int(ones[-1]) if len(ones) else -1
        assert found == self.last_valid_step, (
            f"inconsistency of last_valid_step derived from dense_log_mask = {found} "
            f"and the step() function = {self.last_valid_step}")

    def _log_one_step(self, data_manager, step, env_id=0):
        raise NotImplementedError

    def _update_log_mask(self, data_manager, step):
        raise NotImplementedError

    def _reset_log_mask(self, data_manager):
        raise NotImplementedError


# =====================================================================================
#                                   HIP backend
# =====================================================================================
DEFAULT_FUNCTION_NAMES = [
    # same list as pycuda_function_manager.py:319-332 ...
    "reset_log_mask", "update_log_mask", "log_one_step_in_float", "log_one_step_in_int",
    "reset_in_float_when_done_2d", "reset_in_int_when_done_2d", "reset_in_float_when_done_3d",
    "reset_in_int_when_done_3d", "undo_done_flag_and_reset_timestep", "init_random", "free_random",
    "sample_actions",
    # ... plus what the Numba backend adds (numba_function_manager.py) and the fused reset
    "sample_ou_process", "reset_when_done_fused", "reset_when_done_from_pool",
]


class HIPFunctionManager(CUDAFunctionManager):
    def __init__(self, num_agents: int = 1, num_envs: int = 1, blocks_per_env: int = 1, process_id: int = 0,
                 device_id: Optional[int] = None):
        super().__init__(num_agents=num_agents, num_envs=num_envs, blocks_per_env=blocks_per_env,
                         process_id=process_id)
        if self._blocks_per_env != 1:
            # multi-block replicas need a cross-workgroup spin barrier (env_thread_sync.cu:31-62);
            # out of scope: no BASELINE config has > 1024 agents.
            raise NotImplementedError("the HIP backend supports blocks_per_env == 1 only")
        self._device_id = int(process_id if device_id is None else device_id)
        drv.ensure_init(self._device_id)
        self._module = None
        self._extra_modules = None  # {path: Module} of the other code objects, loaded when a function is first asked for
        self._functions = {}
        self._function_names = []

    # ---- loading
    def load_hip_from_binary_file(self, hsaco: Optional[str] = None, default_functions_included: bool = True):
        assert self._module is None, "the code object has already been loaded, not allowed to load twice"
        self._module = drv.Module(hsaco or drv.HSACO_PATH)
        logging.info(f"loaded the HIP code object {self._module.path}")
        if default_functions_included:
            self.initialize_default_functions()

    load_cuda_from_binary_file = load_hip_from_binary_file  # reference spelling (:115)

    def compile_and_load_hip(self, env_name: Optional[str] = None, template_header_file=None,
                             template_runner_file=None, template_path=None,
                             default_functions_included: bool = True, customized_env_registrar=None,
                             event_messenger=None):
        """(Re)build the in-tree code object if it is stale, exactly once per node, then load it.

        With an `event_messenger` (the reference's own child-process launcher) the protocol is the
        reference's: process 0 builds and sets the Event, the others wait for it
        (pycuda_function_manager.py:170-181,:227-228).  Without one -- ranks started by
        `torch.distributed.run`, which shares no multiprocessing.Event -- every rank takes an
        exclusive file lock around the staleness check, so the first one in builds and the rest find
        a fresh code object; who builds is independent of the device a rank drives."""
        from warp_drive_amd import build as wd_build

        if event_messenger is not None:
            if self._process_id > 0:
                event_messenger.wait(timeout=120)
                if not event_messenger.is_set():
                    raise Exception(f"Process {self._process_id} fails to get the successful compilation message ... ")
            else:
                wd_build.build_kernels()
                event_messenger.set()
        else:
            wd_build.build_kernels_locked()
        self.load_hip_from_binary_file(default_functions_included=default_functions_included)

    compile_and_load_cuda = compile_and_load_hip

    # ---- kernels
    def initialize_default_functions(self):
        self.initialize_functions(DEFAULT_FUNCTION_NAMES)
        self._default_functions_initialized = True

    def initialize_functions(self, func_names: Optional[list] = None):
        assert self._module is not None, "load the code object first: load_hip_from_binary_file()"
        for fname in func_names or []:
            if fname in self._functions:
                continue
            module = self._module_of(fname) or self._module
            self._functions[fname] = module.get_function(fname)  # (raises with the name when nobody has it)
            self._function_names.append(fname)

    def _module_of(self, fname):
        """the loaded code object that holds kernel `fname`: the main one, or -- looked up in the build's manifest and
        loaded on first use -- one of the others (warp_drive_amd/build.py UNITS); None when no object has it"""
        if self._module.has_function(fname):
            return self._module
        if self._extra_modules is None:
            self._extra_modules = {}
        path = drv.code_object_of(fname)
        if path is None or path == self._module.path:
            return None
        if path not in self._extra_modules:
            self._extra_modules[path] = drv.Module(path)
            logging.info(f"loaded the HIP code object {path}")
        return self._extra_modules[path]

    def has_function(self, fname):
        if self._module is None:
            return False
        return self._module.has_function(fname) or drv.code_object_of(fname) is not None

    def code_object_path_of(self, fname):
        """file the kernel `fname` was (or would be) loaded from -- what a counter record is keyed to"""
        if self._module is not None and self._module.has_function(fname):
            return self._module.path
        return drv.code_object_of(fname)

    def global_address(self, name):
        """device address of a __device__ / __constant__ symbol of the MAIN code object (for kernels of the extra
        code objects that read a table the host uploads there)"""
        return self._module.get_global(name)[0]

    def initialize_shared_constants(self, data_manager, constant_names: list):
        """Upload DataManager shared constants into __constant__ symbols (:363-379)."""
        for cname in constant_names:
            value = np.ascontiguousarray(data_manager.shared_constant(cname))
            dst, nbytes = self._module.get_global(cname)
            assert value.nbytes <= nbytes, f"shared constant {cname}: {value.nbytes} B does not fit {nbytes} B"
            drv.memcpy_htod(dst, value)

    def _get_function(self, fname):
        assert fname in self._functions, f"{fname} is not defined"
        return self._functions[fname]

    cuda_function_names = property(lambda self: self._function_names)
    _cuda_function_names = property(lambda self: self._function_names)

    # ---- wave64-aware geometry for replica-packed kernels
    def packed_geometry(self, n_agents: Optional[int] = None, max_threads: int = 512, prefer_large: bool = False):
        """(envs_per_block, block, grid): pack whole replicas into a block of at most `max_threads`
        threads with the fewest idle lanes (105 agents, max 256: ONE replica on 128 threads -- 3 replicas
        on 320 threads would fill 98 % of the lanes but leave 2.6 blocks per CU at 2000 replicas; 5 agents:
        12 replicas per wavefront)."""
        N = int(self._num_agents if n_agents is None else n_agents)
        if N >= max_threads:
            threads = _round_up(N, _WAVE)
            assert threads <= 1024, "more than 1024 agents per replica needs blocks_per_env > 1 (out of scope)"
            return 1, (threads, 1, 1), (self._num_envs, 1)
        best = None
        for epb in range(1, max(1, max_threads // N) + 1):
            threads = _round_up(epb * N, _WAVE)
            if threads > max_threads:
                break
            waste = 1.0 - epb * N / threads
            if best is None or waste < best[0] - 1e-9 or (prefer_large and waste < best[0] + 1e-9):
                best = (waste, epb, threads)
        _, epb, threads = best
        epb = threads // N  # exactly what the kernels derive from blockDim.x
        return epb, (threads, 1, 1), ((self._num_envs + epb - 1) // epb, 1)


def _stream_tag(name):
    return np.int32(zlib.crc32(name.encode()) & 0x7FFFFFFF)


class HIPSampler(CUDASampler):
    """Replaces PyCUDASampler (pycuda_function_manager.py:486-590) and NumbaSampler
    (numba_function_manager.py:248-364)."""

    ROWS_PER_BLOCK = 256
    MAX_DYNAMIC_LDS = 64 * 1024  # dynamic LDS a plain launch may ask for

    def __init__(self, function_manager: HIPFunctionManager):
        super().__init__(function_manager)
        self.sample_actions = function_manager.get_function("sample_actions")
        self.sample_ou_process = function_manager.get_function("sample_ou_process")
        self._rng_state = None
        self._n_threads = function_manager._num_envs * function_manager._num_agents

    def init_random(self, seed: Optional[int] = None):
        if seed is None:
            seed = int(time.time())
            logging.info(f"random seed is not provided, using the current timestamp {seed}")
        seed = np.int32(np.int64(seed) & 0x7FFFFFFF)
        if self._rng_state is None:
            self._rng_state = drv.mem_alloc(4 * (4 + self._n_threads))
        init = self._function_manager.get_function("init_random")
        init(self._rng_state, seed, np.int32(self._n_threads), block=(256, 1, 1),
             grid=(max(1, min(1024, (self._n_threads + 255) // 256)), 1))
        self._random_initialized = True

    @property
    def rng_state(self):
        return self._rng_state

    def categorical_launch(self, distribution_ptr, action_ptr, n_rows, n_actions, use_argmax, tag,
                           out_stride=1, out_offset=0):
        """(function, args, block, grid, shared) of one categorical draw; shared with the
        rollout launch plan."""
        stride = int(n_actions) | 1  # odd row stride: conflict-free LDS reads
        rows = self.ROWS_PER_BLOCK
        while rows > 8 and rows * stride * 4 + 128 > self.MAX_DYNAMIC_LDS:  # long rows: fewer rows per block
            rows //= 2
        assert rows * stride * 4 + 128 <= self.MAX_DYNAMIC_LDS, (
            f"sample_actions stages {rows} rows of {n_actions} probabilities in LDS: more than "
            f"{self.MAX_DYNAMIC_LDS // (8 * 4) - 1} actions per head are not supported")
        grid = max(1, min(8192, (int(n_rows) + rows - 1) // rows))
        args = (self._rng_state, distribution_ptr, action_ptr, drv.DevicePtr(0), np.int32(n_rows),
                np.int32(n_actions), np.int32(use_argmax), np.int32(stride), tag, np.int32(out_stride),
                np.int32(out_offset))
        # (+128 B: the unrolled read-back of a short row runs up to 24 entries past the row's start)
        return self.sample_actions, args, (rows, 1, 1), (grid, 1), rows * stride * 4 + 128

    def sample(self, data_manager, distribution: torch.Tensor, action_name: str, **sample_params):
        assert self._random_initialized, "sample() requires the random seed initialized first, please call init_random()"
        assert torch.is_tensor(distribution)
        assert distribution.is_contiguous(), "distribution is required to be C contiguous"
        assert distribution.dtype == torch.float32
        assert distribution.shape[0] == self._num_envs
        n_agents = int(distribution.shape[1])
        assert data_manager.get_shape(action_name)[1] == n_agents
        n_actions = int(distribution.shape[2])
        n_rows = self._num_envs * n_agents
        assert n_rows <= self._n_threads
        if n_actions > 1:
            assert data_manager.get_shape(f"{action_name}_cum_distr")[2] == n_actions
            fn, args, block, grid, shared = self.categorical_launch(
                distribution, data_manager.device_data(action_name), n_rows, n_actions,
                bool(sample_params.get("use_argmax", False)), _stream_tag(action_name))
            fn(*args, block=block, grid=grid, shared=shared)
        else:
            # deterministic (continuous) action + OU noise, numba_function_manager.py:348-364
            self.sample_ou_process(
                self._rng_state, distribution, data_manager.device_data(action_name),
                data_manager.device_data(f"{action_name}_ou_state"),
                np.float32(sample_params.get("damping", 0.15)), np.float32(sample_params.get("stddev", 0.2)),
                np.float32(sample_params.get("scale", 1.0)), np.int32(n_rows), _stream_tag(action_name),
                block=(256, 1, 1), grid=(max(1, min(4096, (n_rows + 255) // 256)), 1))

    @staticmethod
    def assign(data_manager, actions: np.ndarray, action_name: str):
        """Write actions straight into the action tensor (testing / debugging, :574-590)."""
        assert data_manager.is_data_on_device_via_torch(action_name)
        assert actions.shape == data_manager.get_shape(action_name)
        assert actions.dtype.name == data_manager.get_dtype(action_name)
        t = data_manager.data_on_device_via_torch(action_name)
        t[:] = torch.from_numpy(actions).to(t.device)

    def __del__(self):
        if getattr(self, "_rng_state", None) is not None:
            try:
                self._rng_state.free()
            except Exception:
                pass


class HIPEnvironmentReset(CUDAEnvironmentReset):
    """Replaces PyCUDAEnvironmentReset (pycuda_function_manager.py:593-753) and
    NumbaEnvironmentReset (numba_function_manager.py:367-641)."""

    BLOCK = 256

    def __init__(self, function_manager: HIPFunctionManager):
        super().__init__(function_manager)
        fm = function_manager
        self.reset_fused = fm.get_function("reset_when_done_fused")
        self.reset_from_pool = fm.get_function("reset_when_done_from_pool")
        self.undo = fm.get_function("undo_done_flag_and_reset_timestep")
        self._table = None
        self._table_names = None
        self._retired_tables = []
        self._pool_rng = None

    # ---- custom reset kernels (same contract as the reference)
    def register_custom_reset_function(self, data_manager, reset_function_name=None):
        if reset_function_name is None or not self._function_manager.has_function(reset_function_name):
            return
        self._function_manager.initialize_functions([reset_function_name])
        self._cuda_custom_reset = self._function_manager.get_function(reset_function_name)
        self._cuda_reset_feed = CUDAFunctionFeed(data_manager)

    def custom_reset(self, args: Optional[list] = None, block=None, grid=None):
        assert self._cuda_custom_reset is not None and self._cuda_reset_feed is not None, (
            "Custom Reset function is not defined, call register_custom_reset_function() first")
        assert args is None or isinstance(args, list)
        block = self._block if block is None else block
        grid = self._grid if grid is None else grid
        feed = self._cuda_reset_feed(args) if args else []
        self._cuda_custom_reset(*feed, block=block, grid=grid)

    # ---- deterministic reset: one fused launch
    def _geometry(self):
        return (self.BLOCK, 1, 1), (max(1, min(self._num_envs, 4096)), 1)

    def _build_table(self, data_manager):
        names = list(data_manager.reset_data_list)
        n_envs = int(data_manager.meta_info("n_envs"))
        entries = np.zeros(len(names), dtype=np.dtype([("data", "<u8"), ("ref", "<u8"), ("row", "<i4"), ("pad", "<i4")]))
        for i, name in enumerate(names):
            shape = data_manager.get_shape(name)
            assert shape[0] == n_envs, "reset function assumes the 0th dimension is n_envs"
            dtype = data_manager.get_dtype(name)
            if "float32" not in dtype and "int32" not in dtype:
                raise Exception(f"unknown dtype: {dtype}")
            entries[i] = (int(data_manager.device_data(name)), int(data_manager.device_data(f"{name}_at_reset")),
                          int(np.prod(shape[1:])) if len(shape) > 1 else 1, 0)
        if self._table is not None:
            # launch plans / hipGraphs built earlier hold the old table's address: it stays allocated
            # (and valid for the arrays it lists) until this object goes away
            self._retired_tables.append(self._table)
        self._table = drv.mem_alloc(max(entries.nbytes, 8))
        if entries.nbytes:
            drv.memcpy_htod(self._table, entries.view(np.uint8))
        self._table_names = names

    def fused_launch(self, data_manager, force_reset, undo):
        """(function, args, block, grid) of the fused reset; shared with the rollout plan."""
        if self._table_names != list(data_manager.reset_data_list):
            self._build_table(data_manager)
        block, grid = self._geometry()
        args = (self._table, np.int32(len(self._table_names)), data_manager.device_data("_done_"),
                data_manager.device_data("_timestep_"), np.int32(force_reset), np.int32(undo),
                data_manager.meta_info("n_envs"))
        return self.reset_fused, args, block, grid

    def reset_when_done(self, data_manager, mode: str = "if_done", undo_done_after_reset: bool = True):
        force_reset = self._force_flag(mode)
        if len(data_manager.reset_target_to_pool) == 0:
            # common case: restore every registered array AND clear done/timestep in one launch
            fn, args, block, grid = self.fused_launch(data_manager, force_reset, 1 if undo_done_after_reset else 0)
            fn(*args, block=block, grid=grid)
            return
        super().reset_when_done(data_manager, mode, undo_done_after_reset)

    def reset_when_done_deterministic(self, data_manager, force_reset):
        if len(data_manager.reset_data_list) == 0:
            return
        fn, args, block, grid = self.fused_launch(data_manager, force_reset, 0)
        fn(*args, block=block, grid=grid)

    # ---- reset from a pool (Numba-only in the reference; PyCUDA stubs it, :661-666,:736-742)
    def init_reset_pool(self, data_manager, seed: Optional[int] = None):
        if len(data_manager.reset_target_to_pool) == 0:
            return
        for name, pool_name in data_manager.reset_target_to_pool.items():
            d_shape, p_shape = data_manager.get_shape(name), data_manager.get_shape(pool_name)
            assert data_manager.get_dtype(name) == data_manager.get_dtype(pool_name), (
                f"Inconsistency of dtype is found for data: {name} and its reset pool: {pool_name}")
            assert d_shape[0] == self._num_envs and p_shape[0] > 1
            assert tuple(d_shape[1:]) == tuple(p_shape[1:]), (
                f"Inconsistency of shape is found for data: {name} and its reset pool: {pool_name}")
        if seed is None:
            seed = int(time.time())
        seed = np.int32(np.int64(seed) & 0x7FFFFFFF)
        if self._pool_rng is None:
            self._pool_rng = drv.mem_alloc(4 * (4 + self._num_envs))
        init = self._function_manager.get_function("init_random")
        init(self._pool_rng, seed, np.int32(self._num_envs), block=(256, 1, 1),
             grid=(max(1, min(1024, (self._num_envs + 255) // 256)), 1))
        self._random_initialized = True

    def reset_when_done_from_pool(self, data_manager, force_reset):
        pools = data_manager.reset_target_to_pool
        if len(pools) == 0:
            return
        assert self._random_initialized, (
            "reset_when_done_from_pool() requires the random seed initialized first, please call init_reset_pool()")
        block, grid = self._geometry()
        items = list(pools.items())
        for i, (name, pool_name) in enumerate(items):
            p_shape = data_manager.get_shape(pool_name)
            assert p_shape[0] > 1, "reset function assumes the 0th dimension is n_pool"
            row = int(np.prod(p_shape[1:])) if len(p_shape) > 1 else 1
            # every array of one reset call draws the same pool row per replica; the epoch
            # advances with the last array only
            self.reset_from_pool(self._pool_rng, data_manager.device_data(name), data_manager.device_data(pool_name),
                                 data_manager.device_data("_done_"), np.int32(row), np.int32(p_shape[0]),
                                 np.int32(force_reset), data_manager.meta_info("n_envs"),
                                 np.int32(1 if i == len(items) - 1 else 0), block=block, grid=grid)

    def _undo_done_flag_and_reset_timestep(self, data_manager, force_reset):
        n = self._num_envs
        self.undo(data_manager.device_data("_done_"), data_manager.device_data("_timestep_"), np.int32(force_reset),
                  data_manager.meta_info("n_envs"), block=(256, 1, 1), grid=(max(1, min(4096, (n + 255) // 256)), 1))

    def __del__(self):
        for p in [getattr(self, "_table", None), getattr(self, "_pool_rng", None)] + list(
                getattr(self, "_retired_tables", [])):
            if p is not None:
                try:
                    p.free()
                except Exception:
                    pass


class HIPLogController(CUDALogController):
    """Replaces PyCUDALogController (pycuda_function_manager.py:399-483).

    Besides the arrays registered with `log_data_across_episode=True` (logged by the
    `log_one_step_*` kernels), any device array can be attached after the fact with
    `attach_states()`: its replica slice is then copied device-to-device into a `[T + 1, ...]` buffer
    every step -- what `Trainer.fetch_episode_states` uses instead of the reference's per-tick host
    pulls (trainer_base.py:689-792)."""

    def __init__(self, function_manager: HIPFunctionManager):
        super().__init__(function_manager)
        self._attached = {}  # name -> (DevicePtr of [T + 1, row...], row shape, dtype, row bytes)

    def attach_states(self, data_manager, names):
        T = int(data_manager.meta_info("episode_length"))
        for name in names:
            if name in data_manager.log_data_list or name in self._attached:
                continue
            assert data_manager.is_data_on_device(name), f"{name} is not a valid array name on the GPU!"
            shape = tuple(data_manager.get_shape(name))
            assert shape[0] == int(data_manager.meta_info("n_envs")), "log assumes the 0th dimension is n_envs"
            dtype = np.dtype(data_manager.get_dtype(name))
            row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * dtype.itemsize if len(shape) > 1 else dtype.itemsize
            buf = drv.mem_alloc(max(row_bytes * (T + 1), 8))
            self._attached[name] = (buf, shape[1:], dtype, row_bytes)

    def _log_attached(self, data_manager, step, env_id):
        for name, (buf, _, _, row_bytes) in self._attached.items():
            src = int(data_manager.device_data(name)) + int(env_id) * row_bytes
            drv.memcpy_dtod(int(buf) + int(step) * row_bytes, src, row_bytes)

    def fetch_log(self, data_manager, names=None, last_step=None, check_last_valid_step=True):
        registered = [n for n in (data_manager.log_data_list if names is None else names)
                      if n in data_manager.log_data_list]
        out = super().fetch_log(data_manager, registered, last_step, check_last_valid_step)
        upto = last_step if (last_step is not None and last_step <= self.last_valid_step) else self.last_valid_step
        T = int(data_manager.meta_info("episode_length"))
        for name in (self._attached if names is None else [n for n in names if n in self._attached]):
            buf, row_shape, dtype, row_bytes = self._attached[name]
            host = np.empty((T + 1, *row_shape), dtype=dtype)
            drv.synchronize()
            if host.nbytes:
                drv.memcpy_dtoh(host, buf)
            out[f"{name}_for_log"] = host[: upto + 1]
        return out

    def __del__(self):
        for buf, *_ in getattr(self, "_attached", {}).values():
            try:
                buf.free()
            except Exception:
                pass

    def _log_one_step(self, data_manager, step: int, env_id: int = 0):
        assert env_id < data_manager.meta_info("n_envs")
        fm = self._function_manager
        for name in data_manager.log_data_list:
            shape = data_manager.get_shape(name)
            assert shape[0] == data_manager.meta_info("n_envs"), "log function assumes the 0th dimension is n_envs"
            assert shape[1] == data_manager.meta_info("n_agents"), "log function assumes the 1st dimension is n_agents"
            feature_dim = int(np.prod(shape[2:])) if len(shape) >= 3 else 1
            dtype = data_manager.get_dtype(name)
            if "float" in dtype:
                fn = fm.get_function("log_one_step_in_float")
            elif "int" in dtype:
                fn = fm.get_function("log_one_step_in_int")
            else:
                raise Exception(f"unknown dtype: {dtype}")
            row = int(shape[1]) * feature_dim
            fn(data_manager.device_data(f"{name}_for_log"), data_manager.device_data(name), np.int32(feature_dim),
               np.int32(step), data_manager.meta_info("episode_length"), np.int32(env_id),
               data_manager.meta_info("n_agents"), block=(256, 1, 1), grid=(max(1, min(64, (row + 255) // 256)), 1))
        self._log_attached(data_manager, step, env_id)

    def _update_log_mask(self, data_manager, step: int):
        fn = self._function_manager.get_function("update_log_mask")
        fn(data_manager.device_data("_log_mask_"), np.int32(step), data_manager.meta_info("episode_length"),
           block=(64, 1, 1), grid=(1, 1))
        self.last_valid_step = int(step)

    def _reset_log_mask(self, data_manager):
        fn = self._function_manager.get_function("reset_log_mask")
        fn(data_manager.device_data("_log_mask_"), data_manager.meta_info("episode_length"),
           block=(256, 1, 1), grid=(1, 1))
