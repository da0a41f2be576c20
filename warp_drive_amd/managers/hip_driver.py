"""ctypes binding of libwdhip.so (C-ABI in include/wd_hip.h).

This is the whole FFI surface of the package: the managers above it never touch
hipModule* directly.  It plays the role `pycuda.driver` plays for the reference
(warp_drive/managers/pycuda_managers/pycuda_data_manager.py:13,
pycuda_function_manager.py:17-20).

The product path has NO CPU fallback: if the library or the code object is missing,
or no GPU is visible, the calls raise HipDriverError.
"""
import collections
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.path.join(_CSRC, "libwdhip.so")
# Code objects (warp_drive_amd/build.py UNITS): the MAIN one -- core services, TagGridWorld, Cartpole -- is loaded by
# every function manager; the others (TagContinuous generic / per-K / shape-specialised, the trainer's policy kernels,
# the 5-agent TagGridWorld rollout, the test kernels) are loaded when one of their kernels is first asked for.  Which
# object holds a kernel is read from the manifest the build writes.  WD_HSACO_DIR (experiments only): a directory that
# is searched first, so that a variant of ONE object can be measured against the others unchanged.
_OVERRIDE_DIR = os.environ.get("WD_HSACO_DIR", "")
MANIFEST_PATH = os.path.join(_CSRC, "wd_kernels.manifest.json")


def code_object_path(basename):
    if _OVERRIDE_DIR and os.path.exists(os.path.join(_OVERRIDE_DIR, basename)):
        return os.path.join(_OVERRIDE_DIR, basename)
    return os.path.join(_CSRC, basename)


HSACO_PATH = code_object_path("wd_kernels.hsaco")
_manifest = None


def manifest():
    """kernel name -> code object file name (csrc/wd_kernels.manifest.json)"""
    global _manifest
    if _manifest is None:
        import json

        if not os.path.exists(MANIFEST_PATH):
            raise HipDriverError(f"{MANIFEST_PATH} is missing: build the kernels with warp_drive_amd.build")
        _manifest = json.load(open(MANIFEST_PATH))
    return _manifest


def reload_manifest():
    """forget the cached manifest (a shape-specialised code object was just built: warp_drive_amd.build.build_shape_unit)"""
    global _manifest
    _manifest = None


def code_object_of(kernel_name):
    """path of the code object that holds `kernel_name`, or None when no object of the build has it -- the manifest is
    only a map: an entry whose file is not on disk (a partial build, `build_kernels(only=...)`, a failed unit) counts
    as "not built", so that capability queries answer False instead of failing later at module load"""
    name = manifest().get(kernel_name)
    if name is None:
        return None
    path = code_object_path(name)
    return path if os.path.exists(path) else None


def code_object_sha256(kernel_name):
    """sha256 of the code object `kernel_name` is loaded from (the main object when the manifest does not know it):
    what profiles/pmc_*.json records are keyed to"""
    import hashlib

    return hashlib.sha256(open(code_object_of(kernel_name) or HSACO_PATH, "rb").read()).hexdigest()


def extra_code_objects():
    """every code object of the build except the main one (sorted paths)"""
    return sorted({code_object_path(n) for n in manifest().values()} - {HSACO_PATH})


# every symbol include/wd_hip.h declares (tests assert the library exports them all)
C_ABI_SYMBOLS = (
    "wd_init", "wd_init_with_runtime", "wd_device_count", "wd_device_info", "wd_last_error",
    "wd_version", "wd_malloc", "wd_free", "wd_memcpy_htod", "wd_memcpy_dtoh", "wd_memcpy_dtod",
    "wd_memset", "wd_module_load", "wd_module_load_data", "wd_module_unload", "wd_get_function",
    "wd_get_global", "wd_function_attribute", "wd_launch", "wd_launch_packed", "wd_sync",
    "wd_device_sync", "wd_plan_create", "wd_plan_add", "wd_plan_size", "wd_plan_run",
    "wd_plan_instantiate_graph", "wd_plan_run_graph", "wd_plan_enable_timing", "wd_plan_read_timing",
    "wd_plan_destroy", "wd_event_create",
    "wd_event_record", "wd_event_synchronize", "wd_event_elapsed_ms", "wd_event_destroy",
)


class HipDriverError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()
_initialised_device = None

_vp, _sz, _u32, _i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int


def load_library(path=None):
    """dlopen libwdhip.so and declare prototypes (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise HipDriverError(
                f"{path} is missing: run `python -m warp_drive_amd.build` (or "
                f"__graft_entry__.build()) first; there is no CPU fallback for the HIP backend"
            )
        lib = ctypes.CDLL(path)
        P = ctypes.POINTER
        protos = {
            "wd_init": ([_i32], _i32),
            "wd_init_with_runtime": ([_i32, ctypes.c_char_p], _i32),
            "wd_device_count": ([P(_i32)], _i32),
            "wd_device_info": ([_i32, ctypes.c_char_p, ctypes.c_char_p, P(_i32), P(_sz)], _i32),
            "wd_last_error": ([], ctypes.c_char_p),
            "wd_version": ([], ctypes.c_char_p),
            "wd_malloc": ([_sz, P(_vp)], _i32),
            "wd_free": ([_vp], _i32),
            "wd_memcpy_htod": ([_vp, _vp, _sz, _vp], _i32),
            "wd_memcpy_dtoh": ([_vp, _vp, _sz, _vp], _i32),
            "wd_memcpy_dtod": ([_vp, _vp, _sz, _vp], _i32),
            "wd_memset": ([_vp, _i32, _sz, _vp], _i32),
            "wd_module_load": ([ctypes.c_char_p, P(_vp)], _i32),
            "wd_module_load_data": ([_vp, P(_vp)], _i32),
            "wd_module_unload": ([_vp], _i32),
            "wd_get_function": ([_vp, ctypes.c_char_p, P(_vp)], _i32),
            "wd_get_global": ([_vp, ctypes.c_char_p, P(_vp), P(_sz)], _i32),
            "wd_function_attribute": ([_vp, _i32, P(_i32)], _i32),
            "wd_launch": ([_vp] + [_u32] * 7 + [_vp, P(_vp)], _i32),
            "wd_launch_packed": ([_vp] + [_u32] * 7 + [_vp, _vp, _sz], _i32),
            "wd_sync": ([_vp], _i32),
            "wd_device_sync": ([], _i32),
            "wd_plan_create": ([P(_vp)], _i32),
            "wd_plan_add": ([_vp, _vp] + [_u32] * 7 + [_vp, _sz], _i32),
            "wd_plan_size": ([_vp, P(_i32)], _i32),
            "wd_plan_run": ([_vp, _i32, _vp], _i32),
            "wd_plan_instantiate_graph": ([_vp, _i32, _vp], _i32),
            "wd_plan_run_graph": ([_vp, _i32, _vp], _i32),
            "wd_plan_enable_timing": ([_vp, _i32, _i32, _i32], _i32),
            "wd_plan_read_timing": ([_vp, P(ctypes.c_float), P(_i32)], _i32),
            "wd_plan_destroy": ([_vp], _i32),
            "wd_event_create": ([P(_vp)], _i32),
            "wd_event_record": ([_vp, _vp], _i32),
            "wd_event_synchronize": ([_vp], _i32),
            "wd_event_elapsed_ms": ([_vp, _vp, P(ctypes.c_float)], _i32),
            "wd_event_destroy": ([_vp], _i32),
        }
        for name, (argtypes, restype) in protos.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
        return _lib


def _check(rc, what):
    if rc != 0:
        msg = _lib.wd_last_error().decode(errors="replace") if _lib is not None else ""
        raise HipDriverError(f"{what} failed (code {rc}): {msg}")


def _torch_hip_runtime_path():
    """Path of the libamdhip64 PyTorch-ROCm already mapped into this process, if any."""
    try:
        import torch  # noqa: F401  (importing torch maps its bundled HIP runtime)

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        return cand if os.path.exists(cand) else None
    except Exception:  # torch absent: the C library falls back to the loader path
        return None


def init(device=0):
    """Bind the HIP runtime (sharing PyTorch's instance when present) and select `device`.

    Replaces warp_drive/utils/device_context.py:5-13 and autoinit_pycuda.py: HIP's
    runtime API shares one primary context per device, so no context juggling remains.
    """
    global _initialised_device
    lib = load_library()
    rt = _torch_hip_runtime_path()
    rc = lib.wd_init_with_runtime(int(device), rt.encode() if rt else None)
    _check(rc, f"wd_init(device={device})")
    _initialised_device = int(device)
    return _initialised_device


def ensure_init(device=0):
    if _initialised_device is None or _initialised_device != int(device):
        init(device)


def device_count():
    lib = load_library()
    n = _i32(0)
    _check(lib.wd_device_count(ctypes.byref(n)), "wd_device_count")
    return n.value


def device_info(device=0):
    lib = load_library()
    name = ctypes.create_string_buffer(256)
    arch = ctypes.create_string_buffer(256)
    cus, mem = _i32(0), _sz(0)
    _check(lib.wd_device_info(int(device), name, arch, ctypes.byref(cus), ctypes.byref(mem)),
           "wd_device_info")
    return {"name": name.value.decode(), "gcn_arch": arch.value.decode(),
            "compute_units": cus.value, "total_mem_bytes": mem.value}


def current_stream():
    """The hipStream_t torch is currently issuing work on (0 = legacy default stream)."""
    try:
        import torch

        if torch.cuda.is_available():
            return int(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    return 0


# ------------------------------------------------------------------------ memory
class DevicePtr:
    """A device virtual address.  `owner` keeps whatever backs it alive (a torch tensor, or
    None for wd_malloc blocks which are freed explicitly).  Mirrors the role of
    pycuda DeviceAllocation / CudaTensorHolder (pycuda_data_manager.py:19-26)."""

    __slots__ = ("ptr", "nbytes", "owner", "_owned")

    def __init__(self, ptr, nbytes=0, owner=None, owned=False):
        self.ptr = int(ptr)
        self.nbytes = int(nbytes)
        self.owner = owner
        self._owned = owned

    def __int__(self):
        return self.ptr

    def free(self):
        if self._owned and self.ptr and _lib is not None:
            _lib.wd_free(_vp(self.ptr))
        self.ptr = 0
        self._owned = False

    def __repr__(self):
        return f"DevicePtr(0x{self.ptr:x}, {self.nbytes} B)"


def mem_alloc(nbytes):
    lib = load_library()
    p = _vp(0)
    _check(lib.wd_malloc(int(nbytes), ctypes.byref(p)), f"wd_malloc({nbytes})")
    return DevicePtr(p.value or 0, nbytes, owned=True)


def memcpy_htod(dst, host_array, stream=None):
    a = np.ascontiguousarray(host_array)
    _check(_lib.wd_memcpy_htod(_vp(int(dst)), _vp(a.ctypes.data), a.nbytes,
                               _vp(current_stream() if stream is None else stream)), "wd_memcpy_htod")


def memcpy_dtoh(host_array, src, stream=None):
    assert host_array.flags.c_contiguous
    _check(_lib.wd_memcpy_dtoh(_vp(host_array.ctypes.data), _vp(int(src)), host_array.nbytes,
                               _vp(current_stream() if stream is None else stream)), "wd_memcpy_dtoh")


def memcpy_dtod(dst, src, nbytes, stream=None):
    """device -> device, asynchronous on the stream torch is using"""
    _check(_lib.wd_memcpy_dtod(_vp(int(dst)), _vp(int(src)), int(nbytes),
                               _vp(current_stream() if stream is None else stream)), "wd_memcpy_dtod")


def memset(dst, byte_value, nbytes, stream=None):
    _check(_lib.wd_memset(_vp(int(dst)), int(byte_value), int(nbytes),
                          _vp(current_stream() if stream is None else stream)), "wd_memset")


def synchronize(stream=None):
    _check(_lib.wd_sync(_vp(current_stream() if stream is None else stream)), "wd_sync")


def device_synchronize():
    _check(_lib.wd_device_sync(), "wd_device_sync")


# ------------------------------------------------------------------- code objects
class Module:
    def __init__(self, path):
        lib = load_library()
        if not os.path.exists(path):
            raise HipDriverError(f"code object {path} is missing: build it with warp_drive_amd.build")
        h = _vp(0)
        _check(lib.wd_module_load(path.encode(), ctypes.byref(h)), f"wd_module_load({path})")
        self.handle = h.value
        self.path = path

    def get_function(self, name):
        f = _vp(0)
        _check(_lib.wd_get_function(_vp(self.handle), name.encode(), ctypes.byref(f)),
               f"wd_get_function({name})")
        return Function(f.value, name)

    def has_function(self, name):
        f = _vp(0)
        return _lib.wd_get_function(_vp(self.handle), name.encode(), ctypes.byref(f)) == 0

    def get_global(self, name):
        p, n = _vp(0), _sz(0)
        _check(_lib.wd_get_global(_vp(self.handle), name.encode(), ctypes.byref(p), ctypes.byref(n)),
               f"wd_get_global({name})")
        return DevicePtr(p.value, n.value), n.value


def _pack_args(args):
    """Pack kernel arguments in the amdgpu kernarg layout (natural alignment).

    Accepted: DevicePtr / torch.Tensor / objects with .data_ptr() (8-byte pointer),
    numpy scalars (by their own width), Python bool/int (int32) and float (float32) --
    the reference's 32-bit convention (managers/data_manager.py:263-269, :348-351)."""
    buf = bytearray()
    for a in args:
        if isinstance(a, DevicePtr):
            raw, size = np.uint64(a.ptr).tobytes(), 8
        elif hasattr(a, "data_ptr"):
            raw, size = np.uint64(a.data_ptr()).tobytes(), 8
        elif isinstance(a, np.generic):
            raw, size = a.tobytes(), a.dtype.itemsize
        elif isinstance(a, (bool, int)):
            raw, size = np.int32(a).tobytes(), 4
        elif isinstance(a, float):
            raw, size = np.float32(a).tobytes(), 4
        else:
            raise HipDriverError(f"cannot pass {type(a)} to a kernel")
        pad = (-len(buf)) % size
        buf.extend(b"\0" * pad)
        buf.extend(raw)
    return bytes(buf)


def _dim3(v, n=3):
    v = tuple(int(x) for x in (v if isinstance(v, (tuple, list)) else (v,)))
    return v + (1,) * (n - len(v))


# launches per kernel name through Function.__call__ (plans / graphs replay from C and are not counted): lets a test assert
# WHICH kernels a composed path actually ran instead of trusting a dispatch predicate
LAUNCH_COUNTS = collections.Counter()


class Function:
    """A kernel.  Callable both ways the reference's env classes use
    (example_envs/tag_continuous/tag_continuous.py:842-851):
        fn(*args, block=(bx,1,1), grid=(gx,1))            # PyCUDA style
        fn[grid, block](*args)                            # Numba style
    `shared` = dynamic LDS bytes."""

    def __init__(self, handle, name):
        self.handle = handle
        self.name = name

    def __call__(self, *args, block=(1, 1, 1), grid=(1, 1), shared=0, stream=None):
        packed = _pack_args(args)
        g, b = _dim3(grid), _dim3(block)
        rc = _lib.wd_launch_packed(_vp(self.handle), g[0], g[1], g[2], b[0], b[1], b[2], int(shared),
                                   _vp(current_stream() if stream is None else stream), packed, len(packed))
        _check(rc, f"launch {self.name}")
        LAUNCH_COUNTS[self.name] += 1

    def __getitem__(self, cfg):
        grid, block = cfg[0], cfg[1]
        shared = cfg[2] if len(cfg) > 2 else 0

        def _launch(*args):
            self(*args, block=block, grid=grid, shared=shared)

        return _launch

    def attribute(self, which):
        v = _i32(0)
        _check(_lib.wd_function_attribute(_vp(self.handle), int(which), ctypes.byref(v)),
               "wd_function_attribute")
        return v.value


class LaunchPlan:
    """A fixed sequence of launches replayed from C (optionally as a hipGraph)."""

    def __init__(self):
        load_library()
        h = _vp(0)
        _check(_lib.wd_plan_create(ctypes.byref(h)), "wd_plan_create")
        self.handle = h.value
        self._keep = []

    def add(self, fn, args, block, grid, shared=0):
        packed = _pack_args(args)
        self._keep.append(args)  # keep tensors / DevicePtrs alive
        g, b = _dim3(grid), _dim3(block)
        _check(_lib.wd_plan_add(_vp(self.handle), _vp(fn.handle), g[0], g[1], g[2], b[0], b[1], b[2],
                                int(shared), packed, len(packed)), "wd_plan_add")

    def __len__(self):
        n = _i32(0)
        _check(_lib.wd_plan_size(_vp(self.handle), ctypes.byref(n)), "wd_plan_size")
        return n.value

    def run(self, repeats=1, stream=None):
        _check(_lib.wd_plan_run(_vp(self.handle), int(repeats),
                                _vp(current_stream() if stream is None else stream)), "wd_plan_run")

    def instantiate_graph(self, repeats_per_graph=1, stream=None):
        _check(_lib.wd_plan_instantiate_graph(_vp(self.handle), int(repeats_per_graph),
                                              _vp(current_stream() if stream is None else stream)),
               "wd_plan_instantiate_graph")

    def run_graph(self, launches=1, stream=None):
        _check(_lib.wd_plan_run_graph(_vp(self.handle), int(launches),
                                      _vp(current_stream() if stream is None else stream)), "wd_plan_run_graph")

    def enable_timing(self, entry_index, sample_stride=1, max_samples=64):
        _check(_lib.wd_plan_enable_timing(_vp(self.handle), int(entry_index), int(sample_stride),
                                          int(max_samples)), "wd_plan_enable_timing")

    def read_timing(self):
        """-> (total milliseconds, samples) of the timed entry since the last read"""
        ms, n = ctypes.c_float(0), _i32(0)
        _check(_lib.wd_plan_read_timing(_vp(self.handle), ctypes.byref(ms), ctypes.byref(n)),
               "wd_plan_read_timing")
        return ms.value, n.value

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.wd_plan_destroy(_vp(self.handle))
        except Exception:
            pass
        self.handle = None


class Event:
    def __init__(self):
        load_library()
        h = _vp(0)
        _check(_lib.wd_event_create(ctypes.byref(h)), "wd_event_create")
        self.handle = h.value

    def record(self, stream=None):
        _check(_lib.wd_event_record(_vp(self.handle), _vp(current_stream() if stream is None else stream)),
               "wd_event_record")

    def synchronize(self):
        _check(_lib.wd_event_synchronize(_vp(self.handle)), "wd_event_synchronize")

    def elapsed_ms(self, end):
        ms = ctypes.c_float(0)
        _check(_lib.wd_event_elapsed_ms(_vp(self.handle), _vp(end.handle), ctypes.byref(ms)),
               "wd_event_elapsed_ms")
        return ms.value
