// wd_runtime.cpp -- libwdhip.so: the C-ABI declared in include/wd_hip.h.
//
// Thin, allocation-free-on-the-hot-path wrapper over the HIP module API
// (hipModuleLoad / hipModuleGetFunction / hipModuleGetGlobal /
// hipModuleLaunchKernel).  It replaces the PyCUDA driver-API objects the
// reference's managers use (warp_drive/managers/pycuda_managers/*.py).
//
// The HIP runtime is bound with dlopen/dlsym instead of being linked: PyTorch-ROCm
// ships its own libamdhip64 (+ libhsa-runtime64) inside torch/lib, and a process
// must talk to exactly ONE of them for device pointers and streams to be shared
// with torch tensors (zero-copy obs/reward/done).  wd_init() therefore prefers an
// already-loaded libamdhip64 (RTLD_NOLOAD) and only loads one itself otherwise.
//
// Built with:  g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//              wd_runtime.cpp -o libwdhip.so -ldl       (no -lamdhip64 on purpose)

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "wd_hip.h"

namespace {

thread_local char g_err[1024] = "";

void set_err(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- dynamically bound HIP entry points ---------------------------------------
// Explicit signatures (several runtime functions are overloaded templates in the
// C++ header, so decltype(&name) is ambiguous).
#define WD_HIP_FUNCS(X)                                                                        \
  X(hipInit, hipError_t (*)(unsigned int))                                                     \
  X(hipSetDevice, hipError_t (*)(int))                                                         \
  X(hipGetDeviceCount, hipError_t (*)(int *))                                                  \
  X(hipGetDevicePropertiesR0600, hipError_t (*)(hipDeviceProp_t *, int))                       \
  X(hipGetErrorString, const char *(*)(hipError_t))                                            \
  X(hipGetLastError, hipError_t (*)(void))                                                     \
  X(hipMalloc, hipError_t (*)(void **, size_t))                                                \
  X(hipFree, hipError_t (*)(void *))                                                           \
  X(hipMemcpyAsync, hipError_t (*)(void *, const void *, size_t, hipMemcpyKind, hipStream_t))  \
  X(hipMemsetAsync, hipError_t (*)(void *, int, size_t, hipStream_t))                          \
  X(hipModuleLoad, hipError_t (*)(hipModule_t *, const char *))                                \
  X(hipModuleLoadData, hipError_t (*)(hipModule_t *, const void *))                            \
  X(hipModuleUnload, hipError_t (*)(hipModule_t))                                              \
  X(hipModuleGetFunction, hipError_t (*)(hipFunction_t *, hipModule_t, const char *))          \
  X(hipModuleGetGlobal, hipError_t (*)(hipDeviceptr_t *, size_t *, hipModule_t, const char *)) \
  X(hipFuncGetAttribute, hipError_t (*)(int *, hipFunction_attribute, hipFunction_t))          \
  X(hipModuleLaunchKernel,                                                                     \
    hipError_t (*)(hipFunction_t, unsigned int, unsigned int, unsigned int, unsigned int,      \
                   unsigned int, unsigned int, unsigned int, hipStream_t, void **, void **))   \
  X(hipStreamSynchronize, hipError_t (*)(hipStream_t))                                         \
  X(hipDeviceSynchronize, hipError_t (*)(void))                                                \
  X(hipEventCreate, hipError_t (*)(hipEvent_t *))                                              \
  X(hipEventRecord, hipError_t (*)(hipEvent_t, hipStream_t))                                   \
  X(hipEventSynchronize, hipError_t (*)(hipEvent_t))                                           \
  X(hipEventElapsedTime, hipError_t (*)(float *, hipEvent_t, hipEvent_t))                      \
  X(hipEventDestroy, hipError_t (*)(hipEvent_t))                                               \
  X(hipStreamBeginCapture, hipError_t (*)(hipStream_t, hipStreamCaptureMode))                  \
  X(hipStreamEndCapture, hipError_t (*)(hipStream_t, hipGraph_t *))                            \
  X(hipGraphInstantiate,                                                                       \
    hipError_t (*)(hipGraphExec_t *, hipGraph_t, hipGraphNode_t *, char *, size_t))            \
  X(hipGraphLaunch, hipError_t (*)(hipGraphExec_t, hipStream_t))                               \
  X(hipGraphExecDestroy, hipError_t (*)(hipGraphExec_t))                                       \
  X(hipGraphDestroy, hipError_t (*)(hipGraph_t))

template <class T> using FnT = T;
struct HipApi {
#define X(name, sig) FnT<sig> name = nullptr;
  WD_HIP_FUNCS(X)
#undef X
  void *handle = nullptr;
  bool ready = false;
} g_hip;

int bind_runtime(const char *path) {
  if (g_hip.ready) return 0;
  void *h = nullptr;
  const char *cands[] = {path, "libamdhip64.so.7", "libamdhip64.so", "libamdhip64.so.6", nullptr};
  // 1) an instance already living in the process (PyTorch's bundled copy)
  for (int i = (path ? 0 : 1); cands[i] && !h; ++i) h = dlopen(cands[i], RTLD_NOW | RTLD_NOLOAD);
  // 2) otherwise load one
  for (int i = (path ? 0 : 1); cands[i] && !h; ++i) h = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    set_err("wd_init: cannot find libamdhip64 (%s)", dlerror());
    return WD_ERR_NO_RUNTIME;
  }
  g_hip.handle = h;
#define X(name, sig)                                                   \
  g_hip.name = reinterpret_cast<sig>(dlsym(h, #name));                 \
  if (!g_hip.name) {                                                   \
    set_err("wd_init: libamdhip64 lacks symbol %s", #name);            \
    return WD_ERR_NO_RUNTIME;                                          \
  }
  WD_HIP_FUNCS(X)
#undef X
  g_hip.ready = true;
  return 0;
}

inline int check(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  // the runtime keeps a sticky per-thread "last error" that other users of the same runtime
  // (PyTorch checks it after its own launches) would trip over: this failure is reported
  // through our return code, so consume it here.
  if (g_hip.hipGetLastError) (void)g_hip.hipGetLastError();
  set_err("%s failed: %s (%d)", what, g_hip.hipGetErrorString ? g_hip.hipGetErrorString(e) : "?",
          static_cast<int>(e));
  return static_cast<int>(e);
}

#define WD_REQUIRE_RT()                                        \
  do {                                                         \
    if (!g_hip.ready) {                                        \
      set_err("HIP runtime not bound: call wd_init() first");  \
      return WD_ERR_NO_RUNTIME;                                \
    }                                                          \
  } while (0)

struct PlanEntry {
  hipFunction_t fn;
  uint32_t g[3], b[3], shmem;
  std::vector<unsigned char> args;
};

struct Plan {
  std::vector<PlanEntry> entries;
  hipGraphExec_t exec = nullptr;
  int reps_in_graph = 0;
  // optional event sampling of one entry (bench.py's roofline leg)
  int timed_entry = -1, stride = 1, max_samples = 0, used = 0;
  int group = 1;      // launches bracketed by one event pair (> 1 only for single-launch plans)
  bool open = false;  // a start event was recorded, its end event not yet
  long run_counter = 0;
  std::vector<hipEvent_t> ev;  // 2 * max_samples
};

int launch_packed(hipFunction_t fn, const uint32_t g[3], const uint32_t b[3], uint32_t shmem,
                  hipStream_t stream, const void *buf, size_t bytes) {
  size_t sz = bytes;
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(buf),
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  return check(g_hip.hipModuleLaunchKernel(fn, g[0], g[1], g[2], b[0], b[1], b[2], shmem, stream,
                                           nullptr, config),
               "hipModuleLaunchKernel");
}

}  // namespace

extern "C" {

const char *wd_last_error(void) { return g_err; }
const char *wd_version(void) { return "wdhip 0.1 (gfx950)"; }

int wd_init_with_runtime(int device, const char *path) {
  if (int rc = bind_runtime(path)) return rc;
  if (int rc = check(g_hip.hipInit(0), "hipInit")) return rc;
  return check(g_hip.hipSetDevice(device), "hipSetDevice");
}
int wd_init(int device) { return wd_init_with_runtime(device, nullptr); }

int wd_device_count(int *count) {
  if (!count) return WD_ERR_BAD_ARG;
  if (int rc = bind_runtime(nullptr)) return rc;
  return check(g_hip.hipGetDeviceCount(count), "hipGetDeviceCount");
}

int wd_device_info(int device, char *name, char *gcn_arch, int *cus, size_t *mem) {
  WD_REQUIRE_RT();
  hipDeviceProp_t p;
  if (int rc = check(g_hip.hipGetDevicePropertiesR0600(&p, device), "hipGetDeviceProperties"))
    return rc;
  if (name) { strncpy(name, p.name, 255); name[255] = 0; }
  if (gcn_arch) { strncpy(gcn_arch, p.gcnArchName, 255); gcn_arch[255] = 0; }
  if (cus) *cus = p.multiProcessorCount;
  if (mem) *mem = p.totalGlobalMem;
  return 0;
}

// ---- memory ------------------------------------------------------------------
int wd_malloc(size_t bytes, void **dptr) {
  WD_REQUIRE_RT();
  if (!dptr) return WD_ERR_BAD_ARG;
  return check(g_hip.hipMalloc(dptr, bytes ? bytes : 4), "hipMalloc");
}
int wd_free(void *dptr) {
  WD_REQUIRE_RT();
  return check(g_hip.hipFree(dptr), "hipFree");
}
int wd_memcpy_htod(void *dst, const void *src, size_t bytes, void *stream) {
  WD_REQUIRE_RT();
  if (int rc = check(g_hip.hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                                          static_cast<hipStream_t>(stream)), "hipMemcpyAsync(HtoD)"))
    return rc;
  // pageable host memory: make the call synchronous like pycuda.memcpy_htod
  return check(g_hip.hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}
int wd_memcpy_dtoh(void *dst, const void *src, size_t bytes, void *stream) {
  WD_REQUIRE_RT();
  if (int rc = check(g_hip.hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                                          static_cast<hipStream_t>(stream)), "hipMemcpyAsync(DtoH)"))
    return rc;
  return check(g_hip.hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}
int wd_memcpy_dtod(void *dst, const void *src, size_t bytes, void *stream) {
  WD_REQUIRE_RT();
  return check(g_hip.hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice,
                                    static_cast<hipStream_t>(stream)), "hipMemcpyAsync(DtoD)");
}
int wd_memset(void *dst, int value, size_t bytes, void *stream) {
  WD_REQUIRE_RT();
  return check(g_hip.hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)),
               "hipMemsetAsync");
}

// ---- code objects --------------------------------------------------------------
int wd_module_load(const char *path, void **module) {
  WD_REQUIRE_RT();
  if (!path || !module) return WD_ERR_BAD_ARG;
  hipModule_t m = nullptr;
  if (int rc = check(g_hip.hipModuleLoad(&m, path), "hipModuleLoad")) {
    char tmp[512];
    snprintf(tmp, sizeof(tmp), "%s [%s]", g_err, path);
    set_err("%s", tmp);
    return rc;
  }
  *module = m;
  return 0;
}
int wd_module_load_data(const void *image, void **module) {
  WD_REQUIRE_RT();
  if (!image || !module) return WD_ERR_BAD_ARG;
  hipModule_t m = nullptr;
  if (int rc = check(g_hip.hipModuleLoadData(&m, image), "hipModuleLoadData")) return rc;
  *module = m;
  return 0;
}
int wd_module_unload(void *module) {
  WD_REQUIRE_RT();
  return check(g_hip.hipModuleUnload(static_cast<hipModule_t>(module)), "hipModuleUnload");
}
int wd_get_function(void *module, const char *name, void **function) {
  WD_REQUIRE_RT();
  if (!module || !name || !function) return WD_ERR_BAD_ARG;
  hipFunction_t f = nullptr;
  if (int rc = check(g_hip.hipModuleGetFunction(&f, static_cast<hipModule_t>(module), name),
                     "hipModuleGetFunction")) {
    char tmp[512];
    snprintf(tmp, sizeof(tmp), "%s [kernel '%s']", g_err, name);
    set_err("%s", tmp);
    return rc;
  }
  *function = f;
  return 0;
}
int wd_get_global(void *module, const char *name, void **dptr, size_t *bytes) {
  WD_REQUIRE_RT();
  if (!module || !name || !dptr) return WD_ERR_BAD_ARG;
  hipDeviceptr_t p = nullptr;
  size_t n = 0;
  if (int rc = check(g_hip.hipModuleGetGlobal(&p, &n, static_cast<hipModule_t>(module), name),
                     "hipModuleGetGlobal"))
    return rc;
  *dptr = p;
  if (bytes) *bytes = n;
  return 0;
}
int wd_function_attribute(void *function, int which, int *value) {
  WD_REQUIRE_RT();
  if (!function || !value) return WD_ERR_BAD_ARG;
  hipFunction_attribute a;
  switch (which) {
    case 0: a = HIP_FUNC_ATTRIBUTE_NUM_REGS; break;
    case 1: a = HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES; break;
    case 2: a = HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK; break;
    case 4: a = HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES; break;
    case 5: a = HIP_FUNC_ATTRIBUTE_CONST_SIZE_BYTES; break;
    default: *value = -1; return 0;
  }
  return check(g_hip.hipFuncGetAttribute(value, a, static_cast<hipFunction_t>(function)),
               "hipFuncGetAttribute");
}

// ---- launch -----------------------------------------------------------------
int wd_launch(void *function, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
              uint32_t bz, uint32_t shmem, void *stream, void **params) {
  WD_REQUIRE_RT();
  if (!function) return WD_ERR_BAD_ARG;
  return check(g_hip.hipModuleLaunchKernel(static_cast<hipFunction_t>(function), gx, gy, gz, bx, by,
                                           bz, shmem, static_cast<hipStream_t>(stream), params,
                                           nullptr),
               "hipModuleLaunchKernel");
}
int wd_launch_packed(void *function, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx,
                     uint32_t by, uint32_t bz, uint32_t shmem, void *stream, const void *buf,
                     size_t bytes) {
  WD_REQUIRE_RT();
  if (!function) return WD_ERR_BAD_ARG;
  const uint32_t g[3] = {gx, gy, gz}, b[3] = {bx, by, bz};
  return launch_packed(static_cast<hipFunction_t>(function), g, b, shmem,
                       static_cast<hipStream_t>(stream), buf, bytes);
}
int wd_sync(void *stream) {
  WD_REQUIRE_RT();
  return check(g_hip.hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
}
int wd_device_sync(void) {
  WD_REQUIRE_RT();
  return check(g_hip.hipDeviceSynchronize(), "hipDeviceSynchronize");
}

// ---- launch plans ---------------------------------------------------------------
int wd_plan_create(void **plan) {
  if (!plan) return WD_ERR_BAD_ARG;
  *plan = new Plan();
  return 0;
}
int wd_plan_add(void *plan, void *function, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx,
                uint32_t by, uint32_t bz, uint32_t shmem, const void *buf, size_t bytes) {
  if (!plan || !function) return WD_ERR_BAD_ARG;
  PlanEntry e;
  e.fn = static_cast<hipFunction_t>(function);
  e.g[0] = gx; e.g[1] = gy; e.g[2] = gz;
  e.b[0] = bx; e.b[1] = by; e.b[2] = bz;
  e.shmem = shmem;
  e.args.assign(static_cast<const unsigned char *>(buf),
                static_cast<const unsigned char *>(buf) + bytes);
  static_cast<Plan *>(plan)->entries.push_back(std::move(e));
  return 0;
}
int wd_plan_size(void *plan, int *n) {
  if (!plan || !n) return WD_ERR_BAD_ARG;
  *n = static_cast<int>(static_cast<Plan *>(plan)->entries.size());
  return 0;
}
int wd_plan_run(void *plan, int repeats, void *stream) {
  WD_REQUIRE_RT();
  if (!plan) return WD_ERR_BAD_ARG;
  Plan *p = static_cast<Plan *>(plan);
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int r = 0; r < repeats; ++r, ++p->run_counter) {
    // A plan of ONE launch is timed over `group` consecutive repetitions per event pair: back-to-back
    // launches leave no gap, so elapsed / group is the kernel's average launch duration as a profiler
    // reports it, without the cost of an event pair around every sample.
    const long phase = p->run_counter % p->stride;
    const bool sampling = p->timed_entry >= 0 && (p->open || p->used < p->max_samples);
    for (size_t i = 0; i < p->entries.size(); ++i) {
      auto &e = p->entries[i];
      const bool timed = sampling && static_cast<int>(i) == p->timed_entry;
      if (timed && phase == 0 && !p->open) {
        if (int rc = check(g_hip.hipEventRecord(p->ev[2 * p->used], s), "hipEventRecord")) return rc;
        p->open = true;
      }
      if (int rc = launch_packed(e.fn, e.g, e.b, e.shmem, s, e.args.data(), e.args.size())) return rc;
      if (timed && p->open && phase == p->group - 1) {
        if (int rc = check(g_hip.hipEventRecord(p->ev[2 * p->used + 1], s), "hipEventRecord")) return rc;
        p->open = false;
        ++p->used;
      }
    }
  }
  return 0;
}
int wd_plan_enable_timing(void *plan, int entry_index, int sample_stride, int max_samples) {
  WD_REQUIRE_RT();
  Plan *p = static_cast<Plan *>(plan);
  if (!p) return WD_ERR_BAD_ARG;
  for (auto e : p->ev) (void)g_hip.hipEventDestroy(e);
  p->ev.clear();
  p->used = 0;
  p->run_counter = 0;
  p->timed_entry = -1;
  p->open = false;
  p->group = 1;
  if (entry_index < 0) return 0;
  if (entry_index >= static_cast<int>(p->entries.size()) || sample_stride < 1 || max_samples < 1)
    return WD_ERR_BAD_ARG;
  for (int i = 0; i < 2 * max_samples; ++i) {
    hipEvent_t e = nullptr;
    if (int rc = check(g_hip.hipEventCreate(&e), "hipEventCreate")) return rc;
    p->ev.push_back(e);
  }
  p->timed_entry = entry_index;
  p->stride = sample_stride;
  p->max_samples = max_samples;
  p->group = p->entries.size() == 1 ? (sample_stride < 8 ? sample_stride : 8) : 1;
  return 0;
}
int wd_plan_read_timing(void *plan, float *total_ms, int *n_samples) {
  WD_REQUIRE_RT();
  Plan *p = static_cast<Plan *>(plan);
  if (!p || !total_ms || !n_samples) return WD_ERR_BAD_ARG;
  float total = 0.f;
  for (int i = 0; i < p->used; ++i) {
    if (int rc = check(g_hip.hipEventSynchronize(p->ev[2 * i + 1]), "hipEventSynchronize")) return rc;
    float ms = 0.f;
    if (int rc = check(g_hip.hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]),
                       "hipEventElapsedTime"))
      return rc;
    total += ms;
  }
  *total_ms = total;
  *n_samples = p->used * p->group;  // launches covered by the summed time
  p->used = 0;
  p->open = false;
  return 0;
}
int wd_plan_instantiate_graph(void *plan, int reps, void *stream) {
  WD_REQUIRE_RT();
  if (!plan || reps < 1) return WD_ERR_BAD_ARG;
  Plan *p = static_cast<Plan *>(plan);
  if (p->exec) { (void)g_hip.hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = check(g_hip.hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal),
                     "hipStreamBeginCapture"))
    return rc;
  int rc = wd_plan_run(plan, reps, stream);
  hipGraph_t graph = nullptr;
  int rc2 = check(g_hip.hipStreamEndCapture(s, &graph), "hipStreamEndCapture");
  if (rc) return rc;
  if (rc2) return rc2;
  rc = check(g_hip.hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
  (void)g_hip.hipGraphDestroy(graph);
  if (!rc) p->reps_in_graph = reps;
  return rc;
}
int wd_plan_run_graph(void *plan, int launches, void *stream) {
  WD_REQUIRE_RT();
  Plan *p = static_cast<Plan *>(plan);
  if (!p || !p->exec) { set_err("wd_plan_run_graph: graph not instantiated"); return WD_ERR_BAD_ARG; }
  for (int i = 0; i < launches; ++i)
    if (int rc = check(g_hip.hipGraphLaunch(p->exec, static_cast<hipStream_t>(stream)), "hipGraphLaunch"))
      return rc;
  return 0;
}
int wd_plan_destroy(void *plan) {
  Plan *p = static_cast<Plan *>(plan);
  if (!p) return 0;
  if (p->exec && g_hip.ready) (void)g_hip.hipGraphExecDestroy(p->exec);
  if (g_hip.ready)
    for (auto e : p->ev) (void)g_hip.hipEventDestroy(e);
  delete p;
  return 0;
}

// ---- events ----------------------------------------------------------------------
int wd_event_create(void **event) {
  WD_REQUIRE_RT();
  if (!event) return WD_ERR_BAD_ARG;
  hipEvent_t e = nullptr;
  if (int rc = check(g_hip.hipEventCreate(&e), "hipEventCreate")) return rc;
  *event = e;
  return 0;
}
int wd_event_record(void *event, void *stream) {
  WD_REQUIRE_RT();
  return check(g_hip.hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)),
               "hipEventRecord");
}
int wd_event_synchronize(void *event) {
  WD_REQUIRE_RT();
  return check(g_hip.hipEventSynchronize(static_cast<hipEvent_t>(event)), "hipEventSynchronize");
}
int wd_event_elapsed_ms(void *start, void *stop, float *ms) {
  WD_REQUIRE_RT();
  if (!ms) return WD_ERR_BAD_ARG;
  return check(g_hip.hipEventElapsedTime(ms, static_cast<hipEvent_t>(start),
                                         static_cast<hipEvent_t>(stop)),
               "hipEventElapsedTime");
}
int wd_event_destroy(void *event) {
  WD_REQUIRE_RT();
  return check(g_hip.hipEventDestroy(static_cast<hipEvent_t>(event)), "hipEventDestroy");
}

}  // extern "C"
