/*
 * wd_hip.h -- C-ABI of libwdhip.so, the MI355X-native drop-in boundary for the
 * rollout hot path of salesforce/warp-drive.
 *
 * What it replaces.  The reference's managers drive the GPU through PyCUDA's
 * driver-API objects (or Numba's); each entry point below is the C equivalent a
 * binding (ctypes / cgo / JNI / N-API) would target instead.  Reference file:line
 * of the call each one replaces is given per function (paths relative to
 * /root/reference/warp_drive/managers/pycuda_managers/).
 *
 * Conventions
 *   - extern "C", plain pointers / sizes / 32-bit scalars, no C++ or torch types.
 *   - every function returns 0 on success, otherwise the hipError_t value as int
 *     (or WD_ERR_* below); it never throws.  wd_last_error() gives a thread-local
 *     human-readable message for the most recent failure.
 *   - device pointers are `void*` holding a device virtual address; `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream).  Callers
 *     that share memory with PyTorch pass torch.cuda.current_stream().cuda_stream
 *     so launches are ordered with torch's work (reference: launches go to the
 *     stream-0 PyCUDA shares with torch, pycuda_function_manager.py:563-572).
 *   - the library never frees caller memory; blocks from wd_malloc are freed by
 *     wd_free only.
 *   - the HIP runtime is bound at wd_init() time with dlopen/dlsym: if a
 *     libamdhip64 is already loaded in the process (e.g. PyTorch-ROCm's bundled
 *     copy) that instance is used, so device memory and streams are shared with it;
 *     otherwise libamdhip64.so is loaded from the loader path.  One caller thread
 *     per device is assumed (same as the reference: one process per GPU).
 */
#ifndef WD_HIP_H_
#define WD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WD_ERR_NO_RUNTIME 100001  /* libamdhip64 could not be found / a symbol is missing */
#define WD_ERR_BAD_ARG    100002

/* ---- runtime / device ---------------------------------------------------- */
/* replaces warp_drive/utils/device_context.py:5-13 + autoinit_pycuda (primary-context
 * sharing with torch): binds the HIP runtime and hipSetDevice(device). */
int wd_init(int device);
/* like wd_init but with an explicit path to libamdhip64 (NULL = auto). */
int wd_init_with_runtime(int device, const char *libamdhip64_path);
int wd_device_count(int *count);
/* name must hold >= 256 bytes; gcn_arch e.g. "gfx950:sramecc+:xnack-" */
int wd_device_info(int device, char *name, char *gcn_arch, int *compute_units,
                   size_t *total_mem_bytes);
const char *wd_last_error(void);
const char *wd_version(void);

/* ---- memory (pycuda_data_manager.py:79-126: mem_alloc, memcpy_htod, memcpy_dtoh) */
int wd_malloc(size_t bytes, void **dptr);
int wd_free(void *dptr);
int wd_memcpy_htod(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int wd_memcpy_dtoh(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int wd_memcpy_dtod(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int wd_memset(void *dst_dev, int byte_value, size_t bytes, void *stream);

/* ---- code objects (pycuda_function_manager.py:115-131 module_from_file,
 *      :340-361 get_function, :363-379 get_global + memcpy_htod) ---------------- */
int wd_module_load(const char *hsaco_path, void **module);
int wd_module_load_data(const void *image, void **module);
int wd_module_unload(void *module);
int wd_get_function(void *module, const char *name, void **function);
int wd_get_global(void *module, const char *name, void **dptr, size_t *bytes);
/* which: 0 = VGPRs, 1 = static LDS bytes, 2 = max threads per block, 3 = SGPRs is n/a(-1),
 *        4 = scratch (local) bytes, 5 = const bytes */
int wd_function_attribute(void *function, int which, int *value);

/* ---- launch (pycuda Function.__call__(*args, block=, grid=),
 *      e.g. pycuda_function_manager.py:563-572, :709-734, tag_continuous.py:842-847) */
/* kernel_params: array of n pointers, each to the value of one kernel argument
 * (device pointers are passed by the address of the void* holding them). */
int wd_launch(void *function, uint32_t grid_x, uint32_t grid_y, uint32_t grid_z,
              uint32_t block_x, uint32_t block_y, uint32_t block_z, uint32_t shared_mem_bytes,
              void *stream, void **kernel_params);
/* same, with the arguments already packed in kernel-ABI layout (natural alignment). */
int wd_launch_packed(void *function, uint32_t grid_x, uint32_t grid_y, uint32_t grid_z,
                     uint32_t block_x, uint32_t block_y, uint32_t block_z,
                     uint32_t shared_mem_bytes, void *stream, const void *arg_buffer,
                     size_t arg_bytes);
int wd_sync(void *stream);        /* torch.cuda.synchronize() / Context.synchronize() */
int wd_device_sync(void);

/* ---- launch plans: a fixed sequence of packed launches replayed from C, so a
 * rollout tick (sample heads -> step -> reset) costs no host-language work per
 * launch (the reference pays a Python->driver round trip per kernel,
 * trainer_base.py:392-426). */
int wd_plan_create(void **plan);
int wd_plan_add(void *plan, void *function, uint32_t grid_x, uint32_t grid_y, uint32_t grid_z,
                uint32_t block_x, uint32_t block_y, uint32_t block_z, uint32_t shared_mem_bytes,
                const void *arg_buffer, size_t arg_bytes);
int wd_plan_size(void *plan, int *n_launches);
/* enqueue the whole plan `repeats` times on `stream` (plain launches) */
int wd_plan_run(void *plan, int repeats, void *stream);
/* capture `repeats_per_graph` repetitions into a hipGraph once, then replay it */
int wd_plan_instantiate_graph(void *plan, int repeats_per_graph, void *stream);
int wd_plan_run_graph(void *plan, int graph_launches, void *stream);
/* sample the device duration of entry `entry_index` with HIP events recorded on the
 * launch stream around that launch, every `sample_stride`-th repetition of wd_plan_run
 * (at most max_samples pairs are kept; -1 disables).  A plan that consists of one launch is
 * bracketed over min(sample_stride, 8) consecutive repetitions per event pair, so the event
 * cost is amortised and the result is the average launch duration of back-to-back launches.
 * wd_plan_read_timing synchronises the recorded events and returns the summed milliseconds
 * and the number of launches they cover. */
int wd_plan_enable_timing(void *plan, int entry_index, int sample_stride, int max_samples);
int wd_plan_read_timing(void *plan, float *total_ms, int *n_samples);
int wd_plan_destroy(void *plan);

/* ---- events on the launch stream (bench.py's roofline leg; PerfStats in
 *      training/trainers/trainer_base.py:388-426 uses CUDA events the same way) */
int wd_event_create(void **event);
int wd_event_record(void *event, void *stream);
int wd_event_synchronize(void *event);
int wd_event_elapsed_ms(void *start, void *stop, float *ms);
int wd_event_destroy(void *event);

#ifdef __cplusplus
}
#endif
#endif /* WD_HIP_H_ */
