// wd_test_kernels.hip -- kernels that exist for the test suite and the measurement scripts only; their own code
// object (wd_kernels_test.hsaco), so that nothing of it ships inside the product objects.
#include "wd_common.h"

extern "C" {

// Write-bandwidth probe (scripts/write_pattern_probe.py): every block streams `floats_per_block`
// floats into its own contiguous slice of `out` (slice b starts at b * slice_stride floats), `vec`
// floats per lane per store (1 or 4).  Shows what the HBM write path gives to the rollout's access
// pattern -- thousands of blocks, each writing one replica's contiguous rows -- as opposed to a
// grid-stride fill.
__global__ void wd_write_probe(float *out, long slice_stride, int floats_per_block, int vec, float value) {
  float *dst = out + (long)blockIdx.x * slice_stride;
  if (vec == 4) {
    for (int i = 4 * threadIdx.x; i + 3 < floats_per_block; i += 4 * blockDim.x)
      *(float4 *)(dst + i) = make_float4(value, value, value, value);
  } else {
    for (int i = threadIdx.x; i < floats_per_block; i += blockDim.x) dst[i] = value;
  }
}

// ------------------------------------------------------------- math self-test hook
// Evaluates the device restatements of numpy's float32 routines so the GPU parity
// suite can compare them bit-for-bit with numpy on the host (tests/test_gpu_math.py).
__global__ void wd_test_math(const float *__restrict__ a, const float *__restrict__ b,
                             float *out_sin, float *out_cos, float *out_rem, float *out_sqrt,
                             float *out_div, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s, c;
    wd_np_sincosf(a[i], s, c);
    out_sin[i] = s;
    out_cos[i] = c;
    out_rem[i] = wd_np_remainderf(a[i], b[i]);
    out_sqrt[i] = sqrtf(a[i] * a[i] + b[i] * b[i]);
    out_div[i] = a[i] / b[i];
  }
}

}  // extern "C"
