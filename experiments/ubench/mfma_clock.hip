// Issue rate of v_mfma_f32_32x32x2_f32 and the shader clock while it runs (one wavefront per SIMD,
// 4 independent accumulators), for 64 / 256 / 1024 resident blocks: shader cycles (s_memtime) against
// the 100 MHz real-time counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v16 __attribute__((ext_vector_type(16)));

// MODE 0: A and B operands in VGPRs (compiler's choice); 1: B operand in an AGPR, accumulators in AGPRs
// (the policy kernel's layers 2 and 3: the previous layer's accumulators are the B operands); 2: A and
// B in VGPRs, accumulators in AGPRs
template <int MODE>
__global__ void __launch_bounds__(256, 1) mfma_loop(unsigned long long *out, float *sink, float seed, int reps) {
  v16 a0, a1, a2, a3;
  for (int i = 0; i < 16; ++i) { a0[i] = seed + i; a1[i] = seed - i; a2[i] = seed * i; a3[i] = seed; }
  const float x = seed + threadIdx.x, y = seed * 0.5f + threadIdx.x;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      } else if (MODE == 1) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n"
                     "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
                     : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(x), "a"(y));
      } else {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n"
                     "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3"
                     : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(x), "v"(y));
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    out[2 * w] = c1 - c0;
    out[2 * w + 1] = r1 - r0;
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  if (s == 123.456f) sink[0] = s;
}

int main() {
  unsigned long long *dout; float *dsink;
  CHECK(hipMalloc(&dout, 16 * 4096 * 4)); CHECK(hipMalloc(&dsink, 64));
  const int reps = 52;  // 52 * 32 = 1664 MFMAs: one block of the policy kernel
  for (int mode = 0; mode < 3; ++mode)
  for (int blocks : {64, 256}) {
    for (int it = 0; it < 3; ++it) {
      if (mode == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, dout, dsink, 1.5f, reps);
      if (mode == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, dout, dsink, 1.5f, reps);
      if (mode == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, dout, dsink, 1.5f, reps);
    }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 8);
    CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, us;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); us.push_back(h[2 * w + 1] / 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
    const double c = cyc[cyc.size() / 2], t = us[us.size() / 2];
    printf("mode %d, %5d blocks: %.1f shader cycles per MFMA, %.1f us per 1664 MFMAs, shader clock %.2f GHz\n", mode, blocks,
           c / (reps * 32.0), t, c / t / 1000.0);
  }
  return 0;
}
