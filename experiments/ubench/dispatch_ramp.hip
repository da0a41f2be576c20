// How long does the dispatcher take to start every workgroup of a one-round launch?  Each workgroup stamps the
// 100 MHz real-time counter at its first instruction; the spread between the first and the last start is the ramp
// the TagContinuous tick pays on every launch (2000 workgroups of 128 threads and 20 KB of LDS).
//   hipcc --offload-arch=gfx950 -O2 dispatch_ramp.hip -o dispatch_ramp && ./dispatch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void stamp(unsigned long long *out, int spin) {
  extern __shared__ int lds[];
  const unsigned long long t = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t;
  // stay resident so that the whole grid is one round (as the tick: nobody retires before the last one starts)
  unsigned long long t1 = t;
  while (t1 - t < (unsigned long long)spin) t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 1) lds[0] = (int)t1;
}
int main() {
  unsigned long long *d;
  hipMalloc(&d, 8 * 8192);
  struct { int blocks, threads, lds; } cfg[] = {{2000, 128, 20480}, {1000, 256, 40960}, {500, 512, 81920}, {2000, 128, 1024},
                                                 {2000, 64, 20480}, {4000, 64, 10240}, {667, 320, 53000}, {1024, 256, 0}};
  for (auto &c : cfg) {
    hipFuncSetAttribute((const void *)stamp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<double> ramps;
    for (int it = 0; it < 20; ++it) {
      hipLaunchKernelGGL(stamp, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, 400);  // 4 us resident
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(c.blocks);
      hipMemcpy(h.data(), d, 8 * c.blocks, hipMemcpyDeviceToHost);
      auto mm = std::minmax_element(h.begin(), h.end());
      ramps.push_back((*mm.second - *mm.first) / 100.0);
    }
    std::sort(ramps.begin(), ramps.end());
    printf("%5d workgroups x %3d threads, %6d B LDS: first-to-last start %.2f us (median of 20; min %.2f max %.2f)\n", c.blocks,
           c.threads, c.lds, ramps[10], ramps.front(), ramps.back());
  }
  return 0;
}
