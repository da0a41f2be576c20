// Instruction-rate microbenchmarks for gfx950 (what the TagContinuous kernel leans on).
// Each kernel runs REPS iterations of a 32-instruction unrolled body per wavefront and reports
// shader cycles (s_memtime) per wave-instruction, at 1, 2 and 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define REPS 512
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) bench(unsigned long long *out, float *sink, float seed) {
  __shared__ float2 lds[256];
  lds[threadIdx.x] = make_float2(seed + threadIdx.x, seed);
  __syncthreads();
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f, c = seed * 0.25f;
  unsigned m = 0;
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, c};
  int idx = threadIdx.x & 7;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {  // v_fma_f32, 8 independent chains
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      } else if (MODE == 1) {  // v_med3_f32 chain exactly as the search uses it (descending k)
        asm volatile("v_med3_f32 %7, %6, %7, %8\n v_med3_f32 %6, %5, %6, %8\n v_med3_f32 %5, %4, %5, %8\n v_med3_f32 %4, %3, %4, %8\n"
                     "v_med3_f32 %3, %2, %3, %8\n v_med3_f32 %2, %1, %2, %8\n v_med3_f32 %1, %0, %1, %8\n v_min_f32 %0, %0, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 2) {  // v_min_f32 / v_max_f32 (VOP2), 8 independent
        asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                     "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 3) {  // v_pk_add_f32 / v_pk_mul_f32, 4 independent pairs (8 instrs)
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     "v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
      } else if (MODE == 4) {  // cmp + addc chain through VCC (pass B), 4 pairs = 8 instrs
        asm volatile("v_cmp_le_f32 vcc, %1, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %2, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
                     "v_cmp_le_f32 vcc, %3, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                     : "+v"(m) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc");
      } else if (MODE == 5) {  // v_add_f64 x4 independent + v_cvt_f32_f64 x4  (8 instrs)
        float t0_, t1_, t2_, t3_;
        asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
        asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                     : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        a0 += t0_ + t1_ + t2_ + t3_;
      } else if (MODE == 6) {  // v_sqrt_f32 x8 independent
        asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                     "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 7) {  // dependent v_fma_f32 chain (8 instrs on one register)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                     : "+v"(a0) : "v"(b), "v"(c));
      } else if (MODE == 8) {  // dependent LDS round trips: 8 x (ds_read_b32 -> address)
#pragma unroll
        for (int q = 0; q < 8; ++q) idx = __float_as_int(lds[idx & 255].y) & 255;
      } else if (MODE == 9) {  // 8 independent broadcast ds_read_b64 then one wait
        float2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = lds[(r + q) & 255];
#pragma unroll
        for (int q = 0; q < 8; ++q) a0 += v[q].x;
      } else if (MODE == 10) {  // v_cndmask (VCC read) x8 independent after one cmp
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n"
                     "v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
      } else if (MODE == 11) {  // 64-bit compare + 4 cndmask (one sort-network compare-exchange), x2 = 10 instrs -> counted as 8
        asm volatile("v_cmp_gt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n"
                     "v_cmp_gt_u64 vcc, %1, %0\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %6, vcc"
                     : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) :: "vcc");
      } else if (MODE == 12) {  // v_mul_f64 x4 + v_cvt_f64_f32 x4
        asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));
        asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      } else if (MODE == 13) {  // v_med3_f32 x8 fully independent
        asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                     "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      } else if (MODE == 20) {  // 8 v_cndmask reading a VCC set once OUTSIDE the loop
        asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                     "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 21) {  // 8 v_cmp only (each overwrites VCC)
        asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                     "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                     :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b) : "vcc");
      } else if (MODE == 22) {  // 4 x (v_cmp -> v_cndmask) pairs
        asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                     "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
      } else if (MODE == 23) {  // v_cmp -> 7 independent v_fma (no VCC reader)
        asm volatile("v_cmp_lt_f32 vcc, %7, %8\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                     "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(a7), "v"(b), "v"(c) : "vcc");
      } else if (MODE == 24) {  // v_cmp (e64 -> SGPR pair) + 3 v_cndmask_e64 with that pair, x2
        asm volatile("v_cmp_lt_f32 s[20:21], %0, %6\n v_cndmask_b32 %0, %0, %6, s[20:21]\n v_cndmask_b32 %1, %1, %6, s[20:21]\n v_cndmask_b32 %2, %2, %6, s[20:21]\n"
                     "v_cmp_lt_f32 s[22:23], %3, %6\n v_cndmask_b32 %3, %3, %6, s[22:23]\n v_cndmask_b32 %4, %4, %6, s[22:23]\n v_cndmask_b32 %5, %5, %6, s[22:23]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b) : "s20", "s21", "s22", "s23");
      } else if (MODE == 25) {  // integer VALU: and / lshl / add / bfe / ffbl x8 independent
        asm volatile("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_add_u32 %2, %2, %8\n v_bfe_u32 %3, %3, 3, 7\n"
                     "v_ffbl_b32 %4, %4\n v_xor_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_sub_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 26) {  // v_max_f32 + v_min_f32 pair as a 2-op insertion step, dependent pairs x4
        asm volatile("v_max_f32 %4, %0, %8\n v_min_f32 %1, %1, %4\n v_max_f32 %5, %1, %8\n v_min_f32 %2, %2, %5\n"
                     "v_max_f32 %6, %2, %8\n v_min_f32 %3, %3, %6\n v_max_f32 %7, %3, %8\n v_min_f32 %0, %0, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 27) {  // v_mul_hi_u32 / v_mul_lo_u32 (Philox) x8
        asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                     "v_mul_hi_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 28) {  // v_mov_b32 x8
        asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                     "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 29) {  // ds_write_b32 x8 (stride-71 rows)
#pragma unroll
        for (int q = 0; q < 8; ++q) ((float *)lds)[(threadIdx.x & 63) * 7 + q + (r & 1)] = a0;
      } else if (MODE == 30) {  // v_add_f32 / v_mul_f32 / v_sub_f32 x8 independent (VOP2)
        asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 40) {  // v_cmp -> vcc ; 7 x v_cndmask_e64 with vcc as explicit operand
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n"
                     "v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
      } else if (MODE == 41) {  // v_cmp_e64 -> s[20:21] ; 7 x v_cndmask_e64 s[20:21]
        asm volatile("v_cmp_lt_f32 s[20:21], %8, %9\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n"
                     "v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21");
      } else if (MODE == 42) {  // v_cmp -> vcc ; v_fma ; v_cndmask_e32 vcc   (one instruction between), x2 + 2 fma
        asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, vcc\n v_fma_f32 %2, %2, %4, %5\n"
                     "v_cmp_lt_f32 vcc, %5, %4\n v_fma_f32 %0, %0, %4, %5\n v_cndmask_b32 %3, %3, %4, vcc\n v_fma_f32 %2, %2, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
      } else if (MODE == 43) {  // v_and_b32 with a mask register x8
        asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                     "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 44) {  // v_cmp_u64 -> s[20:21]; 4 cndmask_e64 (one 64-bit compare-exchange) ; 3 fma
        asm volatile("v_cmp_gt_u64 s[20:21], %0, %1\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %2, s[20:21]\n v_cndmask_b32 %4, %4, %5, s[20:21]\n"
                     "v_cndmask_b32 %5, %5, %4, s[20:21]\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %6, %7\n v_fma_f32 %6, %6, %7, %7"
                     : "+v"(d0), "+v"(d1), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) :: "s20", "s21");
      } else if (MODE == 45) {  // v_min_u32 / v_max_u32 x8 independent
        asm volatile("v_min_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n"
                     "v_min_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_max_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 46) {  // v_cmp_e64 -> sgpr pair; v_addc_e64 / v_subb_e64 with explicit carry-in (rank counting), x2 + 2 fma
        asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_addc_co_u32 %2, s[22:23], %2, 0, s[20:21]\n v_subb_co_u32 %3, s[22:23], %3, 0, s[20:21]\n v_fma_f32 %4, %4, %4, %5\n"
                     "v_cmp_lt_u32 s[24:25], %1, %0\n v_addc_co_u32 %3, s[26:27], %3, 0, s[24:25]\n v_subb_co_u32 %2, s[26:27], %2, 0, s[24:25]\n v_fma_f32 %5, %5, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      } else if (MODE == 47) {  // v_lshl_add_u32 / v_add3_u32 / v_mad_u32_u24 (address math) x8
        asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_add3_u32 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_lshl_add_u32 %3, %3, 2, %8\n"
                     "v_add3_u32 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_add3_u32 %7, %7, %8, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 14) {  // v_readlane_b32 + VALU use of the SGPR, 4 pairs
        int s0, s1, s2, s3;
        asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9"
                     : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        asm volatile("v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %5, %1\n v_sub_f32 %2, %6, %2\n v_sub_f32 %3, %7, %3"
                     : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "s"(s1), "s"(s2), "s"(s3));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)] = t1 - t0;
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m + (float)(d0 + d1 + d2 + d3) + p0.x + p1.x + p2.x + p3.y + idx;
  if (s == 123.4567f) sink[0] = s;
}

template <int MODE>
void run(const char *name, unsigned long long *dout, float *dsink) {
  printf("%-44s", name);
  for (int wps : {1, 2, 4}) {  // wavefronts per SIMD: 256-thread blocks = 1 wave per SIMD each
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, dout, dsink, 1.5f);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, dout, dsink, 1.5f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("  %d w/SIMD: %6.2f cyc/instr", wps, med / (REPS * 32.0));
  }
  printf("\n");
}

int main() {
  unsigned long long *dout;
  float *dsink;
  CHECK(hipMalloc(&dout, 8 * 4096 * 4));
  CHECK(hipMalloc(&dsink, 64));
  run<0>("v_fma_f32 x8 independent", dout, dsink);
  run<7>("v_fma_f32 dependent chain", dout, dsink);
  run<13>("v_med3_f32 x8 independent", dout, dsink);
  run<1>("v_med3_f32 insertion chain (7 med3 + min)", dout, dsink);
  run<2>("v_min/v_max_f32 x8 independent", dout, dsink);
  run<3>("v_pk_add/mul_f32 x8 (4 chains)", dout, dsink);
  run<4>("v_cmp + v_addc chain through VCC (8 instrs)", dout, dsink);
  run<10>("v_cmp + 7 v_cndmask (VCC)", dout, dsink);
  run<11>("v_cmp_u64 + 3 cndmask, x2", dout, dsink);
  run<5>("v_add_f64 x4 + v_cvt_f32_f64 x4", dout, dsink);
  run<12>("v_mul_f64 x4 + v_cvt_f64_f32 x4", dout, dsink);
  run<6>("v_sqrt_f32 x8 independent", dout, dsink);
  run<14>("v_readlane x4 + v_sub(sgpr) x4", dout, dsink);
  run<30>("v_add/mul/sub_f32 x8 independent (VOP2)", dout, dsink);
  run<28>("v_mov_b32 x8", dout, dsink);
  run<25>("integer and/lshl/add/bfe/ffbl/xor/or/sub", dout, dsink);
  run<27>("v_mul_hi_u32 / v_mul_lo_u32 x8", dout, dsink);
  run<26>("v_max+v_min dependent pairs", dout, dsink);
  run<20>("8 v_cndmask, VCC set outside the loop", dout, dsink);
  run<21>("8 v_cmp (VCC writes only)", dout, dsink);
  run<22>("4 x (v_cmp -> v_cndmask vcc)", dout, dsink);
  run<23>("v_cmp + 7 v_fma (VCC never read)", dout, dsink);
  run<24>("2 x (v_cmp_e64 sgpr + 3 v_cndmask_e64)", dout, dsink);
  run<40>("v_cmp vcc + 7 v_cndmask_e64 vcc", dout, dsink);
  run<41>("v_cmp_e64 sgpr + 7 v_cndmask_e64 sgpr", dout, dsink);
  run<42>("v_cmp vcc; fma; cndmask_e32 vcc; fma (x2)", dout, dsink);
  run<43>("v_and_b32 x8", dout, dsink);
  run<45>("v_min_u32 / v_max_u32 x8", dout, dsink);
  run<44>("cmp_u64_e64 sgpr + 4 cndmask_e64 + 3 fma", dout, dsink);
  run<46>("cmp_e64 + addc_e64 + subb_e64 + fma (x2)", dout, dsink);
  run<47>("lshl_add / add3 / mad_u24 x8", dout, dsink);
  run<29>("ds_write_b32 x8 (8 per unit)", dout, dsink);
  run<8>("ds_read_b32 dependent chain (8 per unit)", dout, dsink);
  run<9>("ds_read_b64 x8 broadcast + 8 adds (16 per unit)", dout, dsink);
  return 0;
}
