// wd_common.h -- device helpers shared by all gfx950 kernels of the rollout path.
//
// Numerics contract (see DESIGN.md "Parity"): the whole code object is compiled with
// -ffp-contract=off, so a*b+c rounds twice exactly like numpy's separate ufuncs; the
// only fused operations are the explicit __builtin_fmaf calls in wd_np_sincosf(), which
// restates numpy's float32 cos/sin kernel (the reference CPU step calls np.cos/np.sin on
// float32 arrays, example_envs/tag_continuous/tag_continuous.py:370-373).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WD_WAVE 64

// ---------------------------------------------------------------------------------
// numpy float32 sin/cos, bit-exact (Cody-Waite 3-term reduction by pi/2 + minimax
// polynomials; numpy/_core/src/umath/loops_trigonometric.dispatch.*).  Valid for
// |x| <= 71476: directions are kept in [0, 2*pi].
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void wd_np_sincosf(float x, float &sin_out, float &cos_out) {
  const float two_over_pi = 0x1.45f306p-1f;
  const float c1 = -0x1.921fb0p+00f, c2 = -0x1.5110b4p-22f, c3 = -0x1.846988p-48f;
  const float magic = 0x1.800000p+23f;
  // numpy builds this kernel with FP contraction on its FMA dispatch targets: the product
  // x*2/pi is not rounded before the magic add (matters when it lands on k + 0.5).
  float q = __builtin_fmaf(x, two_over_pi, magic);
  q = q - magic;
  float r = __builtin_fmaf(q, c1, x);
  r = __builtin_fmaf(q, c2, r);
  r = __builtin_fmaf(q, c3, r);
  const float r2 = r * r;
  float c = __builtin_fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
  c = __builtin_fmaf(c, r2, 0x1.55553cp-05f);
  c = __builtin_fmaf(c, r2, -0x1.000000p-01f);
  c = __builtin_fmaf(c, r2, 0x1.000000p+00f);
  float s = __builtin_fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
  s = __builtin_fmaf(s, r2, 0x1.11119ap-07f);
  s = __builtin_fmaf(s, r2, -0x1.555556p-03f);
  s = __builtin_fmaf(s, r2, 0.0f);
  s = __builtin_fmaf(s, r, r);
  const int iq = (int)q;
  // sin: quadrant iq ; cos: quadrant iq + 1
  float sv = (iq & 1) == 0 ? s : c;
  if (iq & 2) sv = 0.0f - sv;
  const int ic = iq + 1;
  float cv = (ic & 1) == 0 ? s : c;
  if (ic & 2) cv = 0.0f - cv;
  sin_out = sv;
  cos_out = cv;
}

// numpy remainder (npy_divmodf): result carries the sign of the divisor.
__device__ __forceinline__ float wd_np_remainderf(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.0f) {
    if ((b < 0.0f) != (m < 0.0f)) m += b;
  } else {
    m = copysignf(0.0f, b);
  }
  return m;
}

// ---------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11).  Stateless: the draw
// for (thread, epoch) is a pure function of (seed, thread, epoch, stream tag), so the
// only RNG state in HBM is one 32-bit epoch counter per thread.
// ---------------------------------------------------------------------------------
struct wd_u4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ wd_u4 wd_philox4x32_10(wd_u4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    wd_u4 n;
    n.x = hi1 ^ ctr.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ ctr.w ^ k1;
    n.w = lo0;
    ctr = n;
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

// The draw of the fused ticks of SINGLE-head envs (TagGridWorld, Cartpole): the 32 random bits of (row, epoch) are
// word (epoch & 3) of the Philox block with counter (row, epoch >> 2, stream tag, 4), so a launch that runs T ticks
// pays one Philox call (~100 instructions, 40 of them quarter-rate integer multiplies) per FOUR ticks -- the block
// is cached in registers (`blk`, `blk_quad`; start with blk_quad = 0xffffffff) -- and T single-tick launches draw
// exactly what one T-tick launch draws.  (TagContinuous' two heads share one block per tick: words 0 and 1 of
// counter (row, epoch, stream tag, 3).)  Restated in oracle/core_np.py::single_head_tick_uniform.
__device__ __forceinline__ uint32_t wd_tick_draw(uint32_t row, uint32_t epoch, uint32_t stream_tag, uint32_t k0, uint32_t k1,
                                                 wd_u4 &blk, uint32_t &blk_quad) {
  const uint32_t quad = epoch >> 2;
  if (quad != blk_quad) {
    blk = wd_philox4x32_10(wd_u4{row, quad, stream_tag, 4u}, k0, k1);
    blk_quad = quad;
  }
  // word (epoch & 3) of the block by bit masks: two sign-extended bit fields and three bit-field inserts -- the chain of
  // ternaries on a per-lane index was compiled into three exec-mask branches per draw (a tick of the Cartpole rollout is
  // ~150 instructions: the branches showed)
  const uint32_t m0 = 0u - (epoch & 1u), m1 = 0u - ((epoch >> 1) & 1u);
  const uint32_t lo = (blk.y & m0) | (blk.x & ~m0), hi = (blk.w & m0) | (blk.z & ~m0);
  return (hi & m1) | (lo & ~m1);
}

// uniform in (0, 1], 24 random bits  (curand_uniform's range, random.cu:72)
__device__ __forceinline__ float wd_u01_open_closed(uint32_t bits) {
  return (float)((bits >> 8) + 1u) * 0x1.0p-24f;
}

// A pointer that was READ FROM A TABLE in memory (the reset descriptors) is a generic pointer to the compiler: every
// access through it is a FLAT instruction, whose completion is counted by both memory counters -- waiting for a flat
// load then waits for every global store in flight as well.  The tables only ever hold device-memory addresses: say so.
typedef uint32_t __attribute__((address_space(1))) wd_global_u32;

// Global stores the compiler does NOT track, for the record stores inside a T-tick loop.  The waitcnt pass protects
// the address / data registers of a store until the memory counter says it has left; a loop body that reuses those
// registers on every trip therefore waits for the previous trip's stores (~1 us per tick: the whole cost of a
// Cartpole or TagGridWorld tick).  The hardware reads a store's operands when it issues it (the write-through
// flush of the TagContinuous rows relies on the same), so the registers can be rewritten at once; the s_nop covers
// the one documented hazard (a VALU write to the data registers of a store of more than 64 bits in the next cycle).
// Only for addresses this wavefront does not read back before the kernel ends.
__device__ __forceinline__ void wd_store_untracked(int *p, int v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wd_store_untracked(float *p, float v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wd_store_untracked(float4 *p, float4 v) {
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  const v4f_ q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}

// RNG state layout in HBM (uint32 words): [0]=seed lo, [1]=seed hi, [2]=n_threads,
// [3]=reserved, [4 + tid] = per-thread epoch counter.
#define WD_RNG_HEADER 4

// ---------------------------------------------------------------------------------
// Probability slab of ONE wavefront (the categorical sampler and the fused tick kernels).  The rows of a wavefront's 64 agents are one
// contiguous run of 64*n floats.  It goes global -> LDS directly (global_load_lds_dwordx4: per-lane
// global address, LDS destination = wave-uniform base + lane*16; dword-aligned sources are enough),
// 1 KiB per instruction, fully coalesced, no staging registers, asynchronous until the
// `s_waitcnt vmcnt(0)` before the rows are read back (stride n dwords).  Producer and consumer are
// the same wavefront: no block barrier.
#define WD_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define WD_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
__device__ __forceinline__ void wd_slab_fetch(float *dst, const float *__restrict__ src, int cnt, int lane) {
  const int nvec = cnt >> 2;
  const int nchunk = (nvec + 63) >> 6;  // wave-uniform
  for (int c = 0; c < nchunk; ++c) {
    const int q = c * 64 + lane;
    if (q < nvec) __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * q), WD_LDS_PTR(dst + 256 * c), 16, 0, 0);
  }
  if (lane < (cnt & 3))  // the < 4 floats after the last vector
    __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * nvec + lane), WD_LDS_PTR(dst + 4 * nvec), 4, 0, 0);
}

// inverse CDF on a running float32 sum (random.cu:51-85): number of prefix sums < u, clamped
constexpr int WD_SLAB_CH = 24;  // rows up to this length are read back with all LDS loads in flight
__device__ __forceinline__ int wd_slab_sample(const float *row, int n, float u) {
  int cnt = 0;
  float cum = 0.0f;
  if (n <= WD_SLAB_CH) {
    float p[WD_SLAB_CH];
#pragma unroll
    for (int i = 0; i < WD_SLAB_CH; ++i) p[i] = row[i];  // immediate offsets; entries >= n are the next row's
                                                    // (or, after the last row, table bytes): read, masked off below
    // one subtraction + one funnel shift per entry: the sign of cum - u IS [cum < u] (equal gives +0), shifted into the
    // mask by v_alignbit (m = 2m + sign) -- two full-rate instructions where v_cmp + v_addc (carry in and out) take ~3
    // cycles each on gfx950 (experiments/README.md); the entries past n are dropped with one AND at the end instead of a
    // range check per entry
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < WD_SLAB_CH; ++i) {
      cum = (i == 0) ? p[0] : cum + p[i];
      m = __builtin_amdgcn_alignbit(m, __float_as_uint(cum - u), 31);
    }
    // entry i sits on bit WD_SLAB_CH-1-i
    cnt = __popc(m & (((1u << n) - 1u) << (WD_SLAB_CH - n)));
  } else {
    for (int i = 0; i < n; ++i) {
      cum = (i == 0) ? row[0] : cum + row[i];
      cnt += (cum < u) ? 1 : 0;
    }
  }
  return min(cnt, n - 1);
}
