// mlp_mask_backward.h -- HipLinearMaskBackwardBx3: one hidden layer's input gradient with the ReLU mask of the layer under it.
// Part of the trainer's policy-kernel translation unit (policy_mlp.hip, which holds the design notes, the kernel-argument
// macros and the entries); split by kernel family in round 6 with both code objects (wd_kernels_mlp.hsaco, wd_kernels_update.hsaco)
// byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "mlp_forward_bx3.h"

namespace {

// ---- one hidden layer's INPUT gradient with the ReLU mask of the layer under it, bf16x3:  g_out = [h > 0] * (g_in . W)
// for g_in [R][C] (the gradient with respect to this layer's pre-activations, already masked), W [C out][C in] and h [R][C]
// (the post-ReLU activations of the layer under it).  The update's framework path is a square GEMM (hipBLASLt, ~90 % of the
// f32 matrix peak: 9.3 ms at configs[2]) that writes the unmasked gradient, and a mask pass that reads it back with h and
// writes it again (6 ms): here the product runs on the bf16 matrix cores at float32 accuracy (the forward's arithmetic,
// 2.7 x the f32 matrix rate) and the mask is applied to the accumulators, so 30 GB move once instead of 50.
// Transposed like the forward: G_out^T = W^T . G_in^T, a wavefront owns 32 rows (tile columns); A operand = W^T packed
// with the first-layer mapping over the contraction index (training/update_kernels.py), streamed through the same three
// LDS buffers; B operand = this wavefront's rows of g_in, loaded one k-tile ahead and split in the MFMAs' shadow.
template <int TN>
__device__ __forceinline__ void mlp_mask_backward_bx3(const float *__restrict__ g_in, const float *__restrict__ wpk,
                                                      const float *__restrict__ h_mask, float *__restrict__ g_out,
                                                      long R, float *lds) {
  constexpr int C = 32 * TN, CHUNK = TN * 1536, NC = TN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const long row = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 32 + j;
  const bool valid = row < R;
  const float *const grow = g_in + (valid ? row : R - 1) * C;
  int c = 0;
  auto buffer = [&](int cc) -> float * { return lds + (cc % 3) * CHUNK; };
  auto chunk_src = [&](int cc) -> const float * { return wpk + (size_t)cc * TN * 1536; };
  mlp_fetch_kb(buffer(0), chunk_src(0), 6 * TN, wave, lane);
  // k-tile kt of this lane's row: contraction indices [32 kt + 16 q + 8 h, + 8), q = 0, 1
  mlp_v16 raw[2];
  auto load_slice = [&](int kt, mlp_v16 &dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e4 = 0; e4 < 2; ++e4) {
        const mlp_v4 v = *(const mlp_v4 *)(grow + 32 * kt + 16 * q + 8 * h + 4 * e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[8 * q + 4 * e4 + e] = v[e];
      }
  };
  load_slice(0, raw[0]);
  if (NC > 1) load_slice(1, raw[1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // chunk 0 is published
  if (NC > 1) mlp_fetch_kb(buffer(1), chunk_src(1), 6 * TN, wave, lane);
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 2 < NC) mlp_fetch_kb(buffer(c + 2), chunk_src(c + 2), 6 * TN, wave, lane);
  };
  mlp_bf8 xs[2][3][2];
  mlp_split3(raw[0], xs[0]);
  mlp_v16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[tn][s] = 0.0f;
#pragma unroll
  for (int kt = 0; kt < TN; ++kt) {
    mlp_chunk_bx3<TN>(
        acc, buffer(c), xs[kt & 1], lane, sync,
        [&] { if (kt + 1 < TN) mlp_split3(raw[(kt + 1) & 1], xs[(kt + 1) & 1]); },   // (loaded a chunk ago)
        [&] {
          if (kt + 2 < TN) load_slice(kt + 2, raw[kt & 1]);  // (raw[kt & 1] was split during the previous chunk)
        });
    ++c;
  }
  // Epilogue.  An accumulator lane holds 4-unit runs of ONE row: stored as they are, an instruction writes 32-byte pieces
  // of 32 rows (and reads the mask layer's activations the same way) and the block ends on the memory pipe's issue rate,
  // not on bandwidth.  Where a dead weight buffer leaves room (4.5 KB per wavefront), each 32 x 32 tile goes through LDS
  // and comes back row-major: lane l takes 16 bytes of row l / 8, so an instruction covers whole 128-byte lines of 8 rows.
  constexpr bool STAGED = TN == 8 || TN == 4;
  if constexpr (STAGED) {
    // after the last chunk's barrier nobody reads the buffers of chunks NC - 2 and NC - 3: buffer 0 (TN = 8) / 1 and 2 (TN = 4)
    float *const tile = lds + (TN == 8 ? 0 : CHUNK) + wave * 1152;  // 32 rows x 36 floats
    const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 32;
    const int r8 = lane >> 3, cseg = 4 * (lane & 7);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const mlp_v4 v = {acc[tn][4 * q], acc[tn][4 * q + 1], acc[tn][4 * q + 2], acc[tn][4 * q + 3]};
        *(mlp_v4 *)(tile + j * 36 + 8 * q + 4 * h) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a wavefront's LDS operations execute in order)
      mlp_v4 hm[4], gv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long r = row0 + 8 * it + r8;
        hm[it] = *(const mlp_v4 *)(h_mask + (r < R ? r : R - 1) * C + 32 * tn + cseg);
        gv[it] = *(const mlp_v4 *)(tile + (8 * it + r8) * 36 + cseg);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long r = row0 + 8 * it + r8;
        mlp_v4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hm[it][e] > 0.0f ? gv[it][e] : 0.0f;
        if (r < R) *(mlp_v4 *)(g_out + r * C + 32 * tn + cseg) = v;
      }
    }
  } else if (valid) {
    const float *const hrow = h_mask + row * C;
    float *const orow = g_out + row * C;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      mlp_v4 hm[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) hm[q] = *(const mlp_v4 *)(hrow + 32 * tn + 8 * q + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mlp_v4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hm[q][e] > 0.0f ? acc[tn][4 * q + e] : 0.0f;
        *(mlp_v4 *)(orow + 32 * tn + 8 * q + 4 * h) = v;
      }
    }
  }
}

}  // namespace
