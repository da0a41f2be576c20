// mlp_head_backward.h -- HipHeadBackward / HipHeadBackwardBx3: the output layer's backward fused with the ReLU mask, bias and weight gradient.
// Part of the trainer's policy-kernel translation unit (policy_mlp.hip, which holds the design notes, the kernel-argument
// macros and the entries); split by kernel family in round 6 with both code objects (wd_kernels_mlp.hsaco, wd_kernels_update.hsaco)
// byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "mlp_forward_bx3.h"
#include "mlp_weight_grad.h"

namespace {

// HipHeadBackward: the backward of the output layer fused with the ReLU mask of the hidden layer under it.  out = h2 . W3^T
// + b3 with W3 [W][C] (W = A0 + A1 + 1 <= 64 output rows: all heads' logits and the value; C = 64 / 128 / 256 hidden units)
// and h2 = relu(...) [R][C].  Given g3 = d loss / d out [R][W]:
//     g2[r][j]  = [h2[r][j] > 0] * sum_k g3[r][k] W3[k][j]        (the masked gradient the hidden layer's GEMMs consume)
//     db2[j]    = sum_r g2[r][j]                                   (its bias gradient)
//     dW3[k][j] = sum_r g3[r][k] h2[r][j]                          (the output layer's weight gradient)
// in ONE pass: g3 and h2 are read once, g2 is written once.  The framework path is a [R, W] x [W, C] GEMM that writes the
// unmasked gradient (10 GB at configs[2]), the mask + column-sum pass that reads it back with h2 and writes it again, and
// a skinny [W, R] x [R, C] GEMM that reads h2 a third time (44 TFLOP/s: W = 43 rows do not fill a tile) -- 15 ms of a
// 56 ms update for 0.44 TFLOP that the vector units do in the shadow of the 22 GB this kernel moves.
// One thread per hidden unit (blockDim.x = C): its column of W3 and its 2 W accumulators live in registers; the rows'
// g3 values reach all threads as LDS broadcasts.  `rows_per_block` rows per block; partial sums per block
// (`db2_part` [blocks][C], `dw3_part` [blocks][W][C]) are reduced by the caller in a fixed order.
template <int W>
__device__ __forceinline__ void head_backward_impl(const float *__restrict__ g3, const float *__restrict__ w3,
                                                   const float *__restrict__ h2, float *__restrict__ g2,
                                                   float *__restrict__ db2_part, float *__restrict__ dw3_part, long R,
                                                   int rows_per_block, float *s_g3) {
  constexpr int RT = 32;  // rows per staged tile of g3
  constexpr int WP = (W + 3) & ~3;
  const int C = blockDim.x, j = threadIdx.x;
  const long r_begin = (long)blockIdx.x * rows_per_block, r_end = min(R, r_begin + rows_per_block);
  float wcol[W], dw[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    wcol[k] = w3[(long)k * C + j];
    dw[k] = 0.0f;
  }
  float db = 0.0f;
  for (long r0 = r_begin; r0 < r_end; r0 += RT) {
    const int rows = (int)min((long)RT, r_end - r0);
    __syncthreads();  // (the previous tile is consumed)
    for (int q = j; q < rows * W; q += C) {  // rows padded to whole 16-byte vectors: the broadcasts below are ds_read_b128
      const int r = (int)(((float)q + 0.5f) * (1.0f / (float)W));  // q / W (exact for these sizes)
      s_g3[r * WP + (q - r * W)] = g3[r0 * W + q];
    }
    float h[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) h[r] = (r < rows) ? h2[(r0 + r) * C + j] : 0.0f;  // (all loads of the tile in flight)
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < RT; ++r) {
      if (r >= rows) break;  // block-uniform
      const float4 *const gv4 = (const float4 *)(s_g3 + r * WP);  // the same address in every lane: LDS broadcasts
      float gv[WP];
#pragma unroll
      for (int q = 0; q < WP / 4; ++q) {
        const float4 v = gv4[q];
        gv[4 * q] = v.x; gv[4 * q + 1] = v.y; gv[4 * q + 2] = v.z; gv[4 * q + 3] = v.w;
      }
      float x = 0.0f;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        x = fmaf(gv[k], wcol[k], x);
        dw[k] = fmaf(gv[k], h[r], dw[k]);
      }
      const float m = h[r] > 0.0f ? x : 0.0f;
      g2[(r0 + r) * C + j] = m;
      db += m;
    }
  }
  db2_part[(long)blockIdx.x * C + j] = db;
#pragma unroll
  for (int k = 0; k < W; ++k) dw3_part[((long)blockIdx.x * W + k) * C + j] = dw[k];
}

// ---- HipHeadBackwardBx3: the same three results on the bf16 matrix cores (bf16x3: float32-accurate), 256 hidden units ----
// The kernel above is bound by its 2 W float32 FMAs per row and unit on the vector units (9.2 ms at configs[2] for the 22 GB
// it moves).  Here a block of four wavefronts is persistent over a slab of rows, in steps of 32 rows staged in LDS by
// LDS-direct loads (g3: the step's 32 W floats as one flat run; h2: 32 rows of 260 floats), and wavefront w owns hidden units
// [64 w, 64 w + 64):
//   g2^T tile [64 units x 32 rows] = W3^T . g3^T   A = W3^T, the three bf16 terms of this wavefront's 64 units resident in
//                                   registers (packed by the host in register-image order, zero for k >= W); B = the step's
//                                   g3 rows, lane = row, 8 consecutive k (reads past a row's W values meet zero weights);
//                                   masked with h2 read back from the stage in accumulator layout, stored, and summed per
//                                   lane into the bias-gradient partials (reduced across rows once, at the end);
//   db3 [W] += column sums of the step's g3 (8 rows per wavefront): the output layer's bias gradient rides along;
//   dW3 tile [W (<= 64) x 64 units] += g3^T . h2  contraction over the step's 32 rows: both operands are read from the
//                                   stage TRANSPOSED (lane = column, 8 consecutive rows), as in weight_grad_bx3.
// R and rows_per_block are multiples of 32 (the caller runs the last R % 32 rows through the framework).
template <int W, int NS>
__device__ __forceinline__ void head_backward_bx3(const float *__restrict__ g3, const mlp_bf8 *__restrict__ w3pk,
                                                  const float *__restrict__ h2, float *__restrict__ g2,
                                                  float *__restrict__ db2_part, float *__restrict__ dw3_part,
                                                  float *__restrict__ db3_part, long R, long rows_per_block,
                                                  unsigned char *lds) {
  constexpr int C = 256, ROW = C + 4, KS = (W + 15) / 16, OT = (W + 31) / 32;
  constexpr int G3MAX = 31 * W + (32 * OT > 16 * KS ? 32 * OT : 16 * KS) - 1;  // the last float of the stage any lane reads
  constexpr int G3P = G3MAX / 256 + 1;                                             // KB pieces of g3 per step
  constexpr int G3Q = (G3P + 3) / 4;  // ... per wavefront
  constexpr int G3F = 256 * 4 * G3Q, STAGE = G3F + 32 * ROW, IPW = G3Q + 8;
  static_assert((NS - 2) * IPW < 64, "vmcnt");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = r_begin + rows_per_block < R ? r_begin + rows_per_block : R;
  const int steps = r_begin < r_end ? (int)((r_end - r_begin) >> 5) : 0;
  float *const stages = (float *)lds;
  const long g3_vectors = (R * W) >> 2;

  auto issue = [&](int s) {
    float *const dst = stages + (s % NS) * STAGE;
    const long r0 = r_begin + 32l * s;
#pragma unroll
    for (int q = 0; q < G3Q; ++q) {
      const int k = wave + 4 * q;
      long v = ((r0 * W) >> 2) + 64 * k + lane;
      v = v < g3_vectors ? v : g3_vectors - 1;
      __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(g3 + 4 * v), WD_LDS_PTR(dst + 256 * k), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = wave + 4 * q;
      const long r = r0 + row < R ? r0 + row : R - 1;
      __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(h2 + r * C + 4 * lane), WD_LDS_PTR(dst + G3F + row * ROW), 16, 0, 0);
    }
  };

  // W3^T of this wavefront's units: [tile][k step][term]
  mlp_bf8 w3r[2][KS][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int term = 0; term < 3; ++term) w3r[t][ks][term] = w3pk[(((wave * 2 + t) * KS + ks) * 3 + term) * 64 + lane];
  mlp_v16 accw[OT][2], gsum[2];
  float g3sum = 0.0f;  // lane k < W: column k of g3 over this wavefront's 8 rows of every step (the output layer's bias gradient)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) accw[ot][0][s] = accw[ot][1][s] = 0.0f;
    gsum[0][s] = gsum[1][s] = 0.0f;
  }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  wg_wait_loads<(NS - 2) * IPW>();
  __builtin_amdgcn_s_barrier();  // step 0 is in LDS
  constexpr int GT[6] = {2, 0, 1, 1, 0, 0}, XT[6] = {0, 2, 1, 0, 1, 0};  // (first, second operand's term), ascending product size
  for (int s = 0; s < steps; ++s) {
    issue(s + NS - 1);  // into the stage read during step s - 1
    const float *const g3s = stages + (s % NS) * STAGE, *const h2s = g3s + G3F;
    const long r0 = r_begin + 32l * s;
    float v[8];
    // ---- g2^T = W3^T . g3^T
    mlp_bf8 g3b[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = g3s[c * W + 16 * ks + 8 * h + e];
      wg_split3(v, g3b[ks]);
    }
    mlp_v16 accx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) accx[t][i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          accx[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3r[t][ks][GT[m]], g3b[ks][XT[m]], accx[t], 0, 0, 0);
    // ---- dW3 += g3^T . h2 over the step's 32 rows (two k steps of 16)
    mlp_bf8 g3a[OT][2][3], h2b[2][2][3];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g3s[(16 * kk + 8 * h + e) * W + 32 * ot + c];  // (column >= W: somebody's value, a row of dW3 nobody reads)
        wg_split3(v, g3a[ot][kk]);
      }
#pragma unroll
      for (int ut = 0; ut < 2; ++ut) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = h2s[(16 * kk + 8 * h + e) * ROW + 64 * wave + 32 * ut + c];
        wg_split3(v, h2b[ut][kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int ut = 0; ut < 2; ++ut)
            accw[ot][ut] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3a[ot][kk][GT[m]], h2b[ut][kk][XT[m]], accw[ot][ut], 0, 0, 0);
    if (lane < W) {
#pragma unroll
      for (int r = 0; r < 8; ++r) g3sum += g3s[(8 * wave + r) * W + lane];
    }
    // ---- mask, store, bias partials: accumulator register 4 q + e of lane (c, h) = unit 8 q + 4 h + e of the tile, row c
    float *const orow = g2 + (r0 + c) * C + 64 * wave + 4 * h;
    const float *const hrow = h2s + c * ROW + 64 * wave + 4 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const mlp_v4 hm = *(const mlp_v4 *)(hrow + 32 * t + 8 * q);
        mlp_v4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = hm[e] > 0.0f ? accx[t][4 * q + e] : 0.0f;
          gsum[t][4 * q + e] += o[e];
        }
        *(mlp_v4 *)(orow + 32 * t + 8 * q) = o;
      }
    wg_wait_loads<(NS - 2) * IPW>();  // (loads only -- but the stores above count too: see the note at the entry point)
    __builtin_amdgcn_s_barrier();
  }
  // ---- results.  dW3 tile register i of lane (c, h): row k = 32 ot + (i & 3) + 8 (i >> 2) + 4 h, unit 64 wave + 32 ut + c
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = 32 * ot + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < W) dw3_part[((long)blockIdx.x * W + k) * C + 64 * wave + 32 * ut + c] = accw[ot][ut][i];
      }
  if (lane < W) db3_part[((long)blockIdx.x * 4 + wave) * W + lane] = g3sum;
  // bias partials: sum over the 32 rows (lanes c) of every (tile, register, h), through LDS -- once the LDS-direct loads of
  // the steps past the slab's end (issued to keep the wait counts uniform) have landed in the stages this reuses
  wg_wait_loads<0>();
  __syncthreads();
  float *const red = (float *)lds + wave * (2 * 16 * 64);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(t * 16 + i) * 64 + lane] = gsum[t][i];
  __syncthreads();
  {
    // thread (wave, lane) -> unit 64 wave + lane: tile t = lane >> 5, inside it u = lane & 31: h = (u >> 2) & 1, register
    // i = 4 (u >> 3) + (u & 3)
    const int t = lane >> 5, u = lane & 31, hh = (u >> 2) & 1, i = 4 * (u >> 3) + (u & 3);
    float acc = 0.0f;
    for (int cc = 0; cc < 32; ++cc) acc += red[(t * 16 + i) * 64 + 32 * hh + cc];
    db2_part[(long)blockIdx.x * C + 64 * wave + lane] = acc;
  }
}

}  // namespace
