// mlp_weight_grad.h -- HipWeightGradBx3: a layer's weight gradient over the training batch on the bf16 matrix cores.
// Part of the trainer's policy-kernel translation unit (policy_mlp.hip, which holds the design notes, the kernel-argument
// macros and the entries); split by kernel family in round 6 with both code objects (wd_kernels_mlp.hsaco, wd_kernels_update.hsaco)
// byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "mlp_forward_bx3.h"

namespace {

// ---- weight gradient of a layer over a training batch: dW[o][i] = sum over rows of G[row][o] * X[row][i] ----------------
// ~1e7 rows, a 256 x 256 (or 256 x 71) result: the contraction is the long dimension.  As float32 GEMMs (batched over row
// slices, then summed) hipBLASLt runs this AT the f32 matrix peak -- 9.0 ms for the square layer at configs[2], 5.0 ms for
// the first layer (profiles/r05_update_kernels.txt) -- so the only way down is cheaper arithmetic: the forward's bf16x3
// (six bf16 MFMAs per float32 product, 2.7 x the f32 matrix rate, float32-accurate), which leaves the kernel bound by
// reading G and X once (20 GB / 12.8 GB).
// One persistent block per CU owns a contiguous slab of rows and the WHOLE result: 4 wavefronts x (PA x PB) 32 x 32
// accumulator tiles.  The operands sit row-major in memory with the contraction index (the row) outermost, and an MFMA
// operand wants 8 contraction indices of ONE column in a lane: a transpose.  It happens in LDS: steps of 16 rows are
// streamed global -> LDS by LDS-direct loads (no registers, so the stream runs NS - 1 steps = ~100 KB per CU ahead of the
// MFMAs -- a first version that loaded operand registers directly, one step ahead, was latency-bound at 2.4 TB/s), and
// lane (c, h) of a wavefront reads column c, rows 8 h .. 8 h + 7 of its tiles back (row stride padded so that the two lane
// halves hit different banks), splits them into the three bf16 terms and feeds the MFMAs.  Both operands use the same row
// order inside a step, which is all a contraction needs.  One barrier per step.  `ones_col` (>= 0): that column of X
// reads as 1.0 -- its result column is the bias gradient (column sums of G).  R and rows_per_block are multiples of 32
// (the caller adds the last R % 32 rows itself): no step is partial, steps come in pairs, the loop body has no branch, and the compiler is free
// to interleave the MFMAs with the next step's splits.
// Result: partial[block][o][32 TOB] (summed over blocks by the caller: a fixed order, no atomics).
// (mlp_split3 for one half: 8 values)
__device__ __forceinline__ void wg_split3(const float (&x)[8], mlp_bf8 (&out)[3]) {
  mlp_u4 w[3];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float r0 = x[2 * p], r1 = x[2 * p + 1];
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      const mlp_f2 r = {r0, r1};
      const unsigned t = __builtin_bit_cast(unsigned, __builtin_convertvector(r, mlp_bf2));
      w[term][p] = t;
      if (term < 2) {  // exact; two scalar subtractions: a packed one costs more than two issue slots beside MFMAs
        r0 = r0 - __builtin_bit_cast(float, t << 16);
        r1 = r1 - __builtin_bit_cast(float, t & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int term = 0; term < 3; ++term) out[term] = __builtin_bit_cast(mlp_bf8, w[term]);
}

template <int V> struct wg_int { static constexpr int value = V; };
template <int N> __device__ __forceinline__ void wg_wait_loads() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int TOA, int TOB, int WA, int WB, int NS>
__device__ __forceinline__ void weight_grad_bx3(const float *__restrict__ G, const float *__restrict__ X,
                                                float *__restrict__ partial, long R, int ci, int ones_col,
                                                long rows_per_block, unsigned char *lds) {
  constexpr int NW = WA * WB;  // wavefronts of the block
  static_assert((NW == 4 || NW == 8) && TOA == 8 && TOA % WA == 0 && TOB % WB == 0, "the wavefronts tile the 256-row result");
  constexpr int PA = TOA / WA, PB = TOB / WB, CO = 32 * TOA, CIP = 32 * TOB;
  constexpr bool XROWS = TOB == 8;        // 256-wide X: staged row by row like G; narrower: a step's 16 rows as one flat run
  constexpr int ROW = CO + 4;             // floats per staged row: 8 rows further = 32 banks further
  constexpr int GSTAGE = 16 * ROW, XSTAGE = XROWS ? 16 * ROW : 8 * 256, STAGE = GSTAGE + XSTAGE;
  constexpr int IPW = (16 / NW) * (XROWS ? 2 : 1) + (XROWS ? 0 : 8 / NW);  // load instructions per wavefront and step
  static_assert((NS - 2) * IPW < 64, "vmcnt");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int wa = wave / WB, wb = wave % WB;
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = r_begin + rows_per_block < R ? r_begin + rows_per_block : R;
  const int steps = r_begin < r_end ? (int)((r_end - r_begin + 15) >> 4) : 0;
  float *const stages = (float *)lds;
  const long x_vectors = (R * ci) >> 2;  // 16-byte vectors in the R rows of X (R % 32 == 0)

  auto issue = [&](int s) {  // step s: rows r_begin + 16 s .. + 15 -> stage s % NS (addresses clamped into the arrays)
    float *const dst = stages + (s % NS) * STAGE;
    const long r0 = r_begin + 16l * s;
#pragma unroll
    for (int q = 0; q < 16 / NW; ++q) {
      const int row = wave + NW * q;
      const long r = r0 + row < R ? r0 + row : R - 1;
      __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(G + r * CO + 4 * lane), WD_LDS_PTR(dst + row * ROW), 16, 0, 0);
      if (XROWS)
        __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(X + r * CO + 4 * lane), WD_LDS_PTR(dst + GSTAGE + row * ROW), 16, 0, 0);
    }
    if (!XROWS) {
#pragma unroll
      for (int q = 0; q < 8 / NW; ++q) {
        const int k = wave + NW * q;
        long v = ((r0 * ci) >> 2) + 64 * k + lane;
        v = v < x_vectors ? v : x_vectors - 1;
        __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(X + 4 * v), WD_LDS_PTR(dst + GSTAGE + 256 * k), 16, 0, 0);
      }
    }
  };

  // tile operand q of this wavefront (q < PA: its q-th row tile of G, else a column tile of X) of step s -> three bf16 terms
  mlp_bf8 a[2][PA][3], b[2][PB][3];  // [0]: the operands of the step in the MFMAs, [1]: of the next step
  auto split_unit = [&](int s, int q, auto which) {
    constexpr int P = decltype(which)::value;
    const float *const src = stages + (s % NS) * STAGE;
    float v[8];
    if (q < PA) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(8 * h + e) * ROW + 32 * (wa * PA + q) + c];
      wg_split3(v, a[P][q]);
    } else {
      const int col = 32 * (wb * PB + q - PA) + c;
      if (XROWS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[GSTAGE + (8 * h + e) * ROW + col];
      } else {
        const float fill = col == ones_col ? 1.0f : 0.0f;
        const float *const xs = src + GSTAGE + 8 * h * ci + (col < ci ? col : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = col < ci ? xs[e * ci] : fill;
      }
      wg_split3(v, b[P][q - PA]);
    }
  };

  mlp_v16 acc[PA][PB];
#pragma unroll
  for (int ta = 0; ta < PA; ++ta)
#pragma unroll
    for (int tb = 0; tb < PB; ++tb)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[ta][tb][s] = 0.0f;
  // Software pipeline: during step s the MFMAs run on operand registers split during step s - 1, and the operands of step
  // s + 1 are read from LDS and split BETWEEN them (one tile operand per MF / UNITS MFMAs) -- VALU work in the matrix
  // pipe's shadow; without it every wavefront of the block alternated between a split phase and an MFMA phase in step with
  // the others (one barrier per step), and neither unit was busy half the time.
  constexpr int UNITS = PA + PB, MF = 6 * PA * PB, PER = (MF + UNITS - 1) / UNITS;
#pragma unroll
  for (int s = 0; s < NS; ++s) issue(s);
  wg_wait_loads<(NS - 1) * IPW>();
  __builtin_amdgcn_s_barrier();  // step 0 is in LDS
#pragma unroll
  for (int q = 0; q < UNITS; ++q) split_unit(0, q, wg_int<0>{});
  wg_wait_loads<(NS - 2) * IPW>();
  __builtin_amdgcn_s_barrier();  // step 1 is in LDS, nobody reads stage 0 any more
  // (G term, X term) in ascending size of the partial product
  constexpr int GT[6] = {2, 0, 1, 1, 0, 0}, XT[6] = {0, 2, 1, 0, 1, 0};
  auto step = [&](int s, auto parity) {  // MFMAs on operand set P, the next step's operands split into set 1 - P
    constexpr int P = decltype(parity)::value;
    issue(s + NS);  // into the stage that held step s
#pragma unroll
    for (int q = 0; q < UNITS; ++q) {  // (nested so that each loop's unrolled size stays under the compiler's limit)
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = q * PER + k, m = i / (PA * PB), ta = (i / PB) % PA, tb = i % PB;
        if (i < MF)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][ta][GT[m]], b[P][tb][XT[m]], acc[ta][tb], 0, 0, 0);
      }
      split_unit(s + 1, q, wg_int<1 - P>{});  // (past the slab's last step: rows some other block owns, never used)
    }
    wg_wait_loads<(NS - 2) * IPW>();  // this wavefront's part of step s + 2 has landed; its reads of step s + 1 are done
    __builtin_amdgcn_s_barrier();     // (not __syncthreads(): its fence makes the compiler wait for EVERY LDS-direct load)
  };
  for (int s = 0; s < steps; s += 2) {  // (slabs are multiples of 32 rows: an even number of steps)
    step(s, wg_int<0>{});
    step(s + 1, wg_int<1>{});
  }
  // accumulator register s of lane (c, h): result row (s & 3) + 8 (s >> 2) + 4 h, column c of the tile
  float *const out = partial + (size_t)blockIdx.x * CO * CIP;
#pragma unroll
  for (int ta = 0; ta < PA; ++ta)
#pragma unroll
    for (int tb = 0; tb < PB; ++tb)
#pragma unroll
      for (int s = 0; s < 16; ++s)
        out[(32 * (wa * PA + ta) + (s & 3) + 8 * (s >> 2) + 4 * h) * CIP + 32 * (wb * PB + tb) + c] = acc[ta][tb][s];
}

}  // namespace
