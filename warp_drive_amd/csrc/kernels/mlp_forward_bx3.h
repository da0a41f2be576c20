// mlp_forward_bx3.h -- the same forward with every float32 product emulated on the bf16 matrix cores (bf16x3), activations stored for the update on request.
// Part of the trainer's policy-kernel translation unit (policy_mlp.hip, which holds the design notes, the kernel-argument
// macros and the entries); split by kernel family in round 6 with both code objects (wd_kernels_mlp.hsaco, wd_kernels_update.hsaco)
// byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "mlp_forward.h"

namespace {

// =====================================================================================================================
//   bf16x3: the same network with every float32 product emulated on the bf16 matrix cores (`trainer.policy_arithmetic`)
// =====================================================================================================================
// v_mfma_f32_32x32x2_f32 runs at the float32 VECTOR rate: 1/16 of the bf16 matrix rate (MI355X_MICROARCH.md).  Every
// float32 x is split EXACTLY into three bf16 terms, x = x_hi + x_mid + x_lo (+ a residual below 2^-24 |x|: each term is
// the round-to-nearest bf16 of what the previous ones left, and those subtractions are exact in float32), and a product
// w . x is the sum of the six partial products that reach 2^-24 of it:
//     w_hi x_hi + (w_hi x_mid + w_mid x_hi) + (w_mid x_mid + w_hi x_lo + w_lo x_hi)        [dropped: <= 2^-24 |w x| each]
// each exact in float32 (8 x 8 significant bits), accumulated in float32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32
// cycles cover 16 contraction indices where the float32 form needs 8 of 64 cycles -- 2.7 x the rate for an error of
// the size of float32 rounding itself.  NOT bit-identical to the float32-MFMA path (neither is that one to the
// framework's GEMMs: summation order); the gates are the same: probabilities within 2e-6 of the PyTorch network
// (tests/test_gpu_policy_kernel.py), sampled actions draw for draw on those probabilities.
// The weights are split once per optimizer step on the host (training/policy_kernel.py::pack, [kt][term][tile][k half]
// [lane][8 bf16]: 6 KB per 32 x 32 tile and k-tile); the activations in registers after every layer's ReLU
// (v_cvt_pk_bf16_f32: ~5.5 VALU instructions per value, 128 values per lane and layer).
//
// Weight stream: THREE LDS buffers of one k-tile and the hand-over barrier in the MIDDLE of a chunk's MFMAs.  With two
// buffers the barrier sits at the chunk boundary, where the matrix pipe has nothing queued -- ~1 000 cycles of dead
// time, 19 times per block (float32 path, stamped) -- and it would weigh three times as much against MFMAs that take
// a third of the time.  Here, inside chunk c: first group of MFMAs; wait for this wavefront's pieces of chunk c + 1
// (issued a whole chunk earlier); barrier = chunk c + 1 is published AND every wavefront has left chunk c - 1, so its
// buffer takes the fetch of chunk c + 2, issued right there; remaining MFMAs, whose operands were read from LDS before
// the barrier.  A chunk boundary is then just the next LDS read.
typedef __bf16 mlp_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 mlp_bf2 __attribute__((ext_vector_type(2)));
typedef float mlp_f2 __attribute__((ext_vector_type(2)));

// x[0 .. 15] -> out[term][k half] (8 bf16 each): element e of half q is x[8 q + e].  Each output is assembled as 4 dwords
// (a conversion instruction's packed pair IS an operand register) and the exact residuals are two SCALAR subtractions: a
// packed one costs more than two issue slots beside MFMAs (MI355X_MICROARCH: +26 cycles per two in an MFMA gap).
typedef unsigned mlp_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mlp_split3(const mlp_v16 &x, mlp_bf8 (&out)[3][2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    mlp_u4 w[3];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float r0 = x[8 * q + 2 * p], r1 = x[8 * q + 2 * p + 1];
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const mlp_f2 r = {r0, r1};
        const unsigned t = __builtin_bit_cast(unsigned, __builtin_convertvector(r, mlp_bf2));
        w[term][p] = t;
        if (term < 2) {
          r0 = r0 - __builtin_bit_cast(float, t << 16);          // exact
          r1 = r1 - __builtin_bit_cast(float, t & 0xffff0000u);  // exact
        }
      }
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) out[term][q] = __builtin_bit_cast(mlp_bf8, w[term]);
  }
}

// `pieces` KB of packed weights global -> LDS, this wavefront's share (1 KB per instruction)
__device__ __forceinline__ void mlp_fetch_kb(float *buf, const float *src, int pieces, int wave, int lane) {
  const int rounds = pieces / (int)(blockDim.x >> 6);
  for (int r = 0; r < rounds; ++r) {
    const int v0 = (wave * rounds + r) * 64;  // first 16-byte vector of this instruction (wave-uniform)
    __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * (v0 + lane)), WD_LDS_PTR(buf + 4 * v0), 16, 0, 0);
  }
}

// One chunk = one k-tile (32 contraction indices) of a layer with TN output tiles: acc[tn] += W_chunk[tn] . B, B = the
// three-term split `b` of this k-tile's activations.  Output tiles in PAIRS (MFMAs alternate between two accumulators:
// an instruction between two MFMAs on the same accumulator costs ~43 cycles, between different ones ~6); the LDS
// operand reads of the next pair are issued before the MFMAs of the current one.  `sync` runs after the first pair.
// `fill` runs inside the scheduling region of the first pair's MFMAs: independent VALU / store work (the NEXT k-tile's
// activation split) that the matrix pipe's 32-cycle issue gaps absorb.
template <int TN, typename Sync, typename Fill, typename Late>
__device__ __forceinline__ void mlp_chunk_bx3(mlp_v16 (&acc)[TN], const float *buf, const mlp_bf8 (&b)[3][2], int lane,
                                              Sync sync, Fill fill, Late late) {
  constexpr int G = TN < 2 ? 1 : 2, NG = TN / G;
  const mlp_bf8 *const w = (const mlp_bf8 *)buf;  // [term][tn][k half][lane]
  mlp_bf8 a[2][G][3][2];
#define MLP3_READ(gi_)                                                                                  \
  _Pragma("unroll") for (int t = 0; t < G; ++t)                                                         \
  _Pragma("unroll") for (int term = 0; term < 3; ++term)                                                \
  _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                         \
      a[(gi_) & 1][t][term][q] = w[((term * TN + (gi_) * G + t) * 2 + q) * 64 + lane];
  MLP3_READ(0)
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    if (gi + 1 < NG) MLP3_READ(gi + 1)
    // (no scheduling fences here: left alone the compiler interleaves the reads, the fill work and the MFMAs of a pair
    // a little better than fenced regions did -- 318 -> 307 us per launch, scripts/fwd_ab.sh; hand-placed
    // sched_group_barrier pipelines: 309)
    if (gi == 0) fill();
    // (w term, x term) in ascending size of the partial product: lo x hi, hi x lo, mid x mid, mid x hi, hi x mid, hi x hi
    constexpr int WT[6] = {2, 0, 1, 1, 0, 0}, XT[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int t = 0; t < G; ++t)
                    acc[gi * G + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gi & 1][t][WT[m]][q], b[XT[m]][q], acc[gi * G + t], 0, 0, 0);
    if (gi == 0) {
      sync();
      late();  // (global loads issued here have a whole chunk until the next hand-over's vmcnt(0))
    }
  }
#undef MLP3_READ
}

template <int TN, typename Sync, typename Fill>
__device__ __forceinline__ void mlp_chunk_bx3(mlp_v16 (&acc)[TN], const float *buf, const mlp_bf8 (&b)[3][2], int lane,
                                              Sync sync, Fill fill) {
  mlp_chunk_bx3<TN>(acc, buf, b, lane, sync, fill, [] {});
}

// one layer's post-ReLU activations of this wavefront's 32 agents -> row-major [row][H]: register s of tile tn holds
// hidden unit 32 tn + (s & 3) + 8 (s >> 2) + 4 h of agent j, so registers 4 q .. 4 q + 3 are 16 contiguous bytes
// (and the two lane halves of an agent 32): 4 * TN 16-byte stores per lane
template <int TN>
__device__ __forceinline__ void mlp_store_activations(float *dst, const mlp_v16 (&acc)[TN], bool valid, int h) {
  if (!valid) return;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const mlp_v4 v = {acc[tn][4 * q], acc[tn][4 * q + 1], acc[tn][4 * q + 2], acc[tn][4 * q + 3]};
      *(mlp_v4 *)(dst + 32 * tn + 8 * q + 4 * h) = v;   // (non-temporal stores here: no difference, docs/rounds/r06.md)
    }
}

template <int TN1, int TN2, int KT1>
__device__ __forceinline__ void mlp_impl_bx3(const MlpArgs &p, float *lds) {
  constexpr int TN3 = 2;
  constexpr int TNMAX = TN1 > TN2 ? TN1 : TN2;
  constexpr int CHUNK = TNMAX * 1536;  // floats per LDS buffer: 6 KB per output tile
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int g = ((int)(blockIdx.x * (blockDim.x >> 6) + wave) - p.tile0) * 32 + j;
  const bool valid = g < p.n_rows;
  const int gc = valid ? g : p.n_rows - 1;
  const int env = gc / p.n_pol, ag = gc - env * p.n_pol;
  const long src_row = (long)env * p.N + (p.agent_ids ? p.agent_ids[ag] : p.id0 + ag);

  // the chunk stream: KT1 k-tiles of layer 1, TN1 of layer 2, then the output layer's TN2 k-tiles KG at a time (its
  // k-tiles are only TN3 = 2 output tiles wide: one per chunk would put a hand-over after every 24 MFMAs); chunk c lives
  // in buffer c % 3
  constexpr int KG = (TNMAX / TN3 < TN2) ? TNMAX / TN3 : TN2;  // output-layer k-tiles per chunk (they fill a buffer)
  static_assert(TN2 % KG == 0, "the output layer's k-tiles split evenly into chunks");
  constexpr int NC = KT1 + TN1 + TN2 / KG;
  int c = 0;  // (compile-time after unrolling)
  auto chunk_src = [&](int cc) -> const float * {
    return cc < KT1 ? p.w1 + (size_t)cc * TN1 * 1536
                    : cc < KT1 + TN1 ? p.w2 + (size_t)(cc - KT1) * TN2 * 1536
                                     : p.w3 + (size_t)(cc - KT1 - TN1) * KG * TN3 * 1536;
  };
  auto chunk_pieces = [&](int cc) -> int { return 6 * (cc < KT1 ? TN1 : cc < KT1 + TN1 ? TN2 : TN3 * KG); };
  auto buffer = [&](int cc) -> float * { return lds + (cc % 3) * CHUNK; };

  mlp_fetch_kb(buffer(0), chunk_src(0), chunk_pieces(0), wave, lane);
  // this lane's part of its observation row: k-tile kt, k half q: features [32 kt + 16 q + 8 h, + 8)
  mlp_bf8 x1[KT1][3][2];
  {
    const float *row = p.obs + src_row * p.F;
    float *out = nullptr;
    if (p.obs_out && valid) {
      const long long t = p.batch_row ? p.batch_row[(long)env * p.batch_row_stride] : 0;
      out = p.obs_out + ((long)t * p.n_rows + g) * p.F;
    }
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) {
      mlp_v16 feat;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
          const int f0 = 32 * kt + 16 * q + 8 * h + 4 * e4;
          mlp_v4 v = {0.0f, 0.0f, 0.0f, 0.0f};
          if (f0 + 4 <= p.F) {
            v = *(const mlp_v4u *)(row + f0);
            if (out) *(mlp_v4u *)(out + f0) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (f0 + e < p.F) {
                v[e] = row[f0 + e];
                if (out) out[f0 + e] = v[e];
              }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) feat[8 * q + 4 * e4 + e] = v[e];
        }
      mlp_split3(feat, x1[kt]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pieces of chunk 0 (and its row)
  __syncthreads();                                   // chunk 0 is published
  if (NC > 1) mlp_fetch_kb(buffer(1), chunk_src(1), chunk_pieces(1), wave, lane);

  // inside chunk c, after its first MFMAs (see the header): publish chunk c + 1, fetch chunk c + 2
  auto sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 2 < NC) mlp_fetch_kb(buffer(c + 2), chunk_src(c + 2), chunk_pieces(c + 2), wave, lane);
  };

  mlp_v16 acc1[TN1], acc2[TN2], acc3[TN3];
  // ---- layer 1
  const auto nothing = [] {};
  mlp_init<TN1>(acc1, p.b1, h);
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
    mlp_chunk_bx3<TN1>(acc1, buffer(c), x1[kt], lane, sync, nothing);
    ++c;
  }
  // ---- layers 2 and 3 consume the previous layer's activations one 32-row tile (= one k-tile) at a time: tile kt + 1
  // is ReLU'd and split into its three bf16 terms INSIDE the MFMAs of chunk kt (the matrix pipe's issue gaps absorb the
  // VALU work); only tile 0 of a layer is prepared in the open.  The ReLU'd activations stay in their accumulator
  // registers: when the update wants them (h1_out / h2_out) they are stored after the LAST hand-over of the kernel --
  // every hand-over waits on vmcnt(0), which counts stores too, so a store issued earlier would be waited for (measured:
  // stores spread over the layers cost as much as one burst, +90 us per tick); after the last one nothing waits and
  // the 66 MB a round of blocks writes drains under the epilogue and the next block's first layer.
  mlp_bf8 xs[2][3][2];  // the current and the next tile's split
  auto prepare = [&](mlp_v16 &tile, mlp_bf8 (&out)[3][2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[r] = fmaxf(tile[r], 0.0f);
    mlp_split3(tile, out);
  };
  // ---- layer 2
  mlp_init<TN2>(acc2, p.b2, h);
  prepare(acc1[0], xs[0]);
#pragma unroll
  for (int kt = 0; kt < TN1; ++kt) {
    mlp_chunk_bx3<TN2>(acc2, buffer(c), xs[kt & 1], lane, sync,
                       [&] { if (kt + 1 < TN1) prepare(acc1[kt + 1], xs[(kt + 1) & 1]); });
    ++c;
  }
  // ---- output layer
  mlp_init<TN3>(acc3, p.b3, h);
  prepare(acc2[0], xs[0]);
#pragma unroll
  for (int kt = 0; kt < TN2; ++kt) {
    const auto fill = [&] { if (kt + 1 < TN2) prepare(acc2[kt + 1], xs[(kt + 1) & 1]); };
    const float *const wk = buffer(c) + (kt % KG) * TN3 * 1536;  // this k-tile inside its chunk
    if (kt % KG == 0) mlp_chunk_bx3<TN3>(acc3, wk, xs[kt & 1], lane, sync, fill);      // (hand-over once per chunk)
    else mlp_chunk_bx3<TN3>(acc3, wk, xs[kt & 1], lane, nothing, fill);
    if (kt % KG == KG - 1) ++c;
  }
  if (p.h1_out || p.h2_out) {
    const long long t_row = p.batch_row ? p.batch_row[(long)env * p.batch_row_stride] : 0;
    if (p.h1_out) mlp_store_activations<TN1>(p.h1_out + ((long)t_row * p.n_rows + g) * (32 * TN1), acc1, valid, h);
    if (p.h2_out) mlp_store_activations<TN2>(p.h2_out + ((long)t_row * p.n_rows + g) * (32 * TN2), acc2, valid, h);
  }
  mlp_epilogue<TN3>(p, lds, acc3, g, valid, src_row, wave, lane, j, h);
}

}  // namespace
