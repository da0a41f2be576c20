// policy_mlp.hip -- the rollout's policy forward as ONE kernel.
//
// The reference evaluates its FullyConnected policy (models/fully_connected.py:46-120: MLP trunk,
// one softmax head per action dimension, a value head) with framework GEMMs between the env ticks
// (trainer_base.py:392-405).  For 200 000 observation rows of 71 floats that is three GEMMs, each
// followed by element-wise kernels, with the 256-wide activations written to and read back from
// HBM in between (2 x 205 MB per layer).  Here one launch reads the observation rows in place,
// keeps every activation in registers and writes the probabilities straight into the sampler's
// [E, N, A] tensors (and, optionally, the observation rows into the training batch).
//
// Arithmetic: float32 in, float32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit
// an fmaf chain, no reduced precision) -- the result differs from the framework's GEMMs by summation
// order only.
//
// Layout: everything is computed TRANSPOSED, H^T = W . X^T, a wavefront owning 32 agents (the
// 32 columns of its tiles) and all rows (hidden units) of them.  The accumulator of a 32x32 tile
// holds, in lane (j, h) (j = lane & 31 = column, h = lane >> 5), register s: row
// (s & 3) + 8 * (s >> 2) + 4 * h.  That is exactly the shape of a B operand of the NEXT layer's
// MFMA (lane (j, h) supplies B[k][j] for "its" k of the step) if step s contracts over the rows
// rho(s, 0), rho(s, 1) -- so the activations never leave the registers and never get transposed; the
// order of the contraction index is folded into the (host-side, once per weight update) packing of
// the weights instead.  Weights stream through LDS in chunks of one k-tile (32 contraction indices
// x all output rows: 4 KB per 32x32 tile, packed so that a lane reads the A operands of four
// consecutive steps with one ds_read_b128), double-buffered with global_load_lds, shared by the four
// wavefronts of a block.
#include "wd_common.h"

namespace {

typedef float mlp_v16 __attribute__((ext_vector_type(16)));
typedef float mlp_v4 __attribute__((ext_vector_type(4)));
typedef float mlp_v4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte access

// row inside a 32-row tile of accumulator register s, lane half h
__device__ __forceinline__ int mlp_row(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// one chunk of packed weights (n_tiles x 4 KB) global -> LDS, split over the block's wavefronts (1, 2 or 4)
__device__ __forceinline__ void mlp_fetch(float *buf, const float *src, int n_tiles, int wave, int lane) {
  // 16-byte vectors: n_tiles * 256; each wavefront moves its share, 64 vectors per instruction
  const int rounds = n_tiles * 4 / (int)(blockDim.x >> 6);
  for (int r = 0; r < rounds; ++r) {
    const int v0 = (wave * rounds + r) * 64;  // first vector of this instruction (wave-uniform)
    __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * (v0 + lane)), WD_LDS_PTR(buf + 4 * v0), 16, 0, 0);
  }
}

// acc[tn] += W_chunk[tn] . B for steps [4 * S4_BEGIN, 4 * S4_END) of one k-tile (16 steps; B operand of
// step s = bfrag[s])
template <int TN, int S4_BEGIN, int S4_END>
__device__ __forceinline__ void mlp_ktile(mlp_v16 (&acc)[TN], const float *buf, const mlp_v16 &bfrag, int lane) {
  // operands of G output tiles x 4 steps per LDS read group; the reads of the next group are issued
  // before the MFMAs of the current one (one wavefront per SIMD: nobody else covers the LDS latency)
  constexpr int G = TN < 2 ? TN : 2;
  constexpr int GPS = TN / G;                   // groups per four steps
  constexpr int G0 = S4_BEGIN * GPS, G1 = S4_END * GPS;
  mlp_v4 a[3][G];  // three groups in flight: the reads run two groups (16 MFMAs) ahead
#define MLP_READ_GROUP(gi_)                                                                             \
  {                                                                                                     \
    const int r4 = (gi_) / GPS, r0 = ((gi_) % GPS) * G;                                                 \
    _Pragma("unroll") for (int t = 0; t < G; ++t)                                                       \
        a[(gi_) % 3][t] = *(const mlp_v4 *)(buf + (((r0 + t) * 4 + r4) * 64 + lane) * 4);              \
  }
  MLP_READ_GROUP(G0)
  if (G0 + 1 < G1) MLP_READ_GROUP(G0 + 1)
#pragma unroll
  for (int gi = G0; gi < G1; ++gi) {
    const int s4 = gi / GPS, t0 = (gi % GPS) * G;
    if (gi + 2 < G1) MLP_READ_GROUP(gi + 2)
    // (the scheduler otherwise sinks the reads to just before their first use -- fewer live registers,
    // and an LDS round trip of dead matrix-pipe time per group)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < G; ++t)
        acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi % 3][t][e], bfrag[4 * s4 + e], acc[t0 + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef MLP_READ_GROUP
}

// accumulators start from the bias (packed per lane half: [tile][h][16]) instead of zero
template <int TN>
__device__ __forceinline__ void mlp_init(mlp_v16 (&acc)[TN], const float *bias_packed, int h) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const mlp_v4 *bp = (const mlp_v4 *)(bias_packed + (tn * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const mlp_v4 b = bp[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[tn][4 * q + e] = b[e];
    }
  }
}

template <int TN>
__device__ __forceinline__ void mlp_relu(mlp_v16 (&acc)[TN]) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[tn][s] = fmaxf(acc[tn][s], 0.0f);
}

struct MlpArgs {
  const float *obs;       // [E * N, F] observation rows (the env's own array)
  int F, N;               // row length, agents per replica
  const int *agent_ids;   // [n_pol] agents of this policy inside a replica; null: the range id0 .. id0 + n_pol - 1
  int id0;
  int n_pol, n_rows;      // n_rows = E * n_pol
  const float *w1, *b1, *w2, *b2, *w3, *b3;  // packed (see training/policy_kernel.py)
  int A0, A1;             // sizes of the softmax heads (A1 = 0: one head); the value is output row A0 + A1
  float *probs0, *probs1; // [E, N, A0], [E, N, A1]
  float *values;          // [n_rows] or null
  float *obs_out;         // [T, n_rows, F] training-batch copy of the rows, or null
  const long long *batch_row;  // device counter: which T-row of obs_out (null: row 0)
  int batch_row_stride;   // 0: one counter for the launch; 1: one per replica (all equal: HipRolloutRecord advances
                          // each replica's own, so no kernel needs a cross-block hand-over to advance a shared one)
  // ---- actions drawn in the epilogue (two heads; rng_state null: no sampling).  Same counters, same search as the
  // env's fused tick (tag_continuous.hip::tc_sample_heads): Philox counter (row, epoch, stream_tag, 3), words 0 / 1 for
  // the two heads, inverse CDF on the float32 running sum of the probabilities this kernel would have written
  uint32_t *rng_state;    // seed words + one epoch counter per (replica, agent) row
  int *actions;           // [E * N, 2] the env's `sampled_actions`
  int *act_out;           // [T, n_rows, 2] training-batch copy, or null
  int stream_tag;
  int tile0;              // first 32-row tile of THIS policy in the launch (several policies share one launch)
  // ---- what the UPDATE of an on-policy trainer would otherwise recompute (bf16x3 path; null: not stored): row t of
  // [T, n_rows, H] post-ReLU activations of the two hidden layers and of [T, n_rows, A0 + A1 + 1] outputs (the logits
  // of each head shifted by the head's maximum -- softmax, log-probabilities and entropy do not see the shift -- then
  // the value).  The weights do not change between a rollout and its update, so the update's forward pass is a read.
  float *h1_out, *h2_out, *logits_out;
};

// ---- what follows the output layer, shared by both arithmetic paths: softmax per head, the actions drawn from the
// LDS tile (when asked for), probabilities / value to HBM (when asked for).  acc3: the logits^T tiles (+ the value).
template <int TN3>
__device__ __forceinline__ void mlp_epilogue(const MlpArgs &p, float *lds, mlp_v16 (&acc3)[TN3], int g, bool valid,
                                             long src_row, int wave, int lane, int j, int h) {
  // ---- softmax per head over the rows of a column: a lane holds half of the rows, its partner
  // (lane ^ 32) the other half.  Straight-line code (selects, exp for every register): with one
  // wavefront per SIMD every skipped-over branch costs as much as the work it skips.
  const int r1 = p.A0, r2 = p.A0 + p.A1;  // head 0: rows [0, r1), head 1: [r1, r2), value: row r2
  const float NEG = -__builtin_inff();
  float m0 = NEG, m1 = NEG;
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      const float x = acc3[tn][s];
      m0 = fmaxf(m0, (r < r1) ? x : NEG);
      m1 = fmaxf(m1, (r >= r1 && r < r2) ? x : NEG);
    }
  m0 = fmaxf(m0, __shfl_xor(m0, 32));
  m1 = fmaxf(m1, __shfl_xor(m1, 32));
  if (p.A1 == 0) m1 = 0.0f;  // (no second head: keep the arithmetic below finite)
  constexpr int TS = 65;  // tile stride (odd: conflict-free column writes)
  __syncthreads();        // every wavefront is done with the weight buffers: the tiles below reuse them
  float *const tile = lds + wave * (32 * TS + 32);
  int *const tile_rows = (int *)(tile + 32 * TS);  // destination row of every agent of the tile (-1: none)
  if (p.logits_out) {
    // the outputs the update's objective works on: per head the logits minus the head's maximum, then the value; out
    // through the LDS tile so that every store instruction writes (parts of) whole rows
#pragma unroll
    for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int r = 32 * tn + mlp_row(s, h);
        tile[j * TS + r] = acc3[tn][s] - ((r < r1) ? m0 : (r < r2) ? m1 : 0.0f);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int W = r2 + 1, g0 = g - j;  // floats per row; first policy-local row of the tile
    const long long t = p.batch_row ? p.batch_row[(long)(min(g0, p.n_rows - 1) / p.n_pol) * p.batch_row_stride] : 0;
    float *const dst = p.logits_out + ((long)t * p.n_rows + g0) * W;  // the tile's 32 rows are contiguous
    const int n = min(32, p.n_rows - g0) * W;
    const float inv_w = 1.0f / (float)W;
    for (int q = lane; q < n; q += 64) {
      const int a = (int)(((float)q + 0.5f) * inv_w);  // q / W (exact for these sizes)
      dst[q] = tile[a * TS + (q - a * W)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // (the probabilities overwrite the tile next)
  }
  float z0 = 0.0f, z1 = 0.0f, value = 0.0f;
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      const float x = acc3[tn][s];
      const bool in0 = r < r1, in1 = r >= r1 && r < r2;
      value = (r == r2) ? x : value;
      // (v_exp_f32: ~1 ulp of 2^t, t = (x - m) log2 e <= 0; rows outside the heads: anything finite)
      const float e = __expf(fminf(x - (in0 ? m0 : m1), 0.0f));
      z0 += in0 ? e : 0.0f;
      z1 += in1 ? e : 0.0f;
      acc3[tn][s] = e;
    }
  z0 += __shfl_xor(z0, 32);
  z1 += __shfl_xor(z1, 32);
  const float inv0 = 1.0f / z0, inv1 = 1.0f / fmaxf(z1, 1.0e-30f);
  // The probabilities leave through LDS so that every store instruction writes whole rows: a lane
  // holds single elements of its agent's rows, and storing them directly is 64 separate 4-byte
  // segments per instruction.  The tile [32 agents][64 rows (+1)] of a wavefront reuses the weight
  // buffers once every wavefront is done with them (the host sizes the LDS for 4 tiles as well).
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      tile[j * TS + r] = acc3[tn][s] * ((r < r1) ? inv0 : inv1);
    }
  if (h == 0) tile_rows[j] = valid ? (int)src_row : -1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (p.rng_state) {
    // lane (agent of the tile, head): 32 agents x 2 heads = the wavefront.  The tile row holds both heads'
    // probabilities (stride 65: lanes of different agents hit different banks)
    const int ag = lane & 31, head = lane >> 5;
    const int row = tile_rows[ag];
    if (row >= 0) {
      const uint32_t epoch = p.rng_state[WD_RNG_HEADER + row];
      const wd_u4 rnd = wd_philox4x32_10(wd_u4{(uint32_t)row, epoch, (uint32_t)p.stream_tag, 3u}, p.rng_state[0],
                                         p.rng_state[1]);
      const int a = wd_slab_sample(tile + ag * TS + (head ? r1 : 0), head ? p.A1 : p.A0,
                                   wd_u01_open_closed(head ? rnd.y : rnd.x));
      p.actions[2 * (long)row + head] = a;
      if (p.act_out) {
        const long long t = p.batch_row ? p.batch_row[(long)(row / p.N) * p.batch_row_stride] : 0;
        p.act_out[2 * ((long)t * p.n_rows + (g - j + ag)) + head] = a;
      }
      if (head == 0) p.rng_state[WD_RNG_HEADER + row] = epoch + 1u;
    }
  }
#pragma unroll 1
  for (int head = 0; head < 2; ++head) {
    const int A = head ? p.A1 : p.A0, off = head ? r1 : 0;
    if (A == 0) break;
    float *const out = head ? p.probs1 : p.probs0;
    if (out == nullptr) continue;  // (the actions were drawn above: nobody reads the probabilities)
    const int per_pass = 64 / A;                  // agents per store instruction (heads are <= 63 wide)
    const int sub = (int)(((float)lane + 0.5f) / (float)A), col = lane - sub * A;  // lane -> (agent of the pass, column)
    for (int a0 = 0; a0 < 32; a0 += per_pass) {
      const int ag = a0 + sub;
      if (sub < per_pass && ag < 32) {
        const int row = tile_rows[ag];
        if (row >= 0) out[(long)row * A + col] = tile[ag * TS + off + col];
      }
    }
  }
  // (the value is row r2: it sits in exactly one register of one lane half)
  if (valid && p.values && ((r2 >> 2) & 1) == h && r2 < 32 * TN3) p.values[g] = value;
}

// TN1 / TN2: hidden widths / 32; KT1: ceil(F / 32)
template <int TN1, int TN2, int KT1>
__device__ __forceinline__ void mlp_impl(const MlpArgs &p, float *lds) {
  constexpr int TN3 = 2;  // output rows padded to 64: all head logits + the value
  constexpr int CHUNK = (TN1 > TN2 ? TN1 : TN2) * 1024;  // floats per LDS buffer
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  float *const buf0 = lds, *const buf1 = lds + CHUNK;
  const int g = ((int)(blockIdx.x * (blockDim.x >> 6) + wave) - p.tile0) * 32 + j;  // policy-local row of this lane's column
  const bool valid = g < p.n_rows;
  const int gc = valid ? g : p.n_rows - 1;
  const int env = gc / p.n_pol, a = gc - env * p.n_pol;
  const long src_row = (long)env * p.N + (p.agent_ids ? p.agent_ids[a] : p.id0 + a);

  // first weight chunk, then this lane's part of its observation row: features
  // [32 kt + 16 h, 32 kt + 16 h + 16) of k-tile kt (zero past the end of the row)
  mlp_fetch(buf0, p.w1, TN1, wave, lane);
  mlp_v16 feat[KT1];
  {
    const float *row = p.obs + src_row * p.F;
    float *out = nullptr;
    if (p.obs_out && valid) {
      const long long t = p.batch_row ? p.batch_row[(long)env * p.batch_row_stride] : 0;
      out = p.obs_out + ((long)t * p.n_rows + g) * p.F;
    }
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * kt + 16 * h + 4 * q;
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
        for (int e = 0; e < 4; ++e) feat[kt][4 * q + e] = v[e];
      }
  }

  // chunk c of the stream lives in buf[c & 1]; while it is consumed the next one is fetched.  The
  // fetch instructions come AFTER the first quarter of the chunk's MFMAs: at a chunk boundary the
  // matrix pipe has nothing queued, so whatever is issued before the first MFMA is dead time
  // (stamped build: ~1 200 cycles per boundary with the fetch first, 19 boundaries per block).
  int c = 0;
#define MLP_CHUNK(TN, acc, bfrag, next_src, next_tiles, have_next)                          \
  {                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wavefront's part of chunk c */  \
    __syncthreads(); /* everybody's part; and nobody reads the other buffer any more */     \
    const float *const cur = (c & 1) ? buf1 : buf0;                                         \
    mlp_ktile<TN, 0, 1>(acc, cur, bfrag, lane);                                             \
    if (have_next) mlp_fetch((c & 1) ? buf0 : buf1, (next_src), (next_tiles), wave, lane);  \
    mlp_ktile<TN, 1, 4>(acc, cur, bfrag, lane);                                             \
    ++c;                                                                                    \
  }

  // ---- layer 1: H1^T = relu(W1 . X^T + b1)
  // (every layer's accumulators start from its bias; the loads are issued a layer ahead so that
  // nobody waits for them -- one wavefront per SIMD has nothing else to run meanwhile)
  mlp_v16 acc1[TN1], acc2[TN2], acc3[TN3];
  mlp_init<TN1>(acc1, p.b1, h);
  mlp_init<TN2>(acc2, p.b2, h);
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
    const bool last = kt == KT1 - 1;
    MLP_CHUNK(TN1, acc1, feat[kt], last ? p.w2 : p.w1 + (size_t)(kt + 1) * TN1 * 1024, last ? TN2 : TN1, true)
  }
  mlp_relu<TN1>(acc1);

  // ---- layer 2: H2^T = relu(W2 . H1^T + b2)
  mlp_init<TN3>(acc3, p.b3, h);
#pragma unroll
  for (int kt = 0; kt < TN1; ++kt) {
    const bool last = kt == TN1 - 1;
    MLP_CHUNK(TN2, acc2, acc1[kt], last ? p.w3 : p.w2 + (size_t)(kt + 1) * TN2 * 1024, last ? TN3 : TN2, true)
  }
  mlp_relu<TN2>(acc2);

  // ---- output layer: logits^T (and the value) = W3 . H2^T + b3
#pragma unroll
  for (int kt = 0; kt < TN2; ++kt) {
    const bool last = kt == TN2 - 1;
    MLP_CHUNK(TN3, acc3, acc2[kt], p.w3 + (size_t)(kt + 1) * TN3 * 1024, TN3, !last)
  }
#undef MLP_CHUNK

  mlp_epilogue<TN3>(p, lds, acc3, g, valid, src_row, wave, lane, j, h);
}

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

#define WD_MLP_PARAMS                                                                                 \
  const float *obs, int F, int N, const int *agent_ids, int id0, int n_pol, int n_rows, const float *w1,       \
      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int A0,    \
      int A1, float *probs0, float *probs1, float *values, float *obs_out, const long long *batch_row
#define WD_MLP_PACK()                                                                                 \
  MlpArgs p;                                                                                          \
  p.obs = obs; p.F = F; p.N = N; p.agent_ids = agent_ids; p.id0 = id0; p.n_pol = n_pol; p.n_rows = n_rows;         \
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.A0 = A0; p.A1 = A1;             \
  p.probs0 = probs0; p.probs1 = probs1; p.values = values; p.obs_out = obs_out; p.batch_row = batch_row; \
  p.batch_row_stride = 0;                                                                              \
  p.rng_state = nullptr; p.actions = nullptr; p.act_out = nullptr; p.stream_tag = 0; p.tile0 = 0;   \
  p.h1_out = nullptr; p.h2_out = nullptr; p.logits_out = nullptr;

// HipPolicyMlpAct_*: ALL policies of a rollout tick in ONE launch (blocks [0, first_block_b) serve policy A, the rest
// policy B; first_block_b >= gridDim.x: one policy) with the actions drawn in the epilogue: the probabilities never
// leave the chip (probs0 / probs1 null) and the env's tick is its step + reset entry on given actions
// (`TickA`, tag_continuous.hip).  One launch instead of one per policy: a policy with few rows (10 000 tagger rows
// = a third of a round of blocks) no longer costs a whole round.
#define WD_MLP_ACT_PARAMS                                                                             \
  const float *obs, int F, int N, int A0, int A1, float *probs0, float *probs1, const long long *batch_row, \
      uint32_t *rng_state, int *actions, int stream_tag, int first_block_b,                           \
      const int *a_agent_ids, int a_id0, int a_n_pol, int a_n_rows, const float *a_w1, const float *a_b1, \
      const float *a_w2, const float *a_b2, const float *a_w3, const float *a_b3, float *a_obs_out, int *a_act_out, \
      float *a_h1_out, float *a_h2_out, float *a_logits_out,                                          \
      const int *b_agent_ids, int b_id0, int b_n_pol, int b_n_rows, const float *b_w1, const float *b_b1, \
      const float *b_w2, const float *b_b2, const float *b_w3, const float *b_b3, float *b_obs_out, int *b_act_out, \
      float *b_h1_out, float *b_h2_out, float *b_logits_out
#define WD_MLP_ACT_PACK()                                                                             \
  const bool second = (int)blockIdx.x >= first_block_b; /* block-uniform: scalar selects */           \
  MlpArgs p;                                                                                          \
  p.obs = obs; p.F = F; p.N = N; p.A0 = A0; p.A1 = A1; p.probs0 = probs0; p.probs1 = probs1;          \
  p.values = nullptr; p.batch_row = batch_row; p.batch_row_stride = 1; p.rng_state = rng_state; p.actions = actions; \
  p.stream_tag = stream_tag; p.tile0 = second ? first_block_b * (int)(blockDim.x >> 6) : 0;           \
  p.agent_ids = second ? b_agent_ids : a_agent_ids; p.id0 = second ? b_id0 : a_id0;                   \
  p.n_pol = second ? b_n_pol : a_n_pol; p.n_rows = second ? b_n_rows : a_n_rows;                      \
  p.w1 = second ? b_w1 : a_w1; p.b1 = second ? b_b1 : a_b1; p.w2 = second ? b_w2 : a_w2;              \
  p.b2 = second ? b_b2 : a_b2; p.w3 = second ? b_w3 : a_w3; p.b3 = second ? b_b3 : a_b3;              \
  p.obs_out = second ? b_obs_out : a_obs_out; p.act_out = second ? b_act_out : a_act_out;            \
  p.h1_out = second ? b_h1_out : a_h1_out; p.h2_out = second ? b_h2_out : a_h2_out;                   \
  p.logits_out = second ? b_logits_out : a_logits_out;

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

extern "C" {
// HipPolicyMlp_<H1>x<H2>_k<KT1>: hidden widths H1, H2; observation rows of up to 32 * KT1 floats.
// 64, 128 or 256 threads per block (wavefronts x 32 rows), dynamic LDS = max(2 * max(H1, H2) / 32 * 4096,
// wavefronts * (32 * 65 + 32) * 4) bytes.
#define WD_MLP_KERNEL(H1, H2, KT1)                                                                    \
  __global__ void __launch_bounds__(256, 1) HipPolicyMlp_##H1##x##H2##_k##KT1(WD_MLP_PARAMS) {        \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                          \
    WD_MLP_PACK();                                                                                    \
    mlp_impl<H1 / 32, H2 / 32, KT1>(p, (float *)mlp_smem);                                            \
  }                                                                                                   \
  __global__ void __launch_bounds__(256, 1) HipPolicyMlpAct_##H1##x##H2##_k##KT1(WD_MLP_ACT_PARAMS) { \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                          \
    WD_MLP_ACT_PACK();                                                                                \
    mlp_impl<H1 / 32, H2 / 32, KT1>(p, (float *)mlp_smem);                                            \
  }                                                                                                   \
  /* the bf16x3 arithmetic: same arguments, weights packed as three bf16 terms, dynamic LDS = 3 buffers */ \
  /* of max(H1, H2) / 32 * 6144 bytes                                                                 */ \
  __global__ void __launch_bounds__(256, 1) HipPolicyMlpBx3_##H1##x##H2##_k##KT1(WD_MLP_PARAMS) {     \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                          \
    WD_MLP_PACK();                                                                                    \
    mlp_impl_bx3<H1 / 32, H2 / 32, KT1>(p, (float *)mlp_smem);                                        \
  }                                                                                                   \
  __global__ void __launch_bounds__(256, 1) HipPolicyMlpActBx3_##H1##x##H2##_k##KT1(WD_MLP_ACT_PARAMS) { \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                          \
    WD_MLP_ACT_PACK();                                                                                \
    mlp_impl_bx3<H1 / 32, H2 / 32, KT1>(p, (float *)mlp_smem);                                        \
  }
// HipLinearMaskBackwardBx3_<C>: g_out = [h > 0] * (g_in . W), C x C layer; 256 or 512 threads (a wavefront = 32 rows; 512:
// a weight chunk fetched into LDS serves 8 wavefronts instead of 4), dynamic LDS = 3 * C / 32 * 6144 bytes
#define WD_MLP_MASK_BACKWARD(CC)                                                                                      \
  __global__ void __launch_bounds__(512, 1) HipLinearMaskBackwardBx3_##CC(const float *g_in, const float *wpk,        \
                                                                          const float *h_mask, float *g_out, long R) { \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                                          \
    mlp_mask_backward_bx3<CC / 32>(g_in, wpk, h_mask, g_out, R, (float *)mlp_smem);                                   \
  }
// HipWeightGradBx3_<CO>x<CIP>: partial[block] = G[slab]^T . X[slab]; 64 * WA * WB threads, one block per CU, dynamic LDS
// = WD_WEIGHT_GRAD_STAGES stages of 16 staged rows of G (260 floats each) and of X (the same, or 2048 floats for CIP < 256)
#define WD_WEIGHT_GRAD_STAGES 4
#define WD_WEIGHT_GRAD(CO_, CIP_, WA_, WB_)                                                                           \
  __global__ void __launch_bounds__(64 * WA_ * WB_, 1)                                                                \
      HipWeightGradBx3_##CO_##x##CIP_(const float *G, const float *X, float *partial, long R, int ci, int ones_col,   \
                                      long rows_per_block) {                                                          \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                                          \
    weight_grad_bx3<CO_ / 32, CIP_ / 32, WA_, WB_, WD_WEIGHT_GRAD_STAGES>(G, X, partial, R, ci, ones_col,             \
                                                                          rows_per_block, mlp_smem);                  \
  }
// The file is compiled as two code objects (build.py): -DWD_MLP_PART=1 the rollout's kernels (policy forward, record),
// -DWD_MLP_PART=2 the update's (returns, objective, the backward passes): half the compile time each, side by side.
#if !defined(WD_MLP_PART) || WD_MLP_PART == 2
WD_WEIGHT_GRAD(256, 256, 2, 2)
WD_WEIGHT_GRAD(256, 96, 4, 1)
WD_MLP_MASK_BACKWARD(256)
WD_MLP_MASK_BACKWARD(128)
WD_MLP_MASK_BACKWARD(64)
#endif
#if !defined(WD_MLP_PART) || WD_MLP_PART == 1
WD_MLP_KERNEL(256, 256, 1)
WD_MLP_KERNEL(256, 256, 2)
WD_MLP_KERNEL(256, 256, 3)
WD_MLP_KERNEL(128, 128, 1)
WD_MLP_KERNEL(128, 128, 2)
WD_MLP_KERNEL(128, 128, 3)
WD_MLP_KERNEL(64, 64, 1)
WD_MLP_KERNEL(64, 64, 2)
WD_MLP_KERNEL(64, 64, 3)
#endif

#if !defined(WD_MLP_PART) || WD_MLP_PART == 1
// HipRolloutRecord: the trainer's per-tick bookkeeping as ONE launch (it used to be ~20 framework kernels per tick:
// index_select / index_copy_ per policy and array, the episodic-reward sums -- 100 us of the 670 us tick at
// configs[2]).  After the env tick: row t of every policy's reward batch and of the done batch, the running episodic
// reward per (replica, agent), and -- for replicas that finished on this tick -- the per-replica sums the "Mean
// episodic reward" metric is made of (trainer_base.py:408-426, :514-601 keep the same quantities on the host).
// One block per replica; t = batch_row[replica], which the block advances itself: every replica carries its own copy
// of the batch row, so nothing is handed over between blocks (a shared counter advanced by "the last block to
// finish" cost 2000 serialised atomics: 48 us per tick).  `slot[a]` = policy * 65536 + index of agent a inside its
// policy.
__global__ void HipRolloutRecord(const float *__restrict__ rewards, const int *__restrict__ done, int n_agents,
                                 int n_envs, const int *__restrict__ slot, long long *batch_row,
                                 int *done_batch, float *ep_count, float *reward_batch_a, float *ep_reward_a,
                                 float *ep_sum_a, int n_pol_a, float *reward_batch_b, float *ep_reward_b,
                                 float *ep_sum_b, int n_pol_b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rec_smem[];
  float *const s_total = (float *)rec_smem;  // [n_agents] episodic reward of the agents of a replica that just finished
  const int e = blockIdx.x, tid = threadIdx.x;
  const long long t = batch_row[e];
  const bool finished = done[e] > 0;
  for (int a = tid; a < n_agents; a += blockDim.x) {
    const int sl = slot[a], pol = sl >> 16, la = sl & 0xffff;
    const int n_pol = pol ? n_pol_b : n_pol_a;
    const float r = rewards[(long)e * n_agents + a];
    (pol ? reward_batch_b : reward_batch_a)[((long)t * n_envs + e) * n_pol + la] = r;
    float *const acc = (pol ? ep_reward_b : ep_reward_a) + (long)e * n_pol + la;
    const float total = *acc + r;
    *acc = finished ? 0.0f : total;
    if (finished) s_total[a] = total;
  }
  if (tid == 0) done_batch[(long)t * n_envs + e] = done[e];
  if (finished) {  // block-uniform, once per episode and replica
    __syncthreads();
    if (tid < 2 && (tid == 0 || n_pol_b > 0)) {  // thread p: policy p's agents, in agent order (deterministic sum)
      float sum = 0.0f;
      for (int a = 0; a < n_agents; ++a)
        if ((slot[a] >> 16) == tid) sum += s_total[a];
      (tid ? ep_sum_b : ep_sum_a)[e] += sum / (float)(tid ? n_pol_b : n_pol_a);
    }
    if (tid == 0) ep_count[e] += 1.0f;
  }
  // the replica's row counter advances only after EVERY wavefront of the block has read it (a later wavefront of a
  // multi-wavefront block must not see t + 1)
  __syncthreads();
  if (tid == 0) batch_row[e] = t + 1;
}

#endif  // WD_MLP_PART 1
#if !defined(WD_MLP_PART) || WD_MLP_PART == 2
// HipDiscountedReturns: the bootstrapped discounted returns of a training batch (reference a2c.py:80-95),
//     R[T-1] = done[T-1] ? r[T-1] : V[T-1],   R[t] = r[t] + ((1 - done[t]) * gamma) * R[t+1]
// one thread per (replica, agent) walking its T steps backwards; V = column `v_col` of the network's output rows of width
// `w`.  Also writes R - V (the advantages, when the objective does not normalise).  The framework form is a Python loop
// over T with four small kernels per step: launch-bound, 1.3 ms per policy at T = 50 whatever the batch.  Same operations
// in the same order in float32 (this object is compiled with -ffp-contract=off): bit-identical.
__global__ void __launch_bounds__(256) HipDiscountedReturns(const float *__restrict__ rewards, const int *__restrict__ done,
                                                            const float *__restrict__ out, int w, int v_col, float gamma,
                                                            int T, int E, int n, float *__restrict__ returns,
                                                            float *__restrict__ advantages) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (replica, agent)
  if (i >= (long)E * n) return;
  const int e = (int)(i / n);
  const long stride = (long)E * n;
  float R = 0.0f;
  for (int t = T - 1; t >= 0; --t) {
    const long at = (long)t * stride + i;
    const float d = done[(long)t * E + e] > 0 ? 1.0f : 0.0f, r = rewards[at], v = out[at * w + v_col];
    if (t == T - 1) R = d * r + (1.0f - d) * v;
    else R = r + ((1.0f - d) * gamma) * R;
    returns[at] = R;
    advantages[at] = R - v;
  }
}

// HipPolicyGradientHead: everything between the network's output and its gradient in ONE pass over the batch.  The
// A2C / PPO objective (reference algorithms/policygradient/a2c.py:97-194, ppo.py:150-228) on `out` [R][W] (W = A0 + A1
// + 1: the logits of the two heads -- A1 = 0: one head -- then the value) is
//     loss = mean(-logp(a) * adv) + vf_coeff * mean((v - ret)^2) - ent_coeff * sum_heads mean(H(p_head))
// (PPO, single epoch: ratio = exp(logp - logp.detach()) = 1, the same gradient; its VALUE is -mean(adv)), with adv / ret
// precomputed per row (discounted returns, normalisation: small [T, E, n] tensors).  The framework spends ~60
// element-wise / reduction kernels over [R, 21] tensors on it, forward and backward (R = 1e7 rows at configs[2]: ~30 ms
// of a 110 ms update); the gradient has a closed form,
//     d loss / d z_h[j] = (adv * (p_h[j] - [j == a_h]) + ent_coeff * p_h[j] * (log p_h[j] + H_h)) / R,
//     d loss / d v = 2 vf_coeff (v - ret) / R,
// so one kernel reads `out`, `actions`, `adv`, `ret` and writes `grad` [R][W] plus four partial sums per block
// (sum logp * adv, sum of the heads' entropies, sum (v - ret)^2, sum adv) from which the host forms the loss and the
// logged metrics.  256 rows per block, staged through LDS (coalesced row-major copies in and out; a thread then
// owns one row: stride W floats, conflict-free for odd W).
__global__ void __launch_bounds__(256) HipPolicyGradientHead(const float *__restrict__ out, const int *__restrict__ actions,
                                                             const float *__restrict__ adv, const float *__restrict__ ret,
                                                             float *__restrict__ grad, float *__restrict__ sums, int R,
                                                             int A0, int A1, float inv_R, float ent_coeff, float vf_coeff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pg_smem[];
  float *const tile = (float *)pg_smem;  // [256][W]
  const int W = A0 + A1 + 1, n_heads = A1 > 0 ? 2 : 1;
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 256;
  const int rows = (int)min((long)256, (long)R - row0);
  const int n = rows * W;
  // the block's rows are one contiguous run: 16-byte vectors where the run is aligned (whole blocks of a contiguous tensor:
  // 256 W floats), single floats otherwise -- a dword per lane and instruction made the copies, not the arithmetic, the
  // kernel's time
  const float *const src = out + row0 * W;
  float *const dst = grad + row0 * W;
  const int n4 = ((((size_t)src | (size_t)dst) & 15) == 0) ? n >> 2 : 0;
  for (int q = tid; q < n4; q += 256) ((float4 *)tile)[q] = ((const float4 *)src)[q];
  for (int q = 4 * n4 + tid; q < n; q += 256) tile[q] = src[q];
  __syncthreads();
  float s_pg = 0.0f, s_ent = 0.0f, s_vf = 0.0f, s_adv = 0.0f;
  if (tid < rows) {
    float *const z = tile + tid * W;
    const long row = row0 + tid;
    const float a = adv[row];
    float logp_taken = 0.0f;
#pragma unroll 1
    for (int hd = 0; hd < n_heads; ++hd) {
      float *const zh = z + (hd ? A0 : 0);
      const int A = hd ? A1 : A0;
      const int taken = actions[row * n_heads + hd];
      float m = -__builtin_inff();
      for (int j = 0; j < A; ++j) m = fmaxf(m, zh[j]);
      float total = 0.0f;
      for (int j = 0; j < A; ++j) total += expf(zh[j] - m);
      const float lse = m + logf(total);
      float H = 0.0f;
      for (int j = 0; j < A; ++j) {
        const float lp = zh[j] - lse;
        H -= expf(lp) * lp;
      }
      logp_taken += zh[min(max(taken, 0), A - 1)] - lse;
      for (int j = 0; j < A; ++j) {
        const float lp = zh[j] - lse, pj = expf(lp);
        zh[j] = (a * (pj - (j == taken ? 1.0f : 0.0f)) + ent_coeff * pj * (lp + H)) * inv_R;
      }
      s_ent += H;
    }
    const float d = z[W - 1] - ret[row];
    z[W - 1] = 2.0f * vf_coeff * d * inv_R;
    s_pg = logp_taken * a;
    s_vf = d * d;
    s_adv = a;
  }
  __syncthreads();
  for (int q = tid; q < n4; q += 256) ((float4 *)dst)[q] = ((const float4 *)tile)[q];
  for (int q = 4 * n4 + tid; q < n; q += 256) dst[q] = tile[q];
  // block sums (wave shuffles, then the four wavefronts' partials through LDS, in a fixed order: deterministic)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_pg += __shfl_down(s_pg, off);
    s_ent += __shfl_down(s_ent, off);
    s_vf += __shfl_down(s_vf, off);
    s_adv += __shfl_down(s_adv, off);
  }
  __syncthreads();  // (the tile is free again)
  if ((tid & 63) == 0) {
    float *const w = tile + (tid >> 6) * 4;
    w[0] = s_pg; w[1] = s_ent; w[2] = s_vf; w[3] = s_adv;
  }
  __syncthreads();
  if (tid < 4) sums[(long)blockIdx.x * 4 + tid] = tile[tid] + tile[4 + tid] + tile[8 + tid] + tile[12 + tid];
}

// HipReluBackwardColumnSums: g = gx * [y > 0] (the ReLU mask of a hidden layer's backward) AND the column sums of g
// (that layer's bias gradient) in one pass: the framework's threshold_backward + the two-stage column sum read the
// [R, C] gradient twice more (10 GB each at configs[2]).  C in {16 .. 256} with C / 4 a divisor of 256; block = 256
// threads = C / 4 column quads x 1024 / C row phases; `rows_per_block` rows per block; partial[blockIdx.x][C] holds the block's sums (summed
// over the blocks by the caller).  In place when g == gx.
__global__ void __launch_bounds__(256) HipReluBackwardColumnSums(const float *__restrict__ gx, const float *__restrict__ y,
                                                                 float *__restrict__ g, float *__restrict__ partial,
                                                                 long R, int C, int rows_per_block) {
  __shared__ float s_part[1024];  // [rows_per_pass][C]: 256 / (C / 4) row phases x C columns = 1024 floats
  const int tid = threadIdx.x, quads = C >> 2;
  const int rows_per_pass = 256 / quads;  // (C = 256: 64 lanes per row, 4 rows per pass; C / 4 divides 256)
  const int cq = tid % quads, rp = tid / quads;
  const long r_begin = (long)blockIdx.x * rows_per_block, r_end = min(R, r_begin + rows_per_block);
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (long r = r_begin + rp; r < r_end; r += rows_per_pass) {
    const float4 a = *(const float4 *)(gx + r * C + 4 * cq), b = *(const float4 *)(y + r * C + 4 * cq);
    float4 o;
    o.x = b.x > 0.0f ? a.x : 0.0f; o.y = b.y > 0.0f ? a.y : 0.0f;
    o.z = b.z > 0.0f ? a.z : 0.0f; o.w = b.w > 0.0f ? a.w : 0.0f;
    *(float4 *)(g + r * C + 4 * cq) = o;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  *(float4 *)(s_part + rp * C + 4 * cq) = acc;
  __syncthreads();
  if (tid < C) {  // the row phases of a column, in a fixed order
    float sum = 0.0f;
    for (int k = 0; k < rows_per_pass; ++k) sum += s_part[k * C + tid];
    partial[(long)blockIdx.x * C + tid] = sum;
  }
}

// HipHeadBackwardBx3_W<W>: 256 hidden units, 256 threads, one block per CU; dynamic LDS = WD_HEAD_BACKWARD_STAGES stages of
// (the step's g3 pieces + 32 rows of 260 floats of h2).  The g2 stores count in vmcnt like the loads, in issue order, so
// the wait before a barrier is for "this step's stores and the loads before them": stage s + 1 has landed either way.
#define WD_HEAD_BACKWARD_STAGES 3
#define WD_HEAD_BACKWARD_BX3(WW)                                                                                      \
  __global__ void __launch_bounds__(256, 1) HipHeadBackwardBx3_W##WW(const float *g3, const void *w3pk, const float *h2, \
                                                                     float *g2, float *db2_part, float *dw3_part,       \
                                                                     float *db3_part, long R, long rows_per_block) {    \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                                          \
    head_backward_bx3<WW, WD_HEAD_BACKWARD_STAGES>(g3, (const mlp_bf8 *)w3pk, h2, g2, db2_part, dw3_part, db3_part, R, \
                                                   rows_per_block, mlp_smem);                                         \
  }
WD_HEAD_BACKWARD_BX3(43)
WD_HEAD_BACKWARD_BX3(6)
WD_HEAD_BACKWARD_BX3(3)
#define WD_HEAD_BACKWARD(WW)                                                                                          \
  __global__ void __launch_bounds__(256) HipHeadBackward_W##WW(const float *g3, const float *w3, const float *h2,     \
                                                               float *g2, float *db2_part, float *dw3_part, long R,   \
                                                               int rows_per_block) {                                  \
    __shared__ __attribute__((aligned(16))) float s_g3[32 * ((WW + 3) & ~3)];                                         \
    head_backward_impl<WW>(g3, w3, h2, g2, db2_part, dw3_part, R, rows_per_block, s_g3);                              \
  }
// (one entry per output width the trainer's configs use: 21 + 21 + 1, 5 + 1, 2 + 1; others take the framework path)
WD_HEAD_BACKWARD(43)
WD_HEAD_BACKWARD(6)
WD_HEAD_BACKWARD(3)
#endif  // WD_MLP_PART 2
}
