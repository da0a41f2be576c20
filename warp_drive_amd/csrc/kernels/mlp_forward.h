// mlp_forward.h -- the rollout's policy forward in float32 on the matrix cores (v_mfma_f32_32x32x2_f32), its argument block and the epilogue (softmax, sampling, outputs) both arithmetic paths share.
// Part of the trainer's policy-kernel translation unit (policy_mlp.hip, which holds the design notes, the kernel-argument
// macros and the entries); split by kernel family in round 6 with both code objects (wd_kernels_mlp.hsaco, wd_kernels_update.hsaco)
// byte-identical before / after.
#pragma once
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

}  // namespace
