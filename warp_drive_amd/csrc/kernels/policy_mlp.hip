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
#pragma once
#include "wd_common.h"

namespace {

typedef float mlp_v16 __attribute__((ext_vector_type(16)));
typedef float mlp_v4 __attribute__((ext_vector_type(4)));
typedef float mlp_v4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte access

// row inside a 32-row tile of accumulator register s, lane half h
__device__ __forceinline__ int mlp_row(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// one chunk of packed weights (n_tiles x 4 KB) global -> LDS, split over the block's four wavefronts
__device__ __forceinline__ void mlp_fetch(float *buf, const float *src, int n_tiles, int wave, int lane) {
  // 16-byte vectors: n_tiles * 256; each wavefront moves a quarter, 64 vectors per instruction
  const int rounds = n_tiles;  // (n_tiles * 256 / 4) / 64
  for (int r = 0; r < rounds; ++r) {
    const int v0 = (wave * rounds + r) * 64;  // first vector of this instruction (wave-uniform)
    __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * (v0 + lane)), WD_LDS_PTR(buf + 4 * v0), 16, 0, 0);
  }
}

// acc[tn] += W_chunk[tn] . B   for one k-tile: 16 steps, B operand of step s = bfrag[s]
template <int TN>
__device__ __forceinline__ void mlp_ktile(mlp_v16 (&acc)[TN], const float *buf, const mlp_v16 &bfrag, int lane) {
  // four output tiles at a time: 16 operand registers in flight instead of 4 * TN
  constexpr int G = TN < 4 ? TN : 4;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
    for (int t0 = 0; t0 < TN; t0 += G) {
      mlp_v4 a[G];
#pragma unroll
      for (int t = 0; t < G; ++t) a[t] = *(const mlp_v4 *)(buf + (((t0 + t) * 4 + s4) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < G; ++t)
          acc[t0 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][e], bfrag[4 * s4 + e], acc[t0 + t], 0, 0, 0);
    }
}

// acc += bias (packed per lane half: [tile][h][16]), optional ReLU
template <int TN, bool RELU>
__device__ __forceinline__ void mlp_bias(mlp_v16 (&acc)[TN], const float *bias_packed, int h) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const mlp_v4 *bp = (const mlp_v4 *)(bias_packed + (tn * 2 + h) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const mlp_v4 b = bp[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[tn][4 * q + e] + b[e];
        acc[tn][4 * q + e] = RELU ? fmaxf(v, 0.0f) : v;
      }
    }
  }
}

struct MlpArgs {
  const float *obs;       // [E * N, F] observation rows (the env's own array)
  int F, N;               // row length, agents per replica
  const int *agent_ids;   // [n_pol] agents of this policy inside a replica
  int n_pol, n_rows;      // n_rows = E * n_pol
  const float *w1, *b1, *w2, *b2, *w3, *b3;  // packed (see training/policy_kernel.py)
  int A0, A1;             // sizes of the softmax heads (A1 = 0: one head); the value is output row A0 + A1
  float *probs0, *probs1; // [E, N, A0], [E, N, A1]
  float *values;          // [n_rows] or null
  float *obs_out;         // [T, n_rows, F] training-batch copy of the rows, or null
  const long long *batch_row;  // device counter: which T-row of obs_out (null: row 0)
};

// TN1 / TN2: hidden widths / 32; KT1: ceil(F / 32)
template <int TN1, int TN2, int KT1>
__device__ __forceinline__ void mlp_impl(const MlpArgs &p, float *lds) {
  constexpr int TN3 = 2;  // output rows padded to 64: all head logits + the value
  constexpr int CHUNK = (TN1 > TN2 ? TN1 : TN2) * 1024;  // floats per LDS buffer
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
  float *const buf0 = lds, *const buf1 = lds + CHUNK;
  const int g = (blockIdx.x * 4 + wave) * 32 + j;  // policy-local row of this lane's column
  const bool valid = g < p.n_rows;
  const int gc = valid ? g : p.n_rows - 1;
  const int env = gc / p.n_pol, a = gc - env * p.n_pol;
  const long src_row = (long)env * p.N + p.agent_ids[a];

  // first weight chunk, then this lane's part of its observation row: features
  // [32 kt + 16 h, 32 kt + 16 h + 16) of k-tile kt (zero past the end of the row)
  mlp_fetch(buf0, p.w1, TN1, wave, lane);
  mlp_v16 feat[KT1];
  {
    const float *row = p.obs + src_row * p.F;
    float *out = nullptr;
    if (p.obs_out && valid) {
      const long long t = p.batch_row ? *p.batch_row : 0;
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

  // chunk c of the stream lives in buf[c & 1]; while it is consumed the next one is fetched
  int c = 0;
#define MLP_NEXT_CHUNK(next_src, next_tiles, have_next)                                  \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wavefront's part of chunk c */ \
  __syncthreads(); /* everybody's part; and nobody reads the other buffer any more */    \
  if (have_next) mlp_fetch((c & 1) ? buf0 : buf1, (next_src), (next_tiles), wave, lane); \
  const float *const cur = (c & 1) ? buf1 : buf0;                                        \
  ++c;

  // ---- layer 1: H1^T = relu(W1 . X^T + b1)
  mlp_v16 acc1[TN1];
#pragma unroll
  for (int tn = 0; tn < TN1; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc1[tn][s] = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
    const bool last = kt == KT1 - 1;
    MLP_NEXT_CHUNK(last ? p.w2 : p.w1 + (size_t)(kt + 1) * TN1 * 1024, last ? TN2 : TN1, true)
    mlp_ktile<TN1>(acc1, cur, feat[kt], lane);
  }
  mlp_bias<TN1, true>(acc1, p.b1, h);

  // ---- layer 2: H2^T = relu(W2 . H1^T + b2)
  mlp_v16 acc2[TN2];
#pragma unroll
  for (int tn = 0; tn < TN2; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc2[tn][s] = 0.0f;
#pragma unroll
  for (int kt = 0; kt < TN1; ++kt) {
    const bool last = kt == TN1 - 1;
    MLP_NEXT_CHUNK(last ? p.w3 : p.w2 + (size_t)(kt + 1) * TN2 * 1024, last ? TN3 : TN2, true)
    mlp_ktile<TN2>(acc2, cur, acc1[kt], lane);
  }
  mlp_bias<TN2, true>(acc2, p.b2, h);

  // ---- output layer: logits^T (and the value) = W3 . H2^T + b3
  mlp_v16 acc3[TN3];
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc3[tn][s] = 0.0f;
#pragma unroll
  for (int kt = 0; kt < TN2; ++kt) {
    const bool last = kt == TN2 - 1;
    MLP_NEXT_CHUNK(p.w3 + (size_t)(kt + 1) * TN3 * 1024, TN3, !last)
    mlp_ktile<TN3>(acc3, cur, acc2[kt], lane);
  }
#undef MLP_NEXT_CHUNK
  mlp_bias<TN3, false>(acc3, p.b3, h);

  // ---- softmax per head over the rows of a column: a lane holds half of the rows, its partner
  // (lane ^ 32) the other half
  const int r1 = p.A0, r2 = p.A0 + p.A1;  // head 0: rows [0, r1), head 1: [r1, r2), value: row r2
  float m0 = -__builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      const float x = acc3[tn][s];
      if (r < r1) m0 = fmaxf(m0, x);
      else if (r < r2) m1 = fmaxf(m1, x);
    }
  m0 = fmaxf(m0, __shfl_xor(m0, 32));
  m1 = fmaxf(m1, __shfl_xor(m1, 32));
  float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      const float x = acc3[tn][s];
      if (r < r1) { const float e = expf(x - m0); acc3[tn][s] = e; z0 += e; }
      else if (r < r2) { const float e = expf(x - m1); acc3[tn][s] = e; z1 += e; }
    }
  z0 += __shfl_xor(z0, 32);
  z1 += __shfl_xor(z1, 32);
  if (!valid) return;
  float *const o0 = p.probs0 + src_row * p.A0;
  float *const o1 = p.A1 ? p.probs1 + src_row * p.A1 : nullptr;
#pragma unroll
  for (int tn = 0; tn < TN3; ++tn)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int r = 32 * tn + mlp_row(s, h);
      const float x = acc3[tn][s];
      if (r < r1) o0[r] = x / z0;
      else if (r < r2) o1[r - r1] = x / z1;
      else if (r == r2 && p.values) p.values[g] = x;
    }
}

}  // namespace

#define WD_MLP_PARAMS                                                                                 \
  const float *obs, int F, int N, const int *agent_ids, int n_pol, int n_rows, const float *w1,       \
      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int A0,    \
      int A1, float *probs0, float *probs1, float *values, float *obs_out, const long long *batch_row
#define WD_MLP_PACK()                                                                                 \
  MlpArgs p;                                                                                          \
  p.obs = obs; p.F = F; p.N = N; p.agent_ids = agent_ids; p.n_pol = n_pol; p.n_rows = n_rows;         \
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.A0 = A0; p.A1 = A1;             \
  p.probs0 = probs0; p.probs1 = probs1; p.values = values; p.obs_out = obs_out; p.batch_row = batch_row;

extern "C" {
// HipPolicyMlp_<H1>x<H2>_k<KT1>: hidden widths H1, H2; observation rows of up to 32 * KT1 floats.
// 256 threads per block (4 wavefronts x 32 rows), dynamic LDS = 2 * max(H1, H2) / 32 * 4096 bytes.
#define WD_MLP_KERNEL(H1, H2, KT1)                                                                    \
  __global__ void __launch_bounds__(256, 1) HipPolicyMlp_##H1##x##H2##_k##KT1(WD_MLP_PARAMS) {        \
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];                          \
    WD_MLP_PACK();                                                                                    \
    mlp_impl<H1 / 32, H2 / 32, KT1>(p, (float *)mlp_smem);                                            \
  }
WD_MLP_KERNEL(256, 256, 1)
WD_MLP_KERNEL(256, 256, 2)
WD_MLP_KERNEL(256, 256, 3)
WD_MLP_KERNEL(128, 128, 1)
WD_MLP_KERNEL(128, 128, 2)
WD_MLP_KERNEL(128, 128, 3)
WD_MLP_KERNEL(64, 64, 1)
WD_MLP_KERNEL(64, 64, 2)
WD_MLP_KERNEL(64, 64, 3)
}
