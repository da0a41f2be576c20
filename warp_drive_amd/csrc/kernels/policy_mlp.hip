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
#include "mlp_forward.h"
#include "mlp_forward_bx3.h"
#include "mlp_mask_backward.h"
#include "mlp_weight_grad.h"
#include "mlp_head_backward.h"

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
