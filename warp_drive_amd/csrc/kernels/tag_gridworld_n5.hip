// tag_gridworld_n5.hip -- the TagGridWorld T-tick rollout (HipTagGridWorldRollout in tag_gridworld.hip) specialised
// for the shape every BASELINE config uses: N = 5 agents (4 taggers + the runner), full observations (F = 21),
// blocks of ONE wavefront = 12 replicas.  Its own code object (csrc/wd_kernels_gw5.hsaco): the host picks it when the
// shape matches (envs/tag_gridworld.py::tick_launch), the general kernel serves everything else.
//
// Why: at BASELINE configs[1] (1000 replicas) the rollout is 84 single-wavefront blocks, one per CU -- a tick is one
// wavefront's dependent chain (~530 instructions at ~5 cycles each, two block barriers, ~20 dependent LDS round
// trips: 2.14 us per tick).  What a block of one wavefront does not need:
//   * barriers and LDS vote flags: LDS operations of ONE wavefront execute in issue order, so a lane's read sees any
//     earlier write of another lane; "did any replica finish" is a ballot.  (The wavefront-scope fence at the top
//     of the tick loop is the language-level statement of the same thing; it compiles to nothing here -- the
//     instruction stream is identical with and without it, checked by diffing the assembly -- because every access
//     goes through the same image pointer and already may alias the others);
//   * positions in LDS for the tag check: the runner's cell reaches the replica's five lanes through one
//     ds_bpermute, "some tagger stands on it" is a ballot + a shift;
//   * the per-replica time step in LDS: every lane keeps its replica's in a register;
//   * float divisions per tick (x / L, y / L, t / episode_length: ~15 instructions each, correctly rounded): all
//     quotients that can occur are tabulated in LDS once per launch WITH THE SAME DIVISION, so the values are the
//     reference's bit for bit (tag_gridworld.py:208-214, :273);
//   * index arithmetic with runtime N, F, epb;
//   * the mid-launch write of restored rows to the global arrays: this kernel writes the state after the last tick to
//     every array it restores (positions, observations, time step), so only its LDS / register copies are restored
//     inside the loop.  The host checks that the registered reset arrays are exactly those three.
// Semantics, recording and random draws are those of HipTagGridWorldRollout (same arguments + the action table's
// device address, which lives in the main code object); parity: tests/test_gpu_gridworld.py, same test, same oracle.
//
// LIVE POLICY (`HipTagGridWorldRollout_N5_H32` / `_H64`, the TagGridWorld counterpart of
// HipClassicControlCartPoleEnvRollout_H32 / _H64 in cartpole.hip): every tick evaluates a small network on the agent's
// current observation row instead of reading fixed probabilities -- policy forward, sampling, step, restart and
// recording of a whole training batch in ONE launch (the reference runs a framework forward + sampler + step + reset
// launches + three synchronisations per tick, trainer_base.py:383-428).  Two policies, as in the reference's
// TagGridWorld training config (run_configs/tag_gridworld.yaml: "tagger" for agents 0 - 3, "runner" for agent 4): two
// hidden layers of H = 32 or 64 ReLU units + one softmax head of 5 actions each.  Packed weights per policy
// (training/policy_kernel.py::pack_gridworld_policy): W0 [H][24] (rows of the 21 inputs, padded to 24 floats so that
// every row starts on a 16-byte boundary; the padding is never read), b0 [H], W1 [H][H], b1 [H], Wp [5][H], bp [5],
// float32, the block padded to a multiple of four floats.  Both sets live in LDS for the whole launch; a lane reads
// its own policy's.  Arithmetic: acc = bias, then one fmaf per input in index order; softmax with the maximum
// subtracted, expf, one division per action -- restated in oracle/tag_gridworld_np.py::policy_probabilities.
// Measured (profiles/r05_prepared_items_first_call.txt, 1000 replicas, 20 ticks per launch): 10.9 us per tick at
// H = 32, 30.1 at H = 64 -- the trainer's per-tick path costs 390 - 490 us per tick of the same replicas.
#include "wd_common.h"
#include "tag_gridworld_rewards.h"

namespace {

struct Gw5ResetEntry {  // same layout as wd_reset_entry in wd_core.hip
  wd_global_u32 *data;
  const wd_global_u32 *ref;
  int row_elems;
  int pad_;
};

constexpr int GW5_N = 5, GW5_F = 21, GW5_EPB = 12;
constexpr int GW5_ROW = GW5_N * GW5_F;        // 105 floats: one replica's observation rows
constexpr int GW5_IMG = GW5_EPB * GW5_ROW;    // 1260 floats: the block's observation image
constexpr int GW5_MAX_COORD = 63;             // cells per axis - 1 the quotient table (and the packed cell) holds
constexpr int GW5_IN_STRIDE = 24;             // floats per row of W0 (21 inputs + padding)
constexpr int GW5_ACTIONS = 5;

// floats of one policy's packed weights, rounded up to whole 16-byte vectors (the second policy starts aligned)
__host__ __device__ constexpr int gw5_policy_floats(int H) {
  return (H * GW5_IN_STRIDE + H + H * H + H + GW5_ACTIONS * H + GW5_ACTIONS + 3) & ~3;
}

// running float32 sums of the action probabilities of one agent: w = its policy's packed weights (LDS), x = its
// observation row (LDS, 21 floats)
template <int H>
__device__ __forceinline__ void gw5_policy_cum(const float *w, const float *x, float (&cumv)[8]) {
  const float *W0 = w, *b0 = W0 + H * GW5_IN_STRIDE, *W1 = b0 + H, *b1 = W1 + H * H, *Wp = b1 + H, *bp = Wp + GW5_ACTIONS * H;
  float in[GW5_F];
#pragma unroll
  for (int j = 0; j < GW5_F; ++j) in[j] = x[j];
  float h1[H], h2[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    float acc = b0[i];
#pragma unroll
    for (int j = 0; j < 20; j += 4) {
      const float4 wr = *(const float4 *)(W0 + i * GW5_IN_STRIDE + j);
      acc = fmaf(wr.x, in[j], acc); acc = fmaf(wr.y, in[j + 1], acc);
      acc = fmaf(wr.z, in[j + 2], acc); acc = fmaf(wr.w, in[j + 3], acc);
    }
    acc = fmaf(W0[i * GW5_IN_STRIDE + 20], in[20], acc);
    h1[i] = fmaxf(acc, 0.0f);
  }
#pragma unroll
  for (int i = 0; i < H; ++i) {
    float acc = b1[i];
#pragma unroll
    for (int j = 0; j < H; j += 4) {
      const float4 wr = *(const float4 *)(W1 + i * H + j);
      acc = fmaf(wr.x, h1[j], acc); acc = fmaf(wr.y, h1[j + 1], acc);
      acc = fmaf(wr.z, h1[j + 2], acc); acc = fmaf(wr.w, h1[j + 3], acc);
    }
    h2[i] = fmaxf(acc, 0.0f);
  }
  float logit[GW5_ACTIONS], m = -__builtin_inff();
#pragma unroll
  for (int a = 0; a < GW5_ACTIONS; ++a) {
    float acc = bp[a];
#pragma unroll
    for (int j = 0; j < H; j += 4) {
      const float4 wr = *(const float4 *)(Wp + a * H + j);
      acc = fmaf(wr.x, h2[j], acc); acc = fmaf(wr.y, h2[j + 1], acc);
      acc = fmaf(wr.z, h2[j + 2], acc); acc = fmaf(wr.w, h2[j + 3], acc);
    }
    logit[a] = acc;
    m = fmaxf(m, acc);
  }
  float e[GW5_ACTIONS], sum = 0.0f;
#pragma unroll
  for (int a = 0; a < GW5_ACTIONS; ++a) {
    e[a] = expf(logit[a] - m);
    sum += e[a];
  }
  float cum = 0.0f;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    if (a < GW5_ACTIONS) {
      const float p = e[a] / sum;
      cum = (a == 0) ? p : cum + p;
    }
    cumv[a] = cum;
  }
}


// H = 0: fixed probabilities (`probs`); H = 32 / 64: the live policies
template <int H>
__device__ __forceinline__ void gw5_rollout(
    int *states_x_arr, int *states_y_arr, int *actions_arr, int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner, double step_cost_for_tagger,
    int use_full_observation, int world_boundary, int *env_timestep_arr, int episode_length, int n_agents, int n_envs,
    uint32_t *rng_state, const float *probs, int n_actions, const void *reset_table, int n_reset_arrays,
    int stream_tag, int ticks, float *obs_batch, int *action_batch, float *reward_batch, int *done_batch,
    int reset_cache_dwords, const int *action_table, const float *policy_tagger, const float *policy_runner,
    float *gw5_smem) {
  const int CD = reset_cache_dwords;                        // dwords per replica in the restore cache
  float *const s_obs = gw5_smem;                            // [12][5][21] the block's observation image (16-byte aligned)
  uint32_t *const s_cache = (uint32_t *)(s_obs + GW5_IMG);  // [12][CD] the rows finished replicas are restored from
  float *const s_div = (float *)(s_cache + GW5_EPB * CD);   // [64] c / L
  float *const s_tn = s_div + GW5_MAX_COORD + 1;            // [episode_length + 1] t / episode_length
  // the two policies' packed weights behind the tables, on a 16-byte boundary (host: the same arithmetic)
  float *const s_pol = s_tn + ((episode_length + 1 + 3) & ~3);  // [2][gw5_policy_floats(H)]: tagger, runner
  GW_REWARD_TABLE(wall_hit_penalty, tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger);
  const int lane = threadIdx.x;                             // (blocks are one wavefront)
  const int el = lane / GW5_N, ag = lane - el * GW5_N;      // local replica (12 = none), agent
  const Gw5ResetEntry *const table = (const Gw5ResetEntry *)reset_table;
  const uint32_t k0 = rng_state[0], k1 = rng_state[1];
  int act_dx[5], act_dy[5];  // the action table, once per launch (scalar registers)
#pragma unroll
  for (int i = 0; i < 5; ++i) { act_dx[i] = action_table[2 * i]; act_dy[i] = action_table[2 * i + 1]; }
  {  // every quotient a tick can need, computed with the division the reference's expression compiles to
    const float L = (float)world_boundary;
    if (lane <= world_boundary) s_div[lane] = (float)lane / L;
    for (int q = lane; q <= episode_length; q += 64) s_tn[q] = (float)q / (float)episode_length;
    if constexpr (H > 0) {
      for (int q = lane; q < gw5_policy_floats(H); q += 64) {
        s_pol[q] = policy_tagger[q];
        s_pol[gw5_policy_floats(H) + q] = policy_runner[q];
      }
    }
  }

  // Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8), and a block's action / reward / done rows
  // are 240 / 240 / 48 bytes: with replicas in blockIdx order every such row shares its first and last cache line with
  // a block on ANOTHER XCD, whose L2 cannot merge the two halves.  Give each XCD a contiguous range of replica groups
  // instead (a bijection of [0, gridDim.x) for any grid size); what a replica computes does not depend on its block.
  // Measured (profiles/r06_ab_gridworld_block_order.txt): 20 000 replicas x 50 ticks 145.0 -> 127.8 us, 100 000 x 50
  // 728 -> 688 us, 1 000 x 100 (84 blocks) equal within the run-to-run spread.
#ifndef WD_GW5_BLOCK_ORDER_PLAIN
  const int xcd = blockIdx.x & 7, nq = gridDim.x >> 3, nr = gridDim.x & 7;
  const int group0 = xcd * nq + min(xcd, nr) + (blockIdx.x >> 3);
#else
  const int group0 = blockIdx.x;
#endif
  for (int env0 = group0 * GW5_EPB; env0 < n_envs; env0 += gridDim.x * GW5_EPB) {
    const int env = env0 + el;
    const bool active = (el < GW5_EPB) && (env < n_envs);
    const int idx = env * GW5_N + ag;
    const int envs_here = min(GW5_EPB, n_envs - env0);
    const int n_out = envs_here * GW5_ROW;
    float *const obs_blk = obs_arr + (long)env0 * GW5_ROW;
    int x = 0, y = 0, t = 0;
    uint32_t epoch0 = 0u;
    wd_u4 blk = wd_u4{0u, 0u, 0u, 0u};  // the Philox block of four consecutive ticks (wd_tick_draw)
    uint32_t blk_quad = 0xffffffffu;
    float cumv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cumv[i] = 0.0f;
    if (active) {
      x = states_x_arr[idx];
      y = states_y_arr[idx];
      t = env_timestep_arr[env];
      epoch0 = rng_state[WD_RNG_HEADER + idx];
      if constexpr (H == 0) {
        const float *row = probs + (long)idx * n_actions;
        float cum = 0.0f;  // the running float32 sums of the (fixed) probabilities, once per launch
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < n_actions) cum = (i == 0) ? row[0] : cum + row[i];
          cumv[i] = cum;
        }
      }
    }
    for (int q = lane; q < n_out; q += 64) s_obs[q] = obs_blk[q];  // the observation the first action is sampled on
    // the rows finished replicas are restored from: per registered array one flat, coalesced copy of the block's rows
    int off_x = 0, off_y = 0, off_obs = 0;
    {
      int off = 0;
      for (int r = 0; r < n_reset_arrays; ++r) {
        const Gw5ResetEntry ent = table[r];
        const int re = ent.row_elems;
        if ((size_t)ent.data == (size_t)states_x_arr) off_x = off;
        if ((size_t)ent.data == (size_t)states_y_arr) off_y = off;
        if ((size_t)ent.data == (size_t)obs_arr) off_obs = off;
        const wd_global_u32 *const src = ent.ref + (long)env0 * re;
        const float inv_re = 1.0f / (float)re;
        for (int q = lane; q < envs_here * re; q += 64) {
          const int e = (int)(((float)q + 0.5f) * inv_re);  // q / re (exact for these sizes)
          s_cache[e * CD + off + (q - e * re)] = src[q];
        }
        off += re;
      }
    }
    __syncthreads();
    // every value loaded above is consumed HERE: the wait for a load whose first use is inside the tick loop is placed
    // inside the loop and -- the memory counter returns in order -- waits for the previous tick's stores on every trip
    asm volatile("" : "+v"(x), "+v"(y), "+v"(t), "+v"(epoch0));
    if constexpr (H == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(cumv[i]));
    }
    const float *const my_policy = s_pol + ((ag == GW5_N - 1) ? gw5_policy_floats(H) : 0);  // runner : tagger
    float last_reward = 0.0f;
    int last_action = 0, last_done = 0;
    float *const rep = s_obs + min(el, GW5_EPB - 1) * GW5_ROW;  // this lane's replica's five rows
    const uint32_t *const my_cache = s_cache + min(el, GW5_EPB - 1) * CD;
    const int runner_lane = min(el * GW5_N + GW5_N - 1, 63);

    for (int k = 0; k < ticks; ++k) {
      // the previous tick's image writes / restores, before this tick's record reads (the fixed-policy entry's statement of
      // what holds anyway -- LDS operations of one wavefront execute in issue order; it compiles to nothing there)
      if constexpr (H == 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // ---- record the observation of this tick (flat, coalesced; none of the record stores is tracked).  The usual
      // case -- the block's slice of the row is a whole number of 16-byte vectors on a 16-byte boundary -- reads its
      // (up to) five vectors per lane with all LDS reads in flight, then stores them; a loop of read / wait / store
      // is five LDS round trips one after the other
      float *const brow = obs_batch + ((long)k * n_envs + env0) * GW5_ROW;
      if ((((size_t)brow & 15) | (size_t)(n_out & 3)) == 0) {  // block-uniform
        const int nvec = n_out >> 2;  // 315 for a full block
        const float4 *const img4 = (const float4 *)s_obs;
        float4 v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = img4[min(lane + 64 * i, GW5_IMG / 4 - 1)];  // (clamped into the image)
#pragma unroll
        for (int i = 0; i < 5; ++i)
          if (lane + 64 * i < nvec) wd_store_untracked((float4 *)brow + lane + 64 * i, v[i]);
      } else {
        for (int q = lane; q < n_out; q += 64) wd_store_untracked(brow + q, s_obs[q]);
      }
      bool hit = false;
      int a = 0;
      if (active) {
        const int n_act = (H > 0) ? GW5_ACTIONS : n_actions;
        // ---- sample (random.cu:51-85), the draw of tick k of T single-tick launches
        const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)idx, epoch0 + (uint32_t)k, (uint32_t)stream_tag, k0, k1,
                                                        blk, blk_quad));
        // the LIVE policy on this tick's observation row (the image still holds what was recorded above).  AFTER the
        // draw: placed in front of it (the Philox refill is a branch) the scheduler hoists the network's ~1 500 LDS
        // operand reads over the whole block and the allocator spills 5 483 registers at H = 64 (13 x the tick time)
        if constexpr (H > 0) gw5_policy_cum<H>(my_policy, rep + ag * GW5_F, cumv);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cnt += (i < n_act && cumv[i] < u) ? 1 : 0;
        a = min(cnt, n_act - 1);
        wd_store_untracked(action_batch + ((long)k * n_envs * GW5_N + idx), a);
        // ---- movement :152-173
        int ddx = act_dx[0], ddy = act_dy[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) { ddx = (a == i) ? act_dx[i] : ddx; ddy = (a == i) ? act_dy[i] : ddy; }
        const int ux = x + ddx, uy = y + ddy;
        const int cx = min(max(ux, 0), world_boundary), cy = min(max(uy, 0), world_boundary);
        hit = (ux != cx) || (uy != cy);  // :163-170
        x = cx;
        y = cy;
        t += 1;  // :295
      }
      // ---- tag check :175-178: does a tagger stand on the runner's cell?  (all 64 lanes take part in the exchange)
      const int cell = x | (y << 8);
      const int runner_cell = __shfl(cell, runner_lane);
      const unsigned long long on_runner = __ballot(active && (ag < GW5_N - 1) && (cell == runner_cell));
      const bool tag = ((unsigned)(on_runner >> (min(el, GW5_EPB - 1) * GW5_N)) & 0xfu) != 0u;
      const bool fin = active && ((t >= episode_length) || tag);  // :314
      if (active) {
        if (ag == 0) wd_store_untracked(done_batch + ((long)k * n_envs + env), fin ? 1 : 0);
        last_done = fin ? 1 : 0;
        last_reward = GW_REWARD(ag < GW5_N - 1, tag, hit);
        wd_store_untracked(reward_batch + ((long)k * n_envs * GW5_N + idx), last_reward);
        last_action = a;
        // ---- the image: only the positions and the time change from tick to tick (the type and "is me" columns are
        // constants that arrived with the image and return with a restore); this lane's agent is column ag (x) and
        // 5 + ag (y) of its replica's five rows, the time is column 20 of its own row
        const float fx = s_div[x], fy = s_div[y], tnorm = s_tn[t];
#pragma unroll
        for (int i = 0; i < GW5_N; ++i) {
          rep[i * GW5_F + ag] = fx;
          rep[i * GW5_F + GW5_N + ag] = fy;
        }
        rep[ag * GW5_F + 4 * GW5_N] = tnorm;
      }
      // ---- restore finished replicas: register and LDS copies only (see the header)
      unsigned long long fm = __ballot(fin);  // wave-uniform
      if (fm == 0ull) continue;
      if (fin) {
        x = (int)my_cache[off_x + ag];
        y = (int)my_cache[off_y + ag];
        t = 0;
      }
      while (fm != 0ull) {
        const int e = ((__ffsll((long long)fm) - 1) * 13) >> 6;  // lane / 5 for lanes < 64
        fm &= ~(0x1full << (e * GW5_N));
        for (int q = lane; q < GW5_ROW; q += 64) s_obs[e * GW5_ROW + q] = __uint_as_float(s_cache[e * CD + off_obs + q]);
      }
    }
    // ---- what the launch leaves in the per-tick arrays: the state after its last tick
    if (active) {
      states_x_arr[idx] = x;
      states_y_arr[idx] = y;
      rewards_arr[idx] = last_reward;
      actions_arr[idx] = last_action;
      rng_state[WD_RNG_HEADER + idx] = epoch0 + (uint32_t)ticks;
      if (ag == 0) {
        done_arr[env] = last_done;
        env_timestep_arr[env] = t;
      }
    }
    for (int q = lane; q < n_out; q += 64) obs_blk[q] = s_obs[q];
    __syncthreads();  // (the next trip overwrites the image and the cache)
  }
}

}  // namespace

#define GW5_PARAMS                                                                                                    \
  int *states_x_arr, int *states_y_arr, int *actions_arr, int *done_arr, float *rewards_arr, float *obs_arr,          \
      double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner,                           \
      double step_cost_for_tagger, int use_full_observation, int world_boundary, int *env_timestep_arr,               \
      int episode_length, int n_agents, int n_envs, uint32_t *rng_state, const float *probs, int n_actions,           \
      const void *reset_table, int n_reset_arrays, int stream_tag, int ticks, float *obs_batch, int *action_batch,    \
      float *reward_batch, int *done_batch, int reset_cache_dwords, const int *action_table
#define GW5_ARGS                                                                                                      \
  states_x_arr, states_y_arr, actions_arr, done_arr, rewards_arr, obs_arr, wall_hit_penalty, tag_reward_for_tagger,   \
      tag_penalty_for_runner, step_cost_for_tagger, use_full_observation, world_boundary, env_timestep_arr,           \
      episode_length, n_agents, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks,  \
      obs_batch, action_batch, reward_batch, done_batch, reset_cache_dwords, action_table

extern "C" __global__ void __launch_bounds__(64) HipTagGridWorldRollout_N5(GW5_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float gw5_smem[];
  gw5_rollout<0>(GW5_ARGS, nullptr, nullptr, gw5_smem);
}
// the same arguments + the packed weights of the two policies; dynamic LDS = the fixed-policy kernel's, its time
// table rounded up to 16 bytes, + 2 * gw5_policy_floats(H) floats (envs/tag_gridworld.py::tick_launch)
#define GW5_POLICY_ENTRY(HH)                                                                                          \
  extern "C" __global__ void __launch_bounds__(64) HipTagGridWorldRollout_N5_H##HH(                                   \
      GW5_PARAMS, const float *policy_tagger, const float *policy_runner) {                                           \
    extern __shared__ __attribute__((aligned(16))) float gw5_smem[];                                                  \
    gw5_rollout<HH>(GW5_ARGS, policy_tagger, policy_runner, gw5_smem);                                                \
  }
GW5_POLICY_ENTRY(32)
GW5_POLICY_ENTRY(64)
