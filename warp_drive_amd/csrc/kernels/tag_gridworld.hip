// tag_gridworld.hip -- TagGridWorld step for gfx950 (integer path, bit-exact).
//
// Semantics follow the reference CPU step, example_envs/tag_gridworld/
// tag_gridworld.py: update_state :152-192, generate_observation :194-275,
// step/done :291-317; argument order follows the reference kernel
// (tag_gridworld_step_pycuda.cu:112-127) with one trailing `n_envs`.
//
// MI355X mapping.  A replica has only N = num_taggers + 1 agents (5 in every
// BASELINE config), so block-per-env would light 5 of 64 lanes.  Instead a block
// packs epb = blockDim.x / N replicas (12 per wavefront at N = 5): thread t serves
// agent t % N of local replica t / N.  Positions are staged in LDS; each thread builds
// its agent's observation row in an LDS image of the block's slice, which is contiguous
// in HBM ([E, N, F] row-major) and leaves with flat, fully coalesced stores.  With the
// reference geometry (block=(N,1,1), grid=(E,1)) epb is simply 1.
#include "wd_common.h"
#include "tag_gridworld_rewards.h"

// action index -> (dx, dy); uploaded by the host like the reference
// (kIndexToActionArr, tag_gridworld_step_pycuda.cu:6; env_cpu_gpu_consistency_checker.py:256-264)
// and pre-initialised to TagGridWorld.step_actions (tag_gridworld.py:104).
__constant__ int kIndexToActionArr[10] = {0, 0, 1, 0, -1, 0, 0, 1, 0, -1};

#define WD_GW_IMAGE_MAX_BYTES 60000  // dynamic LDS budget of one block (default limit 64 KB)

namespace {

struct GwResetEntry {  // same layout as wd_reset_entry in wd_core.hip
  wd_global_u32 *data;
  const wd_global_u32 *ref;
  int row_elems;
  int pad_;
};

struct GwFuse {
  uint32_t *rng_state;       // Philox epoch counters (WD_RNG_HEADER + one word per agent row)
  const float *probs;        // [E, N, n_actions] policy output
  int n_actions;
  const GwResetEntry *reset_table;
  int n_reset_arrays;
  int stream_tag;
};

// one agent's observation row, tag_gridworld.py:194-275
__device__ __forceinline__ void gw_write_row(float *row, const float *fx, const float *fy, int N, int ag, int best,
                                             float tnorm, int use_full_observation) {
  if (use_full_observation) {
    for (int j = 0; j < N; ++j) {
      row[j] = fx[j];
      row[N + j] = fy[j];
      row[2 * N + j] = (j == N - 1) ? 1.0f : 0.0f;
      row[3 * N + j] = (j == ag) ? 1.0f : 0.0f;
    }
    row[4 * N] = tnorm;
  } else {
    const int other = (ag < N - 1) ? N - 1 : best;
    row[0] = fx[ag];
    row[1] = fy[ag];
    row[2] = fx[other];
    row[3] = fy[other];
    row[4] = (ag == N - 1) ? 1.0f : 0.0f;
    row[5] = tnorm;
  }
}

// One trip = epb packed replicas.  Thread t serves agent t % N of local replica t / N and writes that
// agent's whole observation row into an LDS image of the block's [epb*N, F] slice; the image is then
// copied out flat (consecutive lanes -> consecutive floats), so the obs stores are fully coalesced
// and no index arithmetic (the reference decodes (agent, feature) from a flat index with integer
// divisions, tag_gridworld_step_pycuda.cu:29-51) is left in the copy loop.
template <bool FUSED>
__device__ __forceinline__ void gw_step_impl(
    int *states_x_arr, int *states_y_arr, int *actions_arr,
    int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner,
    double step_cost_for_tagger, int use_full_observation, int world_boundary,
    int *env_timestep_arr, int episode_length, int n_agents, int n_envs, const GwFuse &fz,
    int *s_mem) {
  const int N = n_agents;
  const int epb = max(1, (int)blockDim.x / N);  // replicas per block
  const int A = epb * N;
  const int F = use_full_observation ? 4 * N + 1 : 6;
  int *s_x = s_mem;                             // [A] positions after the move
  int *s_y = s_x + A;                           // [A]
  float *s_fx = (float *)(s_y + A);             // [A] x / L  (float32 division, :208-214)
  float *s_fy = s_fx + A;                       // [A]
  int *s_t = (int *)(s_fy + A);                 // [epb] timestep after increment
  int *s_done = s_t + epb;                      // [epb] replica finished on this tick
  // [A][F] image of the block's observation slice, 16-byte aligned (host: lds_bytes).  Pointer arithmetic only: a
  // round trip through an integer loses the LDS address space and every access to the image becomes a FLAT one
  // (whose wait also waits for all global stores in flight: the record copy then ran one store at a time)
  float *s_obs = (float *)(s_done + epb + 2 + ((4 - ((4 * A + 2 * epb + 2) & 3)) & 3));  // (+ 2: the rollout's vote flags)
  // very wide rows (N ~ 64 with full observations) do not fit an LDS image: those blocks write their
  // rows straight to HBM.  The host sizes the dynamic LDS with the same rule (lds_bytes()).
  const bool image = ((((size_t)4 * ((size_t)4 * A + 2 * epb + 2)) + 15) & ~(size_t)15) + (size_t)4 * A * F <= WD_GW_IMAGE_MAX_BYTES;
  const int tid = threadIdx.x, T_ = blockDim.x;
  const int el = tid / N, ag = tid - el * N;
  const float L = (float)world_boundary;
  GW_REWARD_TABLE(wall_hit_penalty, tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger);

  for (int env0 = blockIdx.x * epb; env0 < n_envs; env0 += gridDim.x * epb) {
    const int env = env0 + el;
    const bool active = (el < epb) && (env < n_envs);
    const int idx = env * N + ag;
    const int li = el * N + ag;
    bool hit = false;
    if (active) {
      int a;
      if (FUSED) {
        // ---- sample the action (random.cu:51-85): inverse CDF on a running float32 sum
        if (ag == 0) done_arr[env] = 0;  // a replica that finished (and was reset) last tick
        const uint32_t epoch = fz.rng_state[WD_RNG_HEADER + idx];
        fz.rng_state[WD_RNG_HEADER + idx] = epoch + 1u;
        wd_u4 blk;
        uint32_t blk_quad = 0xffffffffu;
        const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)idx, epoch, (uint32_t)fz.stream_tag, fz.rng_state[0],
                                                        fz.rng_state[1], blk, blk_quad));
        const float *row = fz.probs + (long)idx * fz.n_actions;
        float cum = 0.0f;
        int cnt = 0;
        for (int i = 0; i < fz.n_actions; ++i) {
          cum = (i == 0) ? row[0] : cum + row[i];
          cnt += (cum < u) ? 1 : 0;
        }
        a = min(cnt, fz.n_actions - 1);
        actions_arr[idx] = a;
      } else {
        a = actions_arr[idx];
      }
      // ---- movement :152-173
      const int ux = states_x_arr[idx] + kIndexToActionArr[2 * a];
      const int uy = states_y_arr[idx] + kIndexToActionArr[2 * a + 1];
      const int cx = min(max(ux, 0), world_boundary);
      const int cy = min(max(uy, 0), world_boundary);
      hit = (ux != cx) || (uy != cy);  // :163-170
      states_x_arr[idx] = cx;
      states_y_arr[idx] = cy;
      s_x[li] = cx;
      s_y[li] = cy;
      s_fx[li] = (float)cx / L;
      s_fy[li] = (float)cy / L;
      if (ag == 0) {
        const int t = env_timestep_arr[env] + 1;  // :295
        env_timestep_arr[env] = t;
        s_t[el] = t;
      }
    }
    __syncthreads();
    if (active) {
      // ---- tag check :175-178 and closest tagger :246-261 (first argmin); every agent of the replica
      // evaluates the N-1 taggers itself (LDS broadcasts) instead of waiting for one thread
      const int *x = s_x + el * N, *y = s_y + el * N;
      const int rx = x[N - 1], ry = y[N - 1];
      int tag = 0, best = 0, bd = 0x7fffffff;
      for (int j = 0; j < N - 1; ++j) {
        const int dx = x[j] - rx, dy = y[j] - ry;
        const int d = dx * dx + dy * dy;
        tag |= (d == 0);
        if (d < bd) { bd = d; best = j; }
      }
      const int t = s_t[el];
      if (ag == 0) {
        const bool fin = (t >= episode_length) || tag;  // :314
        if (fin) done_arr[env] = 1;
        s_done[el] = fin ? 1 : 0;
      }
      // ---- rewards :180-187
      rewards_arr[idx] = GW_REWARD(ag < N - 1, tag, hit);
      // ---- this agent's observation row :194-275
      const float tnorm = (float)t / (float)episode_length;
      // (two calls, so the LDS image and the HBM row keep their own address spaces -- a pointer
      // that may be either becomes a flat access)
      if (image) gw_write_row(s_obs + (size_t)li * F, s_fx + el * N, s_fy + el * N, N, ag, best, tnorm, use_full_observation);
      else gw_write_row(obs_arr + (long)idx * F, s_fx + el * N, s_fy + el * N, N, ag, best, tnorm, use_full_observation);
    }
    __syncthreads();
    // ---- flat, coalesced copy of the observation image
    const int envs_here = min(epb, n_envs - env0);
    const int n_out = envs_here * N * F;
    float *dst = obs_arr + (long)env0 * N * F;
    if (image)
      for (int q = tid; q < n_out; q += T_) dst[q] = s_obs[q];
    // ---- fused tick: restore finished replicas in place (reset.cu:9-75 for every registered array;
    // `_done_` stays 1 for the trainer, the next tick clears it).  All writes of this block to these
    // rows are ordered before the copies by the barrier.
    if (FUSED) {
      __syncthreads();
      for (int e = 0; e < envs_here; ++e) {
        if (s_done[e] == 0) continue;  // block-uniform
        for (int r = 0; r < fz.n_reset_arrays; ++r) {
          const GwResetEntry ent = fz.reset_table[r];
          const long base = (long)(env0 + e) * ent.row_elems;
          for (int i = tid; i < ent.row_elems; i += T_) ent.data[base + i] = ent.ref[base + i];
        }
        if (tid == 0) env_timestep_arr[env0 + e] = 0;
      }
    }
    __syncthreads();
  }
}

// ---- T ticks per launch with every tick recorded (fixed-policy rollout).  One tick at 1000 replicas is
// 0.7 MB and ~4 dependent global round trips: a launch per tick is bound by the kernel boundary (8 us).
// Here a block keeps its replicas' positions in registers and their observation image in LDS over
// `ticks` ticks; tick k writes ROW k of the env-level batch tensors obs [T, E, N, F], actions [T, E, N],
// rewards [T, E, N], done [T, E] -- the observation the action was sampled on, the action (same Philox
// draw as tick k of T single-tick launches), the reward and the done flag (trainer_base.py:392-426 records
// exactly these per tick) -- and the per-tick arrays receive the state after the last tick.  A replica that
// finishes is restored in place from the registered `*_at_reset` copies (reset.cu:9-75) and goes on.
// Needs the LDS observation image (rows of up to ~60 floats).  n_actions <= 8.
__device__ __forceinline__ void gw_rollout_impl(
    int *states_x_arr, int *states_y_arr, int *actions_arr, int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner, double step_cost_for_tagger,
    int use_full_observation, int world_boundary, int *env_timestep_arr, int episode_length, int n_agents,
    int n_envs, const GwFuse &fz, int ticks, float *obs_batch, int *action_batch, float *reward_batch,
    int *done_batch, int reset_cache_dwords, int *s_mem) {
  const int N = n_agents;
  const int epb = max(1, (int)blockDim.x / N);
  const int A = epb * N;
  const int F = use_full_observation ? 4 * N + 1 : 6;
  int *s_x = s_mem;
  int *s_y = s_x + A;
  float *s_fx = (float *)(s_y + A);
  float *s_fy = s_fx + A;
  int *s_t = (int *)(s_fy + A);
  int *s_done = s_t + epb;
  int *s_flag = s_done + epb;                    // [2] some replica of the block finished on an even / odd tick
  float *s_obs = (float *)(s_flag + 2 + ((4 - ((4 * A + 2 * epb + 2) & 3)) & 3));  // 16-byte aligned (the record path reads it as float4); pointer arithmetic only, see gw_step_impl
  // The rows finished replicas are restored from, copied into LDS once per trip (`reset_cache_dwords` = the sum of
  // the registered arrays' row lengths, 0 = no room): with 12 replicas per wavefront some replica finishes on every
  // third tick, and the restore used to LOAD the registered rows from HBM inside the tick loop -- ~10 loads per tick
  // and wavefront, each waiting for every store issued before it (the memory counters return in order).
  uint32_t *const s_cache = (uint32_t *)(s_obs + (size_t)A * F);   // [epb][reset_cache_dwords]
  const int tid = threadIdx.x, T_ = blockDim.x;
  const int el = tid / N, ag = tid - el * N;
  const float L = (float)world_boundary;
  GW_REWARD_TABLE(wall_hit_penalty, tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger);
  const uint32_t k0 = fz.rng_state[0], k1 = fz.rng_state[1];
  // the action table (host-uploadable __constant__ memory) once per launch, in scalar registers: indexed by the
  // sampled action inside the tick loop it is a global load that waits for every store of the tick (the memory
  // counters return in order)
  int act_dx[5], act_dy[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { act_dx[i] = kIndexToActionArr[2 * i]; act_dy[i] = kIndexToActionArr[2 * i + 1]; }

  for (int env0 = blockIdx.x * epb; env0 < n_envs; env0 += gridDim.x * epb) {
    // (per trip: a flag left set by the previous trip's last tick must not send this trip's tick 0 into the restore
    // path; the barrier behind the loads below publishes the clear)
    if (tid < 2) s_flag[tid] = 0;
    const int env = env0 + el;
    const bool active = (el < epb) && (env < n_envs);
    const int idx = env * N + ag;
    const int li = el * N + ag;
    const int envs_here = min(epb, n_envs - env0);
    const int n_out = envs_here * N * F;
    float *const obs_blk = obs_arr + (long)env0 * N * F;
    int x = 0, y = 0;
    uint32_t epoch0 = 0u;
    wd_u4 blk = wd_u4{0u, 0u, 0u, 0u};   // the Philox block of four consecutive ticks (wd_tick_draw)
    uint32_t blk_quad = 0xffffffffu;
    float cumv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cumv[i] = 0.0f;
    if (active) {
      x = states_x_arr[idx];
      y = states_y_arr[idx];
      epoch0 = fz.rng_state[WD_RNG_HEADER + idx];
      const float *row = fz.probs + (long)idx * fz.n_actions;
      float cum = 0.0f;  // the running float32 sums of the (fixed) probabilities, once per launch
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < fz.n_actions) cum = (i == 0) ? row[0] : cum + row[i];
        cumv[i] = cum;
      }
      if (ag == 0) s_t[el] = env_timestep_arr[env];
    }
    for (int q = tid; q < n_out; q += T_) s_obs[q] = obs_blk[q];  // the observation the first action is sampled on
    if (reset_cache_dwords > 0) {
      // (per array ONE flat, coalesced copy of the block's rows -- they are contiguous in the `_at_reset` copy -- so
      // that all its loads are in flight together; a loop nest over replicas and arrays is a chain of round trips)
      int off = 0;
      for (int r = 0; r < fz.n_reset_arrays; ++r) {
        const GwResetEntry ent = fz.reset_table[r];
        const int re = ent.row_elems;
        const wd_global_u32 *const src = ent.ref + (long)env0 * re;
        const float inv_re = 1.0f / (float)re;
        for (int q = tid; q < envs_here * re; q += T_) {
          const int e = (int)(((float)q + 0.5f) * inv_re);  // q / re (exact for these sizes)
          s_cache[e * reset_cache_dwords + off + (q - e * re)] = src[q];
        }
        off += re;
      }
    }
    __syncthreads();
    // Every value loaded above is consumed HERE, before the tick loop: the wait for a load whose first use is inside
    // the loop is placed inside the loop, where it is executed on every tick and -- the memory counter returns in
    // order -- also waits for every store of the previous tick (~1 us per tick at 1000 replicas).
    asm volatile("" : "+v"(x), "+v"(y), "+v"(epoch0));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(cumv[i]));
    // (what the per-tick arrays receive after the loop: a store the compiler tracks INSIDE the loop costs a wait for
    // all stores on every trip, executed or not -- its operand registers are protected until the counter drains)
    float last_reward = 0.0f;
    int last_action = 0, last_done = 0;
    for (int k = 0; k < ticks; ++k) {
      // ---- record the observation of this tick (flat, coalesced)
      float *const brow = obs_batch + ((long)k * n_envs + env0) * N * F;
      if ((((size_t)brow & 15) | (size_t)(n_out & 3)) == 0) {  // block-uniform: 16-byte vectors (whenever the block's slice is a multiple of 16 bytes; the image is 16-byte aligned)
        for (int q = tid; q < (n_out >> 2); q += T_) wd_store_untracked((float4 *)brow + q, ((const float4 *)s_obs)[q]);
      } else {
        for (int q = tid; q < n_out; q += T_) wd_store_untracked(brow + q, s_obs[q]);
      }
      float fx = 0.0f, fy = 0.0f;
      bool hit = false;
      int a = 0;
      bool fin_mine = false;
      if (active) {
        // ---- sample (random.cu:51-85), the draw of tick k of T single-tick launches
        const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)idx, epoch0 + (uint32_t)k, (uint32_t)fz.stream_tag, k0, k1,
                                                        blk, blk_quad));
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cnt += (i < fz.n_actions && cumv[i] < u) ? 1 : 0;
        a = min(cnt, fz.n_actions - 1);
        wd_store_untracked(action_batch + ((long)k * n_envs * N + idx), a);
        // ---- movement :152-173
        int ddx = act_dx[0], ddy = act_dy[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) { ddx = (a == i) ? act_dx[i] : ddx; ddy = (a == i) ? act_dy[i] : ddy; }
        const int ux = x + ddx, uy = y + ddy;
        const int cx = min(max(ux, 0), world_boundary), cy = min(max(uy, 0), world_boundary);
        hit = (ux != cx) || (uy != cy);
        x = cx;
        y = cy;
        s_x[li] = cx;
        s_y[li] = cy;
        fx = (float)cx / L;
        fy = (float)cy / L;
        if (!use_full_observation) {
          s_fx[li] = fx;
          s_fy[li] = fy;
        }
        if (ag == 0) s_t[el] += 1;  // :295
      }
      if (tid == 0) s_flag[(k + 1) & 1] = 0;  // (the next tick's flag: nobody reads or sets it before the next barrier)
      __syncthreads();  // positions and time steps are published; every lane is done reading the old image
      if (active) {
        const int *px = s_x + el * N, *py = s_y + el * N;
        const int rx = px[N - 1], ry = py[N - 1];
        int tag = 0, best = 0;
        if (use_full_observation) {  // block-uniform: the closest tagger is an input of the partial observation only
          for (int j = 0; j < N - 1; ++j) tag |= ((px[j] == rx) & (py[j] == ry)) ? 1 : 0;  // tag check :175-178
        } else {
          int bd = 0x7fffffff;
          for (int j = 0; j < N - 1; ++j) {  // tag check :175-178, closest tagger :246-261
            const int dx = px[j] - rx, dy = py[j] - ry;
            const int d = dx * dx + dy * dy;
            tag |= (d == 0);
            if (d < bd) { bd = d; best = j; }
          }
        }
        const int t = s_t[el];
        const bool fin = (t >= episode_length) || tag;  // :314
        fin_mine = fin;
        if (ag == 0) {
          s_done[el] = fin ? 1 : 0;
          wd_store_untracked(done_batch + ((long)k * n_envs + env), fin ? 1 : 0);
        }
        last_done = fin ? 1 : 0;
        last_reward = GW_REWARD(ag < N - 1, tag, hit);
        wd_store_untracked(reward_batch + ((long)k * n_envs * N + idx), last_reward);
        last_action = a;
        const float tnorm = (float)t / (float)episode_length;
        if (use_full_observation) {
          // Only 2 N + 1 of a row's 4 N + 1 values change from tick to tick (the positions and the time; the type
          // and "is me" columns are constants that arrived with the image and return with a reset), and this lane
          // has its agent's two position values in registers: it writes them into columns ag and N + ag of the
          // replica's N rows and the time into its own row -- 2 N + 1 LDS stores, no reads (every lane rebuilding
          // its whole row: 2 N reads + 4 N + 1 stores)
          float *const rep_rows = s_obs + (size_t)el * N * F;
          for (int i = 0; i < N; ++i) {
            rep_rows[i * F + ag] = fx;
            rep_rows[i * F + N + ag] = fy;
          }
          rep_rows[ag * F + 4 * N] = tnorm;
        } else {
          gw_write_row(s_obs + (size_t)li * F, s_fx + el * N, s_fy + el * N, N, ag, best, tnorm, use_full_observation);
        }
      }
      // the new image and the done flags are complete; did ANY replica of the block finish?  (block-uniform; an LDS
      // flag per tick parity -- __syncthreads_or reads the workgroup size from the dispatch packet: a global load
      // whose wait drains every store of the tick)
      if (fin_mine) s_flag[k & 1] = 1;
      __syncthreads();
      if (s_flag[k & 1] == 0) continue;
      // ---- restore finished replicas (block-uniform per replica): global arrays, and the copies this block holds
      for (int e = 0; e < envs_here; ++e) {
        if (s_done[e] == 0) continue;
        int off = 0;
        for (int r = 0; r < fz.n_reset_arrays; ++r) {
          const GwResetEntry ent = fz.reset_table[r];
          const long base = (long)(env0 + e) * ent.row_elems;
          const bool is_obs = ((size_t)ent.data == (size_t)obs_arr);
          // (two copies of the loop, not a select between an LDS and a global pointer: that becomes a FLAT access)
          if (reset_cache_dwords > 0) {  // block-uniform
            const uint32_t *const cached = s_cache + e * reset_cache_dwords + off;
            for (int i = tid; i < ent.row_elems; i += T_) {
              const uint32_t v = cached[i];
              ent.data[base + i] = v;
              if (is_obs) s_obs[(size_t)e * N * F + i] = __uint_as_float(v);
            }
            if (el == e && active) {
              if ((size_t)ent.data == (size_t)states_x_arr) x = (int)cached[ag];
              if ((size_t)ent.data == (size_t)states_y_arr) y = (int)cached[ag];
            }
          } else {
            for (int i = tid; i < ent.row_elems; i += T_) {
              const uint32_t v = ent.ref[base + i];
              ent.data[base + i] = v;
              if (is_obs) s_obs[(size_t)e * N * F + i] = __uint_as_float(v);
            }
            if (el == e && active) {
              if ((size_t)ent.data == (size_t)states_x_arr) x = (int)ent.ref[base + ag];
              if ((size_t)ent.data == (size_t)states_y_arr) y = (int)ent.ref[base + ag];
            }
          }
          off += ent.row_elems;
        }
        if (tid == 0) s_t[e] = 0;
      }
      __syncthreads();
    }
    // ---- what the launch leaves in the per-tick arrays: the state after its last tick
    if (active) {
      states_x_arr[idx] = x;
      states_y_arr[idx] = y;
      rewards_arr[idx] = last_reward;
      actions_arr[idx] = last_action;
      if (ag == 0) done_arr[env] = last_done;
      fz.rng_state[WD_RNG_HEADER + idx] = epoch0 + (uint32_t)ticks;
      if (ag == 0) env_timestep_arr[env] = s_t[el];
    }
    for (int q = tid; q < n_out; q += T_) obs_blk[q] = s_obs[q];
    __syncthreads();
  }
}

}  // namespace

extern "C" {

__global__ void HipTagGridWorldStep(
    int *states_x_arr, int *states_y_arr, int *actions_arr,
    int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner,
    double step_cost_for_tagger, int use_full_observation, int world_boundary,
    int *env_timestep_arr, int episode_length, int n_agents, int n_envs) {
  extern __shared__ __attribute__((aligned(16))) int gw_smem[];
  gw_step_impl<false>(states_x_arr, states_y_arr, actions_arr, done_arr, rewards_arr, obs_arr, wall_hit_penalty,
                      tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger, use_full_observation,
                      world_boundary, env_timestep_arr, episode_length, n_agents, n_envs, GwFuse{}, gw_smem);
}

// Fused rollout tick: sample the action + step + reset finished replicas in ONE launch (the reference
// needs the sampler launch, the step, and one reset launch per registered array, trainer_base.py:392-426).
__global__ void HipTagGridWorldTick(
    int *states_x_arr, int *states_y_arr, int *actions_arr,
    int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner,
    double step_cost_for_tagger, int use_full_observation, int world_boundary,
    int *env_timestep_arr, int episode_length, int n_agents, int n_envs, uint32_t *rng_state,
    const float *probs, int n_actions, const void *reset_table, int n_reset_arrays, int stream_tag) {
  extern __shared__ __attribute__((aligned(16))) int gw_smem[];
  GwFuse fz;
  fz.rng_state = rng_state; fz.probs = probs; fz.n_actions = n_actions;
  fz.reset_table = (const GwResetEntry *)reset_table; fz.n_reset_arrays = n_reset_arrays;
  fz.stream_tag = stream_tag;
  gw_step_impl<true>(states_x_arr, states_y_arr, actions_arr, done_arr, rewards_arr, obs_arr, wall_hit_penalty,
                     tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger, use_full_observation,
                     world_boundary, env_timestep_arr, episode_length, n_agents, n_envs, fz, gw_smem);
}

// T ticks per launch, every tick recorded in env-level batch tensors (see gw_rollout_impl)
__global__ void HipTagGridWorldRollout(
    int *states_x_arr, int *states_y_arr, int *actions_arr,
    int *done_arr, float *rewards_arr, float *obs_arr,
    double wall_hit_penalty, double tag_reward_for_tagger, double tag_penalty_for_runner,
    double step_cost_for_tagger, int use_full_observation, int world_boundary,
    int *env_timestep_arr, int episode_length, int n_agents, int n_envs, uint32_t *rng_state,
    const float *probs, int n_actions, const void *reset_table, int n_reset_arrays, int stream_tag,
    int ticks, float *obs_batch, int *action_batch, float *reward_batch, int *done_batch, int reset_cache_dwords) {
  extern __shared__ __attribute__((aligned(16))) int gw_smem[];
  GwFuse fz;
  fz.rng_state = rng_state; fz.probs = probs; fz.n_actions = n_actions;
  fz.reset_table = (const GwResetEntry *)reset_table; fz.n_reset_arrays = n_reset_arrays;
  fz.stream_tag = stream_tag;
  gw_rollout_impl(states_x_arr, states_y_arr, actions_arr, done_arr, rewards_arr, obs_arr, wall_hit_penalty,
                  tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger, use_full_observation,
                  world_boundary, env_timestep_arr, episode_length, n_agents, n_envs, fz, ticks, obs_batch,
                  action_batch, reward_batch, done_batch, reset_cache_dwords, gw_smem);
}

}  // extern "C"
