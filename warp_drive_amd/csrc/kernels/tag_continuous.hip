// tag_continuous.hip -- TagContinuous step for gfx950.
//
// Semantics: the reference CPU step, example_envs/tag_continuous/tag_continuous.py
//   update_state :339-401, compute_distance :403-420, k_nearest_neighbors :422-444,
//   generate_observation :446-610, compute_reward :612-678, done :880-883.
// Where the reference's own CUDA kernel (tag_continuous_step_pycuda.cu:351-520)
// disagrees with its CPU step the CPU wins: stable (distance, id) neighbour order,
// tag counts accumulated without races, no end-of-game bonus for a runner tagged out
// on the last tick.  Argument order is the reference kernel's (:351-385) plus a
// trailing n_envs; the two O(N^2) global scratch arrays it sorts in HBM
// (neighbor_distances, neighbor_ids_sorted_by_distance; :167-199) are accepted and
// never touched.
//
// MI355X mapping
//   * a block packs `epb` consecutive replicas so that epb*N fills whole wavefronts
//     (N = 105: 3 replicas = 315 of 320 lanes); thread t serves agent t % N of local
//     replica t / N.  The reference geometry block=(N,1,1), grid=(E,1) is epb = 1.
//   * phase 0  coalesced [E,N] loads, float32 kinematics (numpy-exact cos/sin),
//              coalesced stores, post-move state staged in LDS (positions in float32
//              for distances, normalised features as the reference computes them:
//              x,y in float64, speed/acc/dir in float32).
//   * phase 1  K nearest neighbours per agent entirely in registers: candidates are
//              streamed from LDS (wave-uniform address => broadcast, conflict-free) in
//              id order and inserted into a sorted register list; strict '<' makes
//              ties resolve to the lower id, i.e. heapq.nsmallest's stable order.
//   * phase 2  the packed replicas' observation block is contiguous in HBM; it is
//              produced by a block-strided gather from LDS so every store instruction
//              writes 64 consecutive floats (the reference writes one 284-byte-strided
//              row per thread).
//   * phase 3  rewards: each runner scans the taggers (first minimum wins), tag counts
//              go through LDS atomics, float adds are replayed in the CPU's order.
#include "wd_common.h"

namespace {

struct TcArgs {
  float *loc_x, *loc_y, *speed, *direction, *acceleration;
  const int *agent_types;
  float *edge_pen_arr;
  float edge_hit_penalty, grid_length;
  const float *acc_actions, *turn_actions;
  float max_speed;
  int K;
  const float *skill_levels;
  int runner_exits;
  int *sig_arr;
  int use_full_obs;
  float *obs;
  const int *actions;
  int *nearest_ids;
  float *rewards;
  const float *step_rewards;
  int *num_runners;
  float margin, tag_reward, tag_penalty, end_reward;
  int *done, *timestep;
  int N, T, E;
};

// LDS carve-up for `epb` packed replicas (all offsets multiples of 8 bytes).
struct TcLds {
  double *nx, *ny;               // [epb*N] normalised positions (float64, :454)
  float *x, *y;                  // [epb*N] positions after the move
  float *nsp, *nac, *ndir;       // [epb*N] normalised speed / acceleration / direction
  int *sig;                      // [epb*N] still_in_the_game BEFORE this tick's tagging
  int *tagcnt;                   // [epb*N] tags credited to a tagger this tick
  int *types;                    // [N]
  int *tagger_ids;               // [N] ids of taggers, ascending
  int *nbr;                      // [epb*N*K] neighbour ids, -1 = padding
  int *tstep, *nrun, *ntag;      // [epb], [epb], [1]
};

__device__ __forceinline__ TcLds tc_carve(unsigned char *base, int epb, int N, int K) {
  TcLds l;
  const int A = epb * N;
  unsigned char *p = base;
  l.nx = (double *)p; p += sizeof(double) * A;
  l.ny = (double *)p; p += sizeof(double) * A;
  l.x = (float *)p; p += 4 * A;
  l.y = (float *)p; p += 4 * A;
  l.nsp = (float *)p; p += 4 * A;
  l.nac = (float *)p; p += 4 * A;
  l.ndir = (float *)p; p += 4 * A;
  l.sig = (int *)p; p += 4 * A;
  l.tagcnt = (int *)p; p += 4 * A;
  l.types = (int *)p; p += 4 * N;
  l.tagger_ids = (int *)p; p += 4 * N;
  l.nbr = (int *)p; p += 4 * (size_t)A * K;
  l.tstep = (int *)p; p += 4 * epb;
  l.nrun = (int *)p; p += 4 * epb;
  l.ntag = (int *)p; p += 4;
  return l;
}

template <int KMAX>
__device__ __forceinline__ void tc_step_impl(const TcArgs &a, unsigned char *smem) {
  const int N = a.N, K = a.K;
  const int epb = max(1, (int)blockDim.x / N);
  const TcLds l = tc_carve(smem, epb, N, a.use_full_obs ? 0 : K);
  const int tid = threadIdx.x;
  const int el = tid / N, ag = tid - el * N;
  const float two_pi = 6.2831854820251465f;          // float32(2*pi), :356
  const float L = a.grid_length;
  const double diag = (double)L * 1.4142135623730951;  // float32 L * np.sqrt(2) -> f64, :146
  const float sp_div = a.max_speed + 1.0e-10f;         // float32 + float32(eps), :456
  const int F = a.use_full_obs ? 7 * (N - 1) + 1 : 7 * K + 1;

  // agent types and the ascending tagger list are replica-independent
  for (int i = tid; i < N; i += blockDim.x) l.types[i] = a.agent_types[i];
  if (tid == 0) *l.ntag = 0;
  __syncthreads();
  for (int i = tid; i < N; i += blockDim.x) {
    if (l.types[i] == 1) {
      int rank = 0;
      for (int j = 0; j < i; ++j) rank += (l.types[j] == 1);
      l.tagger_ids[rank] = i;
      atomicAdd(l.ntag, 1);
    }
  }
  __syncthreads();
  const int n_taggers = *l.ntag;

  for (int env0 = blockIdx.x * epb; env0 < a.E; env0 += gridDim.x * epb) {
    const int env = env0 + el;
    const bool active = (el < epb) && (env < a.E);
    const int gi = env * N + ag;  // index into [E, N] arrays
    const int li = el * N + ag;   // index into LDS arrays
    float edge_pen = 0.0f;

    // ------------------------------------------------------------ phase 0: move
    if (active) {
      const int sg = a.sig_arr[gi];
      const float s = (float)sg;
      const int a_acc = a.actions[2 * gi + 0], a_turn = a.actions[2 * gi + 1];
      const float d_acc = a.acc_actions[a_acc], d_turn = a.turn_actions[a_turn];
      const float dir = wd_np_remainderf(a.direction[gi] + d_turn, two_pi) * s;  // :355-357
      float acc = a.acceleration[gi] + d_acc;                                     // :359
      const float vmax = a.max_speed * a.skill_levels[ag];                        // :363
      float v = a.speed[gi] + acc;
      v = fminf(fmaxf(v, 0.0f), vmax) * s;                                        // :364-366
      acc = acc * (v > 0.0f ? 1.0f : 0.0f) * (v < vmax ? 1.0f : 0.0f);            // :367
      float sn, cs;
      wd_np_sincosf(dir, sn, cs);
      float px = a.loc_x[gi] + v * cs;                                            // :369-374
      float py = a.loc_y[gi] + v * sn;
      const bool crossed = !((px >= 0.0f) && (px <= L) && (py >= 0.0f) && (py <= L));
      px = fminf(fmaxf(px, 0.0f), L);                                             // :385-391
      py = fminf(fmaxf(py, 0.0f), L);
      edge_pen = a.edge_hit_penalty * (crossed ? 1.0f : 0.0f);                    // :394
      a.loc_x[gi] = px;
      a.loc_y[gi] = py;
      a.speed[gi] = v;
      a.direction[gi] = dir;
      a.acceleration[gi] = acc;
      a.edge_pen_arr[gi] = edge_pen;
      l.x[li] = px;
      l.y[li] = py;
      l.nx[li] = (double)px / diag;   // :462
      l.ny[li] = (double)py / diag;
      l.nsp[li] = v / sp_div;
      l.nac[li] = acc / sp_div;
      l.ndir[li] = dir / two_pi;
      l.sig[li] = sg;
      l.tagcnt[li] = 0;
      if (ag == 0) {
        const int t = a.timestep[env] + 1;  // :800
        a.timestep[env] = t;
        l.tstep[el] = t;
        l.nrun[el] = a.num_runners[env];
      }
    }
    __syncthreads();

    // ------------------------------------------------ phase 1: K nearest neighbours
    if (!a.use_full_obs && active) {
      int *my_nbr = l.nbr + (size_t)li * K;
      if (l.sig[li]) {
        const float xi = l.x[li], yi = l.y[li];
        const float *cx = l.x + el * N, *cy = l.y + el * N;
        const int *csig = l.sig + el * N;
        if (KMAX > 0) {
          float bd[KMAX > 0 ? KMAX : 1];
          int bi[KMAX > 0 ? KMAX : 1];
#pragma unroll
          for (int k = 0; k < KMAX; ++k) { bd[k] = __builtin_inff(); bi[k] = -1; }
          for (int j = 0; j < N; ++j) {
            const float dx = xi - cx[j], dy = yi - cy[j];   // x[agent] - x[other], :411-417
            float d = sqrtf(dx * dx + dy * dy);
            d = (csig[j] != 0 && j != ag) ? d : __builtin_inff();
#pragma unroll
            for (int k = KMAX - 1; k >= 0; --k) {
              const bool pk = d < bd[k];
              const bool pkm1 = (k > 0) ? (d < bd[k > 0 ? k - 1 : 0]) : false;
              bd[k] = pkm1 ? bd[k > 0 ? k - 1 : 0] : (pk ? d : bd[k]);
              bi[k] = pkm1 ? bi[k > 0 ? k - 1 : 0] : (pk ? j : bi[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) my_nbr[k] = bi[k];
        } else {
          // generic K: K passes, each picks the smallest (d, id) key above the previous one
          float pd = -1.0f;
          int pj = -1;
          for (int k = 0; k < K; ++k) {
            float best = __builtin_inff();
            int bj = -1;
            for (int j = 0; j < N; ++j) {
              if (csig[j] == 0 || j == ag) continue;
              const float dx = xi - cx[j], dy = yi - cy[j];
              const float d = sqrtf(dx * dx + dy * dy);
              const bool above = (d > pd) || (d == pd && j > pj);
              if (above && d < best) { best = d; bj = j; }
            }
            my_nbr[k] = bj;
            if (bj < 0) { for (int kk = k + 1; kk < K; ++kk) my_nbr[kk] = -1; break; }
            pd = best;
            pj = bj;
          }
        }
      } else {
        for (int k = 0; k < K; ++k) my_nbr[k] = -1;
      }
    }
    __syncthreads();

    // ------------------------------------------------ phase 2: observations (coalesced)
    {
      const int envs_here = min(epb, a.E - env0);
      const int per_env = N * F;
      const long obs_base = (long)env0 * per_env;
      const int total = envs_here * per_env;
      for (int q = tid; q < total; q += blockDim.x) {
        const int e = q / per_env, r = q - e * per_env;
        const int i = r / F, f = r - i * F;
        const int eb = e * N;
        const int me = eb + i;
        const bool in_game = l.sig[me] != 0;
        const int width = a.use_full_obs ? (N - 1) : K;
        float v = 0.0f;
        if (f == 7 * width) {
          // time: float(t) / episode_length for agents in the game, else 0  (:474,:493,:543)
          v = in_game ? (float)((double)l.tstep[e] / (double)a.T) : 0.0f;
        } else {
          const int c = f / width, k = f - c * width;
          int j;
          bool valid;
          if (a.use_full_obs) {
            j = k + (k >= i ? 1 : 0);
            valid = true;
          } else {
            j = l.nbr[(size_t)me * K + k];
            valid = in_game && (j >= 0);
            j = max(j, 0);
          }
          const int o = eb + j;
          if (c == 5) v = valid ? (float)l.types[j] : 0.0f;
          else if (c == 6) v = valid ? (float)l.sig[o] : 0.0f;
          else if (!valid || !in_game) v = 0.0f;
          else if (c == 0) v = (float)(l.nx[o] - l.nx[me]);
          else if (c == 1) v = (float)(l.ny[o] - l.ny[me]);
          else if (c == 2) v = (float)((double)l.nsp[o] - (double)l.nsp[me]);
          else if (c == 3) v = (float)((double)l.nac[o] - (double)l.nac[me]);
          else v = (float)((double)l.ndir[o] - (double)l.ndir[me]);
        }
        a.obs[obs_base + q] = v;
      }
      if (!a.use_full_obs) {
        const int per_env_k = N * K;
        const long nb_base = (long)env0 * per_env_k;
        for (int q = tid; q < envs_here * per_env_k; q += blockDim.x) a.nearest_ids[nb_base + q] = l.nbr[q];
      }
    }

    // ------------------------------------------------------------ phase 3: rewards
    float rew = 0.0f;
    bool tagged = false, is_runner = false;
    if (active) {
      const int sg = l.sig[li];
      if (sg) { rew += edge_pen; rew += a.step_rewards[ag]; }  // :655-658
      is_runner = (l.types[ag] == 0) && (sg != 0);              // member of self.runners
      if (is_runner) {
        const float xi = l.x[li], yi = l.y[li];
        float best = __builtin_inff();
        int bt = -1;
        for (int t = 0; t < n_taggers; ++t) {  // ascending ids, first minimum wins :643-651
          const int j = l.tagger_ids[t];
          const float dx = xi - l.x[el * N + j], dy = yi - l.y[el * N + j];
          const float d = sqrtf(dx * dx + dy * dy);  // array ** 2 == x*x, :630-641
          if (d < best) { best = d; bt = j; }
        }
        if (bt >= 0 && best < a.margin) {  // :661
          tagged = true;
          atomicAdd(&l.tagcnt[el * N + bt], 1);
          if (a.runner_exits) atomicSub(&l.nrun[el], 1);
        }
      }
    }
    __syncthreads();
    if (active) {
      if (tagged) rew += a.tag_penalty;                            // :664
      const int c = l.tagcnt[li];
      for (int k = 0; k < c; ++k) rew += a.tag_reward;             // :665, one add per tag
      const bool still_runner = is_runner && !(tagged && a.runner_exits);
      if (l.tstep[el] == a.T && still_runner) rew += a.end_reward;  // :674-676
      a.rewards[gi] = rew;
      if (tagged && a.runner_exits) a.sig_arr[gi] = 0;             // :669
      if (ag == 0) {
        const int nr = l.nrun[el];
        a.num_runners[env] = nr;
        if (l.tstep[el] >= a.T || nr == 0) a.done[env] = 1;        // :880-883
      }
    }
    __syncthreads();
  }
}

}  // namespace

#define WD_TC_PARAMS                                                                              \
  float *loc_x_arr, float *loc_y_arr, float *speed_arr, float *direction_arr,                     \
      float *acceleration_arr, const int *agent_types_arr, float *edge_hit_reward_penalty,        \
      float kEdgeHitPenalty, float kGridLength, const float *acceleration_actions_arr,            \
      const float *turn_actions_arr, float kMaxSpeed, int kNumOtherAgentsObserved,                \
      const float *skill_levels_arr, int kRunnerExitsGameAfterTagged, int *still_in_the_game_arr, \
      int kUseFullObservation, float *obs_arr, const int *action_indices_arr,                     \
      float *neighbor_distances_arr, int *neighbor_ids_sorted_by_distance_arr,                    \
      int *nearest_neighbor_ids, float *rewards_arr, const float *step_rewards_arr,               \
      int *num_runners_arr, float kDistanceMarginForReward, float kTagRewardForTagger,            \
      float kTagPenaltyForRunner, float kEndOfGameRewardForRunner, int *done_arr,                 \
      int *env_timestep_arr, int kNumAgents, int kEpisodeLength, int kNumEnvs

#define WD_TC_PACK()                                                                              \
  TcArgs a;                                                                                       \
  a.loc_x = loc_x_arr; a.loc_y = loc_y_arr; a.speed = speed_arr; a.direction = direction_arr;     \
  a.acceleration = acceleration_arr; a.agent_types = agent_types_arr;                             \
  a.edge_pen_arr = edge_hit_reward_penalty; a.edge_hit_penalty = kEdgeHitPenalty;                 \
  a.grid_length = kGridLength; a.acc_actions = acceleration_actions_arr;                          \
  a.turn_actions = turn_actions_arr; a.max_speed = kMaxSpeed; a.K = kNumOtherAgentsObserved;      \
  a.skill_levels = skill_levels_arr; a.runner_exits = kRunnerExitsGameAfterTagged;                \
  a.sig_arr = still_in_the_game_arr; a.use_full_obs = kUseFullObservation; a.obs = obs_arr;       \
  a.actions = action_indices_arr; a.nearest_ids = nearest_neighbor_ids; a.rewards = rewards_arr;  \
  a.step_rewards = step_rewards_arr; a.num_runners = num_runners_arr;                             \
  a.margin = kDistanceMarginForReward; a.tag_reward = kTagRewardForTagger;                        \
  a.tag_penalty = kTagPenaltyForRunner; a.end_reward = kEndOfGameRewardForRunner;                 \
  a.done = done_arr; a.timestep = env_timestep_arr; a.N = kNumAgents; a.T = kEpisodeLength;       \
  a.E = kNumEnvs;                                                                                 \
  (void)neighbor_distances_arr; (void)neighbor_ids_sorted_by_distance_arr;

extern "C" {

// generic entry: any K (and the full-observation mode)
__global__ void HipTagContinuousStep(WD_TC_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  WD_TC_PACK();
  tc_step_impl<0>(a, tc_smem);
}

// register-resident top-K specialisations; the host picks the smallest KMAX >= K
#define WD_TC_SPECIALISE(KM)                                           \
  __global__ void HipTagContinuousStep_K##KM(WD_TC_PARAMS) {           \
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[]; \
    WD_TC_PACK();                                                      \
    tc_step_impl<KM>(a, tc_smem);                                      \
  }
WD_TC_SPECIALISE(2)
WD_TC_SPECIALISE(4)
WD_TC_SPECIALISE(6)
WD_TC_SPECIALISE(8)
WD_TC_SPECIALISE(10)
WD_TC_SPECIALISE(12)
WD_TC_SPECIALISE(16)
WD_TC_SPECIALISE(24)
WD_TC_SPECIALISE(32)

}  // extern "C"
