// tag_continuous.hip -- TagContinuous step / fused rollout tick for gfx950.
//
// Semantics: the reference CPU step, example_envs/tag_continuous/tag_continuous.py
//   update_state :339-401, compute_distance :403-420, k_nearest_neighbors :422-444,
//   generate_observation :446-610, compute_reward :612-678, done :880-883.
// Where the reference's own CUDA kernel (tag_continuous_step_pycuda.cu:351-520)
// disagrees with its CPU step the CPU wins: stable (distance, id) neighbour order,
// tag counts accumulated without races, no end-of-game bonus for a runner tagged out
// on the last tick.  Argument order is the reference kernel's (:351-385) plus trailing
// n_envs, the two action-table lengths and the first replica of the launch; the two O(N^2)
// global scratch arrays it sorts in HBM (neighbor_distances,
// neighbor_ids_sorted_by_distance; :167-199) are accepted and never touched.
//
// Two implementations share the move / reward code:
//
//   tc_fast_impl<KMAX>   N <= 1024 agents per replica, K <= KMAX <= 32 observed neighbours (K <= 16 beyond 512 agents)
//     (the BASELINE shape: N = 105, K = 10; entry points Hip...Step_K<k> / Tick_K<k> up to 128 agents, ..._K<k>_N512 up to
//     512, ..._K<k>_N1024 beyond).
//     block = `epb` whole replicas (105 agents -> 1 replica on 128 threads), thread = agent.
//       fetch    every global LOAD of the tick is issued first (state, step rewards, time step, action
//                tables; the memory counters return in order, so a load issued later would wait for all
//                stores issued before it); fused tick: each wavefront's rows of the two probability
//                tensors go straight into LDS (global_load_lds_dwordx4, 1 KiB per instruction);
//       sample   (fused tick) Philox4x32-10 + inverse CDF on a running float32 sum, both heads;
//       move     float32 kinematics exactly as numpy evaluates them (numpy-exact cos/sin);
//                post-move state staged in LDS (positions; 32-byte feature records);
//       tags     every runner finds its nearest tagger; tag counts through LDS atomics (their
//                block barrier is the one after the gather);
//       search   over the agents still IN THE GAME only (one replica per block: they are packed in
//                ascending id order -- as candidates, so the chain is as long as the live list, and as
//                searchers, so a wavefront without a live searcher skips the search; 54 of 105 agents
//                are in the game on average over an episode of the benchmark policy).  Per searcher,
//                in registers, ONE pass over the candidates: the candidate's packed index rides in
//                the low 7 bits of the squared distance through a v_med3_u32 chain that keeps the
//                K+3 smallest keys (K+2 -- ONE look-ahead entry -- where K == KMAX is known and the replica
//                has at most 128 agents: the BASELINE shape's entries; a near-tie at the cut, ~5e-4 per agent,
//                then goes to tc_zone_resolve); where the first K+1 keys are far enough apart the chain order
//                is the reference's order and the ids are read off the keys, otherwise the exact
//                (sqrt(d^2), id) keys of the first K(+1) entries are ranked by pairwise
//                compare-and-count -- an ISOLATED close pair by one exact compare of the two (round 4);
//                ~1e-7 of the agents repeat the search with the two-pass one (tc_knn_registers: exact K-th
//                distance, compare-mask pass, id-ordered peeling) or, beyond 128 candidates, have the whole
//                wavefront resolve the zone around the cut (tc_zone_resolve).  While at most 64 agents are in
//                the game both wavefronts of a block chain half of the candidates each (tc_merge_sorted).  Replicas
//                of more than 128 agents search inside a radius derived from the previous tick's neighbours while at
//                least 200 agents are in the game (tc_pre_pass1 / tc_pre_pass2, cell-sorted since round 6: one subtraction per candidate of the cell rows around a wavefront, the chain over
//                the candidates inside the radius only, the radius checked afterwards);
//       ids out  packed indices -> agent ids through an LDS table; 16-bit block-local ids per agent
//                row in LDS (entry k -> slot k; out-of-order lanes rewrite their rows by rank); one
//                block barrier; nearest_neighbor_ids rows are converted from them and stored;
//       gather   the block's rows are split evenly over its wavefronts; each WAVEFRONT turns its rows
//                into observation rows inside a private LDS staging buffer, a chunk of rows at a
//                time, and streams every chunk out as one contiguous run of write-through 16-byte
//                stores (the [E, N, F] layout makes a replica's rows contiguous); a wavefront with at most
//                9/16 of its rows live builds and stores the live rows only (tc_gather_rows_sparse);
//       rewards  tag counts -> rewards in the CPU's add order, done flags;
//       reset    (fused tick) finished replicas are restored in place from the registered
//                `*_at_reset` copies.
//     Wave priority falls with the phase (s_setprio 3, 2, 0), so the wavefronts of a SIMD finish together.
//
//   tc_generic_impl      full observations, or K beyond the specialisations (any N <= 1024; entry points
//     HipTagContinuousStep / HipTagContinuousTick): K-pass selection per agent, observation
//     rows written from a block-strided (row, slot) loop (16-byte stores in the
//     full-observation mode).
#include "wd_common.h"

// phases (each header: its own anonymous-namespace block; order = dependency order)
#include "tc_types.h"
#include "tc_fetch.h"
#include "tc_sample.h"
#include "tc_move.h"
#include "tc_tags.h"
#include "tc_reset.h"
#include "tc_knn.h"
#include "tc_rows.h"
#include "tc_fast.h"
#include "tc_generic.h"

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
      int *env_timestep_arr, int kNumAgents, int kEpisodeLength, int kNumEnvs,                    \
      int kNumAccelerationActions, int kNumTurnActions, int *obs_rows_cleared_arr,                 \
      unsigned *knn_prev_arr, int kEnvBegin

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
  a.E = kNumEnvs; a.env_begin = kEnvBegin; a.obs_rows_cleared = obs_rows_cleared_arr;              \
  a.knn_prev = knn_prev_arr;                                                                      \
  (void)neighbor_distances_arr; (void)neighbor_ids_sorted_by_distance_arr;

// Fused rollout tick: sample both action heads + step + reset finished replicas in ONE launch
// (the reference needs 2 sampler launches, the step, and 13 reset launches per tick,
// trainer_base.py:392-426).  Same arguments as the step plus the sampler / reset inputs.
#define WD_TC_FUSE_PARAMS                                                                      \
  , uint32_t *rng_state, const float *probs_acc, const float *probs_turn, const void *reset_table, \
      int n_reset_arrays, int stream_tag
#define WD_TC_FUSE_PACK()                                                                      \
  TcFuse fz;                                                                                   \
  fz.rng_state = rng_state; fz.probs_acc = probs_acc; fz.probs_turn = probs_turn;              \
  fz.actions_out = const_cast<int *>(action_indices_arr);                                      \
  fz.reset_table = (const TcResetEntry *)reset_table; fz.n_reset_arrays = n_reset_arrays;      \
  fz.stream_tag = stream_tag;

// ---- entries.  This file is compiled into SEVERAL code objects (warp_drive_amd/build.py UNITS; a build of all of them
// runs in parallel, and tuning one specialisation rebuilds one small object):
//   (no WD_TC_KM)                      the generic entries HipTagContinuousStep / Tick: any N <= 1024, any K, full
//                                      observations                                        -> wd_kernels_tc.hsaco
//   -DWD_TC_KM=<k> -DWD_TC_WAVES=<w>   the fast entries for K <= k: `_K<k>` up to 128 agents (7 id bits in the search
//        [-DWD_TC_BIG]                 keys), `_K<k>_N512` for 129 .. 512 (blocks of up to eight wavefronts, 9 id
//                                      bits) and, with WD_TC_BIG, `_K<k>_N1024` beyond      -> wd_kernels_tc_k<k>.hsaco
//   -DWD_TC_KM=<k> -DWD_TC_SHAPE_N=<n> -DWD_TC_SHAPE_A=<a> -DWD_TC_SHAPE_THREADS=<t>
//                                      `_K<k>_N<n>A<a>`: ONE shape (n agents, exactly k observed, a-way action heads,
//                                      blocks of t threads: the launch MUST use them, the host asserts it)
//                                      with its sizes as compile-time constants -- what the reference gets for EVERY
//                                      run by templating wkNumberAgents into the source it hands to nvcc
//                                      (template_env_config.h:19-21); built ahead of time for the BASELINE shape
//                                      (105 agents, K = 10, 21-way heads, 128 threads: -3.8 % per tick for the sizes, -3.3 % more for
//                                      the block size, experiments/README.md)
//                                                                                           -> wd_kernels_tc_k10_n105a21.hsaco
// Separate objects, so that work on the big-replica search never moves the registers or the code layout of the
// headline kernel.  Every fast size class has three entries: Step (actions given), Tick (sample both heads + step +
// restore finished replicas) and TickA (actions given -- drawn by the policy forward's epilogue, policy_mlp.hip --
// + step + restore; same arguments as Tick, the sampler's are ignored).
extern "C" {

#define WD_TC_CAT_(a, b) a##b
#define WD_TC_CAT(a, b) WD_TC_CAT_(a, b)
#define WD_TC_SMEM() extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[]

#if !defined(WD_TC_KM)

__global__ void HipTagContinuousStep(WD_TC_PARAMS) {
  WD_TC_SMEM();
  WD_TC_PACK();
  tc_generic_impl<false>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions);
}

__global__ void HipTagContinuousTick(WD_TC_PARAMS WD_TC_FUSE_PARAMS) {
  WD_TC_SMEM();
  WD_TC_PACK();
  WD_TC_FUSE_PACK();
  tc_generic_impl<true>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);
}

#elif defined(WD_TC_SHAPE_N)

// one shape, sizes folded: N, K (exactly KM) and the two head sizes are constants from here on
#define WD_TC_SHAPE_NAME(stem) WD_TC_CAT(WD_TC_CAT(WD_TC_CAT(WD_TC_CAT(WD_TC_CAT(stem, WD_TC_KM), _N), WD_TC_SHAPE_N), A), WD_TC_SHAPE_A)
static_assert(WD_TC_SHAPE_N <= 128, "the shape-specialised entries use the 7-bit-id search of replicas up to 128 agents");
__global__ void __launch_bounds__(512, 4) WD_TC_SHAPE_NAME(HipTagContinuousStep_K)(WD_TC_PARAMS) {
  WD_TC_SMEM();
  WD_TC_PACK();
  a.N = WD_TC_SHAPE_N; a.K = WD_TC_KM;
  tc_fast_impl<WD_TC_KM, false, true, 7>(a, TcFuse{}, tc_smem, WD_TC_SHAPE_A, WD_TC_SHAPE_A);
}
__global__ void __launch_bounds__(512, 4) WD_TC_SHAPE_NAME(HipTagContinuousTick_K)(WD_TC_PARAMS WD_TC_FUSE_PARAMS) {
  WD_TC_SMEM();
  WD_TC_PACK();
  WD_TC_FUSE_PACK();
  a.N = WD_TC_SHAPE_N; a.K = WD_TC_KM;
  tc_fast_impl<WD_TC_KM, true, true, 7>(a, fz, tc_smem, WD_TC_SHAPE_A, WD_TC_SHAPE_A);
}
__global__ void __launch_bounds__(512, 4) WD_TC_SHAPE_NAME(HipTagContinuousTickA_K)(WD_TC_PARAMS WD_TC_FUSE_PARAMS) {
  WD_TC_SMEM();
  WD_TC_PACK();
  WD_TC_FUSE_PACK();
  a.N = WD_TC_SHAPE_N; a.K = WD_TC_KM;
  tc_fast_impl<WD_TC_KM, true, true, 7, false>(a, fz, tc_smem, WD_TC_SHAPE_A, WD_TC_SHAPE_A);
}

#else

#define WD_TC_NAME(stem, suffix) WD_TC_CAT(WD_TC_CAT(stem, WD_TC_KM), suffix)
#define WD_TC_FAST_ENTRIES(suffix, THREADS, WAVES, IDB, SPLIT_EXACT)                                            \
  __global__ void __launch_bounds__(THREADS, WAVES) WD_TC_NAME(HipTagContinuousStep_K, suffix)(WD_TC_PARAMS) {   \
    WD_TC_SMEM();                                                                                                \
    WD_TC_PACK();                                                                                                \
    if (SPLIT_EXACT && a.K == WD_TC_KM)                                                                          \
      tc_fast_impl<WD_TC_KM, false, SPLIT_EXACT, IDB>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions); \
    else                                                                                                         \
      tc_fast_impl<WD_TC_KM, false, false, IDB>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions); \
  }                                                                                                              \
  __global__ void __launch_bounds__(THREADS, WAVES) WD_TC_NAME(HipTagContinuousTick_K, suffix)(WD_TC_PARAMS WD_TC_FUSE_PARAMS) { \
    WD_TC_SMEM();                                                                                                \
    WD_TC_PACK();                                                                                                \
    WD_TC_FUSE_PACK();                                                                                           \
    if (SPLIT_EXACT && a.K == WD_TC_KM)                                                                          \
      tc_fast_impl<WD_TC_KM, true, SPLIT_EXACT, IDB>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);  \
    else                                                                                                         \
      tc_fast_impl<WD_TC_KM, true, false, IDB>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);        \
  }                                                                                                              \
  __global__ void __launch_bounds__(THREADS, WAVES) WD_TC_NAME(HipTagContinuousTickA_K, suffix)(WD_TC_PARAMS WD_TC_FUSE_PARAMS) { \
    WD_TC_SMEM();                                                                                                \
    WD_TC_PACK();                                                                                                \
    WD_TC_FUSE_PACK();                                                                                           \
    tc_fast_impl<WD_TC_KM, true, false, IDB, false>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);   \
  }
// up to 128 agents: the entry branches once on "exactly KM observed" (the unrolled row layout of the usual case)
WD_TC_FAST_ENTRIES(, 512, WD_TC_WAVES, 7, true)
WD_TC_FAST_ENTRIES(_N512, 512, WD_TC_WAVES, 9, false)
#if defined(WD_TC_BIG)
// replicas of 513 .. 1024 agents: blocks of up to sixteen wavefronts (1024 threads: the reference's default geometry
// serves up to 1024 agents per block, managers/function_manager.py:64-67), 10 id bits in the search keys (buckets of
// 1024 ulps of d2: the exactness argument of tc_resolve_keys holds for any bucket width)
WD_TC_FAST_ENTRIES(_N1024, 1024, 4, 10, false)
#endif

#endif

}  // extern "C"
