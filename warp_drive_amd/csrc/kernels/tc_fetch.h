// tc_fetch.h -- fetch: every global load of a tick issued up front, probability slabs straight into LDS; the replica-independent tables.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_types.h"

namespace {

// this wavefront's 64 rows of one head's probability tensor -> LDS (asynchronous: wd_slab_fetch)
__device__ __forceinline__ void tc_fetch_slab(float *slab, const float *probs, const TcArgs &a, int env0, int epb, int N,
                                              int n_actions, int tid) {
  const int rows_here = min(epb, a.E - env0) * N;
  const int r0 = (tid >> 6) * 64, lane = tid & 63;
  const int wrows = max(0, min(64, rows_here - r0));
  wd_slab_fetch(slab + (size_t)r0 * n_actions, probs + ((long)env0 * N + r0) * n_actions, wrows * n_actions, lane);
}

// Replicas of more than 256 agents sample the two heads one after the other from ONE slab (the second head's rows
// are fetched into the same LDS after the first head was sampled: wave-private rows, no block barrier): both slabs
// of a 1005-agent replica with 21-way heads are 169 KB, and at ~510 agents half the LDS means two blocks per CU.
__device__ __forceinline__ bool tc_one_slab(int N) { return N > 256; }

// FUSED: the launch also restores finished replicas; SAMPLE: it also draws the actions (false: they are read from
// `actions`, e.g. drawn by the policy forward's epilogue -- csrc/kernels/policy_mlp.hip)
template <bool FUSED, bool SAMPLE = FUSED>
__device__ __forceinline__ void tc_issue_loads(TcIn &in, const TcArgs &a, const TcFuse &fz, int env0, int epb,
                                               int N, int n_acc, int n_turn, int tid, float *slab_acc,
                                               float *slab_turn, bool want_cleared = false) {
  const int el = tid / N, ag = tid - el * N;
  const int env = env0 + el;
  const bool active = (el < epb) && (env < a.E);
  const int gi = env * N + ag;
  in.sg = 0; in.type = 0; in.dir = in.acc = in.speed = in.x = in.y = in.skill = 0.f;
  in.sampled = make_int2(0, 0);
  in.epoch = 0u;
  in.step_reward = 0.f;
  in.tstep = in.nrun = 0;
  in.tab_acc = in.tab_turn = 0.f;
  in.cleared = 0;
  if (n_acc <= WD_TC_TAB && n_turn <= WD_TC_TAB) {  // (a block has at least 64 threads)
    if (tid < n_acc) in.tab_acc = a.acc_actions[tid];
    if (tid < n_turn) in.tab_turn = a.turn_actions[tid];
  }
  if (active) {
    in.sg = a.sig_arr[gi];
    in.dir = a.direction[gi];
    in.acc = a.acceleration[gi];
    in.speed = a.speed[gi];
    in.x = a.loc_x[gi];
    in.y = a.loc_y[gi];
    in.skill = a.skill_levels[ag];
    in.type = a.agent_types[ag];
    // (the counters return in order: a load issued after the tick's stores would wait for all of them)
    in.step_reward = a.step_rewards[ag];
    if (want_cleared) in.cleared = a.obs_rows_cleared[gi];
    if (ag == 0) {
      in.tstep = a.timestep[env];
      in.nrun = a.num_runners[env];
    }
    if (!SAMPLE) in.sampled = ((const int2 *)a.actions)[gi];
    if (SAMPLE) in.epoch = fz.rng_state[WD_RNG_HEADER + gi];
  }
  if (SAMPLE) {
    // this wavefront's rows of both probability slabs -> LDS (the second one later when they share the LDS)
    tc_fetch_slab(slab_acc, fz.probs_acc, a, env0, epb, N, n_acc, tid);
    if (!tc_one_slab(N)) tc_fetch_slab(slab_turn, fz.probs_turn, a, env0, epb, N, n_turn, tid);
  }
}

// ---- replica-independent tables: ascending tagger list, action tables.
// Returns the number of taggers.  Ends WITHOUT a barrier: the caller's next barrier publishes them.
__device__ __forceinline__ int tc_build_tables(const TcTables &tb, const TcArgs &a, int N, int n_acc, int n_turn,
                                               bool tab_in_lds, const TcIn &in) {
  const int tid = threadIdx.x, T_ = WD_TC_BLOCKDIM;
  const int my_type = in.type;
  if (tab_in_lds) {  // (entries loaded up front, before the probability slabs)
    if (tid < n_acc) tb.acc_tab[tid] = in.tab_acc;
    if (tid < n_turn) tb.turn_tab[tid] = in.tab_turn;
  }
  int n_taggers = 0;
  // rank of a tagger = number of taggers with a smaller id: wave ballots + per-wave counts
  const int wave = tid >> 6, lane = tid & 63, n_waves = (T_ + 63) >> 6;
  if (N <= T_) {  // usual case: one barrier
    // (thread tid < N is agent tid of the block's first replica: its type is among the loads issued up
    // front, BEFORE the probability slabs, so waiting for it does not wait for the slabs)
    const int ty = (tid < N) ? my_type : 0;
    const unsigned long long m = __ballot(ty == 1);
    if (lane == 0) tb.wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w2 = 0; w2 < n_waves; ++w2) {
      const int c = tb.wave_cnt[w2];
      before += (w2 < wave) ? c : 0;
      n_taggers += c;
    }
    if (ty == 1) tb.tagger_ids[before + __popcll(m & ((1ull << lane) - 1ull))] = tid;
  } else {
    for (int base = 0; base < N; base += T_) {
      const int i = base + tid;
      const int ty = (i < N) ? a.agent_types[i] : 0;
      const unsigned long long m = __ballot(ty == 1);
      if (lane == 0) tb.wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int before = n_taggers;
      for (int w2 = 0; w2 < wave; ++w2) before += tb.wave_cnt[w2];
      if (ty == 1) tb.tagger_ids[before + __popcll(m & ((1ull << lane) - 1ull))] = i;
      for (int w2 = 0; w2 < n_waves; ++w2) n_taggers += tb.wave_cnt[w2];
      __syncthreads();
    }
  }
  return n_taggers;
}

}  // namespace
