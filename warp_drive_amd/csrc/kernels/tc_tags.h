// tc_tags.h -- tags / rewards / done (compute_reward, :612-678, :880-883).
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_types.h"

namespace {

// ---- tags: a runner in the game finds its nearest tagger (ascending ids, first minimum wins,
// :643-651) and is tagged when closer than the margin (:661); counts go through LDS atomics.
__device__ __forceinline__ bool tc_find_tag(const TcArgs &a, const TcTables &tb, const float2 *cxy, int *tagcnt_env,
                                            int *nrun_env, int n_taggers, float my_x, float my_y) {
  float best = __builtin_inff();
  int bt = -1;
  constexpr int U = 5;  // taggers per batch: all id reads, then all position reads, in flight together
  for (int t0 = 0; t0 < n_taggers; t0 += U) {
    int j[U];
    float2 pt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = tb.tagger_ids[min(t0 + u, n_taggers - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) pt[u] = cxy[j[u]];  // taggers are never out of the game: real positions
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float dx = my_x - pt[u].x, dy = my_y - pt[u].y;
      const float d = sqrtf(dx * dx + dy * dy);  // array ** 2 == x*x, :630-641
      if (t0 + u < n_taggers && d < best) { best = d; bt = j[u]; }
    }
  }
  if (bt >= 0 && best < a.margin) {
    atomicAdd(&tagcnt_env[bt], 1);
    if (a.runner_exits) atomicSub(nrun_env, 1);
    return true;
  }
  return false;
}

// ---- rewards / done of one agent (:655-678, :880-883); call after the barrier that follows the tags
__device__ __forceinline__ void tc_finish_agent(const TcArgs &a, const TcTables &tb, int el, int ag, int gi, int env,
                                                int sg, bool is_runner, bool tagged, int tagcnt, float edge_pen,
                                                float step_reward, bool fused) {
  float rew = 0.0f;
  if (sg) { rew += edge_pen; rew += step_reward; }              // :655-658
  if (tagged) rew += a.tag_penalty;                             // :664
  for (int k = 0; k < tagcnt; ++k) rew += a.tag_reward;         // :665, one add per tag
  const bool still_runner = is_runner && !(tagged && a.runner_exits);
  if (tb.tstep[el] == a.T && still_runner) rew += a.end_reward; // :674-676
  a.rewards[gi] = rew;
  if (tagged && a.runner_exits) a.sig_arr[gi] = 0;              // :669
  if (ag == 0) {
    const int nr = tb.nrun[el];
    a.num_runners[env] = nr;
    const bool fin = (tb.tstep[el] >= a.T || nr == 0);          // :880-883
    if (fin) a.done[env] = 1;
    if (fused) tb.doneflag[el] = fin ? 1 : 0;
  }
}

}  // namespace
