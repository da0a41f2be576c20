// tc_reset.h -- restore of finished replicas inside the fused tick (reset.cu:9-75).
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_types.h"

namespace {

// ---- fused tick: reset finished replicas in place (reset.cu:9-75 for every registered array).
// `_done_` stays 1 so the trainer can read which replicas finished on this tick; the next tick
// clears it.  Must be entered by the whole block after a barrier that follows every store of the
// tick to these rows (the caller drains its own stores first).
__device__ __forceinline__ void tc_reset_finished(const TcArgs &a, const TcFuse &fz, const TcTables &tb, int env0,
                                                  int epb) {
  const int tid = threadIdx.x, T_ = WD_TC_BLOCKDIM;
  const int envs_here = min(epb, a.E - env0);
  for (int e = 0; e < envs_here; ++e) {
    if (tb.doneflag[e] == 0) continue;  // block-uniform
    for (int r = 0; r < fz.n_reset_arrays; ++r) {
      const TcResetEntry ent = fz.reset_table[r];
      const long base = (long)(env0 + e) * ent.row_elems;
      for (int i = tid; i < ent.row_elems; i += T_) ent.data[base + i] = ent.ref[base + i];
    }
    if (tid == 0) a.timestep[env0 + e] = 0;
  }
}

}  // namespace
