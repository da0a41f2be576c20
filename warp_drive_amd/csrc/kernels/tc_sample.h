// tc_sample.h -- sample: both action heads of an agent from the LDS slabs (fused tick).
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_fetch.h"

namespace {

// ---- fused tick: sample both action heads for this thread's agent (replaces two sample_actions
// launches, random.cu:51-85): inverse CDF on a running float32 sum, one Philox call for both heads.
__device__ __forceinline__ int2 tc_sample_heads(const TcArgs &a, const TcFuse &fz, const TcIn &in, bool active, int gi,
                                                int li, const float *slab_acc, float *slab_turn, int n_acc,
                                                int n_turn, int env0, int epb) {
  int2 sampled = make_int2(0, 0);
  wd_u4 rnd = wd_u4{0u, 0u, 0u, 0u};
  if (active) {
    fz.rng_state[WD_RNG_HEADER + gi] = in.epoch + 1u;
    rnd = wd_philox4x32_10(wd_u4{(uint32_t)gi, in.epoch, (uint32_t)fz.stream_tag, 3u}, fz.rng_state[0],
                           fz.rng_state[1]);
  }
  // every global_load_lds of this wavefront has landed once its vmcnt drains; the rows a lane
  // reads were all fetched by its own wavefront
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (active) sampled.x = wd_slab_sample(slab_acc + (size_t)li * n_acc, n_acc, wd_u01_open_closed(rnd.x));
  if (tc_one_slab(a.N)) {  // block-uniform: the second head's rows replace the first head's (slab_turn == slab_acc)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wavefront's reads of its rows are complete
    __builtin_amdgcn_wave_barrier();
    // Wavefront w's rows start at float 64 * w * n_actions: with heads of EQUAL size the two heads' rows of a
    // wavefront coincide and are wave-private.  With unequal sizes w's turn rows overlap the acceleration rows of
    // its neighbours: every wavefront must have sampled its first head (which also means every acceleration fetch
    // has landed) before anybody fetches the second.  Block-uniform condition.
    if (n_acc != n_turn) __syncthreads();
    tc_fetch_slab(slab_turn, fz.probs_turn, a, env0, epb, a.N, n_turn, threadIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (active) {
    sampled.y = wd_slab_sample(slab_turn + (size_t)li * n_turn, n_turn, wd_u01_open_closed(rnd.y));
    ((int2 *)fz.actions_out)[gi] = sampled;
  }
  return sampled;
}

}  // namespace
