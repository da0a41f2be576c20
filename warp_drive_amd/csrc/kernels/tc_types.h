// tc_types.h -- argument / LDS-table / input structures of the TagContinuous kernels.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_config.h"

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
  int env_begin;  // first replica of this launch (a launch covers replicas [env_begin, E))
  int *obs_rows_cleared;  // [E, N] 1 = the agent's observation row in HBM is all zeros already: rows of agents out
                          // of the game are zeros until the episode restarts (:476-560), so the sparse form of the
                          // row gather clears such a row ONCE instead of rewriting it every tick
  unsigned *knn_prev;     // [E, N, 8] 32 bytes per agent (replicas of more than 128 agents; else unused): the ids (16 bits
                          // each, 0xffff = none) of the K + 3 nearest other agents of the previous tick in search order
                          // -- the hint the prefiltered neighbour search starts from (tc_knn_bound16); any content is
                          // safe (the radius is checked); may be null
};

// extra inputs of the fused rollout tick (sample both action heads -> step -> reset finished
// replicas, ONE launch)
struct TcResetEntry {  // same layout as wd_reset_entry in wd_core.hip (global pointers: wd_common.h, wd_global_u32)
  wd_global_u32 *data;
  const wd_global_u32 *ref;
  int row_elems;
  int pad_;
};
struct TcFuse {
  uint32_t *rng_state;             // Philox epoch counters (WD_RNG_HEADER + one word per agent row)
  const float *probs_acc;          // [E, N, n_acc]  policy output, head 0
  const float *probs_turn;         // [E, N, n_turn] policy output, head 1
  int *actions_out;                // [E, N, 2] sampled_actions
  const TcResetEntry *reset_table; // arrays registered with save_copy_and_apply_at_reset
  int n_reset_arrays;
  int stream_tag;
};

// observation features of one agent after the move, as the reference computes them (:453-470):
// x, y normalised in float64; speed / acceleration / direction normalised in float32 (widened
// to float64 only for the neighbour difference); type and still_in_game packed in one word.
// 32 bytes: a neighbour is fetched with two ds_read_b128.
struct __attribute__((aligned(16))) TcFeat {
  double nx, ny;
  float nsp, nac, ndir;
  int type_sig;  // float bits of the agent type (1.0f = tagger, 0) | bit 0: still_in_the_game before tagging
};

// The fast path keeps the record as two arrays of 16-byte halves: a ds_read_b128 starts on one of the 16 aligned
// four-bank slots of the 64 LDS banks; records of 32 bytes reach only the 8 even slots (a gather of 64 random
// neighbours then takes 8 passes), halves of 16 bytes reach all 16 (4 passes, the minimum for 64 lanes).
struct __attribute__((aligned(16))) TcFeatA { double nx, ny; };
struct __attribute__((aligned(16))) TcFeatB { float nsp, nac, ndir; int type_sig; };
struct TcFeatArrays {
  TcFeatA *a;
  TcFeatB *b;
};
__device__ __forceinline__ TcFeat tc_feat_load(const TcFeatArrays &f, int i) {
  const TcFeatA ha = f.a[i];
  const TcFeatB hb = f.b[i];
  TcFeat r;
  r.nx = ha.nx; r.ny = ha.ny; r.nsp = hb.nsp; r.nac = hb.nac; r.ndir = hb.ndir; r.type_sig = hb.type_sig;
  return r;
}
__device__ __forceinline__ void tc_feat_store(const TcFeatArrays &f, int i, const TcFeat &v) {
  TcFeatA ha; ha.nx = v.nx; ha.ny = v.ny;
  TcFeatB hb; hb.nsp = v.nsp; hb.nac = v.nac; hb.ndir = v.ndir; hb.type_sig = v.type_sig;
  f.a[i] = ha;
  f.b[i] = hb;
}

struct TcCand {
  float d2;
  int id;
};

#define WD_TC_TAB 64      // capacity of the LDS copies of the action tables
#define WD_BIG 1.0e30f    // (x - BIG)^2 overflows to +inf: such a candidate is never selected

__device__ __forceinline__ size_t tc_align16(size_t v) { return (v + 15) & ~(size_t)15; }

// replica-independent tables, alive for the whole launch
struct TcTables {
  int *tagger_ids;   // [N] ascending
  float *acc_tab, *turn_tab;  // action tables (n_acc, n_turn entries; capacity WD_TC_TAB each)
  int *wave_cnt;     // [16] taggers per wavefront (rank computation)
  int *live_cnt;     // [16] agents still in the game per wavefront (compaction of the search, one replica per block)
  int *tstep, *nrun; // [epb]
  float *tfrac;      // [epb] float(t) / episode_length
  int *doneflag;     // [epb] replica finished on this tick (fused tick only)
  int *cell_cnt;     // [64] agents in the game per grid cell (cell-sorted search of replicas of more than 128 agents, tc_fast.h;
                     // behind everything else: the host adds the bytes for those replicas only, envs/tag_continuous.py lds_bytes)
};

__device__ __forceinline__ TcTables tc_carve_tables(unsigned char *p, int epb, int N) {
  TcTables t;
  size_t off = 0;
  t.tagger_ids = (int *)(p + off); off += 4 * (size_t)N;
  t.acc_tab = (float *)(p + off); off += 4 * WD_TC_TAB;
  t.turn_tab = (float *)(p + off); off += 4 * WD_TC_TAB;
  t.wave_cnt = (int *)(p + off); off += 4 * 16;
  t.live_cnt = (int *)(p + off); off += 4 * 16;
  t.tstep = (int *)(p + off); off += 4 * epb;
  t.nrun = (int *)(p + off); off += 4 * epb;
  t.tfrac = (float *)(p + off); off += 4 * epb;
  t.doneflag = (int *)(p + off); off += 4 * epb;
  t.cell_cnt = (int *)(p + off);
  return t;
}

// every global input of one loop trip; issued together so the HBM latency is paid once
struct TcIn {
  int sg, type;
  float dir, acc, speed, x, y, skill;
  int2 sampled;
  uint32_t epoch;
  float step_reward;   // step_rewards[agent]
  int tstep, nrun;     // lane of agent 0: _timestep_ / num_runners of the replica
  float tab_acc, tab_turn;  // entry `tid` of the two action tables (tables of at most WD_TC_TAB entries)
  int cleared;              // obs_rows_cleared[agent] (fast path)
};

}  // namespace
