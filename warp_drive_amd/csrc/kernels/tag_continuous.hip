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
//                K+3 smallest keys; where the first K+1 keys are far enough apart the chain order
//                is the reference's order and the ids are read off the keys, otherwise the exact
//                (sqrt(d^2), id) keys of the first K(+1) entries are ranked by pairwise
//                compare-and-count -- an ISOLATED close pair by one exact compare of the two (round 4);
//                ~1e-7 of the agents repeat the search with the two-pass one (tc_knn_registers: exact K-th
//                distance, compare-mask pass, id-ordered peeling) or, beyond 128 candidates, have the whole
//                wavefront resolve the zone around the cut (tc_zone_resolve).  While at most 64 agents are in
//                the game both wavefronts of a block chain half of the candidates each (tc_merge_sorted).  Replicas
//                of more than 128 agents search inside a radius derived from the previous tick's neighbours while at
//                least 200 agents are in the game (tc_chain_prefiltered: one compare per candidate, the chain over
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

// threads per block: a launch-time value, except in the unit that is built for ONE shape (its host geometry is fixed:
// envs/tag_continuous.py::_geometry), where the number of wavefronts, the replicas per block and every loop over them fold
#if defined(WD_TC_SHAPE_THREADS)
#define WD_TC_BLOCKDIM WD_TC_SHAPE_THREADS
#else
#define WD_TC_BLOCKDIM ((int)blockDim.x)
#endif

// Phase probes (experiments/phase_profile.py): compiled out of the product build.  With -DWD_TC_PROBES (variant
// "prof" of experiments/variant_sets.py) lane 0 of every wavefront of the fast path stamps the shader clock at
// the phase boundaries into 24 slots per wavefront behind a __device__ pointer the harness sets.
#ifdef WD_TC_PROBES
extern "C" { __device__ unsigned long long *tc_prof_g = nullptr; }
#define WD_TC_SLOT(k) ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 24 + (k))
#define WD_TC_PROBE(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) tc_prof_g[WD_TC_SLOT(k)] = __builtin_readcyclecounter(); } while (0)
#define WD_TC_PROBE_RT(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) tc_prof_g[WD_TC_SLOT(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// a counter of the wavefront (callable inside divergent code: the first active lane adds)
#define WD_TC_PROBE_VAL(k, v) do { if (tc_prof_g) { const unsigned long long m_ = __ballot(1);                          \
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m_) - 1)) tc_prof_g[WD_TC_SLOT(k)] += (unsigned long long)(v); } } while (0)
// where the wavefront runs: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) | XCC_ID << 32
#define WD_TC_PROBE_HW(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) { unsigned hw_, xcc_;                        \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                  \
    tc_prof_g[WD_TC_SLOT(k)] = (unsigned long long)hw_ | ((unsigned long long)(xcc_ & 15u) << 32); } } while (0)
#else
#define WD_TC_PROBE_HW(k)
#define WD_TC_PROBE(k)
#define WD_TC_PROBE_RT(k)
#define WD_TC_PROBE_VAL(k, v)
#endif

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
  t.doneflag = (int *)(p + off);
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

// ---- move: float32 kinematics exactly as numpy evaluates update_state (:339-401); stores the new
// state, returns the post-move position, the edge penalty and the observation features.
struct TcMoved {
  float x, y, edge_pen;
  TcFeat ft;
};
__device__ __forceinline__ TcMoved tc_move(const TcArgs &a, const TcTables &tb, const TcIn &in, int2 act, int gi,
                                           bool tab_in_lds) {
  const float two_pi = 6.2831854820251465f;            // float32(2*pi), :356
  const float L = a.grid_length;
  const double diag = (double)L * 1.4142135623730951;  // float32 L * np.sqrt(2) -> f64, :146
  const float sp_div = a.max_speed + 1.0e-10f;         // float32 + float32(eps), :456
  const float s = (float)in.sg;
  // (value select, not pointer select: a pointer that may be LDS or global becomes a flat access)
  float d_acc = tb.acc_tab[min(act.x, WD_TC_TAB - 1)], d_turn = tb.turn_tab[min(act.y, WD_TC_TAB - 1)];
  asm volatile("" : "+v"(d_acc), "+v"(d_turn));  // keeps the two loads from being merged into one flat load
  if (!tab_in_lds) {
    d_acc = a.acc_actions[act.x];
    d_turn = a.turn_actions[act.y];
  }
  const float dir = wd_np_remainderf(in.dir + d_turn, two_pi) * s;            // :355-357
  float acc = in.acc + d_acc;                                                 // :359
  const float vmax = a.max_speed * in.skill;                                  // :363
  float v = in.speed + acc;
  v = fminf(fmaxf(v, 0.0f), vmax) * s;                                        // :364-366
  acc = acc * (v > 0.0f ? 1.0f : 0.0f) * (v < vmax ? 1.0f : 0.0f);            // :367
  float sn, cs;
  wd_np_sincosf(dir, sn, cs);
  float px = in.x + v * cs;                                                   // :369-374
  float py = in.y + v * sn;
  const bool crossed = !((px >= 0.0f) && (px <= L) && (py >= 0.0f) && (py <= L));
  px = fminf(fmaxf(px, 0.0f), L);                                             // :385-391
  py = fminf(fmaxf(py, 0.0f), L);
  TcMoved m;
  m.edge_pen = a.edge_hit_penalty * (crossed ? 1.0f : 0.0f);                  // :394
  a.loc_x[gi] = px;
  a.loc_y[gi] = py;
  a.speed[gi] = v;
  a.direction[gi] = dir;
  a.acceleration[gi] = acc;
  a.edge_pen_arr[gi] = m.edge_pen;
  m.x = px;
  m.y = py;
  m.ft.nx = (double)px / diag;    // :462 (float64 division)
  m.ft.ny = (double)py / diag;
  m.ft.nsp = v / sp_div;          // float32 division (:456-458)
  m.ft.nac = acc / sp_div;
  m.ft.ndir = dir / two_pi;
  m.ft.type_sig = ((in.type & 1) ? 0x3f800000 : 0) | (in.sg ? 1 : 0);
  return m;
}

// ---- the seven observation values of row `me` about neighbour `nb` (:479-560).  float64
// differences for x, y, narrowed to float32 like the reference's device push; speed / acc / dir:
// the reference widens float32 values and subtracts in float64; for float32 operands that rounds
// to exactly the float32 difference (53 >= 2*24+2 bits: double rounding is innocuous).
__device__ __forceinline__ void tc_obs_values(float (&vals)[7], const TcFeat &nb, const TcFeat &me, bool rel,
                                              bool valid) {
  // masked with AND (all-ones / zero) rather than selected: a run of v_cndmask on one condition is
  // several times slower than a run of v_and on gfx950, and the masked value is +0.0 exactly
  unsigned mr = rel ? 0xffffffffu : 0u, mv = valid ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(mr), "+v"(mv));  // (opaque: the compiler would turn the ANDs back into selects)
  vals[0] = __uint_as_float(__float_as_uint((float)(nb.nx - me.nx)) & mr);
  vals[1] = __uint_as_float(__float_as_uint((float)(nb.ny - me.ny)) & mr);
  vals[2] = __uint_as_float(__float_as_uint(nb.nsp - me.nsp) & mr);
  vals[3] = __uint_as_float(__float_as_uint(nb.nac - me.nac) & mr);
  vals[4] = __uint_as_float(__float_as_uint(nb.ndir - me.ndir) & mr);
  const unsigned one = 0x3f800000u, ts = (unsigned)nb.type_sig;
  vals[5] = __uint_as_float(ts & one & mv);
  vals[6] = __uint_as_float((0u - (ts & 1u)) & one & mv);
}

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

// =====================================================================================
//                   fast path: N <= 512, partial observations, K <= KMAX
// =====================================================================================

struct TcP4 {
  float2 p[4];
};
// positions of candidates j .. j+3 (j even; every replica's positions start 16-byte aligned): two
// ds_read_b128 with a wave-uniform address -- half the LDS cycles of four 8-byte reads, and the LDS
// pipe is what bounds pass B otherwise
__device__ __forceinline__ TcP4 tc_load4(const float2 *cxy, int j) {
  // (j is a multiple of 4 and every replica's positions start 16-byte aligned: say so, or a start index the compiler
  // cannot see through turns the two ds_read_b128 into eight ds_read_b32)
  const float4 *const q = (const float4 *)__builtin_assume_aligned(cxy + j, 16);
  const float4 a = q[0], b = q[1];
  TcP4 r;
  r.p[0] = make_float2(a.x, a.y); r.p[1] = make_float2(a.z, a.w);
  r.p[2] = make_float2(b.x, b.y); r.p[3] = make_float2(b.z, b.w);
  return r;
}

// rank of entry k in the reference's order (distance, then id): the entries are in ascending id
// order already, so entry j > i goes first only when it is STRICTLY closer.  Counting (one compare
// and two carry adds per pair) instead of a compare-exchange network: a 64-bit compare-exchange is
// a compare plus four v_cndmask, the slowest instruction class on gfx950 when they come in runs.
template <int KMAX>
__device__ __forceinline__ void tc_rank_entries(const unsigned (&sb)[KMAX], int (&rank)[KMAX]) {
#pragma unroll
  for (int k = 0; k < KMAX; ++k) rank[k] = k;
#pragma unroll
  for (int i = 0; i < KMAX; ++i)
#pragma unroll
    for (int j = i + 1; j < KMAX; ++j) {
      const int c = (sb[j] < sb[i]) ? 1 : 0;
      rank[i] += c;
      rank[j] -= c;
    }
}

template <int KMAX>
__device__ __forceinline__ void tc_knn_registers(const float2 *cxy, int ag, int N, int K, int (&nid)[KMAX],
                                                 int (&rank)[KMAX]) {
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  const float INF = __builtin_inff();
  // candidates are streamed four at a time, the next four positions being read from LDS while the
  // current four are processed (the search is latency-bound otherwise: one LDS round trip per group)

  // A. K+1 smallest squared distances over ALL agents of the replica (self contributes 0,
  //    agents out of the game contribute +inf): B[k] = med3(B[k-1], B[k], d2), one op per slot,
  //    no compares, no ids
  float B[KMAX + 1];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) B[k] = INF;
#define WD_TC_INSERT(d2v)                                                                         \
  do {                                                                                            \
    _Pragma("unroll") for (int k = KMAX; k >= 1; --k) B[k] = __builtin_amdgcn_fmed3f(B[k - 1], B[k], (d2v)); \
    B[0] = fminf(B[0], (d2v));                                                                    \
  } while (0)
  {
    const int ng = N >> 2;
    TcP4 nxt = tc_load4(cxy, 0);
    for (int g = 0; g < ng; ++g) {
      const TcP4 cur = nxt;
      nxt = tc_load4(cxy, 4 * g + 4);  // (the last prefetch lands in the padding behind the replica's positions)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dx = xi - cur.p[u].x, dy = yi - cur.p[u].y;
        const float d2 = dx * dx + dy * dy;
        WD_TC_INSERT(d2);
      }
    }
    for (int j = 4 * ng; j < N; ++j) {
      const float2 pj = cxy[j];
      const float dx = xi - pj.x, dy = yi - pj.y;
      const float d2 = dx * dx + dy * dy;
      WD_TC_INSERT(d2);
    }
  }
#undef WD_TC_INSERT
  __builtin_amdgcn_s_setprio(1);
  // B[k], k = 1..K are the K smallest squared distances to OTHER agents (B[0] is self or a
  // co-located twin).  T2 = the K-th of them.
  float T2 = INF;
#pragma unroll
  for (int k = 1; k <= KMAX; ++k) T2 = (k == K) ? B[k] : T2;
  // The reference orders by float32 sqrt distance and breaks ties by id (heapq.nsmallest is stable,
  // :435-437).  sqrt rounds, so a RANGE [T2lo, T2hi] of squared distances maps to the K-th distance
  // S = sqrtf(T2); it is derived exactly in float64 from the midpoints around S.
  float T2lo, T2hi;
  if (T2 == INF) {          // fewer than K candidates in the game: take them all
    T2lo = INF; T2hi = 3.0e38f;
  } else if (T2 == 0.0f) {  // K twins at distance 0
    T2lo = 0.0f; T2hi = 0.0f;
  } else {
    const float S = sqrtf(T2);
    const float Sup = __uint_as_float(__float_as_uint(S) + 1u), Sdn = __uint_as_float(__float_as_uint(S) - 1u);
    const double mhi = 0.5 * ((double)S + (double)Sup), mlo = 0.5 * ((double)S + (double)Sdn);
    // sqrtf(x) == S  <=>  mlo^2 < x < mhi^2  (midpoints squared are exact in float64 and are
    // never float32 values themselves)
    const double hi2 = mhi * mhi, lo2 = mlo * mlo;
    float th = (float)hi2, tl = (float)lo2;  // round to nearest, then step to the inside
    if ((double)th > hi2) th = __uint_as_float(__float_as_uint(th) - 1u);
    if ((double)tl < lo2) tl = __uint_as_float(__float_as_uint(tl) + 1u);
    T2hi = th;
    T2lo = tl;
  }
  // B. second pass: one 128-bit per-lane mask "inside or below the range".  Each candidate costs
  //    a squared distance, one compare and one shift-in-the-carry add (m = 2m + bit); no
  //    data-dependent addressing.  Candidate b of word w lands on bit (nb-1-b): undone with one
  //    bit-reverse per word.
  unsigned sel[4] = {0u, 0u, 0u, 0u};
  int n_upto = 0;
#define WD_TC_PUSH(m, d2v, thr, op) \
  asm("v_cmp_" op "_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2v), "v"(thr) : "vcc")
  {
#define WD_TC_PUSH4(m, g)                                                       \
  do {                                                                          \
    float d_[4];                                                                \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                             \
      const float dx = xi - (g).p[u].x, dy = yi - (g).p[u].y;                   \
      d_[u] = dx * dx + dy * dy;                                                \
    }                                                                           \
    WD_TC_PUSH(m, d_[0], T2hi, "le"); WD_TC_PUSH(m, d_[1], T2hi, "le");         \
    WD_TC_PUSH(m, d_[2], T2hi, "le"); WD_TC_PUSH(m, d_[3], T2hi, "le");         \
  } while (0)
    int w_first = 0;  // words already done by the interleaved loop below
    if (N >= 96) {
      // three full words at once: three INDEPENDENT compare / carry chains interleaved, so that one
      // chain's carry latency is covered by the other two (a single chain issues a dependent pair
      // per candidate)
      unsigned m3[3] = {0u, 0u, 0u};
#define WD_TC_PUSH12(g0, g1, g2)                                                  \
  do {                                                                            \
    float e_[3][4];                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                               \
      const float ax = xi - (g0).p[u].x, ay = yi - (g0).p[u].y;                   \
      const float bx = xi - (g1).p[u].x, by = yi - (g1).p[u].y;                   \
      const float cx = xi - (g2).p[u].x, cy = yi - (g2).p[u].y;                   \
      e_[0][u] = ax * ax + ay * ay; e_[1][u] = bx * bx + by * by; e_[2][u] = cx * cx + cy * cy; \
    }                                                                             \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                               \
      WD_TC_PUSH(m3[0], e_[0][u], T2hi, "le");                                    \
      WD_TC_PUSH(m3[1], e_[1][u], T2hi, "le");                                    \
      WD_TC_PUSH(m3[2], e_[2][u], T2hi, "le");                                    \
    }                                                                             \
  } while (0)
      TcP4 a0 = tc_load4(cxy, 0), a1 = tc_load4(cxy, 32), a2 = tc_load4(cxy, 64), b0, b1, b2;
      for (int b = 0; b < 32; b += 8) {
        b0 = tc_load4(cxy, b + 4); b1 = tc_load4(cxy, b + 36); b2 = tc_load4(cxy, b + 68);
        WD_TC_PUSH12(a0, a1, a2);
        a0 = tc_load4(cxy, b + 8); a1 = tc_load4(cxy, b + 40); a2 = tc_load4(cxy, b + 72);
        WD_TC_PUSH12(b0, b1, b2);
      }
#undef WD_TC_PUSH12
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const unsigned self_bit = ((ag >> 5) == w) ? (1u << (ag & 31)) : 0u;
        sel[w] = __brev(m3[w]) & ~self_bit;
        n_upto += __popc(sel[w]);
      }
      w_first = 3;
    }
    TcP4 ga = tc_load4(cxy, 32 * w_first), gb;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j0 = 32 * w;
      if (w >= w_first && j0 < N) {  // wave-uniform
        const int nb = min(32, N - j0);
        unsigned mu = 0u;
        int b = 0;
        // two groups of four per trip, ping-pong: the loads of one group are in flight while the
        // other is processed, and no register is copied
        for (; b + 8 <= nb; b += 8) {
          gb = tc_load4(cxy, j0 + b + 4);
          WD_TC_PUSH4(mu, ga);
          ga = tc_load4(cxy, j0 + b + 8);  // (at most 8 entries past the last candidate: padding)
          WD_TC_PUSH4(mu, gb);
        }
        if (b + 4 <= nb) {
          gb = tc_load4(cxy, j0 + b + 4);
          WD_TC_PUSH4(mu, ga);
          ga = gb;
          b += 4;
        }
        for (; b < nb; ++b) {  // (only the last word can have a remainder)
          const float2 pj = cxy[j0 + b];
          const float dx = xi - pj.x, dy = yi - pj.y;
          const float d2 = dx * dx + dy * dy;
          WD_TC_PUSH(mu, d2, T2hi, "le");
        }
        const unsigned self_bit = ((ag >> 5) == w) ? (1u << (ag & 31)) : 0u;
        sel[w] = (__brev(mu) >> (32 - nb)) & ~self_bit;
        n_upto += __popc(sel[w]);
      }
    }
#undef WD_TC_PUSH4
  }
  // Usually exactly K others are inside or below the range and the mask is the answer.  More
  // than K means several candidates share the K-th float32 distance: the reference keeps the
  // lowest ids among them.  Rare (a float32 sqrt tie at the cut), so the "strictly below" mask is
  // only built then.
  if (n_upto > K) {
    unsigned lo[4] = {0u, 0u, 0u, 0u};
    int c_less = 0;  // others strictly below the range
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j0 = 32 * w;
      if (j0 < N) {
        const int nb = min(32, N - j0);
        unsigned mb = 0u;
        for (int b = 0; b < nb; ++b) {
          const float2 pj = cxy[j0 + b];
          const float dx = xi - pj.x, dy = yi - pj.y;
          const float d2 = dx * dx + dy * dy;
          WD_TC_PUSH(mb, d2, T2lo, "lt");
        }
        lo[w] = (__brev(mb) >> (32 - nb)) & sel[w];
        c_less += __popc(lo[w]);
      }
    }
    int quota = K - c_less;  // members of the range still to take, ascending id
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned tie = sel[w] & ~lo[w];
      const int have_t = __popc(tie);
      if (have_t > quota) {  // keep the lowest `quota` set bits
        unsigned kept = 0u;
        for (int q = 0; q < quota; ++q) { const unsigned bit = tie & (0u - tie); kept |= bit; tie ^= bit; }
        tie = kept;
      }
      quota -= min(have_t, quota);
      sel[w] = lo[w] | tie;
    }
  }
#undef WD_TC_PUSH
  // C. peel the (at most K) ids off the mask in ascending order; read their positions (all reads in
  //    flight together), rebuild the distances and form 64-bit keys (float bits of sqrt(d2) << 32 |
  //    id); sort
  int jj[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int which = sel[0] ? 0 : sel[1] ? 1 : sel[2] ? 2 : sel[3] ? 3 : 4;
    const unsigned cur = sel[0] ? sel[0] : sel[1] ? sel[1] : sel[2] ? sel[2] : sel[3];
    jj[k] = (which < 4) ? which * 32 + (__ffs(cur) - 1) : -1;
    const unsigned cleared = cur & (cur - 1u);
    sel[0] = (which == 0) ? cleared : sel[0];
    sel[1] = (which == 1) ? cleared : sel[1];
    sel[2] = (which == 2) ? cleared : sel[2];
    sel[3] = (which == 3) ? cleared : sel[3];
  }
  float2 pp[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) pp[k] = cxy[jj[k] < 0 ? ag : jj[k]];
  unsigned sb[KMAX];  // float bits of the float32 distance (>= 0: they order like unsigned integers)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const float dx = xi - pp[k].x, dy = yi - pp[k].y;
    sb[k] = (jj[k] >= 0) ? __float_as_uint(sqrtf(dx * dx + dy * dy)) : 0x7f800000u;
    nid[k] = jj[k];
  }
  tc_rank_entries<KMAX>(sb, rank);
}

// ---- exact fallback for more than 128 candidates (tc_knn_registers keeps a 128-bit mask): K passes,
// each picks the smallest (float32 distance, index) key above the previous one.  Slow (K x N square
// roots) and rare: only a lane with three candidates inside two key buckets at the cut gets here.
template <int KMAX>
__device__ __forceinline__ void tc_knn_scan(const float2 *cxy, int ag, int N, int K, int (&nid)[KMAX]) {
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  float pd = -1.0f;
  int pj = -1;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    float best = __builtin_inff();
    int bj = -1;
    for (int j = 0; j < N; ++j) {
      const float2 pc = cxy[j];
      const float dx = xi - pc.x, dy = yi - pc.y;
      const float d = sqrtf(dx * dx + dy * dy);
      const bool above = (d > pd) || (d == pd && j > pj);
      if (j != ag && above && d < best) { best = d; bj = j; }
    }
#pragma unroll
    for (int q = 0; q < KMAX; ++q) nid[q] = (q == k) ? bj : nid[q];
    if (bj < 0) break;  // fewer than K candidates (the remaining entries stay -1)
    pd = best;
    pj = bj;
  }
}

// ---- neighbour search in ONE pass over the candidates: the candidate's id rides in the low 7 bits
// of its squared distance (key = d2 bits with the low 7 bits replaced by j; non-negative floats order
// like unsigned integers) and a v_med3_u32 chain keeps the K+3 smallest keys, so the ids come out of
// the chain itself -- no second pass that rebuilds every distance to form a mask, no peeling of the
// mask.  The 7 dropped bits make the chain's order approximate (buckets of 128 ulps of d2); the
// exact answer is rebuilt from it:
//   * with b = bucket of the K-th other agent in chain order, a candidate whose bucket is >= b + 2 is
//     more than 128 ulps of d2 farther than each of the first K, i.e. strictly farther in float32
//     sqrt too: it cannot be among the K nearest.  The answer is a subset of {bucket <= b + 1};
//   * nearly always the first K+1 entries are far enough apart for the chain order to be the exact
//     order (see "apart" below) and nothing more is computed.  Otherwise:
//   * the chain tracks K+2 other agents.  If the last of them has a bucket >= b + 2, the subset is
//     inside the first K+1 tracked entries.  The exact keys (float32 distance, id) of the first K are
//     rebuilt and ranked by counting; when the (K+1)-th sits in the uncertain buckets (~3e-4 per
//     agent) it is ranked against them as well, and the entries of rank < K are the answer, in the
//     reference's order;
//   * otherwise (three candidates within 256 ulps of d2 at the cut: ~1e-7 per agent) the lane
//     returns false and repeats the search with tc_knn_registers.  It has to be that rare: a
//     wavefront that repeats the search does so alone, latency-bound, and the whole launch waits for
//     it (with one look-ahead entry less, ~9 of 4000 wavefronts did, and the tick got 5 us longer).
__device__ __forceinline__ unsigned tc_umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ---- the insertion chain over the candidates [j0, j1) (j0 a multiple of 4): L = self + K others + the look-ahead entries
template <int L, int IDB>
__device__ __forceinline__ void tc_chain_range(const float2 *cxy, float xi, float yi, int j0, int j1, unsigned (&S)[L]) {
  constexpr unsigned IDM = (1u << IDB) - 1u;
#pragma unroll
  for (int k = 0; k < L; ++k) S[k] = 0xffffffffu;
#define WD_TC_INSERT_KEY(d2v, jv)                                                          \
  do {                                                                                     \
    const unsigned key_ = (__float_as_uint(d2v) & ~IDM) | (unsigned)(jv);                  \
    _Pragma("unroll") for (int k = L - 1; k >= 1; --k) S[k] = tc_umed3(S[k - 1], S[k], key_); \
    S[0] = min(S[0], key_);                                                                \
  } while (0)
  const int g0 = j0 >> 2, ng = j1 >> 2;
  // groups of four candidates, two groups per trip, ping-pong: the positions of one group are in flight while the
  // other goes through the chain, and no register is copied (a "load the next group, then rotate" loop is what
  // the optimiser turns back into "load at the top, wait, use" when the start index is not a constant)
#define WD_TC_INSERT_GROUP(grp, gidx)                                  \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                      \
    const float dx = xi - (grp).p[u].x, dy = yi - (grp).p[u].y;        \
    const float d2 = dx * dx + dy * dy;                                \
    WD_TC_INSERT_KEY(d2, 4 * (gidx) + u);                              \
  }
  // the second half of the chain runs at the lowest priority, like the phases after the search
  // (the caller entered at 2): measured 36.5 -> 35.6 us per tick with 1 here, another 0.2 us with
  // 0 here and after the search; dropping after 1/8, 1/4 or 3/4 of the candidates, or not at
  // all, is 0.1 .. 1 us slower
  const int g_mid = (g0 + ng) >> 1;
  TcP4 ga = tc_load4(cxy, 4 * g0), gb;
  int g = g0;
  for (; g + 2 <= ng; g += 2) {
    gb = tc_load4(cxy, 4 * g + 4);
    asm volatile("" ::: "memory");   // (keeps the load above the work below)
    if (g >= g_mid && g < g_mid + 2) __builtin_amdgcn_s_setprio(0);
    WD_TC_INSERT_GROUP(ga, g);
    ga = tc_load4(cxy, 4 * g + 8);   // (the last prefetch lands in the padding behind the replica's positions)
    asm volatile("" ::: "memory");
    WD_TC_INSERT_GROUP(gb, g + 1);
  }
  if (g < ng) {
    if (g >= g_mid) __builtin_amdgcn_s_setprio(0);
    WD_TC_INSERT_GROUP(ga, g);
  }
#undef WD_TC_INSERT_GROUP
  for (int j = max(4 * ng, j0); j < j1; ++j) {
    const float2 pj = cxy[j];
    const float dx = xi - pj.x, dy = yi - pj.y;
    const float d2 = dx * dx + dy * dy;
    WD_TC_INSERT_KEY(d2, j);
  }
#undef WD_TC_INSERT_KEY
}

// ---- the L smallest of the union of two ascending lists of L keys (this lane's S and the partner's P), ascending.
// Both lists are padded to W = 16 (32, 64) entries with 0xffffffff -- still ascending --, then
// c[k] = min(S[k], P[W-1-k]) are the W smallest of the 2W (an ascending against a descending sequence: the result
// is bitonic) and a bitonic merge network sorts them: log2(W) x W/2 compare-exchanges (64 min / max for L <= 16)
// against L x L median-of-three for inserting the partner's keys one by one.  (Padding AFTER the min step would
// not do: a bitonic sequence followed by maxima is not bitonic.)
__device__ __forceinline__ void tc_cex(unsigned &a, unsigned &b) {
  const unsigned lo = min(a, b), hi = max(a, b);
  a = lo;
  b = hi;
}
template <int L>
__device__ __forceinline__ void tc_merge_sorted(unsigned (&S)[L], const unsigned (&P)[L]) {
  constexpr int W = (L <= 8) ? 8 : (L <= 16) ? 16 : (L <= 32) ? 32 : 64;
  unsigned c[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const int q = W - 1 - k;  // partner entry
    c[k] = (k < L && q < L) ? min(S[k], P[q]) : (k < L) ? S[k] : (q < L) ? P[q] : 0xffffffffu;
  }
#pragma unroll
  for (int stride = W / 2; stride >= 1; stride >>= 1)
#pragma unroll
    for (int k = 0; k < W; ++k)
      if ((k & stride) == 0) tc_cex(c[k], c[k + stride]);
#pragma unroll
  for (int k = 0; k < L; ++k) S[k] = c[k];
}

// ---- PREFILTERED search for replicas of more than 128 agents (round 4; one replica per block, K <= 12, used while at
// least WD_TC_PRE_MIN_LIVE agents are in the game).  The chain costs 13 median-of-three (~3 cycles each with the VALU
// saturated) + 6 cheap instructions per candidate and searcher and is VALU-bound on all sixteen wavefronts of a
// 1005-agent replica: 80 % of its tick.  Agents move little per tick, so the searcher's K + 3 nearest others of the
// PREVIOUS tick (32 bytes per agent in HBM, `knn_prev`) give a radius that holds the K nearest now (tc_knn_bound16):
//   pass 1  every candidate: squared distance and ONE compare against the radius, shifted into a per-lane bit mask
//           (v_cmp + v_addc: mask = 2 mask + bit) -- 7 instructions, 5 of them float32 add / mul at ~1.2 cycles;
//   pass 2  the candidates whose bit is set (~15 per lane) go through the chain: 128 candidates = four mask words at
//           a time, word by word every lane pops its own lowest set bit (lanes that ran out insert the pad position
//           at +inf), as many trips as the fullest lane of the wavefront needs, U candidates per trip with the
//           next trip's positions in flight.
// At ~100 candidates this was measured and NOT adopted (a wash: pass 2 does not shrink with the number of
// candidates); the break-even is at a few hundred.  What comes out is the K set the full chain
// gives: the result is accepted only if the K-th other agent found lies at least TWO key buckets inside the radius
// (`held` at the call site), so every candidate that was NOT listed is strictly farther in float32 distance than the
// K-th -- it can neither enter the K set nor tie with its last member -- and every candidate that can is listed and
// ranked exactly by tc_resolve_keys.  The look-ahead entries may differ from the full chain's (an unlisted
// candidate reads +inf there), so the 'apart' / 'simple' shortcuts can fire where the full chain would have run the
// exact ranking: that changes the work, not the K set, because a look-ahead entry only ever decides whether keys
// INSIDE the listed range need the exact comparison, and an entry at +inf says "no tie beyond here", which is true.
// That the radius really held the K nearest is CHECKED afterwards (the K-th other agent found must lie inside it), so
// the content of `knn_prev` is only a hint: stale, restored or overwritten rows cost time (the wavefront repeats the
// search with the full chain), never exactness.
#define WD_TC_PRE_MIN_LIVE 200

// The radius: 1.15 x (K + 3) / n x the LARGEST current squared distance to the n remembered agents that are still in
// the game (their positions read NaN otherwise: v_max_f32 skips a NaN); none when fewer than 5 are.  With all K + 3 in
// the game this lists a few more than K + 3 candidates, so the chain refills the remembered set with the K + 3 nearest
// every tick; after remembered agents were tagged out the radius grows by the share that is missing.  A heuristic on
// purpose (the radius that provably holds K candidates lists about K of them, leaves no spares to remember, and the
// next tag leaves a stand-in from across the arena as the bound: experiments/offline/knn_prefilter_sim2.py), checked by
// the caller.  Returns the bits of the radius, 0x7f800000 = none.
template <int KMAX>
__device__ __forceinline__ unsigned tc_knn_bound16(const float2 *xy_by_id, int pad_id, float xi, float yi, uint4 pa, uint4 pb,
                                                   int K) {
  constexpr int M = KMAX + 3;
  static_assert(M <= 15, "the remembered neighbours are sixteen 16-bit ids per agent, K + 3 of them in use");
  float2 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13, p14;
#define WD_TC_PREV_POS(k, vec, word) \
  if (k < M) p##k = xy_by_id[min((vec.word >> (16 * (k & 1))) & 0xffffu, (unsigned)pad_id)]  // out of the game / none / garbage: NaN
  WD_TC_PREV_POS(0, pa, x); WD_TC_PREV_POS(1, pa, x); WD_TC_PREV_POS(2, pa, y); WD_TC_PREV_POS(3, pa, y);
  WD_TC_PREV_POS(4, pa, z); WD_TC_PREV_POS(5, pa, z); WD_TC_PREV_POS(6, pa, w); WD_TC_PREV_POS(7, pa, w);
  WD_TC_PREV_POS(8, pb, x); WD_TC_PREV_POS(9, pb, x); WD_TC_PREV_POS(10, pb, y); WD_TC_PREV_POS(11, pb, y);
  WD_TC_PREV_POS(12, pb, z); WD_TC_PREV_POS(13, pb, z); WD_TC_PREV_POS(14, pb, w);
#undef WD_TC_PREV_POS
  float far = 0.0f;
  unsigned n = 0u;
#define WD_TC_PREV_DIST(k)                                                                                  \
  if (k < M) {                                                                                              \
    const float dx = xi - p##k.x, dy = yi - p##k.y;                                                         \
    const float d2 = dx * dx + dy * dy;                                                                     \
    far = fmaxf(far, d2); /* (maxnum: a NaN operand is ignored) */                                          \
    asm("v_cmp_o_f32 vcc, %1, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n) : "v"(d2) : "vcc");       \
  }
  WD_TC_PREV_DIST(0) WD_TC_PREV_DIST(1) WD_TC_PREV_DIST(2) WD_TC_PREV_DIST(3) WD_TC_PREV_DIST(4)
  WD_TC_PREV_DIST(5) WD_TC_PREV_DIST(6) WD_TC_PREV_DIST(7) WD_TC_PREV_DIST(8) WD_TC_PREV_DIST(9)
  WD_TC_PREV_DIST(10) WD_TC_PREV_DIST(11) WD_TC_PREV_DIST(12) WD_TC_PREV_DIST(13) WD_TC_PREV_DIST(14)
#undef WD_TC_PREV_DIST
  const float T = far * (1.15f * (float)(K + 3)) * __builtin_amdgcn_rcpf((float)n);
  return (n >= 5u) ? __float_as_uint(T) : 0x7f800000u;
}

// S: the L smallest keys among the candidates with d2 <= Tf, ascending; returns the (L+1)-th (one more id to remember)
template <int L, int IDB, int U>
__device__ __forceinline__ unsigned tc_chain_prefiltered(const float2 *cxy, float xi, float yi, int N, float Tf, int pad_idx,
                                                         unsigned (&S)[L]) {
  constexpr unsigned IDM = (1u << IDB) - 1u;
  unsigned extra = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < L; ++k) S[k] = 0xffffffffu;
#define WD_TC_MASK_PUSH(m, d2v) \
  asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2v), "v"(Tf) : "vcc")
#define WD_TC_MASK_PUSH4(m, grp)                                       \
  do {                                                                 \
    float d_[4];                                                       \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                    \
      const float dx = xi - (grp).p[u].x, dy = yi - (grp).p[u].y;      \
      d_[u] = dx * dx + dy * dy;                                       \
    }                                                                  \
    WD_TC_MASK_PUSH(m, d_[0]); WD_TC_MASK_PUSH(m, d_[1]);              \
    WD_TC_MASK_PUSH(m, d_[2]); WD_TC_MASK_PUSH(m, d_[3]);              \
  } while (0)
#define WD_TC_POP2(ix, px)                                                                     \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                              \
    const bool have_ = (mw != 0u);                                                             \
    ix[u] = have_ ? (unsigned)(top - (__ffs(mw) - 1)) : (unsigned)pad_idx;                     \
    mw &= mw - 1u;                                                                             \
    px[u] = cxy[ix[u]];                                                                        \
  }
#ifdef WD_TC_PROBES
  int probe_trips = 0;
#endif
  for (int c0 = 0; c0 < N; c0 += 128) {  // wave-uniform
    // ---- pass 1 of this chunk: candidate b of word w (candidates c0 + 32 w .. + nb - 1) ends on bit nb - 1 - b
    unsigned mask[4] = {0u, 0u, 0u, 0u};
    TcP4 ga = tc_load4(cxy, c0), gb;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j0 = c0 + 32 * w;
      if (j0 < N) {  // wave-uniform
        const int nb = min(32, N - j0);
        unsigned mu = 0u;
        int b = 0;
        for (; b + 8 <= nb; b += 8) {  // two groups of four per trip, ping-pong
          gb = tc_load4(cxy, j0 + b + 4);
          WD_TC_MASK_PUSH4(mu, ga);
          ga = tc_load4(cxy, j0 + b + 8);  // (at most 8 entries past the last candidate: padding)
          WD_TC_MASK_PUSH4(mu, gb);
        }
        if (b + 4 <= nb) {
          gb = tc_load4(cxy, j0 + b + 4);
          WD_TC_MASK_PUSH4(mu, ga);
          ga = gb;
          b += 4;
        }
        for (; b < nb; ++b) {  // (only the last word can have a remainder)
          const float2 pj = cxy[j0 + b];
          const float dx = xi - pj.x, dy = yi - pj.y;
          const float d2 = dx * dx + dy * dy;
          WD_TC_MASK_PUSH(mu, d2);
        }
        mask[w] = mu;
      }
    }
    // ---- pass 2 of this chunk
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j0 = c0 + 32 * w;
      if (j0 < N) {  // wave-uniform
        const int top = j0 + min(32, N - j0) - 1;  // the candidate on bit 0
        unsigned mw = mask[w];
        unsigned idx[U], idn[U];
        float2 pj[U], pn[U];
        bool more = __ballot(mw != 0u) != 0ull;  // wave-uniform: as many trips as the fullest lane needs
        if (more) {
          WD_TC_POP2(idn, pn);
          while (more) {
#ifdef WD_TC_PROBES
            ++probe_trips;
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) { idx[u] = idn[u]; pj[u] = pn[u]; }
            more = __ballot(mw != 0u) != 0ull;
            if (more) { WD_TC_POP2(idn, pn); }
            asm volatile("" ::: "memory");  // (keeps the reads above the work below)
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const float dx = xi - pj[u].x, dy = yi - pj[u].y;
              const float d2 = dx * dx + dy * dy;
              const unsigned key_ = (__float_as_uint(d2) & ~IDM) | idx[u];
              extra = tc_umed3(S[L - 1], extra, key_);
#pragma unroll
              for (int k = L - 1; k >= 1; --k) S[k] = tc_umed3(S[k - 1], S[k], key_);
              S[0] = min(S[0], key_);
            }
          }
        }
      }
    }
  }
#ifdef WD_TC_PROBES
  WD_TC_PROBE_VAL(18, probe_trips);
#endif
#undef WD_TC_POP2
#undef WD_TC_MASK_PUSH4
#undef WD_TC_MASK_PUSH
  return extra;
}

// ---- exact resolution for ONE searcher by the WHOLE wavefront (replicas of more than 128 agents; the K-pass scan
// above took ~1.5 ms for a 1005-agent replica -- ten times the rest of the tick -- and a launch of 2000 replicas hit it
// in ~11 wavefronts, so the launch waited for it on every tick).  The chain already located the cut: the answer lies
// in the key buckets <= `zone_hi` (= the bucket of the K-th other agent + 1).  Lane l looks at candidates l, l + 64,
// ...; the candidates inside the zone are packed (ballot + mbcnt: ascending index order) into a list in the
// wavefront's staging buffer; each lane builds the exact (float32 distance, index) key of one listed candidate and
// counts the smaller keys (LDS broadcast reads); rank r < K writes its index to out[r].  ~25 instructions per 64
// candidates + ~4 per listed candidate.  Returns the number of candidates in the zone (> 64: not resolved, the
// caller falls back to the scan -- a pile of agents on one spot).
__device__ __forceinline__ int tc_zone_resolve(const float2 *cxy, int n_cand, float sx, float sy, int self, unsigned zone_hi,
                                               int idb, int K, unsigned char *scratch, int lane) {
  unsigned short *const zl = (unsigned short *)scratch;                  // [64] candidate indices inside the zone
  unsigned long long *const keys = (unsigned long long *)(scratch + 128);  // [64] exact keys
  unsigned short *const out = (unsigned short *)(scratch + 128 + 512);     // [K] the K nearest in the reference's order
  int cnt = 0;  // wave-uniform
  for (int j0 = 0; j0 < n_cand; j0 += 64) {
    const int j = j0 + lane;
    bool in = false;
    if (j < n_cand) {
      const float2 pj = cxy[j];
      const float dx = sx - pj.x, dy = sy - pj.y;
      const float d2 = dx * dx + dy * dy;
      in = (j != self) && ((__float_as_uint(d2) >> idb) <= zone_hi);
    }
    const unsigned long long m = __ballot(in);
    const int at = cnt + __popcll(m & ((1ull << lane) - 1ull));
    if (in && at < 64) zl[at] = (unsigned short)j;
    cnt += __popcll(m);
  }
  if (cnt > 64) return cnt;
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  unsigned long long mine = ~0ull;
  if (lane < cnt) {
    const int j = zl[lane];
    const float2 pj = cxy[j];
    const float dx = sx - pj.x, dy = sy - pj.y;
    mine = ((unsigned long long)__float_as_uint(sqrtf(dx * dx + dy * dy)) << 32) | (unsigned)j;
    keys[lane] = mine;
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  int r = 0;
  for (int u = 0; u < cnt; ++u) r += (keys[u] < mine) ? 1 : 0;  // (wave-uniform address: a broadcast read)
  if (lane < cnt && r < K) out[r] = (unsigned short)(mine & 0xffffu);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  return cnt;
}

// nid / rank have KMAX + 1 entries: entry k is one of the K nearest iff rank[k] < K
// `in_order`: entry k is the k-th nearest for every k < K (rank[k] == k), and all K of them exist
// IDB = id bits in the key: 7 for up to 128 candidates, 9 for up to 512, 10 for up to 1024 (buckets of 2^IDB ulps of d2;
// the argument above holds for any bucket width: two buckets apart is more than 2^IDB ulps of d2, i.e.
// at least 2^(IDB-1) - 1 ulps of the float32 distance)
// S: the L smallest keys in ascending order (tc_chain_all / tc_chain_prefiltered); o: the same without the
// agent's own entry (the caller remembers their ids for the next tick's bound)
template <int KMAX, int IDB, int L>
__device__ __forceinline__ bool tc_resolve_keys(const float2 *cxy, int ag, int K, const unsigned (&S)[L],
                                                unsigned (&o)[L - 1], int (&nid)[KMAX + 1], int (&rank)[KMAX + 1],
                                                bool &in_order) {
  static_assert(L >= KMAX + 3, "self + K others + two look-ahead entries");
  constexpr unsigned IDM = (1u << IDB) - 1u;
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  // drop the agent's own entry (d2 = 0 exactly: key == ag).  It is the first entry unless a twin with
  // a lower id sits on the same spot.
  // (values first: with two producers of S -- the full and the prefiltered chain -- a select between two ELEMENTS of S
  // becomes a select between their addresses, which keeps the two elements in scratch memory for the whole search)
  unsigned sv[L];
#pragma unroll
  for (int k = 0; k < L; ++k) {
    sv[k] = S[k];
    asm volatile("" : "+v"(sv[k]));
  }
  if (__ballot(sv[0] != (unsigned)ag) == 0ull) {  // wave-uniform
#pragma unroll
    for (int k = 0; k < L - 1; ++k) o[k] = sv[k + 1];
  } else {
    bool after = false;
#pragma unroll
    for (int k = 0; k < L - 1; ++k) {
      after = after || (sv[k] == (unsigned)ag);
      o[k] = after ? sv[k + 1] : sv[k];
    }
  }
  // Chain order IS the reference's order wherever neighbouring keys are >= 383 apart: then their
  // buckets differ by two or more (the ids in the low bits move a key by < 128), so the squared
  // distances differ by more than 128 ulps and the float32 distances strictly.  When that holds for
  // the first K entries and the one after them (all but ~0.2 % of the agents) the low bits of the
  // first K keys are the answer as they stand -- no positions re-read, no square roots, no ranking.
  unsigned gap = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) gap = min(gap, o[k + 1] - o[k]);  // (two slots without a candidate are 0 apart)
  unsigned oKth = o[KMAX - 1];  // the K-th other agent in chain order
#pragma unroll
  for (int k = 0; k < KMAX - 1; ++k) oKth = (k == K - 1) ? o[k] : oKth;
  const bool apart = (gap >= 3u * (IDM + 1u) - 1u) && (oKth < 0x7f800000u);  // (and K others are in the game at all)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    nid[k] = (k < K) ? (int)(o[k] & IDM) : -1;
    rank[k] = k;
  }
  nid[KMAX] = -1;
  rank[KMAX] = KMAX;
  bool exact = true;
  in_order = apart;
  // A lane that is not `apart` nearly always has ONE pair of neighbouring keys that is too close, with clear gaps on
  // either side of it: only that pair's order is open, and it is settled by comparing the two exact keys
  // (float32 distance, index) -- a handful of instructions instead of the ranking of all K entries below, which the
  // whole wavefront used to run with a few lanes active (+4.8 k cycles for the 4 % of the wavefronts that held such
  // a lane: exactly the wavefronts the launch ends with, profiles/r04_phase_profile_*.txt).  The ranking remains
  // for runs of three or more close keys, a close pair at the cut with a close look-ahead entry behind it, and
  // fewer than K agents in the game.
  bool simple = false;
  if (!apart && oKth < 0x7f800000u) {
    constexpr unsigned THR = 3u * (IDM + 1u) - 1u;
    unsigned cm = 0u;  // bit k: keys k and k + 1 are close (k = K: the pair behind the cut)
#pragma unroll
    for (int k = 0; k <= KMAX; ++k)
      if (k <= K) cm |= ((o[k + 1] - o[k] < THR) ? 1u : 0u) << k;
    unsigned rel = cm & ((1u << K) - 1u);
    const bool look_close = ((cm >> K) & 1u) != 0u;
    simple = ((rel & (rel >> 1)) == 0u) && !(((rel >> (K - 1)) & 1u) != 0u && look_close);
    if (simple) {
      WD_TC_PROBE_VAL(23, 1);
      while (rel) {
        const int q = __ffs(rel) - 1;  // the pair (q, q + 1)
        rel &= rel - 1u;
        unsigned ka = o[0], kb = o[1];
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
          ka = (q == k) ? o[k] : ka;
          kb = (q == k) ? o[k + 1] : kb;
        }
        const int ia = (int)(ka & IDM), ib = (int)(kb & IDM);
        const float2 pa = cxy[ia], pb = cxy[ib];
        const float ax = xi - pa.x, ay = yi - pa.y, bx = xi - pb.x, by = yi - pb.y;
        const unsigned sa = __float_as_uint(sqrtf(ax * ax + ay * ay)), sb = __float_as_uint(sqrtf(bx * bx + by * by));
        // (indices are in ascending id order: the later entry goes first only when it is strictly closer)
        if (sb < sa || (sb == sa && ib < ia)) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) nid[k] = (k == q) ? ib : (k == q + 1 && k < K) ? ia : nid[k];
        }
      }
      in_order = true;
    }
  }
  if (!apart && !simple) {
    WD_TC_PROBE_VAL(22, 1);
    // the K-th, (K+1)-th and (K+2)-th other agent in chain order
    unsigned oK = o[KMAX - 1], oExtra = o[KMAX], oLook = o[KMAX + 1];
#pragma unroll
    for (int k = 0; k < KMAX - 1; ++k) {
      oK = (k == K - 1) ? o[k] : oK;
      oExtra = (k == K - 1) ? o[k + 1] : oExtra;
      oLook = (k == K - 1) ? o[k + 2] : oLook;
    }
    const unsigned INVALID = 0x7f800000u;  // agents out of the game sit at +inf; unused slots are above
    const unsigned cut = (oK >> IDB) + 2u;   // first bucket that is certainly outside
    exact = (oK >= INVALID) || ((oLook >> IDB) >= cut);
    // positions of the first K entries (all reads in flight together), exact keys, ranks
    float2 pp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const bool valid = (k < K) && (o[k] < INVALID);
      nid[k] = valid ? (int)(o[k] & IDM) : -1;
      pp[k] = cxy[valid ? nid[k] : ag];
    }
    unsigned long long key64[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float dx = xi - pp[k].x, dy = yi - pp[k].y;
      const unsigned sb = (nid[k] >= 0) ? __float_as_uint(sqrtf(dx * dx + dy * dy)) : 0x7f800000u;
      key64[k] = ((unsigned long long)sb << 32) | (unsigned)nid[k];
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) rank[k] = k;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
#pragma unroll
      for (int j = i + 1; j < KMAX; ++j) {
        const int c = (key64[j] < key64[i]) ? 1 : 0;
        rank[i] += c;
        rank[j] -= c;
      }
    nid[KMAX] = -1;
    rank[KMAX] = KMAX;
    // the (K+1)-th entry is inside the uncertain buckets (~3e-4 per agent): it competes with the first K
    if (oK < INVALID && oExtra < INVALID && (oExtra >> IDB) < cut) {
      const int idE = (int)(oExtra & IDM);
      const float2 pe = cxy[idE];
      const float dx = xi - pe.x, dy = yi - pe.y;
      const unsigned long long keyE = ((unsigned long long)__float_as_uint(sqrtf(dx * dx + dy * dy)) << 32) | (unsigned)idE;
      int rE = K;
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const int c = (i < K && keyE < key64[i]) ? 1 : 0;
        rank[i] += c;
        rE -= c;
      }
      nid[KMAX] = idE;
      rank[KMAX] = rE;
    }
  }
  return exact;
}

// Stream `n` dwords from a wavefront's staging buffer to global memory as one contiguous run.
// The producer placed dword i of the run at stage[mis + i], mis = (address of dst / 4) & 3, so the
// 16-byte vectors of the run are 16-byte aligned in LDS and in memory alike; the <= 3 dwords before
// the first / after the last aligned vector go out as single dwords.
__device__ __forceinline__ void tc_flush_run(const float *stage, float *dst, int n, int lane) {
  const int mis = (int)(((size_t)dst >> 2) & 3);
  const int head = min(n, (4 - mis) & 3);
  const int nvec = (n - head) >> 2;
  const int tail0 = head + 4 * nvec;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f *sv = (const v4f *)(stage + mis + head);
  v4f *dv = (v4f *)(dst + head);
  // write-through (sc1) 16-byte stores: the rows go to memory as they are produced instead of piling
  // up dirty in the L2 until the kernel-boundary write-back (47.0 -> 45.3 us per tick; 16-byte sc1
  // stores cost the same as plain ones, narrower ones do not).  Three vectors per lane per trip, LDS
  // reads in flight together; whole 64-lane groups are stored under wave-uniform branches, only the
  // last partial group is exec-masked.
#define WD_TC_STORE_WT(ptr, val) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(val) : "memory")
  for (int base = 0; base < nvec; base += 192) {
    const int nrem = nvec - base;  // wave-uniform
    const int q = base + lane;
    const v4f v0 = sv[min(q, nvec - 1)], v1 = sv[min(q + 64, nvec - 1)], v2 = sv[min(q + 128, nvec - 1)];
    if (nrem >= 64) WD_TC_STORE_WT(&dv[q], v0);
    else if (lane < nrem) WD_TC_STORE_WT(&dv[q], v0);
    if (nrem >= 128) WD_TC_STORE_WT(&dv[q + 64], v1);
    else if (lane + 64 < nrem) WD_TC_STORE_WT(&dv[q + 64], v1);
    if (nrem >= 192) WD_TC_STORE_WT(&dv[q + 128], v2);
    else if (lane + 128 < nrem) WD_TC_STORE_WT(&dv[q + 128], v2);
  }
#undef WD_TC_STORE_WT
  if (lane < head) dst[lane] = stage[mis + lane];
  if (lane < n - tail0) dst[tail0 + lane] = stage[mis + tail0 + lane];
}

// nearest_neighbor_ids rows from the block-local 16-bit ids in LDS: n dwords starting at `dst`, dword i =
// id i of the run (0xffff -> -1; block-local -> replica-local when a block holds several replicas);
// aligned 16-byte write-through stores, single dwords before / after the aligned part.
__device__ __forceinline__ void tc_flush_ids(const unsigned short *src, int *dst, int n, int lane, int row0, int N,
                                             float invK, float invN, bool one_replica) {
  const int mis = (int)(((size_t)dst >> 2) & 3);
  const int head = min(n, (4 - mis) & 3);
  const int nvec = (n - head) >> 2;
  const int tail0 = head + 4 * nvec;
  // replica-local id of run element i holding block-local id v (v < 0: none)
  auto local = [&](int v, int i) -> int {
    if (one_replica) return v;
    const int row = row0 + (int)(((float)i + 0.5f) * invK);  // i / K, exact (see the gather)
    const int sub = (int)(((float)row + 0.5f) * invN) * N;
    return v - (v >= 0 ? sub : 0);
  };
  typedef int v4i __attribute__((ext_vector_type(4)));
  for (int q = lane; q < nvec; q += 64) {
    const int i = head + 4 * q;
    const int r0 = (int)(short)src[i], r1 = (int)(short)src[i + 1], r2 = (int)(short)src[i + 2],
              r3 = (int)(short)src[i + 3];
    const v4i v = {local(r0, i), local(r1, i + 1), local(r2, i + 2), local(r3, i + 3)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
  }
  if (lane < head) dst[lane] = local((int)(short)src[lane], lane);
  if (lane < n - tail0) dst[tail0 + lane] = local((int)(short)src[tail0 + lane], tail0 + lane);
}

// rows of a wavefront's staging buffer: the host sizes the buffer with the same formula
// (envs/tag_continuous.py: lds_bytes_fast)
#define WD_TC_STAGE_TARGET 5400  // bytes of rows per wavefront (19 rows of 71 floats); half of it for blocks of more
                                 // than four wavefronts (replicas of more than 256 agents), whose LDS also holds
                                 // the larger replica
__device__ __forceinline__ int tc_stage_rows(int row_dwords, int n_waves) {
  const int target = (n_waves > 4) ? WD_TC_STAGE_TARGET / 2 : WD_TC_STAGE_TARGET;
  return max(1, min(64, target / (4 * row_dwords)));
}

// LDS of the fast path.  The per-trip area doubles as the two probability slabs of the fused tick,
// which are dead before the move phase writes it.
struct TcFastLds {
  TcFeatArrays feat;     // [A] + [A] observation features, two 16-byte halves per agent
  float2 *xy;            // [epb][NP] positions after the move (x = +BIG for agents out of the game); NP = N rounded up
                         // to a multiple of 4, plus 8 entries of padding that the search's prefetches may read
  int *sig;              // [A] still_in_the_game before this tick's tagging
  int *tagcnt;           // [A] tags credited to a tagger this tick
  float2 *xyc;           // one replica per block: [NP] positions of the agents IN THE GAME, packed in ascending id order
                         // (the candidates and the searchers of the neighbour search); else == xy
  short *cid;            // one replica per block: [1 + N] cid[1 + c] = id of the c-th agent in the game, cid[0] = -1
  unsigned short *ids;   // [A][K] block-local neighbour indices (0xffff = none)
  float *stage;          // [n_waves][stage_dwords] wave-private staging buffers
  int stage_dwords;
  TcTables tb;
};

__device__ __forceinline__ TcFastLds tc_carve_fast(unsigned char *p0, int epb, int N, int K, int n_waves,
                                                   size_t min_area_bytes, bool compact) {
  TcFastLds l;
  const size_t A = (size_t)epb * N;
  const int F = 7 * K + 1;
  size_t off = 0;
  l.feat.a = (TcFeatA *)(p0 + off); off += sizeof(TcFeatA) * A;
  l.feat.b = (TcFeatB *)(p0 + off); off += sizeof(TcFeatB) * A;
  l.xy = (float2 *)(p0 + off); off += 8 * (size_t)epb * (((N + 3) & ~3) + 8);  // see TcFastLds::xy
  l.sig = (int *)(p0 + off); off += 4 * A;
  l.tagcnt = (int *)(p0 + off); off += 4 * A;
  l.xyc = l.xy;
  l.cid = nullptr;
  if (compact) {  // (the host adds the same bytes: envs/tag_continuous.py lds_bytes)
    off = tc_align16(off);
    l.xyc = (float2 *)(p0 + off); off += 8 * (size_t)(((N + 3) & ~3) + 8);
    l.cid = (short *)(p0 + off); off += tc_align16(2 * ((size_t)N + 1));
  }
  l.ids = (unsigned short *)(p0 + off); off = tc_align16(off + 2 * A * K);
  l.stage_dwords = (int)(tc_align16((size_t)4 * tc_stage_rows(F, n_waves) * F) / 4) + 4 + 16;  // + the list of live rows (64 bytes)
  l.stage = (float *)(p0 + off); off += (size_t)4 * l.stage_dwords * n_waves;
  off = tc_align16(off > min_area_bytes ? off : min_area_bytes);
  l.tb = tc_carve_tables(p0 + off, epb, N);
  return l;
}

// ---- observation rows of one wavefront, SPARSE form (chosen per wavefront when at most 9/16 of its rows
// belong to agents in the game: late in an episode; the dense form -- contiguous chunks of rows, every row
// computed and written -- is cheaper per row but moves every byte): rows [wrow0, wrow0 + wrows) of the block.
// A row of an agent that is out of the game is all zeros until the episode restarts (:476-560): it is
// cleared ONCE, on the first tick the agent is out (bit 1 of l.sig / obs_rows_cleared remember it), and
// costs nothing afterwards -- under the benchmark's own policy half of the rows, on average over an
// episode.  Rows of agents IN the game are built in the wavefront's private LDS staging buffer, `rs`
// rows at a time in packed order (the wavefront's list of live rows maps the packed ordinal to the
// row): work item = (live row, neighbour slot) -> 7 values at c*K + k of the row image; then the time
// column; then every row image leaves as 16-byte write-through stores.
//   A row image starts `mis` dwords into its slot, mis = (row address / 4) & 3, so that its 16-byte
// vectors are aligned in LDS and in memory alike (rows are 4 * F bytes, F odd: the alignment changes
// from row to row); the slot's pads are zeroed.  The <= 3 dwords at either end of a row share a
// 16-byte line with the neighbouring row:
//   * neighbour out of the game: its row is (or is being) cleared, so the whole line is stored with
//     zeros in the neighbour's part;
//   * neighbour in the game and built in the same chunk (the next slot): the lower row stores the line,
//     merged (OR) with the first vector of the next slot; the upper row skips its first vector;
//   * neighbour unknown (other wavefront / other block) or in another chunk: single dwords, own part only.
__device__ __forceinline__ void tc_store_own_dwords(float *rowp, int F, int d0, const float (&v)[4], bool on) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (on && d0 + e >= 0 && d0 + e < F) rowp[d0 + e] = v[e];
}

__device__ __forceinline__ void tc_gather_rows_sparse(const TcArgs &a, const TcFastLds &l, const TcTables &tb, float *stage,
                                               int env0, int wrow0, int wrows, int lane, int K, int N, float invK,
                                               float invN) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int F = 7 * K + 1;
  const int SL = (F + 6) & ~3;       // dwords per row slot (mis + F <= SL)
  const int NV = SL >> 2;            // 16-byte vectors per slot
  const int cap = l.stage_dwords - 16;
  const int rs = min(min(64, cap / SL), 192 / K);  // rows per chunk (at most 3 items per lane)
  const int RPR = 64 / NV;           // rows per flush round (NV <= 58 for K <= 32)
  const float invNV = 1.0f / (float)NV;
  constexpr int U = 3;
  int rr[U], kk[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int t = lane + 64 * u;
    rr[u] = (int)(((float)t + 0.5f) * invK);  // t / K (exact: the quotient is >= 0.5/K away from an integer)
    kk[u] = t - rr[u] * K;
  }
  const int fsub = (int)(((float)lane + 0.5f) * invNV), fv = lane - fsub * NV;  // flush: (row of the round, vector)
  const int sgv = (lane < wrows) ? l.sig[wrow0 + lane] : 2;
  const unsigned long long lmask = __ballot((sgv & 1) != 0);  // rows to build (a wavefront gathers at most 64 rows)
  unsigned long long zmask = __ballot(sgv == 0);              // rows to clear: out of the game, not cleared yet
  unsigned char *const rowlist = (unsigned char *)(stage + cap);
  if (sgv & 1) rowlist[__popcll(lmask & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  float *const obs_w = a.obs + ((long)env0 * N + wrow0) * F;
  const unsigned bdw = (unsigned)((size_t)obs_w >> 2);
  const unsigned short *const idw = l.ids + (size_t)wrow0 * K;
  const int n_rows = __popcll(lmask);
  // ---- rows of agents that left the game since the last tick: zeros, straight from registers
  while (zmask) {  // wave-uniform, rare
    const int r = __ffsll((long long)zmask) - 1;
    zmask &= zmask - 1ull;
    float *const rowp = obs_w + (long)r * F;
    const int mis = (int)((bdw + (unsigned)(r * F)) & 3u);
    const int d0 = 4 * lane - mis;
    const float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (lane < NV) {
      if (d0 >= 0 && d0 + 4 <= F) {
        const v4f q = {0.0f, 0.0f, 0.0f, 0.0f};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(rowp + d0), "v"(q) : "memory");
      } else {
        tc_store_own_dwords(rowp, F, d0, z, true);
      }
    }
  }
  // ---- rows of agents in the game
  for (int j0 = 0; j0 < n_rows; j0 += rs) {
    const int rc = min(rs, n_rows - j0);
    const int items = rc * K;
    if (lane < rc) {  // zero the pads of the slot (first vector, last two vectors)
      const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
      v4f *const sl = (v4f *)(stage + lane * SL);
      sl[0] = zero;
      sl[NV - 2] = zero;
      sl[NV - 1] = zero;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (64 * u < items) {  // wave-uniform
        if (lane + 64 * u < items) {
          const int r = rowlist[j0 + rr[u]];  // row of the item inside the wavefront's rows
          const unsigned jq = idw[r * K + kk[u]];
          const TcFeat me = tc_feat_load(l.feat, wrow0 + r);
          // no neighbour in this slot: the agent's own record stands in, so every difference
          // below is +0.0 without a select
          const bool valid = (jq != 0xffffu);
          const TcFeat nb = tc_feat_load(l.feat, valid ? (int)jq : wrow0 + r);
          unsigned mv = valid ? 0xffffffffu : 0u;
          asm volatile("" : "+v"(mv));  // (opaque: keeps the AND below from being turned into selects)
          const unsigned ts = (unsigned)nb.type_sig & mv;
          const int mis = (int)((bdw + (unsigned)(r * F)) & 3u);
          float *o = stage + rr[u] * SL + mis + kk[u];
          o[0] = (float)(nb.nx - me.nx);   // float64 difference, narrowed (:560)
          o[K] = (float)(nb.ny - me.ny);
          o[2 * K] = nb.nsp - me.nsp;      // float32 operands: the float64 difference rounds to this
          o[3 * K] = nb.nac - me.nac;
          o[4 * K] = nb.ndir - me.ndir;
          o[5 * K] = __uint_as_float(ts & 0x3f800000u);
          o[6 * K] = __uint_as_float((0u - (ts & 1u)) & 0x3f800000u);
        }
      }
    }
    if (lane < rc) {
      // time column: float(t) / episode_length (agents in the game, :474,:493,:543)
      const int r = rowlist[j0 + lane];
      const int e_m = (int)(((float)(wrow0 + r) + 0.5f) * invN);
      const int mis = (int)((bdw + (unsigned)(r * F)) & 3u);
      stage[lane * SL + mis + 7 * K] = tb.tfrac[e_m];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- flush: RPR rows per round, lane = (row of the round, 16-byte vector of its slot); three rounds
    // per trip so that three independent chains of LDS reads are in flight
    for (int sb = 0; sb < rc; sb += 3 * RPR) {
      constexpr int W = 3;
      int slot[W], r[W], d0[W];
      bool on[W];
#pragma unroll
      for (int u = 0; u < W; ++u) {
        slot[u] = sb + u * RPR + fsub;
        on[u] = (fsub < RPR) && (slot[u] < rc);
        slot[u] = min(slot[u], rc - 1);
        r[u] = rowlist[j0 + slot[u]];
      }
      v4f q[W], nx[W];
#pragma unroll
      for (int u = 0; u < W; ++u) {
        const int mis = (int)((bdw + (unsigned)(r[u] * F)) & 3u);
        d0[u] = 4 * fv - mis;  // row-relative index of the vector's first dword
        q[u] = *(const v4f *)(stage + slot[u] * SL + 4 * fv);
        nx[u] = *(const v4f *)(stage + min(slot[u] + 1, rc - 1) * SL);  // first vector of the next slot (merge)
      }
#pragma unroll
      for (int u = 0; u < W; ++u) {
        if (sb + u * RPR < rc) {  // wave-uniform
          float *const rowp = obs_w + (long)r[u] * F;
          const bool head_part = on[u] && (d0[u] < 0), tail_part = on[u] && (d0[u] < F) && (d0[u] + 4 > F);
          // the neighbouring rows: in the game?  known at all (inside this wavefront's rows)?
          const bool prev_known = (r[u] > 0), next_known = (r[u] + 1 < wrows);
          const bool prev_live = prev_known && ((lmask >> (r[u] - 1)) & 1ull);
          const bool next_live = next_known && ((lmask >> (r[u] + 1)) & 1ull);
          const bool merge_next = tail_part && next_live && (slot[u] + 1 < rc);
          unsigned mm = merge_next ? 0xffffffffu : 0u;
          asm volatile("" : "+v"(mm));  // (AND mask, not four selects)
          v4f o = q[u];
          o.x = __uint_as_float(__float_as_uint(o.x) | (__float_as_uint(nx[u].x) & mm));
          o.y = __uint_as_float(__float_as_uint(o.y) | (__float_as_uint(nx[u].y) & mm));
          o.z = __uint_as_float(__float_as_uint(o.z) | (__float_as_uint(nx[u].z) & mm));
          o.w = __uint_as_float(__float_as_uint(o.w) | (__float_as_uint(nx[u].w) & mm));
          const bool skip = head_part && prev_live && (slot[u] > 0);            // stored by the row below
          const bool own_only = (head_part && !skip && (prev_live || !prev_known)) ||
                                (tail_part && !merge_next && (next_live || !next_known));
          const bool full = on[u] && (d0[u] < F) && !skip && !own_only;
          if (full) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(rowp + d0[u]), "v"(o) : "memory");
          if (__ballot(own_only) != 0ull) {  // wave-uniform: a row at the edge of the wavefront's rows or of the chunk
            const float vals[4] = {o.x, o.y, o.z, o.w};
            tc_store_own_dwords(rowp, F, d0[u], vals, own_only);
          }
        }
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}


// EXACTK: K == KMAX, known at compile time (row offsets become immediates, the K-dependent selects fold away)
template <int KMAX, bool FUSED, bool EXACTK, int IDB, bool SAMPLE = FUSED>
__device__ __forceinline__ void tc_fast_impl(const TcArgs &a, const TcFuse &fz, unsigned char *smem, int n_acc,
                                             int n_turn) {
  const int N = a.N, K = EXACTK ? KMAX : a.K;
  const int F = 7 * K + 1;
  const int tid = threadIdx.x, T_ = WD_TC_BLOCKDIM;
  const int epb = max(1, T_ / N);
  // (readfirstlane: the wavefront index is uniform, but only the hardware knows -- without it every loop whose
  // bounds depend on it is compiled as a divergent loop)
  const int n_waves = (T_ + 63) >> 6, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const size_t slab_acc_bytes = tc_align16((size_t)4 * epb * N * n_acc);
  // One replica per block (more than 64 agents: the BASELINE shape): the neighbour search runs over
  // the agents that are still IN THE GAME only, packed in ascending id order -- as candidates (the
  // chain is as long as the live list, not N) and as searchers (searcher lane c works for the c-th
  // live agent, so a wavefront whose lanes are all >= the live count skips the search).  Under the
  // reference's own benchmark policy (uniform random actions) 54 of 105 agents are in the game on
  // average over a 500-tick episode (105 at the start, ~27 at the end).  Packing preserves the id
  // order, so ties break exactly as before; ids are translated back through `cid`.
  const bool compact = (epb == 1);
  const size_t slab_turn_bytes = tc_align16((size_t)4 * epb * N * n_turn);
  const bool one_slab = tc_one_slab(N);
  const TcFastLds l = tc_carve_fast(smem, epb, N, K, n_waves,
                                    !SAMPLE ? 0 : one_slab ? max(slab_acc_bytes, slab_turn_bytes) : slab_acc_bytes + slab_turn_bytes,
                                    compact);
  const TcTables &tb = l.tb;
  float *const slab_acc = (float *)smem, *const slab_turn = (float *)(smem + (one_slab ? 0 : slab_acc_bytes));
  float *const stage = l.stage + (size_t)wave * l.stage_dwords;
  const int el = tid / N, ag = tid - el * N;
  const float invK = 1.0f / (float)K, invN = 1.0f / (float)N;
  const int NP = ((N + 3) & ~3) + 8;  // stride of a replica's positions in LDS (16-byte aligned pairs + padding)

  // ONE trip per block (the host launches ceil(replicas / epb) blocks): every pointer argument is
  // used once and dies, which is what keeps the kernel inside 128 VGPRs / 104 SGPRs.
  // All global loads go out before anything else: the table set-up below (a dependent global load +
  // barrier) then runs in their shadow.
  const int env0 = a.env_begin + blockIdx.x * epb;
  // Wave priority falls with the phase (3: fetch .. tags, 2: first half of the search, 0: the rest
  // of it and everything after, with a short stretch at 1 where the ids come out of the keys): a
  // wavefront that is behind wins VALU arbitration over one that is ahead, so the wavefronts of a
  // SIMD finish together.  The default oldest-first arbitration keeps leaders ahead
  // and leaves the last wavefront of every SIMD running alone, latency-bound (measured with the
  // two-pass search: 48.6 -> 44.4 us per tick; the schedule was re-tuned for the one-pass search,
  // experiments/README.md).
  __builtin_amdgcn_s_setprio(3);
  WD_TC_PROBE_RT(16); WD_TC_PROBE(0); WD_TC_PROBE_HW(21);
  TcIn in;
  tc_issue_loads<FUSED, SAMPLE>(in, a, fz, env0, epb, N, n_acc, n_turn, tid, slab_acc, slab_turn, true);
  const bool tab_in_lds = (n_acc <= WD_TC_TAB) && (n_turn <= WD_TC_TAB);
  const int n_taggers = tc_build_tables(tb, a, N, n_acc, n_turn, tab_in_lds, in);
  if (env0 >= a.E) return;  // whole block (no barrier is skipped by part of a block)
  WD_TC_PROBE(1);

  const int env = env0 + el;
  const bool active = (el < epb) && (env < a.E);
  const int gi = env * N + ag;  // index into [E, N] arrays
  const int li = tid;           // index into LDS arrays (= el * N + ag)
  const int agents_here = min(epb, a.E - env0) * N;
  int2 sampled = in.sampled;
  const unsigned long long live_mask = __ballot(active && in.sg != 0);
  if (compact && lane == 0) tb.live_cnt[wave] = __popcll(live_mask);
  if (FUSED) {
    if (active && ag == 0) a.done[env] = 0;  // a replica that finished (and was reset) last tick
    if (SAMPLE) sampled = tc_sample_heads(a, fz, in, active, gi, li, slab_acc, slab_turn, n_acc, n_turn, env0, epb);
  }
  WD_TC_PROBE(2);
  __syncthreads();  // tables are published; every wavefront is done with the slabs
  WD_TC_PROBE(3);
  // packed index of this lane's agent among the agents in the game, and their number
  int my_c = ag, n_live = N;
  if (compact) {
    int before = 0;
    n_live = 0;
    for (int w2 = 0; w2 < n_waves; ++w2) {
      const int c = tb.live_cnt[w2];
      before += (w2 < wave) ? c : 0;
      n_live += c;
    }
    my_c = before + __popcll(live_mask & ((1ull << lane) - 1ull));
  }
  // the prefiltered search (tc_chain_prefiltered): on for big replicas while enough agents are in the game
  constexpr bool PRE = (IDB != 7) && (KMAX <= 12);
  bool pre_on = false;  // block-uniform
  uint4 hint_a = make_uint4(~0u, ~0u, ~0u, ~0u), hint_b = hint_a;
  if constexpr (PRE) {
    pre_on = compact && (a.knn_prev != nullptr) && (n_live >= WD_TC_PRE_MIN_LIVE) && (l.stage_dwords >= 512);
    if (pre_on && active && in.sg != 0) {  // (in flight during the move)
      const uint4 *const h = (const uint4 *)(a.knn_prev + (size_t)(env * N + ag) * 8);
      hint_a = h[0];
      hint_b = h[1];
    }
  }

  // ------------------------------------------------------------ move
  float edge_pen = 0.0f, my_x = 0.0f, my_y = 0.0f;
  const int sg = in.sg;
  const bool is_runner = active && (in.type == 0) && (sg != 0);  // member of self.runners
  if (active) {
    const TcMoved m = tc_move(a, tb, in, sampled, gi, tab_in_lds);
    edge_pen = m.edge_pen; my_x = m.x; my_y = m.y;
    // agents out of the game are pushed to +BIG for the neighbour search only; every other
    // consumer (taggers are never out of the game) reads real positions
    // (with the prefilter on: NaN -- only its bound reads the entry of an agent that is out of the game then)
    l.xy[el * NP + ag] = make_float2(sg ? m.x : (PRE && pre_on ? __builtin_nanf("") : WD_BIG), m.y);
    if (compact) {
      if (sg) {
        l.xyc[my_c] = make_float2(m.x, m.y);
        l.cid[1 + my_c] = (short)ag;
        if (PRE && pre_on) {  // the hint goes to the lane that searches for this agent: slot my_c & 63 of wavefront my_c >> 6
          uint4 *const slot = (uint4 *)(l.stage + (size_t)(my_c >> 6) * l.stage_dwords) + 2 * (my_c & 63);
          slot[0] = hint_a;
          slot[1] = hint_b;
        }
      }
      if (ag == 0) {
        l.cid[0] = -1;
        if (PRE && pre_on) {
          l.xyc[n_live] = make_float2(WD_BIG, WD_BIG);         // the pad candidate of pass 2: a position at +inf
          l.xy[N] = make_float2(__builtin_nanf(""), 0.0f);     // what a remembered id of 0xffff (none) reads
        }
      }
    }
    tc_feat_store(l.feat, li, m.ft);
    // bit 0: in the game before this tick's tagging; bit 1: the observation row in HBM is all zeros already
    l.sig[li] = (sg ? 1 : 0) | (in.cleared ? 2 : 0);
    // after this tick's gather (either form) the row of an agent out of the game is zeros, the row of one in it is not
    if ((in.cleared != 0) != (sg == 0)) a.obs_rows_cleared[gi] = sg ? 0 : 1;
    l.tagcnt[li] = 0;
    if (ag == 0) {
      const int t = in.tstep + 1;  // :800
      a.timestep[env] = t;
      tb.tstep[el] = t;
      tb.tfrac[el] = (float)((double)t / (double)a.T);  // float(t) / episode_length, :474
      tb.nrun[el] = in.nrun;
    }
  }
  WD_TC_PROBE(4);
  __syncthreads();
  WD_TC_PROBE(5);

  // ------------------------------------------------------------ tags (counts are read after the
  // barrier that follows the gather)
  bool tagged = false;
  if (is_runner)
    tagged = tc_find_tag(a, tb, l.xy + el * NP, l.tagcnt + el * N, &tb.nrun[el], n_taggers, my_x, my_y);

  // ------------------------------------------------------------ search
  WD_TC_PROBE(6);
  int nid[KMAX + 1], rank[KMAX + 1];  // entry k is one of the K nearest iff rank[k] < K
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) { nid[k] = -1; rank[k] = k; }
  // in_order: slot k of the agent's row is entry k (true as well for agents that are not searched
  // for: all their entries are "none")
  bool in_order = true;
  __builtin_amdgcn_s_setprio(2);
  // searcher lane `ag` works for the ag-th agent in the game (packed) or for its own agent
  const bool searcher = compact ? (tid < n_live) : (active && sg != 0);  // (tid == ag for these lanes)
  const float2 *const sxy = compact ? l.xyc : l.xy + el * NP;
  const int n_cand = compact ? n_live : N;
  int row_agent = ag;  // the agent whose row this lane's search fills
  // TWO WAVEFRONTS PER SEARCHER while at most 64 agents are in the game (70 % of an episode of the benchmark
  // policy): the searchers then fit the first wavefront and the second one used to wait at the barrier below for
  // the whole search -- with two of the four wavefronts of a SIMD idle the chain is bound by the issue latency of a
  // single wavefront (~10 cycles per instruction), not by the VALU.  Now lane i of BOTH wavefronts works for searcher
  // i: wavefront 0 runs the chain over the first half of the candidates, wavefront 1 over the second half; wavefront
  // 1 hands its L keys over through its staging buffer (dead until the gather) and wavefront 0 merges the two sorted
  // lists (tc_merge_sorted: the L smallest of the union are exactly what one chain over all candidates keeps).
  constexpr int L = KMAX + 3;  // self + K others + two look-ahead entries
  const bool split = compact && (n_waves == 2) && (n_live <= 64) && (n_live >= 16) &&
                     (l.stage_dwords >= 64 * L);                          // block-uniform
  const int j_half = split ? (((n_live + 7) >> 3) << 2) : n_cand;        // first candidate of wavefront 1's half
  const bool helper = split && (wave == 1);                               // wave-uniform
  unsigned S[L];
  bool prefiltered = false;           // wave-uniform
  unsigned extra = 0xffffffffu;       // the (L+1)-th key (prefiltered search only)
  if constexpr (PRE) {
    if (pre_on) {  // block-uniform
      unsigned Tb = 0u;
      float sx = 0.0f, sy = 0.0f;
      if (searcher) {
        const uint4 *const slot = (const uint4 *)stage + 2 * lane;
        const uint4 pa = slot[0], pb = slot[1];
        sx = sxy[ag].x; sy = sxy[ag].y;
        Tb = tc_knn_bound16<KMAX>(l.xy, N, sx, sy, pa, pb, K);
      }
      WD_TC_PROBE(7);
      // every searcher of the wavefront has a radius -- and there IS a searcher: a wavefront without one (tid >=
      // n_live: up to 11 of 16 at ~300 agents in the game) would run pass 1 over every candidate for nothing and
      // compete for the VALU with the searching wavefronts of its SIMD; it goes straight to the barrier instead
      if (__ballot(searcher) != 0ull && __ballot(searcher && Tb == 0x7f800000u) == 0ull) {
        // (lanes without a searcher: radius -1, nothing listed; they only take part in the wave-wide votes)
        // (candidates popped per trip: the fullest lane of a 32-candidate word holds ~2 at 1000 agents, ~4 at 500)
        constexpr int POPS = (IDB == 10) ? 1 : 2;
        extra = tc_chain_prefiltered<L, IDB, POPS>(sxy, sx, sy, n_cand, searcher ? __uint_as_float(Tb) : -1.0f, n_cand, S);
        // the radius held the K nearest iff the K-th other agent found (entry K with the agent's own) lies at least
        // two key buckets inside it: everything that was not listed is then past the buckets tc_resolve_keys looks at
        unsigned sK = S[KMAX];
#pragma unroll
        for (int k = 1; k < KMAX; ++k) sK = (k == K) ? S[k] : sK;
        const bool held = (sK >> IDB) + 2u <= (Tb >> IDB);
        prefiltered = __ballot(searcher && !held) == 0ull;
        WD_TC_PROBE_VAL(19, prefiltered ? 1 : 2);
        if (!prefiltered) extra = 0xffffffffu;
      }
      WD_TC_PROBE(8);
    }
  }
  if (!prefiltered && (searcher || (helper && lane < n_live))) {
    // one pass with packed keys (this wavefront's share of the candidates)
    const int me = helper ? lane : ag;
    tc_chain_range<L, IDB>(sxy, sxy[me].x, sxy[me].y, helper ? j_half : 0, helper ? n_cand : j_half, S);
  }
  WD_TC_PROBE(9);
  if (split) {  // block-uniform
    if (helper && lane < n_live) {
#pragma unroll
      for (int k = 0; k < L; ++k) ((unsigned *)stage)[64 * k + lane] = S[k];
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    if (searcher) {
      const unsigned *const theirs = (const unsigned *)(l.stage + (size_t)l.stage_dwords);  // wavefront 1's buffer
      unsigned P[L];
#pragma unroll
      for (int k = 0; k < L; ++k) P[k] = theirs[64 * k + lane];
      tc_merge_sorted<L>(S, P);
    }
  }
  bool exact = true;
  unsigned zone_hi = 0u;
  if (searcher) {
    // a lane with three candidates inside 256 ulps at the cut (~1e-7 per agent) repeats the search with the
    // two-pass one (up to 128 candidates) / has the whole wavefront resolve it (more)
    unsigned o[L - 1];
    __builtin_amdgcn_s_setprio(1);
    exact = tc_resolve_keys<KMAX, IDB, L>(sxy, ag, K, S, o, nid, rank, in_order);
    WD_TC_PROBE(10);
    {  // the last key bucket the answer can come from: the K-th other agent's + 1
      unsigned oKth = o[KMAX - 1];
#pragma unroll
      for (int k = 0; k < KMAX - 1; ++k) oKth = (k == K - 1) ? o[k] : oKth;
      zone_hi = (oKth >> IDB) + 1u;
    }
    if constexpr (PRE) {
      if (pre_on) {  // remember the K + 3 nearest others (agent ids, 16 bits each; 0xffff = none) for the next tick's radius
        unsigned w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          unsigned pair = 0u;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int k = 2 * q + h;
            const unsigned key = (k < L - 1) ? o[k < L - 1 ? k : 0] : (k == L - 1) ? extra : 0xffffffffu;
            const unsigned id = (key >= 0x7f800000u) ? 0xffffu : (unsigned)(unsigned short)l.cid[1 + (key & ((1u << IDB) - 1u))];
            pair |= id << (16 * h);
          }
          w[q] = pair;
        }
        uint4 *const h = (uint4 *)(a.knn_prev + (size_t)(env * N + l.cid[1 + ag]) * 8);
        h[0] = make_uint4(w[0], w[1], w[2], w[3]);
        h[1] = make_uint4(w[4], w[5], w[6], w[7]);
      }
    }
    if (!exact && (IDB == 7 || n_cand <= 128)) {
      WD_TC_PROBE_VAL(20, 1);
      int nid2[KMAX], rank2[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { nid2[k] = -1; rank2[k] = k; }
      tc_knn_registers<KMAX>(sxy, ag, n_cand, K, nid2, rank2);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { nid[k] = nid2[k]; rank[k] = rank2[k]; }
      nid[KMAX] = -1;
      rank[KMAX] = KMAX;
      in_order = false;
      exact = true;
    }
  }
  if constexpr (IDB != 7) {
    // more than 128 candidates: the lanes that need the exact resolution get it from the whole wavefront, one
    // after the other (tc_zone_resolve)
    unsigned long long need = __ballot(searcher && !exact);  // wave-uniform
    if (need != 0ull) {
      WD_TC_PROBE_VAL(20, 1);
      unsigned long long unresolved = 0ull;
      const float mx = searcher ? sxy[ag].x : 0.0f, my = searcher ? sxy[ag].y : 0.0f;
      while (need != 0ull) {
        const int fl = __ffsll((long long)need) - 1;
        need &= need - 1ull;
        const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), fl));
        const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my), fl));
        const unsigned zh = (unsigned)__builtin_amdgcn_readlane((int)zone_hi, fl);
        const int self = __builtin_amdgcn_readlane(ag, fl);
        const int cnt = tc_zone_resolve(sxy, n_cand, sx, sy, self, zh, IDB, K, (unsigned char *)stage, lane);
        if (cnt > 64) {
          unresolved |= 1ull << fl;
        } else {
          const unsigned short *const out = (const unsigned short *)((const unsigned char *)stage + 128 + 512);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const int v = (k < K) ? (int)out[k] : -1;
            if (lane == fl) { nid[k] = v; rank[k] = k; }
          }
          if (lane == fl) { nid[KMAX] = -1; rank[KMAX] = KMAX; in_order = false; }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      if ((unresolved >> lane) & 1ull) {  // more than 64 candidates inside the zone: the K-pass scan, this lane alone
        int nid2[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) nid2[k] = -1;
        tc_knn_scan<KMAX>(sxy, ag, n_cand, K, nid2);  // (entries in the reference's order)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { nid[k] = nid2[k]; rank[k] = k; }
        nid[KMAX] = -1;
        rank[KMAX] = KMAX;
        in_order = false;
      }
    }
  }
  if (searcher && compact) {  // packed indices -> agent ids (cid[0] = -1 stands for "none")
    row_agent = l.cid[1 + ag];
#pragma unroll
    for (int k = 0; k <= KMAX; ++k) nid[k] = l.cid[1 + nid[k]];
  }
  __builtin_amdgcn_s_setprio(0);
  WD_TC_PROBE(11);

  // ------------------------------------------------------------ ids out: block-local 16-bit neighbour
  // ids per agent row in LDS (0xffff = none), read by the gather and turned into the
  // `nearest_neighbor_ids` rows after the barrier.  Entry k goes to slot k at fixed offsets; the few
  // lanes whose entries are not in order (a near-tie, fewer than K agents in the game) then rewrite
  // their rows by rank.
  {
    const int ebase = el * N;
    const bool any_out_of_order = __ballot(!in_order) != 0ull;  // wave-uniform
    if (active && sg == 0) {  // out of the game: no neighbours
      unsigned short *const own = l.ids + (size_t)li * K;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) own[k] = 0xffff;
    }
    if (searcher) {
      unsigned short *const idrow = l.ids + (size_t)(ebase + row_agent) * K;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)  // (an in-order row holds K ids)
        if (k < K) idrow[k] = (unsigned short)(ebase + nid[k]);
      if (any_out_of_order && !in_order) {
#pragma unroll
        for (int k = 0; k <= KMAX; ++k)  // (K of the KMAX + 1 entries have a rank < K)
          if (rank[k] < K) idrow[rank[k]] = (unsigned short)(nid[k] < 0 ? 0xffff : ebase + nid[k]);
      }
    }
  }
  // ------------------------------------------------------------ gather: the block's rows are split
  // evenly over its wavefronts (105 agents: 53 + 52 rows instead of 64 + 41: one chunk less on the
  // longer side), so a wavefront also gathers rows whose neighbours another wavefront found
  __syncthreads();
  const int rpw = (agents_here + n_waves - 1) / n_waves;
  const int wrow0 = wave * rpw;
  const int wrows = max(0, min(rpw, agents_here - wrow0));
  // nearest_neighbor_ids [E, N, K]: this wavefront's rows, straight from the 16-bit LDS copies
  tc_flush_ids(l.ids + (size_t)wrow0 * K, a.nearest_ids + ((long)env0 * N + wrow0) * K, wrows * K, lane, wrow0, N,
               invK, invN, epb == 1);
  WD_TC_PROBE(12);
  // the sparse form pays when few rows are live (late in an episode); wave-uniform choice
  const int n_live_rows = __popcll(__ballot(lane < wrows && (l.sig[wrow0 + lane] & 1)));
  if (n_live_rows * 16 <= wrows * 9) {
    tc_gather_rows_sparse(a, l, tb, stage, env0, wrow0, wrows, lane, K, N, invK, invN);
  } else {
    // observation rows, R rows per chunk: work item = (row, neighbour slot) -> 7 values at
    // row*F + c*K + k of the chunk image; then the time column; then the chunk leaves as one run.
    // A chunk holds at most 192 items (tc_stage_rows), i.e. at most 3 per lane; their (row, slot)
    // split is the same for every chunk and is worked out once.
    const int R = tc_stage_rows(F, n_waves);
    constexpr int U = 3;
    int rr[U], so[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = lane + 64 * u;
      rr[u] = (int)(((float)t + 0.5f) * invK);  // t / K (exact: the quotient is >= 0.5/K away from an integer)
      so[u] = rr[u] * F + (t - rr[u] * K);      // offset of the item's first value in the chunk image
    }
    float *const obs_w = a.obs + ((long)env0 * N + wrow0) * F;
    for (int r0 = 0; r0 < wrows; r0 += R) {
      const int rc = min(R, wrows - r0);
      float *const dst = obs_w + (long)r0 * F;
      const int mis = (int)(((size_t)dst >> 2) & 3);
      const int items = rc * K;
      const unsigned short *const idp = l.ids + (size_t)(wrow0 + r0) * K;  // ids of item t: idp[t]
      const int fp = wrow0 + r0;
      // ids, then feature records, all reads of a lane's items in flight together.  A lane whose item
      // index is past the end recomputes the LAST item and writes the same values to the same place:
      // straight-line code (exec-mask branches would cost more than the duplicate work)
      int tt[U];
      unsigned jq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { tt[u] = min(lane + 64 * u, items - 1); jq[u] = idp[tt[u]]; }
      const bool clamped2 = lane + 128 >= items, clamped1 = lane + 64 >= items, clamped0 = lane >= items;
      TcFeat me[U], nb[U];
      int off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool cl = (u == 0) ? clamped0 : (u == 1) ? clamped1 : clamped2;
        // (row, offset) of the item: precomputed for unclamped lanes, recomputed for the last item
        const int r_last = rc - 1, o_last = r_last * F + (K - 1);
        const int r = cl ? r_last : rr[u];
        off[u] = cl ? o_last : so[u];
        me[u] = tc_feat_load(l.feat, fp + r);
        // no neighbour (or the agent is out of the game): its own record stands in, so every
        // difference below is +0.0 without a select
        const bool valid = ((me[u].type_sig & 1) != 0) && (jq[u] != 0xffffu);
        nb[u] = tc_feat_load(l.feat, valid ? (int)jq[u] : fp + r);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool valid = ((me[u].type_sig & 1) != 0) && (jq[u] != 0xffffu);
        unsigned mv = valid ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(mv));  // (opaque: keeps the AND below from being turned into selects)
        const unsigned ts = (unsigned)nb[u].type_sig & mv;
        float *o = stage + mis + off[u];
        o[0] = (float)(nb[u].nx - me[u].nx);   // float64 difference, narrowed (:560)
        o[K] = (float)(nb[u].ny - me[u].ny);
        o[2 * K] = nb[u].nsp - me[u].nsp;      // float32 operands: the float64 difference rounds to this
        o[3 * K] = nb[u].nac - me[u].nac;
        o[4 * K] = nb[u].ndir - me[u].ndir;
        o[5 * K] = __uint_as_float(ts & 0x3f800000u);
        o[6 * K] = __uint_as_float((0u - (ts & 1u)) & 0x3f800000u);
      }
      if (lane < rc) {
        // time column: float(t) / episode_length for agents in the game, else 0 (:474,:493,:543)
        const int m = wrow0 + r0 + lane;
        const int e_m = (int)(((float)m + 0.5f) * invN);
        stage[mis + lane * F + 7 * K] = (l.sig[m] & 1) ? tb.tfrac[e_m] : 0.0f;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      tc_flush_run(stage, dst, rc * F, lane);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  WD_TC_PROBE(13);
  __syncthreads();  // every runner's tag is counted
  WD_TC_PROBE(14);

  // ------------------------------------------------------------ rewards / done
  if (active) tc_finish_agent(a, tb, el, ag, gi, env, sg, is_runner, tagged, l.tagcnt[li], edge_pen, in.step_reward, FUSED);
  if (FUSED) {
    __syncthreads();  // doneflag
    bool any = false;
    for (int e = 0; e < min(epb, a.E - env0); ++e) any = any || (tb.doneflag[e] != 0);
    if (any) {  // block-uniform, rare (once per episode)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores are complete ...
      __syncthreads();                                  // ... before any wavefront rewrites the rows
      tc_reset_finished(a, fz, tb, env0, epb);
    }
  }
  WD_TC_PROBE(15); WD_TC_PROBE_RT(17);
}

// =====================================================================================
//                generic path: any N <= 1024, any K, full observations
// =====================================================================================
struct TcGenLds {
  TcFeat *feat;      // [A]
  int *nbr;          // [A][K] neighbour ids (env-local, -1 = none)
  float2 *xy;        // [A]
  int *sig;          // [A]
  int *tagcnt;       // [A]
  TcTables tb;
};

__device__ __forceinline__ TcGenLds tc_carve_generic(unsigned char *p0, int epb, int N, int K, size_t min_area_bytes) {
  TcGenLds l;
  const size_t A = (size_t)epb * N;
  size_t off = 0;
  l.feat = (TcFeat *)(p0 + off); off += sizeof(TcFeat) * A;
  l.nbr = (int *)(p0 + off); off += tc_align16(4 * A * (size_t)max(K, 1));
  l.xy = (float2 *)(p0 + off); off += 8 * A;
  l.sig = (int *)(p0 + off); off += 4 * A;
  l.tagcnt = (int *)(p0 + off); off += 4 * A;
  off = tc_align16(off > min_area_bytes ? off : min_area_bytes);
  l.tb = tc_carve_tables(p0 + off, epb, N);
  return l;
}

// K passes, each picks the smallest (float32 distance, id) key above the previous one
__device__ __forceinline__ void tc_knn_generic(const float2 *cxy, const int *csig, int *out, int ag, int N, int K) {
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  float pd = -1.0f;
  int pj = -1;
  for (int k = 0; k < K; ++k) {
    float best = __builtin_inff();
    int bj = -1;
    for (int j = 0; j < N; ++j) {
      if (csig[j] == 0 || j == ag) continue;
      const float2 pc = cxy[j];
      const float dx = xi - pc.x, dy = yi - pc.y;
      const float d = sqrtf(dx * dx + dy * dy);
      const bool above = (d > pd) || (d == pd && j > pj);
      if (above && d < best) { best = d; bj = j; }
    }
    out[k] = bj;
    if (bj < 0) {
      for (int kk = k + 1; kk < K; ++kk) out[kk] = -1;
      break;
    }
    pd = best;
    pj = bj;
  }
}


template <bool FUSED>
__device__ __forceinline__ void tc_generic_impl(const TcArgs &a, const TcFuse &fz, unsigned char *smem, int n_acc,
                                                int n_turn) {
  const int N = a.N, K = a.use_full_obs ? 0 : a.K;
  const int W = a.use_full_obs ? (N - 1) : K;  // columns per feature
  const int F = 7 * W + 1;
  const int tid = threadIdx.x, T_ = WD_TC_BLOCKDIM;
  const int epb = max(1, T_ / N);
  const size_t slab_acc_bytes = tc_align16((size_t)4 * epb * N * n_acc);
  const size_t slab_turn_bytes = tc_align16((size_t)4 * epb * N * n_turn);
  const bool one_slab = tc_one_slab(N);
  const TcGenLds l = tc_carve_generic(smem, epb, N, K,
                                      !FUSED ? 0 : one_slab ? max(slab_acc_bytes, slab_turn_bytes) : slab_acc_bytes + slab_turn_bytes);
  const TcTables &tb = l.tb;
  float *const slab_acc = (float *)smem, *const slab_turn = (float *)(smem + (one_slab ? 0 : slab_acc_bytes));
  const int el = tid / N, ag = tid - el * N;

  int env0 = a.env_begin + blockIdx.x * epb;
  TcIn in;
  tc_issue_loads<FUSED>(in, a, fz, env0, epb, N, n_acc, n_turn, tid, slab_acc, slab_turn);
  const bool tab_in_lds = (n_acc <= WD_TC_TAB) && (n_turn <= WD_TC_TAB);
  const int n_taggers = tc_build_tables(tb, a, N, n_acc, n_turn, tab_in_lds, in);
  if (env0 >= a.E) return;

  while (true) {
    const int env = env0 + el;
    const bool active = (el < epb) && (env < a.E);
    const int gi = env * N + ag;
    const int li = el * N + ag;
    int2 sampled = in.sampled;
    if (FUSED) {
      if (active && ag == 0) a.done[env] = 0;
      sampled = tc_sample_heads(a, fz, in, active, gi, li, slab_acc, slab_turn, n_acc, n_turn, env0, epb);
    }
    __syncthreads();

    // ------------------------------------------------------------ move
    float edge_pen = 0.0f, my_x = 0.0f, my_y = 0.0f;
    const int sg = in.sg;
    const bool is_runner = active && (in.type == 0) && (sg != 0);
    if (active) {
      const TcMoved m = tc_move(a, tb, in, sampled, gi, tab_in_lds);
      edge_pen = m.edge_pen; my_x = m.x; my_y = m.y;
      l.xy[li] = make_float2(m.x, m.y);
      l.feat[li] = m.ft;
      l.sig[li] = sg;
      l.tagcnt[li] = 0;
      if (ag == 0) {
        const int t = in.tstep + 1;
        a.timestep[env] = t;
        tb.tstep[el] = t;
        tb.tfrac[el] = (float)((double)t / (double)a.T);
        tb.nrun[el] = in.nrun;
      }
    }
    __syncthreads();

    // ------------------------------------------------------------ tags + K nearest neighbours
    bool tagged = false;
    if (is_runner)
      tagged = tc_find_tag(a, tb, l.xy + el * N, l.tagcnt + el * N, &tb.nrun[el], n_taggers, my_x, my_y);
    if (!a.use_full_obs && active) {
      int *out = l.nbr + (size_t)li * K;
      if (sg) tc_knn_generic(l.xy + el * N, l.sig + el * N, out, ag, N, K);
      else for (int k = 0; k < K; ++k) out[k] = -1;
    }
    __syncthreads();

    // ------------------------------------------------------------ observations
    // One work item = (agent row m, neighbour slot k): it reads the neighbour id once, then the 7
    // features of that neighbour and of the agent, and writes the 7 columns {c*W + k} of the row.
    {
      const int agents_here = min(epb, a.E - env0) * N;
      const int items = agents_here * W;
      float *obs_blk = a.obs + (long)env0 * N * F;
      const int Wd = max(W, 1);
      if (a.use_full_obs && (W & 3) == 0 && W > 0) {
        // Full observations: rows are 7 runs of W consecutive floats, and the phase is bound by the
        // store path (612 MB per tick at N = 105).  One work item = (row, group of four consecutive
        // slots): 28 values, seven 16-byte stores.  The groups follow the 16-byte grid of MEMORY, not
        // the slot index: a row starts at a dword-aligned address (F is odd), so group g of a row whose
        // start is `mis` dwords past a 16-byte boundary covers slots 4g - mis .. 4g - mis + 3 (W is a
        // multiple of 4: the same shift aligns all seven runs).  Every full group is then ONE aligned
        // 16-byte store per run; only the clipped groups at the two ends of a run use dword stores.
        const int ng = (W >> 2) + 1;  // groups per row, the clipped ones included
        int g = tid % ng, mq = tid / ng, iq = mq % N;
        const int sg = T_ % ng, smq = T_ / ng, siq = smq % N;
        for (int t = tid; t < agents_here * ng; t += T_) {
          const int ebase = mq - iq;
          const bool in_game = l.sig[mq] != 0;
          const TcFeat me = l.feat[mq];
          float *const row = obs_blk + (long)mq * F;
          const int mis = (int)(((size_t)row >> 2) & 3);
          const int s0 = 4 * g - mis;  // first slot of the group (< 0 / > W - 4: clipped)
          float v[7][4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int kcol = min(max(s0 + kk, 0), W - 1);
            const TcFeat nb = l.feat[ebase + kcol + (kcol >= iq ? 1 : 0)];
            float vals[7];
            tc_obs_values(vals, nb, me, in_game, true);  // type / still_in_game columns are always filled
#pragma unroll
            for (int c = 0; c < 7; ++c) v[c][kk] = vals[c];
          }
          if (s0 >= 0 && s0 + 3 < W) {
#pragma unroll
            for (int c = 0; c < 7; ++c) {
              // non-temporal: 612 MB per tick stream through; measured 192 us (plain) -> 160 us, the
              // round-1 slot-indexed (dword-aligned) quads 178 us; write-through (sc1) 475 us here
              typedef float v4f __attribute__((ext_vector_type(4)));
              const v4f quad = {v[c][0], v[c][1], v[c][2], v[c][3]};
              __builtin_nontemporal_store(quad, (v4f *)(row + c * W + s0));
            }
          } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              if (s0 + kk >= 0 && s0 + kk < W) {
#pragma unroll
                for (int c = 0; c < 7; ++c) row[c * W + s0 + kk] = v[c][kk];
              }
          }
          g += sg;
          const int carry = (g >= ng) ? 1 : 0;
          g -= carry ? ng : 0;
          mq += smq + carry;
          iq += siq + carry;
          iq -= (iq >= N) ? N : 0;
        }
      } else {
        int k = tid % Wd, m = tid / Wd;         // block-local agent row and slot of the first item
        int i = m % N;                          // agent id inside its replica
        const int sk = T_ % Wd, sm = T_ / Wd, si = sm % N;
        for (int t = tid; t < items; t += T_) {
          const int ebase = m - i;              // first agent of this row's replica
          const bool in_game = l.sig[m] != 0;
          int j;
          bool valid;
          if (a.use_full_obs) {
            j = k + (k >= i ? 1 : 0);
            valid = true;
          } else {
            j = l.nbr[(size_t)m * K + k];
            valid = in_game && (j >= 0);
            j = max(j, 0);
          }
          const TcFeat nb = l.feat[ebase + j], me = l.feat[m];
          float vals[7];
          tc_obs_values(vals, nb, me, valid && in_game, valid);
          float *row = obs_blk + (long)m * F;
#pragma unroll
          for (int c = 0; c < 7; ++c) row[c * W + k] = vals[c];
          k += sk;
          const int carry = (k >= W) ? 1 : 0;
          k -= carry ? W : 0;
          m += sm + carry;
          i += si + carry;
          i -= (i >= N) ? N : 0;
        }
      }
      // time column: float(t) / episode_length for agents in the game, else 0 (:474,:493,:543)
      for (int m0 = tid; m0 < agents_here; m0 += T_)
        obs_blk[(long)m0 * F + 7 * W] = (l.sig[m0] != 0) ? tb.tfrac[m0 / N] : 0.0f;
      if (!a.use_full_obs && K > 0) {
        int *nb_blk = a.nearest_ids + (long)env0 * N * K;
        for (int q = tid; q < agents_here * K; q += T_) nb_blk[q] = l.nbr[q];
      }
    }

    // ------------------------------------------------------------ rewards / done
    if (active) tc_finish_agent(a, tb, el, ag, gi, env, sg, is_runner, tagged, l.tagcnt[li], edge_pen, in.step_reward, FUSED);
    __syncthreads();  // (also: all stores of the tick to this replica's rows are issued)
    if (FUSED) tc_reset_finished(a, fz, tb, env0, epb);
    env0 += gridDim.x * epb;
    if (env0 >= a.E) break;
    __syncthreads();
    tc_issue_loads<FUSED>(in, a, fz, env0, epb, N, n_acc, n_turn, tid, slab_acc, slab_turn);
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
