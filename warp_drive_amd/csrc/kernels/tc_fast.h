// tc_fast.h -- tc_fast_impl: one tick of whole replicas per block (N <= 1024, partial observations, K <= KMAX).
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_fetch.h"
#include "tc_sample.h"
#include "tc_move.h"
#include "tc_tags.h"
#include "tc_reset.h"
#include "tc_knn.h"
#include "tc_rows.h"

namespace {

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
  // Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8).  A replica's rows of the state arrays are
  // 4 * N bytes, so with replicas in blockIdx order the first and last cache line of every such row is shared with a
  // block on ANOTHER XCD, whose L2 cannot merge the two halves before they go to memory.  Replicas of <= 128 agents
  // get a contiguous range of replica groups per XCD instead (a bijection of [0, gridDim.x) for any grid size; what a
  // replica computes does not depend on its block): headline tick 25.30 -> 24.81 us, 16 000 replicas 216 -> 210 us.
  // Replicas of > 128 agents keep blockIdx order: their rows are long, and the same remap measured 2.5 % SLOWER at
  // 500 replicas of 505 / 1005 agents (profiles/r06_ab_tc_block_order.txt).
  int group = blockIdx.x;
#ifndef WD_TC_BLOCK_ORDER_PLAIN
  if constexpr (IDB == 7) {
    const int bx = blockIdx.x & 7, bq = gridDim.x >> 3, br = gridDim.x & 7;
    group = bx * bq + min(bx, br) + (int)(blockIdx.x >> 3);
  }
#endif
  const int env0 = a.env_begin + group * epb;
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
  if constexpr ((IDB != 7) && (KMAX <= 12)) {  // (the cell-sorted packing's counters, see below)
    if (compact && tid < 64) tb.cell_cnt[tid] = 0;
  }
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
  // the prefiltered search (tc_pre_pass1 / tc_pre_pass2): on for big replicas while enough agents are in the game
  constexpr bool PRE = (IDB != 7) && (KMAX <= 12);
  // equal distances are settled by agent id through the packed-index -> id table wherever the candidates of a block may be
  // packed in another order than ascending id (tc_tie_order, tc_knn.h): replicas of more than 128 agents, one per block
  constexpr bool TIE_TABLE = (IDB != 7);
  bool pre_on = false;  // block-uniform
  int cells_C = 0;      // block-uniform: side of the cell grid the agents in the game are packed by (0: by id)
  float cell_len = 0.0f, cell_inv = 0.0f;
  int my_cell = 0, my_cell_slot = 0;  // this lane's agent: its cell and its arrival number in it
  uint4 hint_a = make_uint4(~0u, ~0u, ~0u, ~0u), hint_b = hint_a;
  if constexpr (PRE) {
    pre_on = compact && (a.knn_prev != nullptr) && (n_live >= WD_TC_PRE_MIN_LIVE) && (l.stage_dwords >= WD_TC_LIST_DWORDS);
    if (pre_on && active && in.sg != 0) {  // (in flight during the move)
      const uint4 *const h = (const uint4 *)(a.knn_prev + (size_t)(env * N + ag) * 8);
      hint_a = h[0];
      hint_b = h[1];
    }
    // ... and then the agents in the game are packed by grid cell, not by id (tc_knn.h "CELL-SORTED packing")
    if (pre_on) {
      cells_C = tc_cell_grid(n_live, K);
      cell_len = a.grid_length / (float)max(cells_C, 1);
      cell_inv = (float)cells_C / a.grid_length;
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
        bool by_cell = false;
        if constexpr (PRE) {
          if (cells_C != 0) {  // its packed place is known after the barrier (counting sort)
            const int2 c = tc_cell_xy(m.x, m.y, cell_inv, cells_C);
            my_cell = tc_cell_place(c, cells_C);
            my_cell_slot = atomicAdd(&tb.cell_cnt[my_cell], 1);
            by_cell = true;
          }
        }
        if (!by_cell) {
          l.xyc[my_c] = make_float2(m.x, m.y);
          l.cid[1 + my_c] = (short)ag;
          if (PRE && pre_on) {  // the hint goes to the lane that searches for this agent: slot my_c & 63 of wavefront my_c >> 6
            uint4 *const slot = (uint4 *)(l.stage + (size_t)(my_c >> 6) * l.stage_dwords) + 2 * (my_c & 63);
            slot[0] = hint_a;
            slot[1] = hint_b;
          }
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
  int cell_incl = 0;  // lane c: agents in the game in cells 0 .. c (every wavefront scans the 64 counters for itself)
  if constexpr (PRE) {
    if (cells_C != 0) {  // block-uniform
      const int cnt = tb.cell_cnt[lane];
      cell_incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(cell_incl, d, 64);
        cell_incl += (lane >= d) ? up : 0;
      }
      my_c = __builtin_amdgcn_ds_bpermute(my_cell << 2, cell_incl - cnt) + my_cell_slot;
      if (active && sg != 0) {
        l.xyc[my_c] = make_float2(my_x, my_y);
        l.cid[1 + my_c] = (short)ag;
        uint4 *const slot = (uint4 *)(l.stage + (size_t)(my_c >> 6) * l.stage_dwords) + 2 * (my_c & 63);
        slot[0] = hint_a;
        slot[1] = hint_b;
      }
    }
  }

  // ------------------------------------------------------------ tags (counts are read after the
  // barrier that follows the gather)
  bool tagged = false;
  if (is_runner)
    tagged = tc_find_tag(a, tb, l.xy + el * NP, l.tagcnt + el * N, &tb.nrun[el], n_taggers, my_x, my_y);
  if constexpr (PRE) {
    if (cells_C != 0) __syncthreads();  // the packed positions, ids and hints are in place
  }

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
  // self + K others + the look-ahead entries: two; ONE where K == KMAX is known and the replica has at most 128 agents (the
  // BASELINE shape's entries): one v_med3_u32 less per candidate, near-ties at the cut go to tc_zone_resolve (tc_resolve_keys)
  constexpr int L = KMAX + ((EXACTK && IDB == 7) ? 2 : 3);
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
      int2 scell = make_int2(0, 0);
      if (searcher) {
        const uint4 *const slot = (const uint4 *)stage + 2 * lane;
        const uint4 pa = slot[0], pb = slot[1];
        sx = sxy[ag].x; sy = sxy[ag].y;
        Tb = tc_knn_bound16<KMAX>(l.xy, N, sx, sy, pa, pb, K);
        if (cells_C != 0) {  // nothing outside the searcher's 3 x 3 cell block is inside the radius
          scell = tc_cell_xy(sx, sy, cell_inv, cells_C);
          Tb = min(Tb, tc_cell_cover2(sx, sy, scell, cell_len, cells_C));
        }
      }
      WD_TC_PROBE(7);
      // every searcher of the wavefront has a radius -- and there IS a searcher: a wavefront without one (tid >=
      // n_live: up to 11 of 16 at ~300 agents in the game) would run pass 1 over every candidate for nothing and
      // compete for the VALU with the searching wavefronts of its SIMD; it goes straight to the barrier instead
      if (__ballot(searcher) != 0ull && __ballot(searcher && Tb == 0x7f800000u) == 0ull) {
        // (lanes without a searcher: radius -1, nothing listed; they only take part in the wave-wide votes)
        // (candidates popped per trip: the fullest lane of a 32-candidate word holds ~2 at 1000 agents, ~4 at 500)
        const float Tf = searcher ? __uint_as_float(Tb) : -1.0f;
#pragma unroll
        for (int k = 0; k < L; ++k) S[k] = 0xffffffffu;
        // Packed by id: ONE run, every candidate.  Packed by cell: the wavefront's searchers hold consecutive packed places,
        // i.e. the cells (x_lo, y_lo) .. (x_hi, y_hi) of the serpentine sweep; per cell row r, the cells within one column of
        // any of them in the rows r - 1 .. r + 1 are one run of packed places (the hull of the columns).  Runs are taken in
        // ascending order, widened to multiples of 4 places, never overlapping (a candidate must not enter the chain
        // twice) and joined where they touch.  All of it is scalar work; pass 1 and pass 2 have ONE call site each:
        // pass 1 lists up to WD_TC_LIST_CAP words of candidates (from one run or several), pass 2 empties the lists.
        const int C = cells_C;
        int r = 0, r_end = -1, run0 = 0, run1 = n_cand, x_lo = 0, y_lo = 0, x_hi = 0, y_hi = 0;
        if (C != 0) {
          const int last = min(63, __builtin_amdgcn_readfirstlane(n_live) - 1 - 64 * wave);
          x_lo = __builtin_amdgcn_readfirstlane(scell.x); y_lo = __builtin_amdgcn_readfirstlane(scell.y);
          x_hi = __builtin_amdgcn_readlane(scell.x, last); y_hi = __builtin_amdgcn_readlane(scell.y, last);
          r = max(0, y_lo - 1);
          r_end = min(C - 1, y_hi + 1);
          run1 = 0;
        }
        const TcPreList pl = tc_pre_list(stage, lane);
        int j = 0, j_end = 0;       // the part of the current run that pass 1 has not seen
        int listed = 0, words = 0;  // per lane: words in its list; wave-uniform: words since the last round of pass 2
        int probe_trips = 0;
        bool finished = false;
        while (!finished) {
          while (words < WD_TC_LIST_CAP) {
            if (j >= j_end) {  // the next run
              bool flush = false;
              int p0 = 0, p1 = 0;
              while (!flush && r <= r_end) {
                int lo = C, hi = -1;
                for (int yy = max(y_lo, r - 1); yy <= min(y_hi, r + 1); ++yy) {
                  // the columns of cell row yy that hold searchers of this wavefront: the sweep runs left to right in even
                  // rows, right to left in odd ones; it enters the wavefront at (x_lo, y_lo) and leaves it at (x_hi, y_hi)
                  const bool fwd = (yy & 1) == 0;
                  const int from = (yy == y_lo) ? x_lo : (fwd ? 0 : C - 1), to = (yy == y_hi) ? x_hi : (fwd ? C - 1 : 0);
                  lo = min(lo, min(from, to));
                  hi = max(hi, max(from, to));
                }
                lo = max(0, lo - 1);
                hi = min(C - 1, hi + 1);
                // (an odd row's cells are packed in descending column order)
                const int first_cell = r * C + ((r & 1) ? C - 1 - hi : lo), last_cell = r * C + ((r & 1) ? C - 1 - lo : hi);
                ++r;
                if (hi < lo) continue;
                const int b = first_cell == 0 ? 0 : __builtin_amdgcn_readlane(cell_incl, max(first_cell, 1) - 1);
                const int e = __builtin_amdgcn_readlane(cell_incl, last_cell);
                const int j0 = max(run1, b & ~3), j1 = min(n_cand, (e + 3) & ~3);
                if (j1 <= j0) continue;
                if (run1 > run0 && j0 != run1) { p0 = run0; p1 = run1; flush = true; run0 = j0; run1 = j1; }
                else if (run1 > run0) run1 = j1;
                else { run0 = j0; run1 = j1; }
              }
              if (!flush) {
                if (run1 <= run0) { finished = true; break; }
                p0 = run0; p1 = run1; run0 = run1;
              }
              j = p0;
              j_end = p1;
            }
            const int j_stop = min(j_end, j + 32 * (WD_TC_LIST_CAP - words));
            tc_pre_pass1(sxy, sx, sy, j, j_stop, Tf, pl, listed);
            words += (j_stop - j + 31) >> 5;
            j = j_stop;
          }
          if (words != 0) tc_pre_pass2<L, IDB>(sxy, sx, sy, pl, listed, S, extra, probe_trips);
          listed = 0;
          words = 0;
        }
        WD_TC_PROBE_VAL(18, probe_trips);
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
    exact = tc_resolve_keys<KMAX, IDB, L, TIE_TABLE>(sxy, ag, K, S, o, nid, rank, in_order, l.cid);
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
    if (!exact && !(EXACTK && IDB == 7) && (IDB == 7 || n_cand <= 128)) {
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
  if constexpr (IDB != 7 || EXACTK) {
    // more than 128 candidates, or the chain with ONE look-ahead entry (L above): the lanes that need the exact resolution
    // get it from the whole wavefront, one after the other (tc_zone_resolve)
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
        // (several replicas per block -- small replicas, never packed: the candidates are the replica's of lane fl)
        const float2 *const zxy = compact ? sxy : l.xy + __builtin_amdgcn_readlane(el, fl) * NP;
        const int cnt = tc_zone_resolve<TIE_TABLE>(zxy, n_cand, sx, sy, self, zh, IDB, K, (unsigned char *)stage, lane, l.cid);
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
        tc_knn_scan<KMAX, TIE_TABLE>(sxy, ag, n_cand, K, nid2, l.cid);  // (entries in the reference's order)
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
    int R = tc_stage_rows(F, n_waves);
    if constexpr (IDB != 7) {  // replicas of more than 128 agents: the buffer is at least the search's lists -- fill it
      R = max(R, min(min(64, (l.stage_dwords - 20) / F), 192 / K));
    }
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

}  // namespace
