// tag_continuous.hip -- TagContinuous step for gfx950.
//
// Semantics: the reference CPU step, example_envs/tag_continuous/tag_continuous.py
//   update_state :339-401, compute_distance :403-420, k_nearest_neighbors :422-444,
//   generate_observation :446-610, compute_reward :612-678, done :880-883.
// Where the reference's own CUDA kernel (tag_continuous_step_pycuda.cu:351-520)
// disagrees with its CPU step the CPU wins: stable (distance, id) neighbour order,
// tag counts accumulated without races, no end-of-game bonus for a runner tagged out
// on the last tick.  Argument order is the reference kernel's (:351-385) plus trailing
// n_envs and the two action-table lengths; the two O(N^2) global scratch arrays it sorts in HBM
// (neighbor_distances, neighbor_ids_sorted_by_distance; :167-199) are accepted and
// never touched.
//
// MI355X mapping (one block = `epb` consecutive replicas, thread t = agent t % N of
// local replica t / N; N = 105 -> 3 replicas fill 315 of 320 lanes; the reference
// geometry block=(N,1,1), grid=(E,1) is simply epb = 1):
//   phase 0  coalesced [E,N] loads, float32 kinematics (numpy-exact cos/sin), coalesced
//            stores; post-move state staged in LDS: float32 positions for distances and
//            the seven observation features per agent as float64 (x,y normalised in
//            float64, speed/acc/dir normalised in float32 then widened -- exactly the
//            reference's dtype flow, so the float64 neighbour difference is bit-exact).
//   phase 1  K nearest neighbours, ~28 VALU ops per candidate instead of a 50-op sorted
//            insertion:
//              A. stream all N candidates from LDS (wave-uniform address: broadcast) and
//                 keep only the K+1 smallest SQUARED distances in registers with a
//                 v_med3_f32 chain (B[k] = med3(B[k-1], B[k], d2): one op per slot, no
//                 compares, no ids, no serial dependency);
//              B. B[K] bounds the K-th neighbour.  Convert it to the float32 distance S
//                 the reference compares (sqrt rounds, so a RANGE [T2lo, T2hi] of squared
//                 distances maps to S), stream the candidates again and append the ones
//                 below the range, plus the first few inside it in id order, to a
//                 K-entry list in LDS -- exactly the reference's K smallest (distance, id)
//                 keys;
//              C. sort those <= K entries by (sqrt(d2), id) with a register sorting
//                 network and leave the ids in LDS.
//   phase 2  the packed replicas' observation block is contiguous in HBM ([E,N,F]); it
//            is produced by a block-strided gather from LDS -- every store instruction
//            writes 64 consecutive floats (the reference writes one 284-byte-strided row
//            per thread); (feature, slot) of a column comes from a small LDS table and
//            the (replica, agent, column) counters advance incrementally: no divisions.
//   phase 3  rewards: each runner scans the taggers (first minimum wins), tag counts go
//            through LDS atomics, float adds are replayed in the CPU's order.
#include "wd_common.h"

// Timing experiments only (scripts/ablate_tc.sh): bit 0 skips phase 1 (neighbour search),
// bit 1 skips phase 2 (observation gather), bit 2 stops phase 1 after part A, bit 3 after
// part B.  Results are wrong when any bit is set; the shipped code object uses 0.
#ifndef WD_TC_ABLATE
#define WD_TC_ABLATE 0
#endif
// -DWD_TC_PROFILE: thread 0 of every block stores s_memtime stamps at the phase boundaries into
// the (otherwise unused) neighbor_distances array, 16 x uint64 per block (scripts/phase_profile.py).
#ifdef WD_TC_PROFILE
#define WD_TC_STAMP(slot) do { if (threadIdx.x == 0 && l.prof) { l.prof[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); \
    if ((slot) == 0) l.prof[blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime(); \
    if ((slot) == 10) l.prof[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define WD_TC_STAMP(slot) do { } while (0)
#endif

// Start cohorts (see tc_step_impl): number of cohorts, block-id shift that selects the cohort,
// start offset between consecutive cohorts in ns.
#ifndef WD_TC_COHORTS
#define WD_TC_COHORTS 1
#endif
#ifndef WD_TC_COHORT_SHIFT
#define WD_TC_COHORT_SHIFT 8
#endif
#ifndef WD_TC_COHORT_NS
#define WD_TC_COHORT_NS 4000
#endif

// Store policy of the observation rows (timing experiments: scripts/store_policy_tc.sh).
// 0 = plain write-back stores, 1 = non-temporal, 2 = system-scope, 3 = agent-scope write-through.
#ifndef WD_TC_OBS_STORE
#define WD_TC_OBS_STORE 0
#endif

namespace {

__device__ __forceinline__ void tc_store_obs(float *p, float v) {
#if WD_TC_OBS_STORE == 1
  __builtin_nontemporal_store(v, p);
#elif WD_TC_OBS_STORE == 2
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif WD_TC_OBS_STORE == 3
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

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
  unsigned long long *prof;  // phase time stamps (profiling builds only)
};

struct TcCand {
  float d2;
  int id;
};

// extra inputs of the fused rollout tick (sample both action heads -> step -> reset finished
// replicas, ONE launch; see HipTagContinuousTick below)
struct TcResetEntry {  // same layout as wd_reset_entry in wd_core.hip
  uint32_t *data;
  const uint32_t *ref;
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
// 48 bytes = three 16-byte slots, so a neighbour is fetched with two ds_read_b128.
struct __attribute__((aligned(16))) TcFeat {
  double nx, ny;
  float nsp, nac, ndir;
  int type_sig;  // bit 0: agent type (1 = tagger), bit 1: still_in_the_game before tagging
  int pad_[2];
};

// LDS carve-up for `epb` packed replicas, A = epb * N agents (offsets multiples of 8).
struct TcLds {
  TcFeat *feat;      // [A] observation features (see TcFeat)
  TcCand *cand;      // [A][K+1] phase-1 lists; afterwards the first K ints of a row = neighbour ids
  float2 *xy;        // [A] positions after the move (x = +BIG for agents out of the game)
  int *sig;          // [A] still_in_the_game before this tick's tagging
  int *tagcnt;       // [A] tags credited to a tagger this tick
  int *types;        // [N]
  int *tagger_ids;   // [N] ascending
  float *acc_tab, *turn_tab;  // action tables (n_acc, n_turn entries; capacity 64 each)
  int *wave_cnt;     // [16] taggers per wavefront (rank computation)
  int *tstep, *nrun; // [epb]
  float *tfrac;      // [epb] float(t) / episode_length
  int *doneflag;     // [epb] replica finished on this tick (fused tick only)
  unsigned long long *prof;
};

#define WD_TC_TAB 64  // capacity of the LDS copies of the action tables

__device__ __forceinline__ size_t tc_align16(size_t v) { return (v + 15) & ~(size_t)15; }

// The per-trip work area (features, lists, positions, flags) doubles as the two probability slabs
// of the fused tick, which are dead before phase 0 writes it: min_area_bytes = both slabs.
__device__ __forceinline__ TcLds tc_carve(unsigned char *p0, int epb, int N, int K, size_t min_area_bytes) {
  // offsets only (no pointer differences: they turn LDS pointers into flat ones and back)
  TcLds l;
  const size_t A = (size_t)epb * N;
  size_t off = 0;
  l.feat = (TcFeat *)(p0 + off); off += sizeof(TcFeat) * A;
  l.cand = (TcCand *)(p0 + off); off += tc_align16(8 * A * (K + 1));
  l.xy = (float2 *)(p0 + off); off += 8 * A;
  l.sig = (int *)(p0 + off); off += 4 * A;
  l.tagcnt = (int *)(p0 + off); off += 4 * A;
  off = tc_align16(off > min_area_bytes ? off : min_area_bytes);
  l.types = (int *)(p0 + off); off += 4 * (size_t)N;
  l.tagger_ids = (int *)(p0 + off); off += 4 * (size_t)N;
  l.acc_tab = (float *)(p0 + off); off += 4 * WD_TC_TAB;
  l.turn_tab = (float *)(p0 + off); off += 4 * WD_TC_TAB;
  l.wave_cnt = (int *)(p0 + off); off += 4 * 16;
  l.tstep = (int *)(p0 + off); off += 4 * epb;
  l.nrun = (int *)(p0 + off); off += 4 * epb;
  l.tfrac = (float *)(p0 + off); off += 4 * epb;
  l.doneflag = (int *)(p0 + off);
  return l;
}

// Copy n floats global -> LDS with every thread's loads in flight before the first LDS write
// (a plain strided copy loop serialises load -> write per trip: ~1 us per trip at 4 waves/SIMD).
// src needs 4-byte alignment only; 16-byte vector loads are used on the aligned middle part.
__device__ __forceinline__ void tc_copy_to_lds(float *dst, const float *__restrict__ src, int n, int tid, int T_) {
  const int head = min(n, (int)(((16u - ((unsigned)(size_t)src & 15u)) & 15u) >> 2));
  const int nvec = (n - head) >> 2;
  const int tail0 = head + 4 * nvec;
  const float4 *v = (const float4 *)(src + head);
  constexpr int U = 6;  // vector loads kept in flight per thread per round
  for (int q0 = 0; q0 < nvec; q0 += U * T_) {
    float4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + u * T_ + tid;
      if (q < nvec) r[u] = v[q];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + u * T_ + tid;
      if (q < nvec) {
        float *d = dst + head + 4 * q;
        d[0] = r[u].x; d[1] = r[u].y; d[2] = r[u].z; d[3] = r[u].w;
      }
    }
  }
  if (tid < head) dst[tid] = src[tid];
  if (tid < n - tail0) dst[tail0 + tid] = src[tail0 + tid];
}

// ---- fused tick: probability slab of ONE wavefront.  The rows of a wavefront's 64 agents are one
// contiguous run of 64*n floats.  It goes global -> LDS directly (global_load_lds_dwordx4: per-lane
// global address, LDS destination = wave-uniform base + lane*16; dword-aligned sources are enough),
// 1 KiB per instruction, fully coalesced, no staging registers, asynchronous until the
// `s_waitcnt vmcnt(0)` before the rows are read back (stride n dwords).  Producer and consumer are
// the same wavefront: no block barrier.  A per-thread row walk instead costs 2n uncoalesced load
// instructions per thread, each touching 64 rows (measured: 13 k of the tick's 74 k cycles).
struct __attribute__((packed, aligned(4))) TcF4u { float x, y, z, w; };  // 16-byte access, dword aligned
#define WD_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define WD_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
__device__ __forceinline__ void tc_slab_fetch(float *dst, const float *__restrict__ src, int cnt, int lane) {
  const int nvec = cnt >> 2;
  const int nchunk = (nvec + 63) >> 6;  // wave-uniform
  for (int c = 0; c < nchunk; ++c) {
    const int q = c * 64 + lane;
    if (q < nvec) __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * q), WD_LDS_PTR(dst + 256 * c), 16, 0, 0);
  }
  if (lane < (cnt & 3))  // the < 4 floats after the last vector
    __builtin_amdgcn_global_load_lds(WD_GLOBAL_PTR(src + 4 * nvec + lane), WD_LDS_PTR(dst + 4 * nvec), 4, 0, 0);
}

// inverse CDF on a running float32 sum (random.cu:51-85): number of prefix sums < u, clamped
constexpr int TC_CH = 24;  // rows up to this length are read back with all LDS loads in flight
__device__ __forceinline__ int tc_slab_sample(const float *row, int n, float u) {
  int cnt = 0;
  float cum = 0.0f;
  if (n <= TC_CH) {
    float p[TC_CH];
#pragma unroll
    for (int i = 0; i < TC_CH; ++i) p[i] = row[i];  // immediate offsets; entries >= n are the next row's
                                                    // (or, after the last row, table bytes): read, never used
#pragma unroll
    for (int i = 0; i < TC_CH; ++i) {
      cum = (i == 0) ? p[0] : cum + p[i];
      cnt += (i < n && cum < u) ? 1 : 0;
    }
  } else {
    for (int i = 0; i < n; ++i) {
      cum = (i == 0) ? row[0] : cum + row[i];
      cnt += (cum < u) ? 1 : 0;
    }
  }
  return min(cnt, n - 1);
}

// every global input of one loop trip; issued together so the HBM latency is paid once
struct TcIn {
  int sg;
  float dir, acc, speed, x, y, skill;
  int2 sampled;
  uint32_t epoch;
};

template <bool FUSED>
__device__ __forceinline__ void tc_issue_loads(TcIn &in, const TcArgs &a, const TcFuse &fz, int env0, int epb,
                                               int N, int n_acc, int n_turn, int tid, float *slab_acc,
                                               float *slab_turn) {
  const int el = tid / N, ag = tid - el * N;
  const int env = env0 + el;
  const bool active = (el < epb) && (env < a.E);
  const int gi = env * N + ag;
  in.sg = 0; in.dir = in.acc = in.speed = in.x = in.y = in.skill = 0.f;
  in.sampled = make_int2(0, 0);
  in.epoch = 0u;
  if (active) {
    in.sg = a.sig_arr[gi];
    in.dir = a.direction[gi];
    in.acc = a.acceleration[gi];
    in.speed = a.speed[gi];
    in.x = a.loc_x[gi];
    in.y = a.loc_y[gi];
    in.skill = a.skill_levels[ag];
    if (!FUSED) in.sampled = ((const int2 *)a.actions)[gi];
    if (FUSED) in.epoch = fz.rng_state[WD_RNG_HEADER + gi];
  }
  if (FUSED) {
    // this wavefront's rows of both probability slabs -> LDS
    const int rows_here = min(epb, a.E - env0) * N;
    const int r0 = (tid >> 6) * 64, lane = tid & 63;
    const int wrows = max(0, min(64, rows_here - r0));
    tc_slab_fetch(slab_acc + (size_t)r0 * n_acc, fz.probs_acc + ((long)env0 * N + r0) * n_acc, wrows * n_acc, lane);
    tc_slab_fetch(slab_turn + (size_t)r0 * n_turn, fz.probs_turn + ((long)env0 * N + r0) * n_turn, wrows * n_turn,
                  lane);
  }
}

#define WD_BIG 1.0e30f  // (x - BIG)^2 overflows to +inf: such a candidate is never selected

// compare-exchange of (distance, id) keys, ascending.  distance >= 0, so its float bits order
// like an unsigned integer and (bits << 32 | id) is one total 64-bit key.
__device__ __forceinline__ void tc_cex(unsigned long long &a, unsigned long long &b) {
  const bool swap = a > b;
  const unsigned long long lo = swap ? b : a, hi = swap ? a : b;
  a = lo;
  b = hi;
}

// Batcher's merge-exchange sorting network (Knuth 5.2.2 Algorithm M) for n keys, built at
// compile time and fully unrolled: 31 compare-exchanges for n = 10 (odd-even transposition
// needs 45).
template <int n>
struct TcNet {
  int a[n * 8 + 1] = {}, b[n * 8 + 1] = {};
  int count = 0;
};

template <int n>
constexpr TcNet<n> tc_make_net() {
  TcNet<n> net;
  int t = 0;
  while ((1 << t) < n) ++t;
  if (t == 0) return net;
  for (int p = 1 << (t - 1); p > 0; p >>= 1) {
    int q = 1 << (t - 1), r = 0, d = p;
    for (;;) {
      for (int i = 0; i + d < n; ++i)
        if ((i & p) == r) { net.a[net.count] = i; net.b[net.count] = i + d; ++net.count; }
      if (q == p) break;
      d = q - p;
      q >>= 1;
      r = p;
    }
  }
  return net;
}

template <int n>
__device__ __forceinline__ void tc_sort_network(unsigned long long (&key)[n]) {
  constexpr TcNet<n> net = tc_make_net<n>();
#pragma unroll
  for (int c = 0; c < net.count; ++c) tc_cex(key[net.a[c]], key[net.b[c]]);
}

// ---- phase 1, register-resident variant (K <= KMAX) --------------------------------
template <int KMAX>
__device__ __forceinline__ void tc_knn_registers(const TcLds &l, int el, int ag, int li, int N, int K) {
  const float2 *cxy = l.xy + el * N;
  TcCand *mine = l.cand + (size_t)li * (K + 1);
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  const float INF = __builtin_inff();

  // A. K+1 smallest squared distances over ALL agents of the replica (self contributes 0,
  //    agents out of the game contribute +inf)
  float B[KMAX + 1];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) B[k] = INF;
  for (int j = 0; j < N; ++j) {
    const float2 pj = cxy[j];
    const float dx = xi - pj.x, dy = yi - pj.y;
    const float d2 = dx * dx + dy * dy;
#pragma unroll
    for (int k = KMAX; k >= 1; --k) B[k] = __builtin_amdgcn_fmed3f(B[k - 1], B[k], d2);
    B[0] = fminf(B[0], d2);
  }
  if (WD_TC_ABLATE & 4) { ((float *)mine)[0] = B[KMAX]; return; }
#ifdef WD_TC_PROFILE
  if (threadIdx.x == 0 && l.prof) l.prof[blockIdx.x * 16 + 4] = __builtin_readcyclecounter();
#endif
  // B[k], k = 1..K are the K smallest squared distances to OTHER agents (B[0] is self or a
  // co-located twin).  T2 = the K-th of them.
  float T2 = INF;
#pragma unroll
  for (int k = 1; k <= KMAX; ++k) T2 = (k == K) ? B[k] : T2;
  // range of squared distances whose float32 sqrt equals S = sqrtf(T2)
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
  unsigned long long key[KMAX];
  if (N <= 128) {
    // B. second pass: one 128-bit per-lane mask "inside or below the range".  Each candidate costs
    //    a squared distance, one compare and one shift-in-the-carry add (m = 2m + bit); no LDS
    //    traffic, no data-dependent addressing.  Candidate b of word w lands on bit (nb-1-b):
    //    undone with one bit-reverse per word.
    unsigned sel[4] = {0u, 0u, 0u, 0u};
    int n_upto = 0;
    // (loop-invariant across trips, but hoisting them costs long-lived registers the kernel does not
    // have at 128 VGPRs: the empty asm statements pin their computation here)
    int ag_here = ag, n_here = N;
    asm volatile("" : "+v"(ag_here));
    asm volatile("" : "+s"(n_here));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j0 = 32 * w;
      if (j0 < n_here) {  // wave-uniform
        const int nb = min(32, n_here - j0);
        unsigned mu = 0u;
        // m = 2m + (d2 <= T2hi): compare into VCC, add with carry-in (2 VALU ops per candidate).
        // Unrolled by hand (loops holding inline asm are not unrolled by the compiler) so the four
        // LDS reads of a group are in flight together.
#define WD_TC_PUSH(m, d2v, thr, op) \
  asm("v_cmp_" op "_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2v), "v"(thr) : "vcc")
        int b = 0;
        for (; b + 4 <= nb; b += 4) {
          const float2 p0 = cxy[j0 + b], p1 = cxy[j0 + b + 1], p2 = cxy[j0 + b + 2], p3 = cxy[j0 + b + 3];
          const float ax = xi - p0.x, ay = yi - p0.y, bx = xi - p1.x, by = yi - p1.y;
          const float cx = xi - p2.x, cy = yi - p2.y, ex = xi - p3.x, ey = yi - p3.y;
          const float d0 = ax * ax + ay * ay, d1 = bx * bx + by * by, d2 = cx * cx + cy * cy, d3 = ex * ex + ey * ey;
          WD_TC_PUSH(mu, d0, T2hi, "le");
          WD_TC_PUSH(mu, d1, T2hi, "le");
          WD_TC_PUSH(mu, d2, T2hi, "le");
          WD_TC_PUSH(mu, d3, T2hi, "le");
        }
        for (; b < nb; ++b) {
          const float2 pj = cxy[j0 + b];
          const float dx = xi - pj.x, dy = yi - pj.y;
          const float d2 = dx * dx + dy * dy;
          WD_TC_PUSH(mu, d2, T2hi, "le");
        }
        const unsigned self_bit = ((ag_here >> 5) == w) ? (1u << (ag_here & 31)) : 0u;
        sel[w] = (__brev(mu) >> (32 - nb)) & ~self_bit;
        n_upto += __popc(sel[w]);
      }
    }
    // Usually exactly K others are inside or below the range and the mask is the answer.  More
    // than K means several candidates share the K-th float32 distance: the reference keeps the
    // lowest ids among them (stable heapq.nsmallest, :435-437).  Rare (a float32 sqrt tie at the
    // cut), so the "strictly below" mask is only built then.
    if (n_upto > K) {
      unsigned lo[4] = {0u, 0u, 0u, 0u};
      int c_less = 0;  // others strictly below the range
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int j0 = 32 * w;
        if (j0 < n_here) {
          const int nb = min(32, n_here - j0);
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
    if (WD_TC_ABLATE & 8) { ((unsigned *)mine)[0] = sel[0] ^ sel[1] ^ sel[2] ^ sel[3]; return; }
#ifdef WD_TC_PROFILE
    if (threadIdx.x == 0 && l.prof) l.prof[blockIdx.x * 16 + 5] = __builtin_readcyclecounter();
#endif
    // C. peel the (at most K) ids off the mask in ascending order, rebuild their distances
    //    and form 64-bit keys (float bits of sqrt(d2) << 32 | id)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int which = sel[0] ? 0 : sel[1] ? 1 : sel[2] ? 2 : sel[3] ? 3 : 4;
      const unsigned cur = sel[0] ? sel[0] : sel[1] ? sel[1] : sel[2] ? sel[2] : sel[3];
      const bool have = which < 4;
      const int j = have ? which * 32 + (__ffs(cur) - 1) : ag;
      const unsigned cleared = cur & (cur - 1u);
      sel[0] = (which == 0) ? cleared : sel[0];
      sel[1] = (which == 1) ? cleared : sel[1];
      sel[2] = (which == 2) ? cleared : sel[2];
      sel[3] = (which == 3) ? cleared : sel[3];
      const float2 pj = cxy[j];
    const float dx = xi - pj.x, dy = yi - pj.y;
      const unsigned long long sbits =
          have ? (unsigned long long)__float_as_uint(sqrtf(dx * dx + dy * dy)) : 0x7f800000ull;
      key[k] = (sbits << 32) | (unsigned long long)(unsigned int)(have ? j : 0xffff);
    }
  } else {
    // B'. more than 128 agents: same selection, collected in a (K+1)-slot LDS row.  Branch-free:
    //     every candidate is written to the next free slot and the slot only advances when it
    //     was selected (the write after the K-th selection stays inside the row).
    int c_less = 0;  // others strictly below the range (counting pre-pass)
    for (int j = 0; j < N; ++j) {
      const float2 pc = cxy[j];
      const float dx = xi - pc.x, dy = yi - pc.y;
      c_less += (j != ag && (dx * dx + dy * dy) < T2lo) ? 1 : 0;
    }
    const int need_tie = K - c_less;
    int cnt = 0, tie_taken = 0;
    for (int j = 0; j < N; ++j) {
      const float2 pj = cxy[j];
    const float dx = xi - pj.x, dy = yi - pj.y;
      const float d2 = dx * dx + dy * dy;
      const bool other = (j != ag);
      const bool less = other && (d2 < T2lo);
      const bool tie = other && !less && (d2 <= T2hi) && (tie_taken < need_tie);
      mine[cnt] = TcCand{d2, j};
      cnt += (less || tie) ? 1 : 0;
      tie_taken += tie ? 1 : 0;
    }
    if (WD_TC_ABLATE & 8) return;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const bool have = k < cnt;
      const TcCand c = mine[have ? k : 0];
      const unsigned long long sbits = have ? (unsigned long long)__float_as_uint(sqrtf(c.d2)) : 0x7f800000ull;
      key[k] = (sbits << 32) | (unsigned long long)(unsigned int)(have ? c.id : 0xffff);
    }
  }
  tc_sort_network<KMAX>(key);
#ifdef WD_TC_PROFILE
  if (threadIdx.x == 0 && l.prof) l.prof[blockIdx.x * 16 + 6] = __builtin_readcyclecounter();
#endif
  int *out = (int *)mine;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) out[k] = (key[k] >> 32) == 0x7f800000ull ? -1 : (int)(unsigned int)key[k];
}

// ---- phase 1, generic K: K passes, each picks the smallest (d, id) key above the previous
__device__ __forceinline__ void tc_knn_generic(const TcLds &l, int el, int ag, int li, int N, int K) {
  const float2 *cxy = l.xy + el * N;
  const int *csig = l.sig + el * N;
  int *out = (int *)(l.cand + (size_t)li * (K + 1));
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

template <int KMAX, bool FUSED>
__device__ __forceinline__ void tc_step_impl(const TcArgs &a, const TcFuse &fz, unsigned char *smem, int n_acc,
                                             int n_turn) {
  const int N = a.N, K = a.use_full_obs ? 0 : a.K;
  const int W = a.use_full_obs ? (N - 1) : K;  // columns per feature
  const int F = 7 * W + 1;
  const int epb = max(1, (int)blockDim.x / N);
  const size_t slab_acc_bytes = tc_align16((size_t)4 * epb * N * n_acc);
  const TcLds l = tc_carve(smem, epb, N, K,
                           FUSED ? slab_acc_bytes + tc_align16((size_t)4 * epb * N * n_turn) : 0);
  float *const slab_acc = (float *)smem, *const slab_turn = (float *)(smem + slab_acc_bytes);
  const int tid = threadIdx.x, T_ = blockDim.x;
  const int el = tid / N, ag = tid - el * N;
  // (profiling builds: a launch over replicas [env_begin, E) stamps the rows of its own blocks)
  const_cast<TcLds &>(l).prof = a.prof ? a.prof + (size_t)(a.env_begin / epb) * 16 : nullptr;
  const int row_ints = 2 * (K + 1);                    // ints per agent row of the id list
  const float two_pi = 6.2831854820251465f;            // float32(2*pi), :356
  const float L = a.grid_length;
  const double diag = (double)L * 1.4142135623730951;  // float32 L * np.sqrt(2) -> f64, :146
  const float sp_div = a.max_speed + 1.0e-10f;         // float32 + float32(eps), :456

#if WD_TC_COHORTS > 1
  // Start cohort: blocks of different cohorts begin WD_TC_COHORT_NS apart, so that the memory-bound
  // phases (probability fetch, observation stores) of one cohort run under the VALU-bound neighbour
  // search of another instead of all blocks marching through the phases in lock step.
  {
    const unsigned cohort = (blockIdx.x >> WD_TC_COHORT_SHIFT) % WD_TC_COHORTS;
    if (cohort) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long wait = (unsigned long long)cohort * (WD_TC_COHORT_NS / 10);
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(2);
    }
  }
#endif
  WD_TC_STAMP(0);
  // the first trip's global loads go out before anything else: the table set-up below (a
  // dependent global load + barrier) then runs in their shadow
  TcIn in;
  tc_issue_loads<FUSED>(in, a, fz, a.env_begin + blockIdx.x * epb, epb, N, n_acc, n_turn, tid, slab_acc, slab_turn);

  // ---- replica-independent tables: agent types, ascending tagger list, action tables
  const bool tab_in_lds = (n_acc <= WD_TC_TAB) && (n_turn <= WD_TC_TAB);
  if (tab_in_lds) {
    for (int i = tid; i < n_acc; i += T_) l.acc_tab[i] = a.acc_actions[i];
    for (int i = tid; i < n_turn; i += T_) l.turn_tab[i] = a.turn_actions[i];
  }
  int n_taggers = 0;
  {
    // rank of a tagger = number of taggers with a smaller id: wave ballots + per-wave counts
    const int wave = tid >> 6, lane = tid & 63, n_waves = (T_ + 63) >> 6;
    if (N <= T_) {  // usual case: one barrier
      const int ty = (tid < N) ? a.agent_types[tid] : 0;
      if (tid < N) l.types[tid] = ty;
      const unsigned long long m = __ballot(ty == 1);
      if (lane == 0) l.wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int before = 0;
      for (int w2 = 0; w2 < n_waves; ++w2) {
        const int c = l.wave_cnt[w2];
        before += (w2 < wave) ? c : 0;
        n_taggers += c;
      }
      if (ty == 1) l.tagger_ids[before + __popcll(m & ((1ull << lane) - 1ull))] = tid;
    } else {
      for (int base = 0; base < N; base += T_) {
        const int i = base + tid;
        const int ty = (i < N) ? a.agent_types[i] : 0;
        if (i < N) l.types[i] = ty;
        const unsigned long long m = __ballot(ty == 1);
        if (lane == 0) l.wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = n_taggers;
        for (int w2 = 0; w2 < wave; ++w2) before += l.wave_cnt[w2];
        if (ty == 1) l.tagger_ids[before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        for (int w2 = 0; w2 < n_waves; ++w2) n_taggers += l.wave_cnt[w2];
        __syncthreads();
      }
    }
  }
  // (tagger_ids / action tables are first read after the barriers inside the replica loop)

  // rotated loop: the loads of trip i+1 are issued at the end of trip i, so `in` is live only
  // from issue to use (never across the body)
  int env0 = a.env_begin + blockIdx.x * epb;
  if (env0 >= a.E) return;  // whole block (no barrier is skipped by part of a block)
  while (true) {
    const int env = env0 + el;
    const bool active = (el < epb) && (env < a.E);
    const int gi = env * N + ag;  // index into [E, N] arrays
    const int li = el * N + ag;   // index into LDS arrays
    float edge_pen = 0.0f, my_x = 0.0f, my_y = 0.0f;
    const int sg = in.sg;
    const float dir_in = in.dir, acc_in = in.acc, speed_in = in.speed, x_in = in.x, y_in = in.y, skill = in.skill;
    int2 sampled = in.sampled;
    WD_TC_STAMP(1);

    // ------------------------------------------- fused tick: sample both action heads
    // (replaces two sample_actions launches, random.cu:51-85): inverse CDF on a running
    // float32 sum, one Philox call for both heads.
    if (FUSED) {
      if (active && ag == 0) a.done[env] = 0;  // a replica that finished (and was reset) last tick
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
      if (active) {
        sampled.x = tc_slab_sample(slab_acc + (size_t)li * n_acc, n_acc, wd_u01_open_closed(rnd.x));
        sampled.y = tc_slab_sample(slab_turn + (size_t)li * n_turn, n_turn, wd_u01_open_closed(rnd.y));
      }
      if (active) ((int2 *)fz.actions_out)[gi] = sampled;
    }
    __syncthreads();  // prologue tables (first iteration) / previous iteration's LDS readers

    WD_TC_STAMP(2);
    // ------------------------------------------------------------ phase 0: move
    if (active) {
      const float s = (float)sg;
      const int2 act = sampled;
      // (value select, not pointer select: a pointer that may be LDS or global becomes a flat access)
      float d_acc = l.acc_tab[min(act.x, WD_TC_TAB - 1)], d_turn = l.turn_tab[min(act.y, WD_TC_TAB - 1)];
      asm volatile("" : "+v"(d_acc), "+v"(d_turn));  // keeps the two loads from being merged into one flat load
      if (!tab_in_lds) {
        d_acc = a.acc_actions[act.x];
        d_turn = a.turn_actions[act.y];
      }
      const float dir = wd_np_remainderf(dir_in + d_turn, two_pi) * s;            // :355-357
      float acc = acc_in + d_acc;                                                 // :359
      const float vmax = a.max_speed * skill;                                     // :363
      float v = speed_in + acc;
      v = fminf(fmaxf(v, 0.0f), vmax) * s;                                        // :364-366
      acc = acc * (v > 0.0f ? 1.0f : 0.0f) * (v < vmax ? 1.0f : 0.0f);            // :367
      float sn, cs;
      wd_np_sincosf(dir, sn, cs);
      float px = x_in + v * cs;                                                   // :369-374
      float py = y_in + v * sn;
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
      my_x = px;
      my_y = py;
      // agents out of the game are pushed to +BIG for the neighbour search only; every other
      // consumer (taggers are never out of the game) reads real positions
      l.xy[li] = make_float2(sg ? px : WD_BIG, py);
      TcFeat ft;
      ft.nx = (double)px / diag;    // :462 (float64 division)
      ft.ny = (double)py / diag;
      ft.nsp = v / sp_div;          // float32 division (:456-458)
      ft.nac = acc / sp_div;
      ft.ndir = dir / two_pi;
      ft.type_sig = (l.types[ag] & 1) | (sg ? 2 : 0);
      ft.pad_[0] = 0; ft.pad_[1] = 0;
      l.feat[li] = ft;
      l.sig[li] = sg;
      l.tagcnt[li] = 0;
      if (ag == 0) {
        const int t = a.timestep[env] + 1;  // :800
        a.timestep[env] = t;
        l.tstep[el] = t;
        l.tfrac[el] = (float)((double)t / (double)a.T);  // float(t) / episode_length, :474
        l.nrun[el] = a.num_runners[env];
      }
    }
    __syncthreads();

    WD_TC_STAMP(3);
    // ------------------------------------------------ phase 1: K nearest neighbours
    if (!a.use_full_obs && active && !(WD_TC_ABLATE & 1)) {
      if (l.sig[li]) {
        if (KMAX > 0) tc_knn_registers<(KMAX > 0 ? KMAX : 1)>(l, el, ag, li, N, K);
        else tc_knn_generic(l, el, ag, li, N, K);
      } else {
        int *out = (int *)(l.cand + (size_t)li * (K + 1));
        for (int k = 0; k < K; ++k) out[k] = -1;
      }
    }
    __syncthreads();

    WD_TC_STAMP(7);
    // ------------------------------------------------ phase 2: observations
    // One work item = (agent row m, neighbour slot k): it reads the neighbour id once, then
    // the 7 features of that neighbour and of the agent, and writes the 7 columns
    // {c*W + k} of the row.  Consecutive lanes hold consecutive k, so every store
    // instruction writes runs of W consecutive floats (whole 256-byte lines in the
    // full-observation mode); only two dependent LDS round trips per 7 outputs.
    if (!(WD_TC_ABLATE & 2)) {
      const int agents_here = min(epb, a.E - env0) * N;
      const int items = agents_here * W;
      float *obs_blk = a.obs + (long)env0 * N * F;
      const int Wd = max(W, 1);
      int k = tid % Wd, m = tid / Wd;         // block-local agent row and slot of the first item
      int i = m % N;                          // agent id inside its replica
      const int sk = T_ % Wd, sm = T_ / Wd, si = sm % N;
      if (KMAX == 0 && a.use_full_obs && (W & 3) == 0 && W > 0 && !(WD_TC_ABLATE & (16 | 64 | 128))) {
        // (generic entry points only: the host launches full observations through them, and the
        // K-specialised kernels have no registers to spare)
        // Full observations: rows are 7 runs of W consecutive floats, and the phase is bound by the
        // number of store instructions (612 MB per tick at N = 105).  One work item = (row, four
        // consecutive slots): 28 values, seven 16-byte stores (dword-aligned addresses) -- 4x fewer
        // store instructions than one float per lane.
        const int nq = W >> 2;
        int q = tid % nq, mq = tid / nq, iq = mq % N;
        const int sq = T_ % nq, smq = T_ / nq, siq = smq % N;
        for (int t = tid; t < agents_here * nq; t += T_) {
          const int ebase = mq - iq;
          const bool in_game = l.sig[mq] != 0;
          const TcFeat me = l.feat[mq];
          float v[7][4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int kcol = 4 * q + kk;
            const TcFeat nb = l.feat[ebase + kcol + (kcol >= iq ? 1 : 0)];
            v[0][kk] = in_game ? (float)(nb.nx - me.nx) : 0.0f;
            v[1][kk] = in_game ? (float)(nb.ny - me.ny) : 0.0f;
            v[2][kk] = in_game ? (nb.nsp - me.nsp) : 0.0f;
            v[3][kk] = in_game ? (nb.nac - me.nac) : 0.0f;
            v[4][kk] = in_game ? (nb.ndir - me.ndir) : 0.0f;
            v[5][kk] = (float)(nb.type_sig & 1);
            v[6][kk] = (float)((nb.type_sig >> 1) & 1);
          }
          float *row = obs_blk + (long)mq * F + 4 * q;
#pragma unroll
          for (int c = 0; c < 7; ++c) *(TcF4u *)(row + c * W) = TcF4u{v[c][0], v[c][1], v[c][2], v[c][3]};
          q += sq;
          const int carry = (q >= nq) ? 1 : 0;
          q -= carry ? nq : 0;
          mq += smq + carry;
          iq += siq + carry;
          iq -= (iq >= N) ? N : 0;
        }
      } else
      for (int t = tid; t < items; t += T_) {
        const int ebase = m - i;              // first agent of this row's replica
        const bool in_game = l.sig[m] != 0;
        int j;
        bool valid;
        if (a.use_full_obs) {
          j = k + (k >= i ? 1 : 0);
          valid = true;   // type / still_in_game columns are filled even for agents out of the game
        } else {
          j = ((const int *)l.cand)[(size_t)m * row_ints + k];
          valid = in_game && (j >= 0);
          j = max(j, 0);
        }
        const int o = ebase + j;
        float *row = obs_blk + (long)m * F;
        const bool rel = valid && in_game;  // relative features only for agents in the game
        TcFeat nb, me;
        if (WD_TC_ABLATE & 64) {  // timing experiment: no LDS feature reads
          nb.nx = 1.0 + o; nb.ny = 2.0; nb.nsp = 0.5f; nb.nac = 0.25f; nb.ndir = 0.125f; nb.type_sig = o;
          me = nb; me.nx = 0.5;
        } else {
          nb = l.feat[o];
          me = l.feat[m];
        }
        float vals[7];
        // float64 differences (:560), narrowed to float32 like the reference's device push
        vals[0] = rel ? (float)(nb.nx - me.nx) : 0.0f;
        vals[1] = rel ? (float)(nb.ny - me.ny) : 0.0f;
        // speed / acc / dir: the reference widens float32 values and subtracts in float64; for
        // float32 operands that rounds to exactly the float32 difference (53 >= 2*24+2 bits:
        // double rounding is innocuous), so no float64 arithmetic is needed here
        vals[2] = rel ? (nb.nsp - me.nsp) : 0.0f;
        vals[3] = rel ? (nb.nac - me.nac) : 0.0f;
        vals[4] = rel ? (nb.ndir - me.ndir) : 0.0f;
        vals[5] = valid ? (float)(nb.type_sig & 1) : 0.0f;
        vals[6] = valid ? (float)((nb.type_sig >> 1) & 1) : 0.0f;
        if (WD_TC_ABLATE & 16) {  // timing experiment: keep the math, drop the stores
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 7; ++c) acc += vals[c];
          if (acc == 123.456f) row[0] = acc;
        } else {
#pragma unroll
          for (int c = 0; c < 7; ++c) {
            if (WD_TC_ABLATE & 128)  // timing experiment: same bytes, fully coalesced (wrong layout)
              tc_store_obs(obs_blk + (long)c * items + t, vals[c]);
            else
              tc_store_obs(row + c * W + k, vals[c]);
          }
        }
        // advance (m, i, k) by the block stride
        k += sk;
        int carry = (k >= W) ? 1 : 0;
        k -= carry ? W : 0;
        m += sm + carry;
        i += si + carry;
        i -= (i >= N) ? N : 0;
      }
      // time column: float(t) / episode_length for agents in the game, else 0 (:474,:493,:543)
      if (!(WD_TC_ABLATE & 32))
      for (int m0 = tid; m0 < agents_here; m0 += T_)
        tc_store_obs(obs_blk + (long)m0 * F + 7 * W, (l.sig[m0] != 0) ? l.tfrac[m0 / N] : 0.0f);
      if (!a.use_full_obs && K > 0 && !(WD_TC_ABLATE & 32)) {
        int *nb_blk = a.nearest_ids + (long)env0 * N * K;
        int k2 = tid % K, m2 = tid / K;
        const int sk2 = T_ % K, sm2 = T_ / K;
        for (int q = tid; q < agents_here * K; q += T_) {
          nb_blk[q] = ((const int *)l.cand)[(size_t)m2 * row_ints + k2];
          k2 += sk2;
          const int carry = (k2 >= K) ? 1 : 0;
          k2 -= carry ? K : 0;
          m2 += sm2 + carry;
        }
      }
    }

    WD_TC_STAMP(9);
    // ------------------------------------------------------------ phase 3: rewards
    float rew = 0.0f;
    bool tagged = false, is_runner = false;
    if (active) {
      const int sg = l.sig[li];
      if (sg) { rew += edge_pen; rew += a.step_rewards[ag]; }  // :655-658
      is_runner = (l.types[ag] == 0) && (sg != 0);              // member of self.runners
      if (is_runner) {
        float best = __builtin_inff();
        int bt = -1;
        for (int t = 0; t < n_taggers; ++t) {  // ascending ids, first minimum wins :643-651
          const int j = l.tagger_ids[t];
          const float2 pt = l.xy[el * N + j];
          const float dx = my_x - pt.x, dy = my_y - pt.y;
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
      if (tagged) rew += a.tag_penalty;                             // :664
      const int c = l.tagcnt[li];
      for (int k = 0; k < c; ++k) rew += a.tag_reward;              // :665, one add per tag
      const bool still_runner = is_runner && !(tagged && a.runner_exits);
      if (l.tstep[el] == a.T && still_runner) rew += a.end_reward;  // :674-676
      a.rewards[gi] = rew;
      if (tagged && a.runner_exits) a.sig_arr[gi] = 0;              // :669
      if (ag == 0) {
        const int nr = l.nrun[el];
        a.num_runners[env] = nr;
        const bool fin = (l.tstep[el] >= a.T || nr == 0);           // :880-883
        if (fin) a.done[env] = 1;
        if (FUSED) l.doneflag[el] = fin ? 1 : 0;
      }
    }
    __syncthreads();
    WD_TC_STAMP(10);
    // ------------------------------------------- fused tick: reset finished replicas in place
    // (reset.cu:9-75 for every registered array).  `_done_` stays 1 so the trainer can read
    // which replicas finished on this tick; the next tick clears it.  All writes of this block
    // to these rows happened before the barrier above.
    if (FUSED) {
      const int envs_here = min(epb, a.E - env0);
      for (int e = 0; e < envs_here; ++e) {
        if (l.doneflag[e] == 0) continue;  // block-uniform
        for (int r = 0; r < fz.n_reset_arrays; ++r) {
          const TcResetEntry ent = fz.reset_table[r];
          const long base = (long)(env0 + e) * ent.row_elems;
          for (int i = tid; i < ent.row_elems; i += T_) ent.data[base + i] = ent.ref[base + i];
        }
        if (tid == 0) a.timestep[env0 + e] = 0;
      }
      __syncthreads();
    }
    env0 += gridDim.x * epb;
    if (env0 >= a.E) break;
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
      int kNumAccelerationActions, int kNumTurnActions, int kEnvBegin

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
  a.E = kNumEnvs; a.env_begin = kEnvBegin;                                                        \
  a.prof = (unsigned long long *)neighbor_distances_arr; (void)neighbor_ids_sorted_by_distance_arr;

extern "C" {

// generic entry: any K (and the full-observation mode)
__global__ void HipTagContinuousStep(WD_TC_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  WD_TC_PACK();
  tc_step_impl<0, false>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions);
}

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

__global__ void HipTagContinuousTick(WD_TC_PARAMS WD_TC_FUSE_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  WD_TC_PACK();
  WD_TC_FUSE_PACK();
  tc_step_impl<0, true>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);
}

// register-resident top-K specialisations (blocks of <= 512 threads); the host picks the
// smallest KMAX >= K and falls back to the generic entry for > 512 agents per replica
#define WD_TC_SPECIALISE(KM, WAVES)                                                                 \
  __global__ void __launch_bounds__(512, WAVES) HipTagContinuousStep_K##KM(WD_TC_PARAMS) {            \
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];                    \
    WD_TC_PACK();                                                                              \
    tc_step_impl<KM, false>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions);   \
  }                                                                                            \
  __global__ void __launch_bounds__(512, WAVES) HipTagContinuousTick_K##KM(WD_TC_PARAMS WD_TC_FUSE_PARAMS) { \
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];                    \
    WD_TC_PACK();                                                                              \
    WD_TC_FUSE_PACK();                                                                         \
    tc_step_impl<KM, true>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions);          \
  }
WD_TC_SPECIALISE(2, 4)
WD_TC_SPECIALISE(4, 4)
WD_TC_SPECIALISE(6, 4)
WD_TC_SPECIALISE(8, 4)
WD_TC_SPECIALISE(10, 4)
WD_TC_SPECIALISE(12, 3)
WD_TC_SPECIALISE(16, 3)
WD_TC_SPECIALISE(24, 2)
WD_TC_SPECIALISE(32, 2)

}  // extern "C"
