// tc_knn.h -- K nearest neighbours of the fast path: one-pass packed-key chain, exact tie resolution, prefiltered search of big replicas, fallbacks.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_types.h"

namespace {

// =====================================================================================
//                   fast path: N <= 512, partial observations, K <= KMAX
// =====================================================================================

// ---- THE tie-break order of a candidate index.  The reference orders neighbours by (float32 distance, agent id)
// (heapq.nsmallest on (distance, id) tuples, tag_continuous.py:422-444).  Every routine below works on candidate INDICES
// (positions in the packed candidate array) and settles equal distances through this one function:
//   TABLE == false  candidates are packed in ascending agent-id order (always so up to 128 agents; the identity folds away)
//   TABLE == true   `tie` = the packed-index -> agent-id table (`TcFastLds::cid`, tie[1 + idx]): candidates may be packed
//                   in ANY order -- e.g. by grid cell (tc_fast.h, replicas of more than 128 agents); null = no table
//                   (several replicas per block: never packed, index = id)
template <bool TABLE>
__device__ __forceinline__ int tc_tie_order(const short *tie, int idx) {
  if constexpr (TABLE) return tie ? (int)tie[1 + idx] : idx;
  else return idx;
}

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

// (candidates in ascending agent-id order only -- tc_tie_order<false>: it peels its masks in index order and ranks equal
// distances by index; the caller uses it up to 128 candidates, which are never packed in another order)
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
template <int KMAX, bool TABLE = false>
__device__ __forceinline__ void tc_knn_scan(const float2 *cxy, int ag, int N, int K, int (&nid)[KMAX],
                                            const short *tie = nullptr) {
  const float xi = cxy[ag].x, yi = cxy[ag].y;
  float pd = -1.0f;
  int pj = -1, pt = -1;  // the previous pick: index and tie-break order
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    float best = __builtin_inff();
    int bj = -1, bt = 0x7fffffff;
    for (int j = 0; j < N; ++j) {
      const float2 pc = cxy[j];
      const float dx = xi - pc.x, dy = yi - pc.y;
      const float d = sqrtf(dx * dx + dy * dy);
      if constexpr (TABLE) {  // smallest (distance, id) key above the previous pick's
        const int tj = tc_tie_order<true>(tie, j);
        const bool above = (d > pd) || (d == pd && tj > pt);
        if (j != ag && above && (d < best || (d == best && tj < bt))) { best = d; bj = j; bt = tj; }
      } else {               // ascending index = ascending id: the first minimum wins
        const bool above = (d > pd) || (d == pd && j > pj);
        if (j != ag && above && d < best) { best = d; bj = j; }
      }
    }
#pragma unroll
    for (int q = 0; q < KMAX; ++q) nid[q] = (q == k) ? bj : nid[q];
    if (bj < 0) break;  // fewer than K candidates (the remaining entries stay -1)
    pd = best;
    pj = bj;
    pt = bt;
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
// 1005-agent replica: 80 % of its tick.  The searcher's K + 3 nearest others of the PREVIOUS tick (32 bytes per agent in
// HBM, `knn_prev`) give a radius that holds the K nearest now (tc_knn_bound16); since round 6 the agents in the game are
// packed by grid cell and the radius is also capped by the searcher's 3 x 3 cell block ("CELL-SORTED packing" below):
//   pass 1  the candidates of the cell rows around the wavefront's searchers: squared distance and ONE subtraction from
//           the radius whose sign is shifted into a per-lane bit mask (v_alignbit) -- 7 full-rate instructions;
//   pass 2  the candidates whose bit is set (~21 per lane) go through the chain: every lane pops its own lowest set bit
//           from its own list of non-empty mask words (tc_pre_pass1 / tc_pre_pass2), a lane leaves the loop when its list
//           is empty: as many trips as the fullest lane of the wavefront has candidates.
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

// The radius, from the CURRENT squared distances to the n remembered agents that are still in the game (their positions
// read NaN otherwise; none when fewer than 5 are):
//   n >= K + 2   1.15 x the THIRD LARGEST of them (round 6).  It is the (n - 2)-th smallest, n - 2 >= K, so K others are
//                inside it for sure; with all K + 3 in the game it lists the K + 1 nearest of the remembered and whoever came
//                closer since -- ~21 candidates instead of the ~35 the largest distance listed: agents move up to a
//                neighbourhood radius per tick, and ONE remembered agent that ran away used to set the radius
//                (experiments/offline/knn_cell_sort_sim.py; pass 2 takes as many trips as the fullest lane lists);
//   otherwise    1.15 x (K + 3) / n x the largest: after remembered agents were tagged out the radius grows by the share
//                that is missing.
// Fewer than K + 3 listed means fewer remembered for the next tick, which falls back to the second form: self-correcting.
// A heuristic on purpose (experiments/offline/knn_prefilter_sim2.py), checked by the caller.  Returns the bits of the
// radius, 0x7f800000 = none.
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
  float far = 0.0f, far2 = 0.0f, far3 = 0.0f;  // the three largest, descending
  unsigned n = 0u;
#define WD_TC_PREV_DIST(k)                                                                                  \
  if (k < M) {                                                                                              \
    const float dx = xi - p##k.x, dy = yi - p##k.y;                                                         \
    const float d2 = dx * dx + dy * dy;                                                                     \
    const float dz = fmaxf(d2, 0.0f); /* (maxnum: a NaN operand is ignored -- out of the game counts as 0) */ \
    far3 = __builtin_amdgcn_fmed3f(far2, far3, dz);                                                         \
    far2 = __builtin_amdgcn_fmed3f(far, far2, dz);                                                          \
    far = fmaxf(far, dz);                                                                                   \
    asm("v_cmp_o_f32 vcc, %1, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n) : "v"(d2) : "vcc");       \
  }
  WD_TC_PREV_DIST(0) WD_TC_PREV_DIST(1) WD_TC_PREV_DIST(2) WD_TC_PREV_DIST(3) WD_TC_PREV_DIST(4)
  WD_TC_PREV_DIST(5) WD_TC_PREV_DIST(6) WD_TC_PREV_DIST(7) WD_TC_PREV_DIST(8) WD_TC_PREV_DIST(9)
  WD_TC_PREV_DIST(10) WD_TC_PREV_DIST(11) WD_TC_PREV_DIST(12) WD_TC_PREV_DIST(13) WD_TC_PREV_DIST(14)
#undef WD_TC_PREV_DIST
  const float T = (n >= (unsigned)(K + 2)) ? 1.15f * far3 : far * (1.15f * (float)(K + 3)) * __builtin_amdgcn_rcpf((float)n);
  return (n >= 5u) ? __float_as_uint(T) : 0x7f800000u;
}

// ---- CELL-SORTED packing (round 6; replicas of more than 128 agents while the prefiltered search is on).  The agents in the
// game are packed by grid cell (C x C cells over the arena, a serpentine sweep; a counting sort through LDS atomics, tc_fast.h) instead
// of by id, so the 64 searchers of a wavefront are neighbours in space and pass 1 of the prefiltered search runs over the
// cell rows around them only: ~300 instead of 1005 candidates at 1005 agents (experiments/offline/knn_cell_sort_sim.py).
// Exactness costs nothing new: a candidate OUTSIDE the 3 x 3 cell block of a searcher is farther than the distance from the
// searcher to the block's border, so the radius handed to the prefilter is min(hint radius, that distance squared) -- such a
// candidate would have failed pass 1's compare anyway, the search is the search over ALL candidates with that radius, and
// the `held` check of the caller (the K-th found lies two key buckets inside the radius) covers it.  Ties are settled by
// agent id through the packed-index -> id table (tc_tie_order<true>), so the packing order is free; within a cell it is the
// order of the LDS atomics, which varies from run to run and changes nothing that leaves the kernel but the look-ahead
// ids remembered in `knn_prev` (a hint).
// Side of the grid for `n_live` agents in the game (0: pack by id, one range): cells of at least 2.2 x the expected distance
// to the K-th neighbour at uniform density (r_K = length x sqrt(K / (pi n))), at most 8 x 8 (one counter per lane).
__device__ __forceinline__ int tc_cell_grid(int n_live, int K) {
  const int c = (int)sqrtf((float)n_live * (3.14159265f / (2.2f * 2.2f)) / (float)K);
  return c >= 4 ? min(c, 8) : 0;
}
// (positions are clipped to [0, length]: the products are in [0, C])
__device__ __forceinline__ int2 tc_cell_xy(float x, float y, float inv, int C) {
  return make_int2(min(C - 1, (int)(x * inv)), min(C - 1, (int)(y * inv)));
}
// place of a cell in the packing order: a SERPENTINE sweep (even rows left to right, odd rows right to left), so that the
// cells of 64 consecutive packed agents stay neighbours where the sweep changes rows -- in row-major order a wavefront
// that wraps holds the end of one row and the start of the next, and its candidate runs are two whole cell rows
// (258 instead of 305 candidates per wavefront at 1005 agents, experiments/offline/knn_cell_sort_sim.py)
__device__ __forceinline__ int tc_cell_place(int2 c, int C) { return c.y * C + ((c.y & 1) ? C - 1 - c.x : c.x); }
// squared distance from (x, y) in cell (cx, cy) to the border of its 3 x 3 block, less a margin of 1/1024 cell (the cell
// index comes from a rounded product, the border from another: both are within a few ulps of the arena's length of the
// exact values, a thousand times less than the margin); the arena's own border does not count.  Bits; 0x7f800000 = none.
__device__ __forceinline__ unsigned tc_cell_cover2(float x, float y, int2 c, float len, int C) {
  float cover = __builtin_inff();
  if (c.x > 0) cover = fminf(cover, x - (float)(c.x - 1) * len);
  if (c.x < C - 1) cover = fminf(cover, (float)(c.x + 2) * len - x);
  if (c.y > 0) cover = fminf(cover, y - (float)(c.y - 1) * len);
  if (c.y < C - 1) cover = fminf(cover, (float)(c.y + 2) * len - y);
  cover = fmaxf(cover - len * (1.0f / 1024.0f), 0.0f);
  return __float_as_uint(cover * cover);
}

// The prefiltered search keeps, per searcher lane, a LIST in the wavefront's staging buffer (dead between the hint read and
// the row gather): the non-empty 32-candidate mask words of pass 1 with the index of the candidate on bit 0, rows of 64
// lanes.  Pass 2 then pops ACROSS words: a lane that has used up a word moves on to its next one in the same trip, so a
// round of pass 2 takes as many trips as the fullest lane has candidates in the whole list (~40 for the ~300 candidates of a wavefront) --
// word by word (rounds 4-5) it took the sum over the words of the fullest lane PER WORD: 80 - 120 trips per search, and more
// the closer the wavefront's searchers sit together (profiles/r06_phase_cells_*.txt).  WD_TC_LIST_CAP words per round.
#define WD_TC_LIST_CAP 11  // (a 1024-agent replica with K = 12 and sixteen such buffers still fits the 160 KB of a workgroup)
#define WD_TC_LIST_DWORDS ((WD_TC_LIST_CAP + 1) * (64 + 32))  // masks (u32) + bit-0 indices (u16), one spare row each
struct TcPreList {
  unsigned *masks;        // [CAP + 1][64], this lane's column
  unsigned short *tops;   // [CAP + 1][64], this lane's column
};
__device__ __forceinline__ TcPreList tc_pre_list(float *stage, int lane) {
  TcPreList pl;
  pl.masks = (unsigned *)stage + lane;
  pl.tops = (unsigned short *)((unsigned *)stage + (WD_TC_LIST_CAP + 1) * 64) + lane;
  return pl;
}

// ---- pass 1 over the candidates [j0, j1) (j0 a multiple of 4; j1 a multiple of 4 or the number of candidates; at most
// CAP - (words listed since the last round) words): squared distance and ONE compare against the radius per candidate,
// shifted into the word's mask (candidate b of a word of nb ends on bit nb - 1 - b); every word is written to slot `cnt` of
// the lane's list, and only a non-empty one keeps the slot.
__device__ __forceinline__ void tc_pre_pass1(const float2 *cxy, float xi, float yi, int j0, int j1, float Tf, const TcPreList &pl,
                                             int &cnt) {
  // The compare is a subtraction: the sign of Tf - d2 is set iff d2 > Tf (d2 == Tf gives +0, a pad position -inf), and
  // v_alignbit shifts it into the word -- two full-rate instructions where v_cmp + v_addc (carry in and out) take ~3 cycles
  // each on gfx950 (experiments/README.md).  The word collects the "outside" bits and is inverted at the end.
#define WD_TC_MASK_PUSH(m, d2v) (m) = __builtin_amdgcn_alignbit((m), __float_as_uint(Tf - (d2v)), 31)
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
  TcP4 ga = tc_load4(cxy, j0), gb;
  for (int jw = j0; jw < j1; jw += 32) {  // wave-uniform
    const int nb = min(32, j1 - jw);
    unsigned mo = 0xffffffffu;  // bit set: outside the radius (or no candidate)
    int b = 0;
    for (; b + 8 <= nb; b += 8) {  // two groups of four per trip, ping-pong
      gb = tc_load4(cxy, jw + b + 4);
      WD_TC_MASK_PUSH4(mo, ga);
      ga = tc_load4(cxy, jw + b + 8);  // (at most 8 entries past the last candidate: padding; else the next word's first group)
      WD_TC_MASK_PUSH4(mo, gb);
    }
    if (b + 4 <= nb) {
      gb = tc_load4(cxy, jw + b + 4);
      WD_TC_MASK_PUSH4(mo, ga);
      ga = gb;
      b += 4;
    }
    for (; b < nb; ++b) {  // (only the last word of the candidates can have a remainder)
      const float2 pj = cxy[jw + b];
      const float dx = xi - pj.x, dy = yi - pj.y;
      const float d2 = dx * dx + dy * dy;
      WD_TC_MASK_PUSH(mo, d2);
    }
    const unsigned mu = ~mo;  // (bits nb and up were shifted in from the all-ones start: clear)
    pl.masks[cnt * 64] = mu;
    pl.tops[cnt * 64] = (unsigned short)(jw + nb - 1);
    cnt += (mu != 0u) ? 1 : 0;
  }
#undef WD_TC_MASK_PUSH4
#undef WD_TC_MASK_PUSH
}

// ---- pass 2 over the lists: the listed candidates go through the chain.  S = the L smallest keys so far, ascending;
// `extra` = the (L+1)-th (one more id to remember); both start at 0xffffffff.  A plain divergent loop: a lane leaves it when
// its list is empty (the wavefront makes as many trips as its fullest lane has candidates), and moving on to the lane's next
// word is a branch, not a run of selects -- the search is VALU-bound at four wavefronts per SIMD, the scalar unit and the
// other wavefronts cover the branches and the LDS round trips (round 6: ~30 instead of ~40 vector instructions per trip
// against the branch-free, software-pipelined form with a pad candidate for lanes that ran out).
template <int L, int IDB>
__device__ __forceinline__ void tc_pre_pass2(const float2 *cxy, float xi, float yi, const TcPreList &pl, int cnt,
                                             unsigned (&S)[L], unsigned &extra, int &probe_trips) {
  constexpr unsigned IDM = (1u << IDB) - 1u;
  unsigned cur = (cnt > 0) ? pl.masks[0] : 0u;
  int top = pl.tops[0], pos1 = 1;
#ifdef WD_TC_PROBES
  int my_trips = 0;
#endif
  while (cur != 0u) {
#ifdef WD_TC_PROBES
    ++my_trips;
#endif
    const unsigned idx = (unsigned)(top - (__ffs(cur) - 1));
    cur &= cur - 1u;
    const float2 pj = cxy[idx];
    if (cur == 0u && pos1 < cnt) {  // on to the lane's next word
      cur = pl.masks[pos1 * 64];
      top = pl.tops[pos1 * 64];
      ++pos1;
    }
    const float dx = xi - pj.x, dy = yi - pj.y;
    const float d2 = dx * dx + dy * dy;
    const unsigned key_ = (__float_as_uint(d2) & ~IDM) | idx;
    extra = tc_umed3(S[L - 1], extra, key_);
#pragma unroll
    for (int k = L - 1; k >= 1; --k) S[k] = tc_umed3(S[k - 1], S[k], key_);
    S[0] = min(S[0], key_);
  }
#ifdef WD_TC_PROBES
  for (int o = 32; o > 0; o >>= 1) my_trips = max(my_trips, __shfl_xor(my_trips, o, 64));  // the wavefront's trips = its fullest lane's
  probe_trips += my_trips;
#endif
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
template <bool TABLE = false>
__device__ __forceinline__ int tc_zone_resolve(const float2 *cxy, int n_cand, float sx, float sy, int self, unsigned zone_hi,
                                               int idb, int K, unsigned char *scratch, int lane, const short *tie = nullptr) {
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
    // (float32 distance, tie-break order, index): the index rides along below the order and never decides
    mine = ((unsigned long long)__float_as_uint(sqrtf(dx * dx + dy * dy)) << 32) |
           (TABLE ? ((unsigned)tc_tie_order<TABLE>(tie, j) << 16) : 0u) | (unsigned)j;
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
// S: the L smallest keys in ascending order (tc_chain_range / tc_pre_pass2); o: the same without the
// agent's own entry (the caller remembers their ids for the next tick's bound)
// L = KMAX + 3: two look-ahead entries behind the K-th other agent (any K <= KMAX).  L = KMAX + 2 (round 6; K == KMAX only:
// the shape-specialised and the exact-K entries of replicas up to 128 agents): ONE look-ahead entry, one v_med3_u32 less per
// candidate and searcher.  The (K+1)-th other agent is then the last thing the chain knows: whenever IT lies inside the
// uncertain buckets (~3e-4 per agent), or forms a close pair with the K-th (~5e-4), what lies behind it is unknown and the
// function returns false -- the caller has the whole wavefront resolve the zone for that lane (tc_zone_resolve: ~150
// instructions, where the two-pass repeat of round 2 cost a whole search and the launch waited for it).
template <int KMAX, int IDB, int L, bool TABLE = false>
__device__ __forceinline__ bool tc_resolve_keys(const float2 *cxy, int ag, int K, const unsigned (&S)[L],
                                                unsigned (&o)[L - 1], int (&nid)[KMAX + 1], int (&rank)[KMAX + 1],
                                                bool &in_order, const short *tie = nullptr) {
  static_assert(L == KMAX + 3 || L == KMAX + 2, "self + K others + two look-ahead entries (one: K == KMAX only)");
  constexpr bool ONE_LOOK = (L == KMAX + 2);
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
    for (int k = 0; k <= (ONE_LOOK ? KMAX - 1 : KMAX); ++k)
      if (k <= K) cm |= ((o[k + 1] - o[k] < THR) ? 1u : 0u) << k;
    unsigned rel = cm & ((1u << K) - 1u);
    // (one look-ahead entry: what follows the (K+1)-th is unknown -- as if it were close)
    const bool look_close = ONE_LOOK ? true : ((cm >> K) & 1u) != 0u;
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
        // (equal float32 distances: the lower agent id goes first)
        if (sb < sa || (sb == sa && tc_tie_order<TABLE>(tie, ib) < tc_tie_order<TABLE>(tie, ia))) {
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
    unsigned oK = o[KMAX - 1], oExtra = o[KMAX], oLook = o[ONE_LOOK ? KMAX : KMAX + 1];
#pragma unroll
    for (int k = 0; k < KMAX - 1; ++k) {
      oK = (k == K - 1) ? o[k] : oK;
      oExtra = (k == K - 1) ? o[k + 1] : oExtra;
      if constexpr (!ONE_LOOK) oLook = (k == K - 1) ? o[k + 2] : oLook;
    }
    const unsigned INVALID = 0x7f800000u;  // agents out of the game sit at +inf; unused slots are above
    const unsigned cut = (oK >> IDB) + 2u;   // first bucket that is certainly outside
    // (one look-ahead entry: oLook IS the (K+1)-th -- certain only if that one is already outside)
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
      key64[k] = ((unsigned long long)sb << 32) | (unsigned)(nid[k] < 0 ? nid[k] : tc_tie_order<TABLE>(tie, nid[k]));
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
    // (one look-ahead entry: that case returned `exact == false` above and is the caller's)
    if (!ONE_LOOK && oK < INVALID && oExtra < INVALID && (oExtra >> IDB) < cut) {
      const int idE = (int)(oExtra & IDM);
      const float2 pe = cxy[idE];
      const float dx = xi - pe.x, dy = yi - pe.y;
      const unsigned long long keyE = ((unsigned long long)__float_as_uint(sqrtf(dx * dx + dy * dy)) << 32) |
                                      (unsigned)tc_tie_order<TABLE>(tie, idE);
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

}  // namespace
