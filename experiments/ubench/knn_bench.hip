// Neighbour-search A/B outside the tick kernel: the product's lane-per-agent search
// (tc_knn_registers) against a TRANSPOSED one (lane = candidate, loop over agents, threshold search
// on the scalar unit; defined below, not in the product) on one replica of 105 agents per 128-thread
// block, 8 blocks per CU (the tick kernel's occupancy), random positions.  Prints shader cycles per
// search per wavefront and checks that both produce the same neighbour lists.
//
// Result on MI355X (round 2): identical lists; lane-per-agent 32.3k cycles per search for the
// 64-agent wavefront, transposed 93-105k.  The transposed loop needs only ~33 vector instructions per
// agent, but ~170 scalar ones (probe bookkeeping, structurised control flow), and a wavefront issues
// one instruction of ANY kind per ~5 cycles: ~200 instructions x 64 agents against ~3.2k for the
// lane-per-agent search.  Not adopted.
#include "../../warp_drive_amd/csrc/kernels/wd_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int KM = 10;
constexpr int REPS = 16;

namespace {
// ---- neighbour search, transposed: lane = CANDIDATE, loop over the wavefront's own agents.
// For blocks that hold one replica of at most 128 agents.  A wavefront keeps the replica's positions
// in four registers (candidates `lane` and `64 + lane`); for one agent at a time (its position comes
// out of those registers with v_readlane, so the loop reads no LDS) every lane forms the squared
// distance to "its" two candidates, and a threshold t with exactly K candidates at or below it is
// searched on the SCALAR unit: count(t) = two ballots + two s_bcnt1.  Per agent that is ~10 vector
// instructions plus two compares per probe, against ~17 vector instructions per CANDIDATE for the
// lane-per-agent search (tc_knn_registers).  The search starts from the threshold that worked on the
// previous tick (`hint`, kept in HBM; only the number of probes depends on it -- ~3 on average, see
// experiments/offline/knn_threshold_search.py) and moves by a secant step on the float bits
// (count grows like t: one candidate per 1/(K + 1/2) of t), then by bisection.
//
// Exactness: the reference takes the K smallest (float32 sqrt distance, id) keys.  {d2 <= t} with
// count K is that set unless two candidates on either side of the cut share a float32 distance; the
// search therefore also requires count(t * (1 + 2^-21)) == K: then the first candidate outside is
// more than 2^-21 (relative) above the last one inside, and their square roots differ by more than
// one ulp.  Agents for which no such t exists (ties at the cut) are returned in the `slow` mask and
// go through tc_knn_registers.  The selected (d2 bits, id) pairs of agent `il` land in
// sbuf[il * K ..], ascending id.
template <int KMAX>
__device__ __forceinline__ unsigned long long tc_knn_transposed(const float2 *cxy, int N, int K, int wave, int lane,
                                                                unsigned long long need, float grid_length,
                                                                int hint_bits, uint2 *sbuf, int &n_found) {
  const int j1 = 64 + lane;
  const float2 p0 = cxy[min(lane, N - 1)], p1 = cxy[min(j1, N - 1)];
  const float vx0 = (lane < N) ? p0.x : WD_BIG, vy0 = p0.y;
  const float vx1 = (j1 < N) ? p1.x : WD_BIG, vy1 = p1.y;
  // candidates in the game (agents out of it sit at x = WD_BIG), minus the agent itself
  const int others = __popcll(__ballot(vx0 < 1.0e29f)) + __popcll(__ballot(vx1 < 1.0e29f)) - 1;
  const bool take_all = others <= K;
  n_found = max(0, min(K, others));
  wave = __builtin_amdgcn_readfirstlane(wave);
  const float own_x = wave ? vx1 : vx0, own_y = wave ? vy1 : vy0;
  const int LO0 = __float_as_int(1.0e-8f);
  const int HI0 = __builtin_amdgcn_readfirstlane(__float_as_int(4.0f * grid_length * grid_length));
  // threshold for a uniform density: K + 1/2 candidates inside the disc
  const int guess0 = __builtin_amdgcn_readfirstlane(__float_as_int(
      grid_length * grid_length * ((float)K + 0.5f) / (3.14159265f * (float)max(others, 1))));
  const int guess = max(LO0 + 1, min(HI0 - 1, guess0));
  const int STEP = (int)(8388608.0f / (0.69314718f * ((float)K + 0.5f)));
  unsigned long long todo = need, slow = 0ull;
  while (todo) {  // wave-uniform
    const int il = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    const float xi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own_x), il));
    const float yi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own_y), il));
    const float dx0 = xi - vx0, dy0 = yi - vy0, dx1 = xi - vx1, dy1 = yi - vy1;
    float d0 = dx0 * dx0 + dy0 * dy0, d1 = dx1 * dx1 + dy1 * dy1;
    // the agent is not its own neighbour: its lane gets a NaN (all ones), which no compare accepts
    if (wave == 0) asm("v_writelane_b32 %0, -1, %1" : "+v"(d0) : "s"(il));
    else asm("v_writelane_b32 %0, -1, %1" : "+v"(d1) : "s"(il));
    int t = 0x7f7fffff;  // FLT_MAX: every candidate in the game
    bool ok = true;
    if (!take_all) {
      int lo = LO0, hi = HI0;  // count(lo) < K < count(hi)
      t = __builtin_amdgcn_readlane(hint_bits, il);
      if ((unsigned)(t - LO0 - 1) >= (unsigned)(HI0 - LO0 - 1)) t = guess;
      ok = false;
      bool open = true;  // false: the bracket closed without a threshold (a tie at the cut)
      // three secant steps at most ...
      for (int it = 0; it < 3; ++it) {
        const float tf = __int_as_float(t);
        const int c = __popcll(__ballot(d0 <= tf)) + __popcll(__ballot(d1 <= tf));
        if (c == K) { ok = true; break; }
        if (c < K) lo = t; else hi = t;
        if (hi - lo <= 1) { open = false; break; }  // (also when K or more twins sit below LO0)
        const int cand = t - (c - K) * STEP + ((c < K) ? (STEP >> 1) : -(STEP >> 1));
        t = ((unsigned)(cand - lo - 1) < (unsigned)(hi - lo - 1)) ? cand : (int)(((unsigned)lo + (unsigned)hi) >> 1);
      }
      // ... then bisection on the float bits (ends: the bracket shrinks every trip)
      while (!ok && open) {
        const float tf = __int_as_float(t);
        const int c = __popcll(__ballot(d0 <= tf)) + __popcll(__ballot(d1 <= tf));
        if (c == K) { ok = true; break; }
        if (c < K) lo = t; else hi = t;
        if (hi - lo <= 1) break;
        t = (int)(((unsigned)lo + (unsigned)hi) >> 1);
      }
      if (ok) {  // nothing within 8 ulps above the threshold (see "Exactness")
        const float tb = __int_as_float(t + 8);
        ok = (__popcll(__ballot(d0 <= tb)) + __popcll(__ballot(d1 <= tb))) == K;
      }
    }
    if (ok) {
      const float tf = __int_as_float(t);
      const bool in0 = d0 <= tf, in1 = d1 <= tf;
      const unsigned long long m0 = __ballot(in0), m1 = __ballot(in1);
      const int pos0 = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
      const int pos1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)m1, (unsigned)__popcll(m0)));
      uint2 *const row = sbuf + il * K;
      if (in0) row[pos0] = make_uint2(__float_as_uint(d0), (unsigned)lane);
      if (in1) row[pos1] = make_uint2(__float_as_uint(d1), (unsigned)j1);
    } else {
      slow |= 1ull << il;
    }
  }
  return slow;
}

#define WD_TC_HINT_BUMP 300000  // float bits (+3.6 %)
// lane-per-agent tail of the transposed search: the agent's selected pairs -> ids and ranks in the
// reference's order (float32 distance, then id)
template <int KMAX>
__device__ __forceinline__ void tc_knn_collect(const uint2 *sbuf, int lane, int K, int n_found, int (&nid)[KMAX],
                                               int (&rank)[KMAX], int &hint_bits) {
  const uint2 *const row = sbuf + lane * K;
  uint2 e[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) e[k] = row[min(k, K - 1)];
  unsigned sb[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool valid = k < n_found;  // wave-uniform
    nid[k] = valid ? (int)e[k].y : -1;
    sb[k] = valid ? __float_as_uint(sqrtf(__uint_as_float(e[k].x))) : 0x7f800000u;
  }
  // next tick's search starts a little above this tick's K-th squared distance
  unsigned far = 0u;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) far = max(far, (k < n_found) ? e[k].x : 0u);
  hint_bits = (int)far + WD_TC_HINT_BUMP;
  tc_rank_entries<KMAX>(sb, rank);
}

}  // namespace

// MODE 0: lane-per-agent; MODE 1: transposed
template <int MODE>
__global__ void __launch_bounds__(128) knn_bench(const float2 *pos, float *hints, int *out_ids, unsigned long long *cycles,
                                                  unsigned *probes_slow, int N, int K, float L, float jitter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NP = ((N + 3) & ~3) + 8;
  float2 *xy = (float2 *)smem;
  uint2 *sbuf_all = (uint2 *)(smem + 8 * NP + 64);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int env = blockIdx.x;
  const bool active = tid < N;
  int nid[KM], rank[KM];
  unsigned long long total = 0;
  unsigned slow_cnt = 0;
  int hint_bits = active ? __float_as_int(hints[env * N + tid]) : 0;
  for (int r = 0; r < REPS; ++r) {
    // positions drift a little every repetition (the hint is one "tick" old)
    __syncthreads();
    if (active) {
      float2 p = pos[env * N + tid];
      p.x += jitter * (float)r * (float)((tid * 7 + env) % 5 - 2);
      p.y += jitter * (float)r * (float)((tid * 3 + env) % 5 - 2);
      xy[tid] = p;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KM; ++k) { nid[k] = -1; rank[k] = k; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
      if (active) tc_knn_registers<KM>(xy, tid, N, K, nid, rank);
    } else if (MODE == 2) {
      bool exact = true;
      int nid1[KM + 1], rank1[KM + 1];
#pragma unroll
      for (int k = 0; k <= KM; ++k) { nid1[k] = -1; rank1[k] = k; }
      bool in_order = true;
      if (active) exact = tc_knn_packed<KM>(xy, tid, N, K, nid1, rank1, in_order);
      if (!exact) {
        tc_knn_registers<KM>(xy, tid, N, K, nid, rank);
      } else {
        // entries of rank < K -> slot `rank`
#pragma unroll
        for (int k = 0; k <= KM; ++k)
          if (rank1[k] < K) {
#pragma unroll
            for (int q = 0; q < KM; ++q)
              if (q == rank1[k]) { nid[q] = nid1[k]; rank[q] = q; }
          }
      }
      slow_cnt += __popcll(__ballot(!exact));
    } else {
      const unsigned long long need = __ballot(active);
      uint2 *sbuf = sbuf_all + wave * 64 * K;
      int n_found;
      const unsigned long long slow = tc_knn_transposed<KM>(xy, N, K, wave, lane, need, L, hint_bits, sbuf, n_found);
      const bool mine_slow = (slow >> lane) & 1ull;
      if (active && !mine_slow) tc_knn_collect<KM>(sbuf, lane, K, n_found, nid, rank, hint_bits);
      if (active && mine_slow) tc_knn_registers<KM>(xy, tid, N, K, nid, rank);
      slow_cnt += __popcll(slow);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    if (active)
#pragma unroll
      for (int k = 0; k < KM; ++k)
        if (rank[k] < K) out_ids[((size_t)r * gridDim.x * N + (size_t)env * N + tid) * K + rank[k]] = nid[k];
  }
  if (lane == 0) {
    cycles[blockIdx.x * 2 + wave] = total;
    probes_slow[blockIdx.x * 2 + wave] = slow_cnt;
  }
  if (active) hints[env * N + tid] = __int_as_float(hint_bits);
}

int main() {
  const int N = 105, K = 10, E = 2048;
  const float L = 20.0f;
  std::vector<float2> h(E * N);
  srand(1234);
  for (auto &p : h) { p.x = L * (rand() / (float)RAND_MAX); p.y = L * (rand() / (float)RAND_MAX); }
  // a few out-of-game agents and twins
  for (int e = 0; e < E; e += 7) { h[e * N + 17].x = 1.0e30f; h[e * N + 80] = h[e * N + 3]; }
  float2 *dpos; float *dhint; int *dout0, *dout1; unsigned long long *dcyc; unsigned *dslow;
  const size_t out_elems = (size_t)REPS * E * N * K;
  CHECK(hipMalloc(&dpos, sizeof(float2) * h.size()));
  CHECK(hipMalloc(&dhint, 4 * E * N));
  CHECK(hipMalloc(&dout0, 4 * out_elems));
  CHECK(hipMalloc(&dout1, 4 * out_elems));
  CHECK(hipMalloc(&dcyc, 8 * E * 2));
  CHECK(hipMalloc(&dslow, 4 * E * 2));
  CHECK(hipMemcpy(dpos, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(dhint, 0, 4 * E * N));
  CHECK(hipMemset(dout0, 0xff, 4 * out_elems));
  CHECK(hipMemset(dout1, 0xff, 4 * out_elems));
  const int lds = 19 * 1024;
  auto report = [&](const char *name) {
    std::vector<unsigned long long> c(E * 2);
    std::vector<unsigned> s(E * 2);
    CHECK(hipMemcpy(c.data(), dcyc, 8 * E * 2, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(s.data(), dslow, 4 * E * 2, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> w0, w1;
    unsigned long long slow = 0;
    for (int e = 0; e < E; ++e) { w0.push_back(c[2 * e]); w1.push_back(c[2 * e + 1]); slow += s[2 * e] + s[2 * e + 1]; }
    std::sort(w0.begin(), w0.end()); std::sort(w1.begin(), w1.end());
    printf("%-34s wave0 (64 agents) %8.0f cycles/search   wave1 (41 agents) %8.0f   slow-path agents %llu\n", name,
           (double)w0[E / 2] / REPS, (double)w1[E / 2] / REPS, slow);
  };
  for (float jitter : {0.0f, 0.05f}) {
    printf("--- jitter %.2f per repetition\n", jitter);
    for (int pass = 0; pass < 2; ++pass) {
      hipLaunchKernelGGL(knn_bench<0>, dim3(E), dim3(128), lds, 0, dpos, dhint, dout0, dcyc, dslow, N, K, L, jitter);
      CHECK(hipDeviceSynchronize());
    }
    report("lane per agent (registers)");
    CHECK(hipMemset(dhint, 0, 4 * E * N));
    hipLaunchKernelGGL(knn_bench<1>, dim3(E), dim3(128), lds, 0, dpos, dhint, dout1, dcyc, dslow, N, K, L, jitter);
    CHECK(hipDeviceSynchronize());
    report("transposed, no hints");
    {
      std::vector<int> a(out_elems), b(out_elems);
      CHECK(hipMemset(dout1, 0xff, 4 * out_elems));
      hipLaunchKernelGGL(knn_bench<2>, dim3(E), dim3(128), lds, 0, dpos, dhint, dout1, dcyc, dslow, N, K, L, jitter);
      hipLaunchKernelGGL(knn_bench<2>, dim3(E), dim3(128), lds, 0, dpos, dhint, dout1, dcyc, dslow, N, K, L, jitter);
      CHECK(hipDeviceSynchronize());
      report("one pass, packed keys");
      CHECK(hipMemcpy(a.data(), dout0, 4 * out_elems, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(b.data(), dout1, 4 * out_elems, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < out_elems; ++i) bad += a[i] != b[i];
      printf("packed keys: %zu of %zu entries differ\n", bad, out_elems);
    }
    hipLaunchKernelGGL(knn_bench<1>, dim3(E), dim3(128), lds, 0, dpos, dhint, dout1, dcyc, dslow, N, K, L, jitter);
    CHECK(hipDeviceSynchronize());
    report("transposed, hints of the last run");
    std::vector<int> a(out_elems), b(out_elems);
    CHECK(hipMemcpy(a.data(), dout0, 4 * out_elems, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), dout1, 4 * out_elems, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < out_elems; ++i) bad += a[i] != b[i];
    printf("neighbour lists: %zu of %zu entries differ\n", bad, out_elems);
  }
  return 0;
}
