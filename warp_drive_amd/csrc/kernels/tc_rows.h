// tc_rows.h -- observation rows / neighbour-id rows: LDS staging, write-through flush, LDS carve-up of the fast path, sparse row gather.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_move.h"

namespace {

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
  // replicas of more than 128 agents: the buffer also holds the prefiltered search's hints (8 dwords per lane) and, after
  // them, its candidate lists (tc_knn.h WD_TC_LIST_DWORDS = 1152; the host adds the same, envs/tag_continuous.py lds_bytes)
  if (compact && N > 128) l.stage_dwords = max(l.stage_dwords, 1152);
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

}  // namespace
