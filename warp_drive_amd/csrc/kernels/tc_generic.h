// tc_generic.h -- tc_generic_impl: any N <= 1024, any K, full observations.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_fetch.h"
#include "tc_sample.h"
#include "tc_move.h"
#include "tc_tags.h"
#include "tc_reset.h"

namespace {

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
