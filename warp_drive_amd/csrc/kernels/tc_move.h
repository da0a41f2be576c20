// tc_move.h -- move: float32 kinematics exactly as numpy evaluates update_state; the observation values of a (row, neighbour) pair.
// Part of the TagContinuous translation unit (tag_continuous.hip, which holds the design notes, the probe macros
// and the kernel entries); split by phase in round 6 with every shipped code object byte-identical before / after.
#pragma once
#include "wd_common.h"
#include "tc_types.h"

namespace {

// ---- move: float32 kinematics exactly as numpy evaluates update_state (:339-401); stores the new
// state, returns the post-move position, the edge penalty and the observation features.
// numpy's float32 remainder by 2 pi (npy_divmodf: fmodf, then the divisor's sign) for the only dividends a heading ever
// produces -- an angle of [0, 2 pi) plus one turn: |a| < 4 pi.  fmodf is EXACT by definition, and for |a| < 2 b it is a or
// a -+ b, the subtraction exact (Sterbenz: b <= |a| < 2 b): five branch-free instructions instead of the library's long
// division (~40 executed instructions and four exec-mask branches per agent and tick).  Anything else -- a heading somebody
// wrote into the state array, an action table with turns beyond 2 pi, NaN -- takes the library path in a cold block.
__device__ __forceinline__ float tc_remainder_two_pi(float a, float b) {
  const float aa = fabsf(a);
  if (__builtin_expect(__ballot(!(aa < 2.0f * b)) != 0ull, 0)) return wd_np_remainderf(a, b);  // (wave-uniform, never in practice)
  float m = aa - ((aa >= b) ? b : 0.0f);   // |fmodf(a, b)|, exact
  m = (a < 0.0f && m != 0.0f) ? b - m : m; // fmodf carries the dividend's sign; a negative remainder gets + b (one rounding,
                                           // the same one: -m + b); zero stays +0 (copysignf(0, b), b > 0)
  return m;
}

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
  const float dir = tc_remainder_two_pi(in.dir + d_turn, two_pi) * s;          // :355-357
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

}  // namespace
