// cartpole.hip -- ClassicControl CartPole Euler step (BASELINE config 5: the
// HBM-roofline ceiling microbenchmark, 68 algorithmic bytes per env-step).
//
// Follows the reference's only device implementation,
// example_envs/single_agent/classic_control/cartpole/cartpole_step_numba.py:5-83
// (its CPU step is third-party gym.CartPoleEnv, absent here).  Numba's type
// inference is restated literally: float32 state and scalars, but the Python
// literal 4.0/3.0 makes the pole-acceleration denominator -- and everything
// downstream of thetaacc -- float64 (:56-60, :63-66).
//
// MI355X mapping: one THREAD per replica (the reference uses one 1-thread block per
// replica, cartpole.py:139-141), 16-byte state/obs accesses, grid-stride, and an
// optional `ticks` loop so many ticks fuse into one launch (a single tick at
// E = 100 000 moves 6.8 MB -- under 1 us of HBM time, i.e. launch-bound).
#include "wd_common.h"

namespace {

struct CpPhysics {
  float gravity, masspole, total_mass, length, polemass_length, force_mag, tau;
  float theta_threshold_radians, x_threshold;
  float inv_total_mass;  // RN(1 / total_mass) (correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt)
};

// x / c for a divisor that is the same on every tick of every replica (the total mass): q = RN(x * y), r = x - c * q
// (exact: one FMA), RN(q + r * y), y = RN(1 / c) -- three instructions instead of the ~10 of the correctly rounded division.
// Equal to RN(x / c) for ALL normal x away from underflow / overflow iff it is for the 2^24 values of one binade and both signs
// (scaling x by a power of two scales q, r and the result exactly): HipCartPoleVerifyInvariantDivide checks exactly that for
// the env's own c, exhaustively, once per env object (no float32 divisor tried so far fails it; the proof is per divisor all the
// same: the classical guarantee has exceptions, and an exhaustive check costs one launch); the launch
// uses this form only with that proof in hand, and only for dividends inside [2^-90, 2^90] (cp_euler<true>).
__device__ __forceinline__ float cp_div_invariant(float x, float c, float y) {
  const float q = x * y;
  const float r = __builtin_fmaf(-c, q, x);
  return __builtin_fmaf(r, y, q);
}

// numpy's float32 sin / cos kernel (wd_np_sincosf) for |x| <= 0.75: the quadrant is 0 -- RN(x * 2/pi + magic) - magic = 0
// exactly (|x * 2/pi| < 0.478) and the three reduction FMAs return x itself -- so the reduction and the quadrant selects drop
// out: the same two polynomials on r = x, bit for bit (checked against wd_np_sincosf for every float32 in [-0.75, 0.75] by
// tests/test_gpu_core.py).  A pole beyond 12 degrees has fallen: the angle of a replica that is still running is below 0.21.
__device__ __forceinline__ void cp_sincos_small(float x, float &sin_out, float &cos_out) {
  const float r2 = x * x;
  float c = __builtin_fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
  c = __builtin_fmaf(c, r2, 0x1.55553cp-05f);
  c = __builtin_fmaf(c, r2, -0x1.000000p-01f);
  c = __builtin_fmaf(c, r2, 0x1.000000p+00f);
  float sn = __builtin_fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
  sn = __builtin_fmaf(sn, r2, 0x1.11119ap-07f);
  sn = __builtin_fmaf(sn, r2, -0x1.555556p-03f);
  sn = __builtin_fmaf(sn, r2, 0.0f);
  sn = __builtin_fmaf(sn, x, x);
  sin_out = sn;
  cos_out = c;
}

// one Euler update, cartpole_step_numba.py:42-78 (float32 state; everything downstream of the
// Python literal 4.0/3.0 in float64, as Numba types it).
// FAST (compile time; the middle ticks of a recorded launch whose PRECONDITIONS the kernel checked once, cp_tick_impl):
// |theta| <= 0.75 on entry -> the quadrant-free sin / cos; the two float32 divisions by the total mass in three
// instructions each.  Bit-identical to the general form: the one data-dependent condition left -- a dividend outside the
// range the division shortcut was proved for (the force term's dividend; the other one is the pole's mass times cos^2) --
// redoes both divisions in a cold block behind a branch that is never taken in practice (the dividends are ~10 and ~0.1).
template <bool FAST = false>
__device__ __forceinline__ bool cp_euler(float4 &s, int action, const CpPhysics &p) {
  float x = s.x, x_dot = s.y, theta = s.z, theta_dot = s.w;
  const float force = (action > 0) ? p.force_mag : -p.force_mag;  // action > 0.5
  float sintheta, costheta;
  if (FAST) cp_sincos_small(theta, sintheta, costheta);
  else wd_np_sincosf(theta, sintheta, costheta);
  const float temp_num = force + p.polemass_length * (theta_dot * theta_dot) * sintheta;
  const float frac_num = p.masspole * (costheta * costheta);
  float temp, frac;
  if (FAST) {
    temp = cp_div_invariant(temp_num, p.total_mass, p.inv_total_mass);
    frac = cp_div_invariant(frac_num, p.total_mass, p.inv_total_mass);
    // |temp_num| inside [2^-90, 2^90) by its exponent field (zero, subnormal, infinite, NaN: outside); frac_num is the pole's
    // mass (inside [2^-60, 2^60]: a precondition of the launch) times cos^2 >= 0.53: always inside
    const bool in_range = ((__float_as_uint(temp_num) & 0x7fffffffu) - 0x12800000u) < 0x5a000000u;
    if (__builtin_expect(__ballot(!in_range) != 0ull, 0)) {
      temp = temp_num / p.total_mass;
      frac = frac_num / p.total_mass;
    }
  } else {
    temp = temp_num / p.total_mass;
    frac = frac_num / p.total_mass;
  }
  const double den = (double)p.length * (4.0 / 3.0 - (double)frac);
  const double thetaacc = (double)(p.gravity * sintheta - costheta * temp) / den;
  const double xacc = (double)temp - (double)p.polemass_length * thetaacc * (double)costheta / (double)p.total_mass;
  x = x + p.tau * x_dot;
  x_dot = (float)((double)x_dot + (double)p.tau * xacc);
  theta = theta + p.tau * theta_dot;
  theta_dot = (float)((double)theta_dot + (double)p.tau * thetaacc);
  s = make_float4(x, x_dot, theta, theta_dot);
  return x < -p.x_threshold || x > p.x_threshold || theta < -p.theta_threshold_radians ||
         theta > p.theta_threshold_radians;
}

// ---- a small policy INSIDE the rollout kernel: two hidden layers of H units + one softmax head, the
// weights in LDS (every lane reads the same address: broadcast), the activations in registers.
// Packed weights (training/policy_kernel.py::pack_rollout_policy): W0 [H][4], b0 [H], W1 [H][H], b1 [H],
// Wp [A][H], bp [A], all float32.  Arithmetic: acc = bias, then fmaf over the inputs in index order;
// softmax with the maximum subtracted, expf, one division per action; restated in
// oracle/cartpole_np.py::policy_probabilities.  Returns the running float32 sums of the probabilities
// (what the inverse-CDF sampler compares the uniform with, random.cu:51-85).
constexpr int CP_MAX_REG_ACTIONS = 8;

template <int H>
__device__ __forceinline__ void cp_policy_cum(const float *w, const float4 &s, int n_actions,
                                              float (&cumv)[CP_MAX_REG_ACTIONS]) {
  const float *W0 = w, *b0 = W0 + 4 * H, *W1 = b0 + H, *b1 = W1 + H * H, *Wp = b1 + H, *bp = Wp + n_actions * H;
  float h1[H], h2[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float4 wr = *(const float4 *)(W0 + 4 * i);
    float acc = b0[i];
    acc = fmaf(wr.x, s.x, acc); acc = fmaf(wr.y, s.y, acc); acc = fmaf(wr.z, s.z, acc); acc = fmaf(wr.w, s.w, acc);
    h1[i] = fmaxf(acc, 0.0f);
  }
#pragma unroll
  for (int i = 0; i < H; ++i) {
    float acc = b1[i];
#pragma unroll
    for (int j = 0; j < H; j += 4) {
      const float4 wr = *(const float4 *)(W1 + i * H + j);
      acc = fmaf(wr.x, h1[j], acc); acc = fmaf(wr.y, h1[j + 1], acc);
      acc = fmaf(wr.z, h1[j + 2], acc); acc = fmaf(wr.w, h1[j + 3], acc);
    }
    h2[i] = fmaxf(acc, 0.0f);
  }
  float logit[CP_MAX_REG_ACTIONS], m = -__builtin_inff();
#pragma unroll
  for (int a = 0; a < CP_MAX_REG_ACTIONS; ++a) {
    logit[a] = -__builtin_inff();
    if (a < n_actions) {
      float acc = bp[a];
#pragma unroll
      for (int j = 0; j < H; j += 4) {
        const float4 wr = *(const float4 *)(Wp + a * H + j);
        acc = fmaf(wr.x, h2[j], acc); acc = fmaf(wr.y, h2[j + 1], acc);
        acc = fmaf(wr.z, h2[j + 2], acc); acc = fmaf(wr.w, h2[j + 3], acc);
      }
      logit[a] = acc;
      m = fmaxf(m, acc);
    }
  }
  float e[CP_MAX_REG_ACTIONS], sum = 0.0f;
#pragma unroll
  for (int a = 0; a < CP_MAX_REG_ACTIONS; ++a) {
    e[a] = (a < n_actions) ? expf(logit[a] - m) : 0.0f;
    sum += e[a];
  }
  float cum = 0.0f;
#pragma unroll
  for (int a = 0; a < CP_MAX_REG_ACTIONS; ++a) {
    const float p = e[a] / sum;
    if (a < n_actions) cum = (a == 0) ? p : cum + p;
    cumv[a] = cum;
  }
}

struct CpResetEntry {  // same layout as wd_reset_entry in wd_core.hip
  wd_global_u32 *data;
  const wd_global_u32 *ref;
  int row_elems;
  int pad_;
};

// Fused rollout tick(s): sample the action + step + reset a finished replica, `ticks` times per launch
// with the state kept in registers (a single tick at E = 100 000 moves 6.8 MB: under 1 us of HBM time,
// so one launch per tick is launch-bound -- SURVEY 8(d) asks the ceiling run to fuse T ticks).  The
// policy's probabilities are read once per launch, so ticks > 1 is a fixed-policy rollout; every tick
// still writes its action, observation, reward, done and timestep.  `_done_` reports the last tick.
// With the four `*_batch` pointers (the trainer's [T, E, ...] batch tensors, training/data_loader.py:
// processed observations, actions, rewards, done flags) tick k of the launch writes ROW k of them --
// the observation the policy acted on, the sampled action, the reward and the done flag, exactly what
// trainer_base.py:392-426 records per tick -- so T ticks move T times the bytes (28 B per env-step,
// all coalesced: the real HBM ceiling run of configs[4]); the per-tick arrays then receive the last
// tick only.  With `policy` (packed weights of a two-hidden-layer MLP of width `hidden` = 32 or 64 and one
// head, in dynamic LDS) the rollout runs with a LIVE policy: every tick evaluates the network on the
// replica's current observation instead of reading `probs` -- a whole training batch of ticks is then
// one launch (the reference runs policy forward, sampler, step and reset as separate launches with
// three host synchronisations per tick, trainer_base.py:392-426).
// (No __restrict__ on the arrays: the reset table aliases them.)
// stores of the tick loop through a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: the per-lane 64-bit
// address arithmetic of four record arrays per tick (a v_mad_u64 and three-instruction adds each) becomes scalar work.
// Untracked like wd_store_untracked (wd_common.h): the loop never reads these addresses back.
__device__ __forceinline__ void cp_store_s(const void *sbase, uint32_t voff, int v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void cp_store_s(const void *sbase, uint32_t voff, float v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void cp_store_s(const void *sbase, uint32_t voff, float4 v) {
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  const v4f_ q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(q), "s"(sbase) : "memory");
}

// BATCH: the four `*_batch` pointers are given (every tick recorded) -- a compile-time copy of the loop for each case, so
// that the tick carries no "is there a batch" branches
struct CpTrue { static constexpr bool value = true, fast = false; };
struct CpFalse { static constexpr bool value = false, fast = false; };
struct CpTrueFast { static constexpr bool value = true, fast = true; };  // a middle tick with cp_euler<true>

// A2: exactly two actions (Cartpole's own action space), known at compile time
template <int H, bool BATCH, bool A2 = false>
__device__ __forceinline__ void cp_tick_impl(float *cp_weights,
    float4 *state_arr, int *action_arr, int *done_arr,
    float *reward_arr, float4 *observation_arr, float gravity,
    float masspole, float total_mass, float length, float polemass_length, float force_mag,
    float tau, float theta_threshold_radians, float x_threshold,
    int *env_timestep_arr, int episode_length, int n_envs, uint32_t *rng_state,
    const float *__restrict__ probs, int n_actions, const void *reset_table, int n_reset_arrays,
    int stream_tag, int ticks, float4 *obs_batch, int *action_batch, float *reward_batch, int *done_batch,
    const float *policy, int hidden, int invariant_divide_ok) {
  const CpPhysics p{gravity, masspole, total_mass, length, polemass_length, force_mag, tau,
                    theta_threshold_radians, x_threshold, 1.0f / total_mass};
  const CpResetEntry *table = (const CpResetEntry *)reset_table;
  const uint32_t k0 = rng_state[0], k1 = rng_state[1];
  if (H > 0) {
    const int n_w = 4 * H + H + H * H + H + n_actions * H + n_actions;
    for (int i = threadIdx.x; i < n_w; i += blockDim.x) cp_weights[i] = policy[i];
    __syncthreads();
  }
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < n_envs; env += gridDim.x * blockDim.x) {
    int t = env_timestep_arr[env];
    float4 s = state_arr[env];
    const uint32_t epoch0 = rng_state[WD_RNG_HEADER + env];
    wd_u4 blk = wd_u4{0u, 0u, 0u, 0u};   // the Philox block of four consecutive ticks (wd_tick_draw)
    uint32_t blk_quad = 0xffffffffu;
    const float *row = probs + (long)env * n_actions;
    // The running float32 sums of the (fixed) probabilities, once per launch and in registers: a load
    // inside the tick loop would wait for every store issued before it (the memory counters return in
    // order) -- 2.9 us per tick at 100 000 replicas instead of 0.5.
    // The rows a finished replica is restored from, preloaded ONCE (the usual registration: `state` and the
    // observation, four floats each).  A random policy ends a Cartpole episode every ~20 ticks, so some lane of a
    // wavefront finishes on nearly every tick; the restore used to LOAD the registered rows inside the tick loop
    // (and reload the state): 12 loads per tick and wavefront, each waiting for every store issued before it (the
    // memory counters return in order) -- 70 % of the wavefronts' cycles were waits (profiles/r04_pmc_mix_cartpole_T50.txt).
    CpResetEntry ent0 = CpResetEntry{nullptr, nullptr, 0, 0}, ent1 = ent0;
    uint4 row0 = make_uint4(0u, 0u, 0u, 0u), row1 = row0;
    bool cached = (n_reset_arrays >= 1) && (n_reset_arrays <= 2);
    if (cached) {
      ent0 = table[0];
      ent1 = (n_reset_arrays == 2) ? table[1] : table[0];
      cached = (ent0.row_elems == 4) && (ent1.row_elems == 4) &&
               ((size_t)ent0.data == (size_t)state_arr || (size_t)ent1.data == (size_t)state_arr);
      if (cached) {
        row0 = ((const uint4 __attribute__((address_space(1))) *)ent0.ref)[env];
        row1 = ((const uint4 __attribute__((address_space(1))) *)ent1.ref)[env];
      }
    }
    float cumv[CP_MAX_REG_ACTIONS];
    if (H == 0) {
      float cum = 0.0f;
#pragma unroll
      for (int i = 0; i < CP_MAX_REG_ACTIONS; ++i) {
        if (i < n_actions) cum = (i == 0) ? row[0] : cum + row[i];
        cumv[i] = cum;
      }
    }
    // Every value loaded above is consumed HERE, before the tick loop: the wait for a load whose first use is inside
    // the loop is placed inside the loop, executed on every tick, and also waits for every store of the previous tick.
    asm volatile("" : "+v"(t), "+v"(s.x), "+v"(s.y), "+v"(s.z), "+v"(s.w));
    asm volatile("" : "+v"(row0.x), "+v"(row0.y), "+v"(row0.z), "+v"(row0.w), "+v"(row1.x), "+v"(row1.y), "+v"(row1.z), "+v"(row1.w));
#pragma unroll
    for (int i = 0; i < CP_MAX_REG_ACTIONS; ++i) asm volatile("" : "+v"(cumv[i]));
    // per-lane byte offsets of this replica's row in the 16-byte and the 4-byte arrays (constant over the ticks); the
    // record arrays' row k starts n_envs elements further on every tick: a scalar pointer
    const uint32_t off16 = 16u * (uint32_t)env, off4 = 4u * (uint32_t)env;
    const unsigned char *ob_k = (const unsigned char *)obs_batch, *ab_k = (const unsigned char *)action_batch,
                        *rb_k = (const unsigned char *)reward_batch, *db_k = (const unsigned char *)done_batch;
    // a batch launch whose restore rows sit in registers keeps the per-tick arrays, the state and the restored rows in
    // registers until the LAST tick: every one of those addresses is written again then (the per-tick arrays and the
    // state on the last tick of every replica, the registered rows = state + observation), so what memory holds after
    // the launch is the same, and a replica that finishes mid-launch costs two register moves instead of seven stores
    // (`cached` is the same in every lane -- the table is one per launch -- but it was computed from vector loads: say so,
    // or the two loops below become divergent control flow and the record pointers vector registers)
    const bool lazy = BATCH && (__builtin_amdgcn_readfirstlane(cached ? 1 : 0) != 0);
    // the state a finished replica restarts from (the preloaded row registered for `state`)
    const uint4 sr0 = ((size_t)ent0.data == (size_t)state_arr) ? row0 : row1;
    const float4 s_restart = make_float4(__uint_as_float(sr0.x), __uint_as_float(sr0.y), __uint_as_float(sr0.z), __uint_as_float(sr0.w));

    // ---- one tick.  MID (compile time): a tick of a `lazy` launch that is not its last -- four record stores, the Euler
    // step, and a finished replica restarts by five selects: no per-tick-array stores, no branch on "finished", no
    // divergent code at all.  Otherwise the general tick (`last` = the launch's last tick).
    auto tick = [&](int k, auto mid_tag, bool last) __attribute__((always_inline)) {
      constexpr bool MID = decltype(mid_tag)::value;
      // ---- sample (random.cu:51-85): inverse CDF on a running float32 sum
      const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)env, epoch0 + (uint32_t)k, (uint32_t)stream_tag, k0, k1,
                                                      blk, blk_quad));
      if (H > 0) cp_policy_cum<(H > 0 ? H : 4)>(cp_weights, s, n_actions, cumv);  // live policy: THIS tick's observation
      int cnt = 0;
      if (A2 || n_actions == 2) {  // (uniform) two compares instead of eight masked ones
        cnt = ((cumv[0] < u) ? 1 : 0) + ((cumv[1] < u) ? 1 : 0);
      } else if (n_actions <= CP_MAX_REG_ACTIONS) {
#pragma unroll
        for (int i = 0; i < CP_MAX_REG_ACTIONS; ++i) cnt += (i < n_actions && cumv[i] < u) ? 1 : 0;
      } else {
        float cum = 0.0f;
        for (int i = 0; i < n_actions; ++i) {
          cum = (i == 0) ? row[0] : cum + row[i];
          cnt += (cum < u) ? 1 : 0;
        }
      }
      const int a = min(cnt, (A2 ? 2 : n_actions) - 1);
      // ---- step
      if (BATCH) cp_store_s(ob_k, off16, s);  // the observation this action was sampled on
      t += 1;
      const bool terminated = cp_euler<decltype(mid_tag)::fast>(s, a, p);
      const bool fin = (t == episode_length) || terminated;
      if (BATCH) {
        cp_store_s(ab_k, off4, a);
        cp_store_s(rb_k, off4, 1.0f);
        cp_store_s(db_k, off4, fin ? 1 : 0);
        ob_k += 16 * (size_t)n_envs; ab_k += 4 * (size_t)n_envs; rb_k += 4 * (size_t)n_envs; db_k += 4 * (size_t)n_envs;
      }
      if (MID) {
        t = fin ? 0 : t;
        s.x = fin ? s_restart.x : s.x; s.y = fin ? s_restart.y : s.y;
        s.z = fin ? s_restart.z : s.z; s.w = fin ? s_restart.w : s.w;
        return;
      }
      // (untracked stores, wd_common.h: a store the compiler tracks inside the loop costs a wait for ALL stores on
      // every trip, executed or not.  The one place that reads such an address back -- the restore of a replica whose
      // reset rows were not preloaded -- drains the counter itself first.)
      if (!BATCH || last || (fin && !lazy)) {
        cp_store_s(action_arr, off4, a);
        cp_store_s(observation_arr, off16, s);
        cp_store_s(reward_arr, off4, 1.0f);
        cp_store_s(done_arr, off4, fin ? 1 : 0);
      }
      if (last || (fin && !lazy)) cp_store_s(state_arr, off16, s);  // otherwise the state stays in registers
      // ---- reset in place (reset.cu:9-75 for every registered array); `_done_` stays set
      if (fin) {
        t = 0;
        if (cached) {  // stores only
          if (last || !lazy) {
            // (after the untracked stores above to the same rows: stores of one wavefront to one address stay in order)
            wd_store_untracked((float4 *)ent0.data + env, make_float4(__uint_as_float(row0.x), __uint_as_float(row0.y), __uint_as_float(row0.z), __uint_as_float(row0.w)));
            if (n_reset_arrays == 2)
              wd_store_untracked((float4 *)ent1.data + env, make_float4(__uint_as_float(row1.x), __uint_as_float(row1.y), __uint_as_float(row1.z), __uint_as_float(row1.w)));
          }
          s = s_restart;
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the untracked stores above have left
          for (int r = 0; r < n_reset_arrays; ++r) {
            const CpResetEntry ent = table[r];
            const long base = (long)env * ent.row_elems;
            for (int i = 0; i < ent.row_elems; ++i) ent.data[base + i] = ent.ref[base + i];
          }
          s = state_arr[env];
          // the reload is consumed HERE, inside the rare branch: otherwise the wait for it lands at the top
          // of the tick loop, where it also waits for every store of the previous tick
          asm volatile("" : "+v"(s.x), "+v"(s.y), "+v"(s.z), "+v"(s.w));
        }
      }
    };
    if (lazy) {  // (uniform) every tick but the last: the branch-free body
      // cp_euler<true>'s preconditions, checked ONCE per launch and wavefront: the division shortcut is proved for this total
      // mass (host flag), a running replica's angle stays inside 0.75 (threshold), and so do the angles the wavefront starts
      // from and restarts from.  A middle tick then only ever sees the restart state or a state that did not terminate.
      const bool small = (p.theta_threshold_radians <= 0.75f) && (fabsf(s.z) <= 0.75f) && (fabsf(s_restart.z) <= 0.75f);
      const bool mass_ok = (p.masspole >= 0x1.0p-60f) && (p.masspole <= 0x1.0p60f);
      const bool fast = (invariant_divide_ok != 0) && mass_ok && (__ballot(!small) == 0ull);  // (wave-uniform)
      if (fast) for (int k = 0; k < ticks - 1; ++k) tick(k, CpTrueFast{}, false);
      else for (int k = 0; k < ticks - 1; ++k) tick(k, CpTrue{}, false);
      tick(ticks - 1, CpFalse{}, true);
    } else {
      for (int k = 0; k < ticks; ++k) tick(k, CpFalse{}, k == ticks - 1);
    }
    env_timestep_arr[env] = t;
    rng_state[WD_RNG_HEADER + env] = epoch0 + (uint32_t)ticks;
  }
}

}  // namespace

extern "C" {

// Exhaustive proof for ONE divisor c (the env's total mass) that cp_div_invariant(x, c, RN(1 / c)) == x / c for every
// float32 x of one binade, both signs (2^24 values; all other normal x follow by scaling): `ok` (preset to 1 by the host)
// is cleared on the first mismatch (a test hands it a reciprocal that is one ulp off: it must notice).  Run once per env
// object (envs/cartpole.py::invariant_divide_ok).
__global__ void HipCartPoleVerifyInvariantDivide(float c, float y, int *ok) {  // y: RN(1 / c), formed by the host
  if (y != 1.0f / c && threadIdx.x == 0 && blockIdx.x == 0) *ok = 0;  // (not what the tick kernels use: 1.0f / total_mass)
  // bits 0 .. 22: the significand; bit 23: a second binade (x 2^40: the scaling argument, spot-checked); bit 24: the sign
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << 25); i += gridDim.x * blockDim.x) {
    const float m = __uint_as_float(0x3f800000u | (i & 0x7fffffu) | ((i >> 24) << 31));
    const float x = ((i >> 23) & 1u) ? m * 0x1.0p40f : m;
    const float want = x / c, got = cp_div_invariant(x, c, y);
    if (__float_as_uint(want) != __float_as_uint(got)) *ok = 0;
  }
}

// cp_sincos_small against wd_np_sincosf for every float32 of [-0.75, 0.75] (the test of the claim in its comment)
__global__ void HipCartPoleVerifySmallAngle(int *ok) {
  const uint32_t top = __float_as_uint(0.75f);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= top; i += gridDim.x * blockDim.x) {
    const float xs[2] = {__uint_as_float(i), __uint_as_float(i | 0x80000000u)};
    for (int k = 0; k < 2; ++k) {
      float s0, c0, s1, c1;
      wd_np_sincosf(xs[k], s0, c0);
      cp_sincos_small(xs[k], s1, c1);
      if (__float_as_uint(s0) != __float_as_uint(s1) || __float_as_uint(c0) != __float_as_uint(c1)) *ok = 0;
    }
  }
}

__global__ void HipClassicControlCartPoleEnvStep(
    float4 *__restrict__ state_arr, const int *__restrict__ action_arr, int *__restrict__ done_arr,
    float *__restrict__ reward_arr, float4 *__restrict__ observation_arr, float gravity,
    float masspole, float total_mass, float length, float polemass_length, float force_mag,
    float tau, float theta_threshold_radians, float x_threshold,
    int *__restrict__ env_timestep_arr, int episode_length, int n_envs) {
  const CpPhysics p{gravity, masspole, total_mass, length, polemass_length, force_mag, tau,
                    theta_threshold_radians, x_threshold, 1.0f / total_mass};
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < n_envs; env += gridDim.x * blockDim.x) {
    const int t = env_timestep_arr[env] + 1;
    env_timestep_arr[env] = t;
    float4 s = state_arr[env];
    const bool terminated = cp_euler(s, action_arr[env], p);
    state_arr[env] = s;
    observation_arr[env] = s;
    reward_arr[env] = 1.0f;
    if (t == episode_length || terminated) done_arr[env] = 1;
  }
}

__global__ void HipClassicControlCartPoleEnvTick(
    float4 *state_arr, int *action_arr, int *done_arr,
    float *reward_arr, float4 *observation_arr, float gravity,
    float masspole, float total_mass, float length, float polemass_length, float force_mag,
    float tau, float theta_threshold_radians, float x_threshold,
    int *env_timestep_arr, int episode_length, int n_envs, uint32_t *rng_state,
    const float *__restrict__ probs, int n_actions, const void *reset_table, int n_reset_arrays,
    int stream_tag, int ticks, float4 *obs_batch, int *action_batch, float *reward_batch, int *done_batch,
    const float *policy, int hidden, int invariant_divide_ok) {
  if (obs_batch && n_actions == 2)  // the recorded two-action rollout (configs[4]): sizes folded
    cp_tick_impl<0, true, true>(nullptr, state_arr, action_arr, done_arr, reward_arr, observation_arr, gravity, masspole, total_mass, length, polemass_length, force_mag, tau, theta_threshold_radians, x_threshold, env_timestep_arr, episode_length, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks, obs_batch, action_batch, reward_batch, done_batch, policy, hidden, invariant_divide_ok);
  else if (obs_batch)
    cp_tick_impl<0, true>(nullptr, state_arr, action_arr, done_arr, reward_arr, observation_arr, gravity, masspole, total_mass, length, polemass_length, force_mag, tau, theta_threshold_radians, x_threshold, env_timestep_arr, episode_length, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks, obs_batch, action_batch, reward_batch, done_batch, policy, hidden, invariant_divide_ok);
  else
    cp_tick_impl<0, false>(nullptr, state_arr, action_arr, done_arr, reward_arr, observation_arr, gravity, masspole, total_mass, length, polemass_length, force_mag, tau, theta_threshold_radians, x_threshold, env_timestep_arr, episode_length, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks, obs_batch, action_batch, reward_batch, done_batch, policy, hidden, invariant_divide_ok);
}

// the rollout with a live policy (weights in dynamic LDS); `hidden` must equal the entry's width
#define WD_CP_ROLLOUT(HH)                                                                          \
  __global__ void __launch_bounds__(256, 2) HipClassicControlCartPoleEnvRollout_H##HH(             \
    float4 *state_arr, int *action_arr, int *done_arr, \
    float *reward_arr, float4 *observation_arr, float gravity, \
    float masspole, float total_mass, float length, float polemass_length, float force_mag, \
    float tau, float theta_threshold_radians, float x_threshold, \
    int *env_timestep_arr, int episode_length, int n_envs, uint32_t *rng_state, \
    const float *__restrict__ probs, int n_actions, const void *reset_table, int n_reset_arrays, \
    int stream_tag, int ticks, float4 *obs_batch, int *action_batch, float *reward_batch, int *done_batch, \
    const float *policy, int hidden, int invariant_divide_ok) {           \
    extern __shared__ __attribute__((aligned(16))) float cp_lds[];                                 \
    if (obs_batch)                                                                                 \
      cp_tick_impl<HH, true>(cp_lds, state_arr, action_arr, done_arr, reward_arr, observation_arr, gravity, masspole, total_mass, length, polemass_length, force_mag, tau, theta_threshold_radians, x_threshold, env_timestep_arr, episode_length, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks, obs_batch, action_batch, reward_batch, done_batch, policy, hidden, invariant_divide_ok); \
    else                                                                                           \
      cp_tick_impl<HH, false>(cp_lds, state_arr, action_arr, done_arr, reward_arr, observation_arr, gravity, masspole, total_mass, length, polemass_length, force_mag, tau, theta_threshold_radians, x_threshold, env_timestep_arr, episode_length, n_envs, rng_state, probs, n_actions, reset_table, n_reset_arrays, stream_tag, ticks, obs_batch, action_batch, reward_batch, done_batch, policy, hidden, invariant_divide_ok); \
  }
WD_CP_ROLLOUT(32)
WD_CP_ROLLOUT(64)

}  // extern "C"
