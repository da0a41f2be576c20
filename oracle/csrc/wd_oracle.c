/*
 * wd_oracle.c -- plain-C restatement of the reference's CPU step() for the
 * rollout hot path.  TEST INFRASTRUCTURE (checker + reported CPU baseline), not
 * product code: nothing under warp_drive_amd/ links or loads this file.
 *
 * Follows (reference paths relative to /root/reference):
 *   example_envs/tag_continuous/tag_continuous.py
 *       update_state            :339-401
 *       compute_distance        :403-420   (np.float32 scalar ** 2  == libm powf(x, 2))
 *       k_nearest_neighbors     :422-444   (heapq.nsmallest == stable by (dist, id))
 *       generate_observation    :446-610
 *       compute_reward          :612-678
 *       step / done             :796-887
 *   example_envs/tag_gridworld/tag_gridworld.py
 *       update_state :152-192, generate_observation :194-275, step :291-317
 *
 * numpy float32 cos/sin: the reference calls np.cos/np.sin on float32 arrays
 * (tag_continuous.py:370-373).  numpy's float32 kernels are its own SIMD
 * Cody-Waite + minimax-polynomial routine (numpy/_core/src/umath/
 * loops_trigonometric.dispatch.*, "max ULP 1.49"), restated scalar in
 * np_sincosf() below with explicit fmaf; tests/test_oracle_golden.py checks it is
 * bit-identical to np.cos/np.sin of the numpy in this image.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -mfma [-fopenmp]   (oracle/build.py)
 * -ffp-contract=off matters: every a*b+c below that is NOT an explicit fmaf must
 * round twice, exactly like numpy's separate ufunc calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */

/* np.float32.__pow__(x, 2) -> npy_powf -> libm powf (tag_continuous.py:414,419).
 * The exponent is read through a volatile so gcc cannot fold powf(x, 2) into x*x
 * (the two differ by 1 ulp for ~0.07 % of inputs). */
static volatile float kTwo = 2.0f;
static inline float powf_2(float x) { return powf(x, kTwo); }

void wdo_powf2(const float *in, float *out, long n) {
  for (long i = 0; i < n; ++i) out[i] = powf_2(in[i]);
}

static inline float np_sincosf(float x, int want_cos) {
  /* valid for |x| <= 71476 (cos) / 117435 (sin); directions live in [0, 2pi] */
  const float two_over_pi = 0x1.45f306p-1f;
  const float c1 = -0x1.921fb0p+00f, c2 = -0x1.5110b4p-22f, c3 = -0x1.846988p-48f;
  const float magic = 0x1.800000p+23f;
  /* numpy's kernel is built with FP contraction on its FMA dispatch targets, so the
   * quadrant is round(fma(x, 2/pi, magic)) - magic: the product is NOT rounded before
   * the magic add.  It matters exactly when x*2/pi rounds to k + 0.5 (e.g. x = 5*pi/4). */
  float q = fmaf(x, two_over_pi, magic);
  q = q - magic;
  float r = fmaf(q, c1, x);
  r = fmaf(q, c2, r);
  r = fmaf(q, c3, r);
  float r2 = r * r;
  float c = fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
  c = fmaf(c, r2, 0x1.55553cp-05f);
  c = fmaf(c, r2, -0x1.000000p-01f);
  c = fmaf(c, r2, 0x1.000000p+00f);
  float s = fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
  s = fmaf(s, r2, 0x1.11119ap-07f);
  s = fmaf(s, r2, -0x1.555556p-03f);
  s = fmaf(s, r2, 0.0f);
  s = fmaf(s, r, r);
  int iq = (int)q + (want_cos ? 1 : 0);
  float out = (iq & 1) == 0 ? s : c;
  if (iq & 2) out = 0.0f - out;
  return out;
}

void wdo_np_cosf(const float *in, float *out, long n) {
  for (long i = 0; i < n; ++i) out[i] = np_sincosf(in[i], 1);
}
void wdo_np_sinf(const float *in, float *out, long n) {
  for (long i = 0; i < n; ++i) out[i] = np_sincosf(in[i], 0);
}

static inline float np_remainderf(float a, float b) {
  /* numpy npy_divmodf: fmod, then fix the sign so the result has b's sign */
  float m = fmodf(a, b);
  if (m != 0.0f) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysignf(0.0f, b);
  }
  return m;
}

/* ------------------------------------------------------- TagContinuous step */

typedef struct {
  int n_envs, n_agents, episode_length;
  int num_other_agents_observed; /* K */
  int use_full_observation;
  int runner_exits_game_after_tagged;
  float grid_length, max_speed, edge_hit_penalty;
  float distance_margin_for_reward;
  float tag_reward_for_tagger, tag_penalty_for_runner, end_of_game_reward_for_runner;
  int n_acc_actions, n_turn_actions;
} wdo_tc_cfg;

#define TC_SCRATCH_BYTES(N) (sizeof(double) * (N) * 2 + sizeof(float) * (N) * 4 + sizeof(int) * (N))

/* One tick for envs [e0, e1).  All arrays are [E, N(,...)] row-major, float32 /
 * int32, the same layout the device uses.  obs is float32 [E, N, F]. */
static void tc_step_range(const wdo_tc_cfg *c, int e0, int e1, float *loc_x, float *loc_y,
                          float *speed, float *direction, float *acceleration,
                          const int *agent_types, float *edge_pen, const float *acc_actions,
                          const float *turn_actions, const float *skill_levels, int *sig,
                          float *obs, const int *actions, float *rewards,
                          const float *step_rewards, int *num_runners, int *done,
                          int *timestep, int *nearest_ids, void *scratch) {
  const int N = c->n_agents, K = c->num_other_agents_observed;
  const int F = c->use_full_observation ? 7 * (N - 1) + 1 : 7 * K + 1;
  const float two_pi = (float)(2 * M_PI); /* python float, weak-promoted to float32 */
  const float L = c->grid_length;
  const double diag = (double)L * sqrt(2.0);        /* :146  float32 * np.float64 -> f64 */
  const float sp_div = c->max_speed + 1e-10f;       /* :456  float32 + float32(eps)      */
  /* scratch: TC_SCRATCH_BYTES(N) supplied by the caller (once per thread), or allocated here */
  void *own = scratch ? NULL : malloc(TC_SCRATCH_BYTES(N));
  double *nx = (double *)(scratch ? scratch : own);
  double *ny = nx + N;
  float *nf = (float *)(nx + 2 * N); /* speed, acc, dir normalised (f32) */
  float *cd = nf + 3 * N;
  int *cid = (int *)(cd + N);

  for (int e = e0; e < e1; ++e) {
    float *x = loc_x + (size_t)e * N, *y = loc_y + (size_t)e * N;
    float *sp = speed + (size_t)e * N, *dr = direction + (size_t)e * N;
    float *ac = acceleration + (size_t)e * N, *ep = edge_pen + (size_t)e * N;
    int *sg = sig + (size_t)e * N;
    const int *act = actions + (size_t)e * N * 2;
    float *rw = rewards + (size_t)e * N;
    float *ob = obs + (size_t)e * N * F;
    timestep[e] += 1; /* :800 */
    const int t = timestep[e];

    /* ---- update_state :339-401 */
    for (int i = 0; i < N; ++i) {
      const float da = acc_actions[act[2 * i + 0]];
      const float dt = turn_actions[act[2 * i + 1]];
      const float s = (float)sg[i];
      float d = np_remainderf(dr[i] + dt, two_pi) * s; /* :355-357 */
      float a = ac[i] + da;                            /* :359 */
      const float vmax = c->max_speed * skill_levels[i];
      float v = sp[i] + a;
      v = fmaxf(v, 0.0f);
      v = fminf(v, vmax);
      v = v * s;                                       /* :364-366 */
      a = a * (v > 0.0f ? 1.0f : 0.0f) * (v < vmax ? 1.0f : 0.0f); /* :367 */
      float px = x[i] + v * np_sincosf(d, 1);          /* :369-374 */
      float py = y[i] + v * np_sincosf(d, 0);
      const int crossed = !((px >= 0) && (px <= L) && (py >= 0) && (py <= L));
      px = fminf(fmaxf(px, 0.0f), L);
      py = fminf(fmaxf(py, 0.0f), L);
      ep[i] = c->edge_hit_penalty * (crossed ? 1.0f : 0.0f); /* :394 */
      x[i] = px; y[i] = py; sp[i] = v; dr[i] = d; ac[i] = a;
    }

    /* ---- generate_observation :446-610 (uses still_in_the_game BEFORE tagging) */
    for (int i = 0; i < N; ++i) {
      nx[i] = (double)x[i] / diag;
      ny[i] = (double)y[i] / diag;
      nf[i] = sp[i] / sp_div;
      nf[N + i] = ac[i] / sp_div;
      nf[2 * N + i] = dr[i] / two_pi;
    }
    const float tfrac = (float)((double)t / (double)c->episode_length);
    for (int i = 0; i < N; ++i) {
      float *o = ob + (size_t)i * F;
      if (c->use_full_observation) {
        const int M = N - 1;
        int k = 0;
        for (int j = 0; j < N; ++j) {
          if (j == i) continue;
          if (sg[i]) {
            o[0 * M + k] = (float)(nx[j] - nx[i]);
            o[1 * M + k] = (float)(ny[j] - ny[i]);
            o[2 * M + k] = (float)((double)nf[j] - (double)nf[i]);
            o[3 * M + k] = (float)((double)nf[N + j] - (double)nf[N + i]);
            o[4 * M + k] = (float)((double)nf[2 * N + j] - (double)nf[2 * N + i]);
          } else {
            o[0 * M + k] = o[1 * M + k] = o[2 * M + k] = o[3 * M + k] = o[4 * M + k] = 0.0f;
          }
          o[5 * M + k] = (float)agent_types[j];
          o[6 * M + k] = (float)sg[j];
          ++k;
        }
        o[7 * M] = sg[i] ? tfrac : 0.0f;
      } else {
        for (int f = 0; f < F; ++f) o[f] = 0.0f; /* :548 init_obs */
        /* optional output: the ids the row was built from, -1 = fewer than K others in the game (:422-444) */
        int *nid = nearest_ids ? nearest_ids + ((size_t)e * N + i) * K : NULL;
        if (nid) for (int k = 0; k < K; ++k) nid[k] = -1;
        if (!sg[i]) continue;
        /* k nearest among others still in the game, stable by (distance, id) :422-444 */
        int cnt = 0;
        for (int j = 0; j < N; ++j) {
          if (j == i || !sg[j]) continue;
          const float dx = x[i] - x[j], dy = y[i] - y[j];
          const float d = sqrtf(powf_2(dx) + powf_2(dy)); /* :409-420 */
          /* insertion keeps ascending (d, id); strict '<' keeps earlier ids first */
          int q;
          if (cnt < K) q = cnt++;
          else if (d < cd[K - 1]) q = K - 1;
          else continue;
          while (q > 0 && d < cd[q - 1]) { cd[q] = cd[q - 1]; cid[q] = cid[q - 1]; --q; }
          cd[q] = d; cid[q] = j;
        }
        for (int k = 0; k < cnt; ++k) {
          const int j = cid[k];
          if (nid) nid[k] = j;
          o[0 * K + k] = (float)(nx[j] - nx[i]);
          o[1 * K + k] = (float)(ny[j] - ny[i]);
          o[2 * K + k] = (float)((double)nf[j] - (double)nf[i]);
          o[3 * K + k] = (float)((double)nf[N + j] - (double)nf[N + i]);
          o[4 * K + k] = (float)((double)nf[2 * N + j] - (double)nf[2 * N + i]);
          o[5 * K + k] = (float)agent_types[j];
          o[6 * K + k] = (float)sg[j];
        }
        o[7 * K] = tfrac;
      }
    }

    /* ---- compute_reward :612-678 */
    for (int i = 0; i < N; ++i) {
      float r = 0.0f;
      if (sg[i]) { r += ep[i]; r += step_rewards[i]; } /* :655-658 */
      rw[i] = r;
    }
    int nr = num_runners[e];
    for (int i = 0; i < N; ++i) { /* runners in id order :660 */
      if (agent_types[i] != 0 || !sg[i]) continue;
      float best = INFINITY; int bt = -1;
      for (int j = 0; j < N; ++j) { /* taggers in id order, first min wins :643-651 */
        if (agent_types[j] != 1) continue;
        const float dx = x[i] - x[j], dy = y[i] - y[j];
        const float d = sqrtf(dx * dx + dy * dy); /* array ** 2 == np.square :630-641 */
        if (d < best) { best = d; bt = j; }
      }
      int still_runner = 1;
      if (best < c->distance_margin_for_reward) { /* :661 */
        rw[i] += c->tag_penalty_for_runner;
        rw[bt] += c->tag_reward_for_tagger;
        if (c->runner_exits_game_after_tagged) { sg[i] = 0; nr -= 1; still_runner = 0; }
      }
      if (t == c->episode_length && still_runner) rw[i] += c->end_of_game_reward_for_runner;
    }
    num_runners[e] = nr;
    done[e] = (t >= c->episode_length) || (nr == 0); /* :880-883 */
  }
  free(own);
}

void wdo_tc_step_ids(const wdo_tc_cfg *c, float *loc_x, float *loc_y, float *speed,
                 float *direction, float *acceleration, const int *agent_types,
                 float *edge_pen, const float *acc_actions, const float *turn_actions,
                 const float *skill_levels, int *sig, float *obs, const int *actions,
                 float *rewards, const float *step_rewards, int *num_runners, int *done,
                 int *timestep, int *nearest_ids, int n_threads) {
  const int E = c->n_envs;
  if (n_threads <= 1) {
    tc_step_range(c, 0, E, loc_x, loc_y, speed, direction, acceleration, agent_types, edge_pen,
                  acc_actions, turn_actions, skill_levels, sig, obs, actions, rewards,
                  step_rewards, num_runners, done, timestep, nearest_ids, NULL);
    return;
  }
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
  for (int b = 0; b < n_threads; ++b) {
    const int e0 = (int)((long)E * b / n_threads), e1 = (int)((long)E * (b + 1) / n_threads);
    tc_step_range(c, e0, e1, loc_x, loc_y, speed, direction, acceleration, agent_types, edge_pen,
                  acc_actions, turn_actions, skill_levels, sig, obs, actions, rewards,
                  step_rewards, num_runners, done, timestep, nearest_ids, NULL);
  }
}

/* cpu_baseline leg of bench.py: `n_ticks` ticks of every replica with the SAME actions each tick (no resets).
 * Replicas are independent, so a thread takes a chunk of replicas through all its ticks (dynamic schedule over
 * chunks, no barrier between ticks, scratch allocated once per thread). */
void wdo_tc_run_ticks(const wdo_tc_cfg *c, float *loc_x, float *loc_y, float *speed,
                      float *direction, float *acceleration, const int *agent_types,
                      float *edge_pen, const float *acc_actions, const float *turn_actions,
                      const float *skill_levels, int *sig, float *obs, const int *actions,
                      float *rewards, const float *step_rewards, int *num_runners, int *done,
                      int *timestep, int n_ticks, int chunk, int n_threads) {
  const int E = c->n_envs;
  if (chunk < 1) chunk = 1;
  const int n_chunks = (E + chunk - 1) / chunk;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
  {
    void *scratch = malloc(TC_SCRATCH_BYTES(c->n_agents));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < n_chunks; ++b) {
      const int e0 = b * chunk, e1 = e0 + chunk < E ? e0 + chunk : E;
      for (int t = 0; t < n_ticks; ++t)
        tc_step_range(c, e0, e1, loc_x, loc_y, speed, direction, acceleration, agent_types, edge_pen,
                      acc_actions, turn_actions, skill_levels, sig, obs, actions, rewards,
                      step_rewards, num_runners, done, timestep, NULL, scratch);
    }
    free(scratch);
  }
}

/* the same without the neighbour-id output */
void wdo_tc_step(const wdo_tc_cfg *c, float *loc_x, float *loc_y, float *speed,
                 float *direction, float *acceleration, const int *agent_types,
                 float *edge_pen, const float *acc_actions, const float *turn_actions,
                 const float *skill_levels, int *sig, float *obs, const int *actions,
                 float *rewards, const float *step_rewards, int *num_runners, int *done,
                 int *timestep, int n_threads) {
  wdo_tc_step_ids(c, loc_x, loc_y, speed, direction, acceleration, agent_types, edge_pen, acc_actions,
                  turn_actions, skill_levels, sig, obs, actions, rewards, step_rewards, num_runners, done,
                  timestep, NULL, n_threads);
}

/* -------------------------------------------------------- TagGridWorld step */

void wdo_gw_step(int n_envs, int n_agents, int episode_length, int world_boundary,
                 int use_full_observation, double wall_hit_penalty, double tag_reward_for_tagger,
                 double tag_penalty_for_runner, double step_cost_for_tagger, int *loc_x,
                 int *loc_y, const int *actions, float *rewards, float *obs, int *done,
                 int *timestep) {
  static const int kAct[10] = {0, 0, 1, 0, -1, 0, 0, 1, 0, -1}; /* :104 */
  const int N = n_agents, L = world_boundary;
  const int F = use_full_observation ? 4 * N + 1 : 6;
  for (int e = 0; e < n_envs; ++e) {
    int *x = loc_x + (size_t)e * N, *y = loc_y + (size_t)e * N;
    float *rw = rewards + (size_t)e * N, *ob = obs + (size_t)e * N * F;
    timestep[e] += 1;
    const double tfrac = (double)timestep[e] / episode_length;
    int tag = 0;
    double pen[1024];
    for (int i = 0; i < N; ++i) {
      const int a = actions[(size_t)e * N + i];
      const int ux = x[i] + kAct[2 * a], uy = y[i] + kAct[2 * a + 1];
      const int cx = ux < 0 ? 0 : (ux > L ? L : ux), cy = uy < 0 ? 0 : (uy > L ? L : uy);
      pen[i] = -1.0 * wall_hit_penalty * ((ux != cx) || (uy != cy) ? 1.0 : 0.0);
      x[i] = cx; y[i] = cy;
    }
    for (int i = 0; i < N - 1; ++i) tag |= (x[i] == x[N - 1] && y[i] == y[N - 1]);
    for (int i = 0; i < N; ++i) {
      double r;
      if (i < N - 1) r = tag ? tag_reward_for_tagger : -1.0 * step_cost_for_tagger;
      else r = tag ? -1.0 * tag_penalty_for_runner : 1.0 * step_cost_for_tagger;
      rw[i] = (float)(r + pen[i]);
    }
    if (use_full_observation) {
      for (int i = 0; i < N; ++i) {
        float *o = ob + (size_t)i * F;
        for (int j = 0; j < N; ++j) {
          o[j] = (float)((double)x[j] / L);
          o[N + j] = (float)((double)y[j] / L);
          o[2 * N + j] = (j == N - 1) ? 1.0f : 0.0f;
          o[3 * N + j] = (j == i) ? 1.0f : 0.0f;
        }
        o[4 * N] = (float)tfrac;
      }
    } else {
      int best = 0; long bd = -1;
      for (int j = 0; j < N - 1; ++j) {
        const long dx = x[j] - x[N - 1], dy = y[j] - y[N - 1], d = dx * dx + dy * dy;
        if (bd < 0 || d < bd) { bd = d; best = j; }
      }
      for (int i = 0; i < N; ++i) {
        float *o = ob + (size_t)i * 6;
        const int other = (i < N - 1) ? N - 1 : best;
        o[0] = (float)((double)x[i] / L); o[1] = (float)((double)y[i] / L);
        o[2] = (float)((double)x[other] / L); o[3] = (float)((double)y[other] / L);
        o[4] = (i == N - 1) ? 1.0f : 0.0f; o[5] = (float)tfrac;
      }
    }
    done[e] = (timestep[e] >= episode_length) || tag;
  }
}
