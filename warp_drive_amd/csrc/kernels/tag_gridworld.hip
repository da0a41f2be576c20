// tag_gridworld.hip -- TagGridWorld step for gfx950 (integer path, bit-exact).
//
// Semantics follow the reference CPU step, example_envs/tag_gridworld/
// tag_gridworld.py: update_state :152-192, generate_observation :194-275,
// step/done :291-317; argument order follows the reference kernel
// (tag_gridworld_step_pycuda.cu:112-127) with one trailing `n_envs`.
//
// MI355X mapping.  A replica has only N = num_taggers + 1 agents (5 in every
// BASELINE config), so block-per-env would light 5 of 64 lanes.  Instead a block
// packs epb = blockDim.x / N replicas (12 per wavefront at N = 5): thread t serves
// agent t % N of local replica t / N.  Positions are staged in LDS; the observation
// block of the packed replicas is contiguous in HBM ([E, N, F] row-major) and is
// written with block-strided, fully coalesced stores instead of one strided row per
// thread (tag_gridworld_step_pycuda.cu:29-51).  With the reference geometry
// (block=(N,1,1), grid=(E,1)) epb is simply 1.
#include "wd_common.h"

// action index -> (dx, dy); uploaded by the host like the reference
// (kIndexToActionArr, tag_gridworld_step_pycuda.cu:6; env_cpu_gpu_consistency_checker.py:256-264)
// and pre-initialised to TagGridWorld.step_actions (tag_gridworld.py:104).
__constant__ int kIndexToActionArr[10] = {0, 0, 1, 0, -1, 0, 0, 1, 0, -1};

extern "C" __global__ void HipTagGridWorldStep(
    int *__restrict__ states_x_arr, int *__restrict__ states_y_arr,
    const int *__restrict__ actions_arr, int *__restrict__ done_arr,
    float *__restrict__ rewards_arr, float *__restrict__ obs_arr, float wall_hit_penalty,
    float tag_reward_for_tagger, float tag_penalty_for_runner, float step_cost_for_tagger,
    int use_full_observation, int world_boundary, int *__restrict__ env_timestep_arr,
    int episode_length, int n_agents, int n_envs) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  const int N = n_agents;
  const int epb = max(1, (int)blockDim.x / N);  // replicas per block
  int *s_x = s_mem;                             // [epb][N]
  int *s_y = s_x + epb * N;                     // [epb][N]
  int *s_t = s_y + epb * N;                     // [epb] timestep after increment
  int *s_tag = s_t + epb;                       // [epb] runner caught?
  int *s_near = s_tag + epb;                    // [epb] closest tagger (partial obs)
  const int F = use_full_observation ? 4 * N + 1 : 6;
  const int tid = threadIdx.x;
  const int el = tid / N, ag = tid - el * N;
  const float L = (float)world_boundary;

  for (int env0 = blockIdx.x * epb; env0 < n_envs; env0 += gridDim.x * epb) {
    const int env = env0 + el;
    const bool active = (el < epb) && (env < n_envs);
    const int idx = env * N + ag;
    float rew = 0.0f;
    if (active) {
      // ---- movement :152-173
      const int a = actions_arr[idx];
      const int ux = states_x_arr[idx] + kIndexToActionArr[2 * a];
      const int uy = states_y_arr[idx] + kIndexToActionArr[2 * a + 1];
      const int cx = min(max(ux, 0), world_boundary);
      const int cy = min(max(uy, 0), world_boundary);
      if (ux != cx || uy != cy) rew = -wall_hit_penalty;  // -1.0 * wall_hit_penalty * hit
      states_x_arr[idx] = cx;
      states_y_arr[idx] = cy;
      s_x[el * N + ag] = cx;
      s_y[el * N + ag] = cy;
      if (ag == 0) {
        const int t = env_timestep_arr[env] + 1;  // :295
        env_timestep_arr[env] = t;
        s_t[el] = t;
      }
    }
    __syncthreads();
    if (active && ag == 0) {
      // ---- tag check :175-178 and closest tagger :246-261 (first argmin)
      const int rx = s_x[el * N + N - 1], ry = s_y[el * N + N - 1];
      int tag = 0, best = 0, bd = 0x7fffffff;
      for (int j = 0; j < N - 1; ++j) {
        const int dx = s_x[el * N + j] - rx, dy = s_y[el * N + j] - ry;
        const int d = dx * dx + dy * dy;
        tag |= (d == 0);
        if (d < bd) { bd = d; best = j; }
      }
      s_tag[el] = tag;
      s_near[el] = best;
      if (s_t[el] >= episode_length || tag) done_arr[env] = 1;  // :314
    }
    __syncthreads();
    if (active) {
      // ---- rewards :180-187
      const int tag = s_tag[el];
      const float base = (ag < N - 1) ? (tag ? tag_reward_for_tagger : -step_cost_for_tagger)
                                      : (tag ? -tag_penalty_for_runner : step_cost_for_tagger);
      rewards_arr[idx] = base + rew;
    }
    // ---- observations :194-275, coalesced over the packed replicas
    const int envs_here = min(epb, n_envs - env0);
    const int per_env = N * F;
    const long obs_base = (long)env0 * per_env;
    for (int q = tid; q < envs_here * per_env; q += blockDim.x) {
      const int e = q / per_env, r = q - e * per_env;
      const int i = r / F, f = r - i * F;
      const int *x = s_x + e * N, *y = s_y + e * N;
      float v;
      if (use_full_observation) {
        const int c = f / N, j = f - c * N;
        if (c == 0) v = (float)x[j] / L;
        else if (c == 1) v = (float)y[j] / L;
        else if (c == 2) v = (j == N - 1) ? 1.0f : 0.0f;
        else if (c == 3) v = (j == i) ? 1.0f : 0.0f;
        else v = (float)s_t[e] / (float)episode_length;
      } else {
        const int other = (i < N - 1) ? N - 1 : s_near[e];
        if (f == 0) v = (float)x[i] / L;
        else if (f == 1) v = (float)y[i] / L;
        else if (f == 2) v = (float)x[other] / L;
        else if (f == 3) v = (float)y[other] / L;
        else if (f == 4) v = (i == N - 1) ? 1.0f : 0.0f;
        else v = (float)s_t[e] / (float)episode_length;
      }
      obs_arr[obs_base + q] = v;
    }
    __syncthreads();
  }
}
