// dummy_env.hip -- `testkernel`, the fixture the reference's manager tests drive
// (example_envs/dummy_env/test_step.cu:9-45; tests/warp_drive/pycuda_tests/
// test_function_manager.py:71-230).  x /= multiplier; y *= multiplier; done when
// step == episode_length or any y >= target; actions[i] = i.
#include "wd_common.h"

extern "C" __global__ void testkernel(float *x, int *y, int *done, int *actions, float multiplier,
                                      int target, int step, int episode_length, int n_agents,
                                      int n_envs) {
  __shared__ int reach_target;
  const int action_dim = 3;
  for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
    if (threadIdx.x == 0) reach_target = 0;
    __syncthreads();
    for (int ag = threadIdx.x; ag < n_agents; ag += blockDim.x) {
      const int index = env * n_agents + ag;
      x[index] = x[index] / multiplier;
      y[index] = (int)((float)y[index] * multiplier);
      if (y[index] >= target) atomicAdd(&reach_target, 1);
      for (int i = 0; i < action_dim; ++i) actions[index * action_dim + i] = i;
    }
    __syncthreads();
    if ((step == episode_length || reach_target > 0) && threadIdx.x == 0) atomicMax(&done[env], 1);
    __syncthreads();
  }
}
