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

extern "C" __global__ void HipClassicControlCartPoleEnvStep(
    float4 *__restrict__ state_arr, const int *__restrict__ action_arr, int *__restrict__ done_arr,
    float *__restrict__ reward_arr, float4 *__restrict__ observation_arr, float gravity,
    float masspole, float total_mass, float length, float polemass_length, float force_mag,
    float tau, float theta_threshold_radians, float x_threshold,
    int *__restrict__ env_timestep_arr, int episode_length, int n_envs) {
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < n_envs;
       env += gridDim.x * blockDim.x) {
    const int t = env_timestep_arr[env] + 1;
    env_timestep_arr[env] = t;
    const float4 s = state_arr[env];
    float x = s.x, x_dot = s.y, theta = s.z, theta_dot = s.w;
    const float force = (action_arr[env] > 0) ? force_mag : -force_mag;  // action > 0.5
    float sintheta, costheta;
    wd_np_sincosf(theta, sintheta, costheta);
    const float temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    const double den = (double)length *
                       (4.0 / 3.0 - (double)(masspole * (costheta * costheta) / total_mass));
    const double thetaacc = (double)(gravity * sintheta - costheta * temp) / den;
    const double xacc =
        (double)temp - (double)polemass_length * thetaacc * (double)costheta / (double)total_mass;
    x = x + tau * x_dot;
    x_dot = (float)((double)x_dot + (double)tau * xacc);
    theta = theta + tau * theta_dot;
    theta_dot = (float)((double)theta_dot + (double)tau * thetaacc);
    const float4 o = make_float4(x, x_dot, theta, theta_dot);
    state_arr[env] = o;
    observation_arr[env] = o;
    const bool terminated = x < -x_threshold || x > x_threshold ||
                            theta < -theta_threshold_radians || theta > theta_threshold_radians;
    reward_arr[env] = 1.0f;
    if (t == episode_length || terminated) done_arr[env] = 1;
  }
}
