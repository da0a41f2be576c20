// wd_core.hip -- core service kernels: action sampler (categorical + OU/Gaussian),
// reset-when-done (per array and fused), undo-done, episode logger.
//
// Replaces reference warp_drive/cuda_includes/core/{random,reset,log}.cu and
// warp_drive/numba_includes/core/{random,reset,pool_reset,log}.py.  Differences that
// matter on MI355X:
//   * sizes are runtime kernel arguments (the reference bakes wkNumberEnvs /
//     wkNumberAgents in at compile time, template_env_config.h:19-21), so one code
//     object serves every configuration;
//   * every kernel is geometry-agnostic: it grid-strides over env replicas and
//     block-strides inside a replica, so both the reference launch shape
//     (block=(agents,1,1), grid=(envs,1), function_manager.py:64-67) and wave64-
//     friendly shapes work;
//   * the sampler stages probability rows through LDS with coalesced loads and keeps
//     the running sum in a register (the reference streams an uncoalesced row per
//     thread and round-trips the prefix sum through global `cum_distr`,
//     random.cu:74-83);
//   * the RNG is counter-based Philox4x32-10: 4 bytes of state traffic per draw
//     instead of curand XORWOW's 48-byte heap-allocated state (random.cu:14-23).
#include "wd_common.h"

extern "C" {

// ------------------------------------------------------------------------------ RNG
// replaces init_random (random.cu:14-23 / numba random.py:29-30).  `state` holds
// WD_RNG_HEADER + n_threads uint32 words.
__global__ void init_random(uint32_t *state, int seed, int n_threads) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) {
    state[0] = (uint32_t)seed;
    state[1] = 0x5bd1e995u;
    state[2] = (uint32_t)n_threads;
    state[3] = 0u;
  }
  for (int i = tid; i < n_threads; i += gridDim.x * blockDim.x) state[WD_RNG_HEADER + i] = 0u;
}

// free_random (random.cu:25-31) has nothing to release: the state is one wd_malloc'ed
// block owned by the host-side sampler.  Kept so the default function list resolves.
__global__ void free_random() {}

// ---------------------------------------------------------------------- categorical
// replaces sample_actions + search_index (random.cu:33-85).  One thread per
// (env, agent) row of `distr` [n_rows, num_actions] (row-major, contiguous).
//   use_argmax : first maximum (strict '<' scan, random.cu:59-66)
//   otherwise  : u ~ U(0,1]; running float32 sum in index order; result = number of
//                prefix sums < u, clamped to num_actions-1 -- the index search_index
//                converges to (its |cum-u|<1e-8 early exit only fires on exact
//                float ties, a measure-zero event).
// `cum_distr` is accepted for signature compatibility and never touched.
// The result goes to action_indices[row * out_stride + out_offset]: (1, 0) is the
// reference's [E, n, 1] per-head array; (n_heads, k) writes head k straight into the
// combined [E, n, n_heads] action tensor, replacing the strided torch copy of
// trainer_base.py:506-512.
// Dynamic LDS: blockDim.x * lds_stride floats (lds_stride odd => conflict-free rows) PLUS 128 bytes:
// rows of at most WD_SLAB_CH (24) entries are read back 24 entries at a time from the row start
// (wd_slab_sample; the entries past the row length are masked off), so the last row's read runs up to
// 92 bytes past the slab.  HIPSampler.categorical_launch adds the padding; any other launcher (C-ABI
// users) must do the same.
__global__ void sample_actions(uint32_t *rng_state, const float *__restrict__ distr,
                               int *__restrict__ action_indices, float *cum_distr, int n_rows,
                               int num_actions, int use_argmax, int lds_stride, int stream_tag,
                               int out_stride, int out_offset) {
  extern __shared__ __attribute__((aligned(16))) float s_rows[];
  (void)cum_distr;
  const int rows_per_block = blockDim.x;
  const uint32_t seed_lo = rng_state[0], seed_hi = rng_state[1];
  if (lds_stride == num_actions && (blockDim.x & 63) == 0 && use_argmax <= 0) {
    // Unpadded rows (the host asks for them when num_actions is odd: a stride of n dwords is then
    // conflict-free as it is): every wavefront moves the contiguous rows of its own 64 threads
    // global -> LDS with global_load_lds (1 KiB per instruction, no staging registers, no index
    // arithmetic) and samples them after its own vmcnt wait -- no block barrier at all.
    const int wave0 = (int)threadIdx.x & ~63, lane = (int)threadIdx.x & 63;
    for (long row0 = (long)blockIdx.x * rows_per_block; row0 < n_rows;
         row0 += (long)gridDim.x * rows_per_block) {
      const long wrow0 = row0 + wave0;
      const int wrows = (int)max(0L, min(64L, (long)n_rows - wrow0));
      float *slab = s_rows + (size_t)wave0 * num_actions;
      wd_slab_fetch(slab, distr + wrow0 * num_actions, wrows * num_actions, lane);
      const long row = wrow0 + lane;
      uint32_t epoch = 0u;
      wd_u4 rnd = wd_u4{0u, 0u, 0u, 0u};
      if (lane < wrows) {
        epoch = rng_state[WD_RNG_HEADER + row];
        rng_state[WD_RNG_HEADER + row] = epoch + 1u;
        rnd = wd_philox4x32_10(wd_u4{(uint32_t)row, epoch, (uint32_t)stream_tag, 0u}, seed_lo, seed_hi);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane < wrows)
        action_indices[row * out_stride + out_offset] =
            wd_slab_sample(slab + (size_t)lane * num_actions, num_actions, wd_u01_open_closed(rnd.x));
      __builtin_amdgcn_wave_barrier();  // (the next trip's fetch overwrites the slab)
    }
    return;
  }
  for (long row0 = (long)blockIdx.x * rows_per_block; row0 < n_rows;
       row0 += (long)gridDim.x * rows_per_block) {
    const int rows_here = min((long)rows_per_block, (long)n_rows - row0);
    const long base = row0 * num_actions;
    const int total = rows_here * num_actions;
    // coalesced slab load -> LDS (row stride padded).  (row, column) of element i advance
    // incrementally with the block stride: no integer division per element
    {
      const int step_r = (int)blockDim.x / num_actions, step_c = (int)blockDim.x - step_r * num_actions;
      int r = (int)threadIdx.x / num_actions, c = (int)threadIdx.x - r * num_actions;
      constexpr int U = 4;  // loads in flight per thread
      for (int i0 = threadIdx.x; i0 < total; i0 += U * blockDim.x) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = distr[base + min(i0 + u * (int)blockDim.x, total - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i0 + u * (int)blockDim.x < total) s_rows[r * lds_stride + c] = v[u];
          r += step_r;
          c += step_c;
          if (c >= num_actions) { c -= num_actions; ++r; }
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < rows_here) {
      const long row = row0 + threadIdx.x;
      const float *p = s_rows + threadIdx.x * lds_stride;
      int result;
      if (use_argmax > 0) {
        float best = p[0];
        result = 0;
        for (int i = 1; i < num_actions; ++i) {
          const float v = p[i];
          if (best < v) { best = v; result = i; }
        }
      } else {
        const uint32_t epoch = rng_state[WD_RNG_HEADER + row];
        rng_state[WD_RNG_HEADER + row] = epoch + 1u;
        const wd_u4 rnd = wd_philox4x32_10(wd_u4{(uint32_t)row, epoch, (uint32_t)stream_tag, 0u},
                                           seed_lo, seed_hi);
        const float u = wd_u01_open_closed(rnd.x);
        float cum = 0.0f;
        int cnt = 0;
        for (int i = 0; i < num_actions; ++i) {
          cum = (i == 0) ? p[0] : cum + p[i];
          cnt += (cum < u) ? 1 : 0;
        }
        result = min(cnt, num_actions - 1);
      }
      action_indices[row * out_stride + out_offset] = result;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------- OU / Gaussian
// replaces sample_ou_process (numba random.py:66-105): ou = (1-damping)*ou +
// stddev*N(0,1); action = distr + scale*ou; scale < 1e-8 passes distr through.
__global__ void sample_ou_process(uint32_t *rng_state, const float *__restrict__ distr,
                                  float *__restrict__ actions, float *__restrict__ ou_states,
                                  float damping, float stddev, float scale, int n_rows,
                                  int stream_tag) {
  const uint32_t seed_lo = rng_state[0], seed_hi = rng_state[1];
  for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
       row += (long)gridDim.x * blockDim.x) {
    if (scale < 1.0e-8f) {
      actions[row] = distr[row];
      continue;
    }
    const uint32_t epoch = rng_state[WD_RNG_HEADER + row];
    rng_state[WD_RNG_HEADER + row] = epoch + 1u;
    const wd_u4 rnd = wd_philox4x32_10(wd_u4{(uint32_t)row, epoch, (uint32_t)stream_tag, 1u},
                                       seed_lo, seed_hi);
    // Box-Muller on two (0,1] uniforms
    const float u1 = wd_u01_open_closed(rnd.x), u2 = wd_u01_open_closed(rnd.y);
    const float normal = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    const float ou = (1.0f - damping) * ou_states[row] + stddev * normal;
    ou_states[row] = ou;
    actions[row] = distr[row] + scale * ou;
  }
}

// -------------------------------------------------------------------------- reset
// One replica's slice of an array is contiguous ([E, ...] row-major), so every
// reset flavour of the reference (reset.cu:9-63, numba reset.py:7-36) is the same
// masked 4-byte copy of `row_elems` words; it is done with block-strided, coalesced
// accesses instead of one thread copying `feature_dim` strided words (reset.cu:33-39).
__device__ __forceinline__ void wd_reset_rows(uint32_t *__restrict__ data,
                                              const uint32_t *__restrict__ ref,
                                              const int *__restrict__ done, int row_elems,
                                              int n_envs, int force_reset) {
  for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
    if (force_reset > 0 || done[env] > 0) {
      const long base = (long)env * row_elems;
      for (int i = threadIdx.x; i < row_elems; i += blockDim.x) data[base + i] = ref[base + i];
    }
  }
}

__global__ void reset_in_float_when_done_2d(float *data, const float *ref, const int *done,
                                            int feature_dim, int force_reset, int n_envs) {
  wd_reset_rows((uint32_t *)data, (const uint32_t *)ref, done, feature_dim, n_envs, force_reset);
}
__global__ void reset_in_int_when_done_2d(int *data, const int *ref, const int *done,
                                          int feature_dim, int force_reset, int n_envs) {
  wd_reset_rows((uint32_t *)data, (const uint32_t *)ref, done, feature_dim, n_envs, force_reset);
}
__global__ void reset_in_float_when_done_3d(float *data, const float *ref, const int *done,
                                            int agent_dim, int feature_dim, int force_reset,
                                            int n_envs) {
  wd_reset_rows((uint32_t *)data, (const uint32_t *)ref, done, agent_dim * feature_dim, n_envs,
                force_reset);
}
__global__ void reset_in_int_when_done_3d(int *data, const int *ref, const int *done,
                                          int agent_dim, int feature_dim, int force_reset,
                                          int n_envs) {
  wd_reset_rows((uint32_t *)data, (const uint32_t *)ref, done, agent_dim * feature_dim, n_envs,
                force_reset);
}

// undo_done_flag_and_reset_timestep (reset.cu:65-75)
__global__ void undo_done_flag_and_reset_timestep(int *done, int *timestep, int force_reset,
                                                  int n_envs) {
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < n_envs;
       env += gridDim.x * blockDim.x) {
    if (force_reset > 0 || done[env] > 0) {
      done[env] = 0;
      timestep[env] = 0;
    }
  }
}

// Fused reset: ONE launch restores every registered array of every finished replica
// and (optionally) clears done/timestep -- the reference issues one launch per array
// plus one for undo, 13 for TagContinuous (pycuda_function_manager.py:686-753).
// `table` = n_arrays entries of {data, ref, row_elems}.
struct wd_reset_entry {  // (global pointers: wd_common.h, wd_global_u32)
  wd_global_u32 *data;
  const wd_global_u32 *ref;
  int row_elems;
  int pad_;
};

__global__ void reset_when_done_fused(const wd_reset_entry *__restrict__ table, int n_arrays,
                                      int *done, int *timestep, int force_reset, int undo,
                                      int n_envs) {
  for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
    const bool hit = force_reset > 0 || done[env] > 0;  // block-uniform
    if (hit) {
      for (int a = 0; a < n_arrays; ++a) {
        const wd_reset_entry e = table[a];
        const long base = (long)env * e.row_elems;
        for (int i = threadIdx.x; i < e.row_elems; i += blockDim.x) e.data[base + i] = e.ref[base + i];
      }
    }
    __syncthreads();  // every thread has read done[env] before it is cleared
    if (hit && undo > 0 && threadIdx.x == 0) {
      done[env] = 0;
      timestep[env] = 0;
    }
  }
}

// Reset from a pool of candidate starts (numba pool_reset.py:9-52): a finished
// replica copies a uniformly drawn row of `pool` [n_pool, row_elems].  All arrays
// that share a launch epoch draw the same row index for a replica, because the
// index depends only on (seed, env, epoch) and the epoch is advanced by the caller
// once per reset call through `advance`.
__global__ void reset_when_done_from_pool(uint32_t *rng_state, uint32_t *data,
                                          const uint32_t *__restrict__ pool,
                                          const int *__restrict__ done, int row_elems, int n_pool,
                                          int force_reset, int n_envs, int advance) {
  const uint32_t seed_lo = rng_state[0], seed_hi = rng_state[1];
  for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
    const bool hit = force_reset > 0 || done[env] > 0;
    const uint32_t epoch = rng_state[WD_RNG_HEADER + env];
    if (hit) {
      const wd_u4 rnd = wd_philox4x32_10(wd_u4{(uint32_t)env, epoch, 0x706f6f6cu, 2u}, seed_lo, seed_hi);
      // p in [0,1): int(p * n_pool)  (pool_reset.py:20-22)
      const float p = (float)(rnd.x >> 8) * 0x1.0p-24f;
      int ref_id = (int)(p * (float)n_pool);
      ref_id = min(ref_id, n_pool - 1);
      const long src = (long)ref_id * row_elems, dst = (long)env * row_elems;
      for (int i = threadIdx.x; i < row_elems; i += blockDim.x) data[dst + i] = pool[src + i];
    }
    __syncthreads();
    if (hit && advance > 0 && threadIdx.x == 0) rng_state[WD_RNG_HEADER + env] = epoch + 1u;
  }
}

// ---------------------------------------------------------------------------- log
// Episode logger (log.cu:11-62): copies one replica's [n_agents, feature_dim] slice
// into row `timestep` of a [T+1, n_agents, feature_dim] buffer.
__global__ void reset_log_mask(int *log_mask, int episode_length) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= episode_length;
       i += gridDim.x * blockDim.x)
    log_mask[i] = 0;
}
__global__ void update_log_mask(int *log_mask, int timestep, int episode_length) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && timestep <= episode_length) log_mask[timestep] = 1;
}
__device__ __forceinline__ void wd_log_rows(uint32_t *log, const uint32_t *data, int row_elems,
                                            int timestep, int episode_length, int env_id) {
  if (timestep > episode_length) return;
  const long dst = (long)timestep * row_elems, src = (long)env_id * row_elems;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_elems; i += gridDim.x * blockDim.x)
    log[dst + i] = data[src + i];
}
__global__ void log_one_step_in_float(float *log, const float *data, int feature_dim, int timestep,
                                      int episode_length, int env_id, int n_agents) {
  wd_log_rows((uint32_t *)log, (const uint32_t *)data, n_agents * feature_dim, timestep,
              episode_length, env_id);
}
__global__ void log_one_step_in_int(int *log, const int *data, int feature_dim, int timestep,
                                    int episode_length, int env_id, int n_agents) {
  wd_log_rows((uint32_t *)log, (const uint32_t *)data, n_agents * feature_dim, timestep,
              episode_length, env_id);
}

}  // extern "C"
