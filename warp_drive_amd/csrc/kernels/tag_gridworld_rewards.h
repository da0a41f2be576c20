// tag_gridworld_rewards.h -- the rewards of a TagGridWorld tick, exactly as the reference's CPU step computes them.
//
// The reference adds `reward_tag + reward_penalty` in float64 (numpy arrays of Python floats,
// example_envs/tag_gridworld/tag_gridworld.py:163-187) and the result is narrowed to float32 when it reaches a
// device array (managers/data_manager.py:263-269).  Adding the float32-NARROWED scalars in float32 (what the
// reference's own CUDA kernel does, tag_gridworld_step_pycuda.cu:200-220) differs from that by one ulp for some
// configurations (-0.01 - 0.1, the shipped run config).  Here the four reward scalars arrive as float64 kernel
// arguments -- the env's Python floats, unnarrowed -- and the eight sums a tick can produce
// {tagger, runner} x {tagged, not} x {hit a wall, not} are formed in float64 and narrowed ONCE, at kernel start:
// bit for bit `float32(reward)` of the CPU step, at the cost of two selects per tick.
#pragma once

// (macros over eight plain `const float` locals, not a struct: a select between two FIELDS of a struct is rewritten
// by the compiler into a load through a selected ADDRESS, which pins the struct in scratch memory -- 36 bytes of
// private segment and a scratch load per tick in the single-wavefront rollout kernel)
#define GW_REWARD_TABLE(wall_hit_penalty, tag_reward_for_tagger, tag_penalty_for_runner, step_cost_for_tagger)          \
  /* reward_tag: tag_reward_for_tagger | -1.0 * step_cost_for_tagger | -1.0 * tag_penalty_for_runner |                  \
     1.0 * step_cost_for_tagger (:180-185); reward_penalty: -1.0 * wall_hit_penalty * hit (:163-170) */                 \
  const double gwr_pen = -1.0 * (wall_hit_penalty);                                                                     \
  const float gwr_tagger_tag = (float)(tag_reward_for_tagger), gwr_tagger_step = (float)(-1.0 * (step_cost_for_tagger)), \
              gwr_runner_tag = (float)(-1.0 * (tag_penalty_for_runner)),                                                 \
              gwr_runner_step = (float)(1.0 * (step_cost_for_tagger)),                                                   \
              gwr_tagger_tag_wall = (float)((tag_reward_for_tagger) + gwr_pen),                                          \
              gwr_tagger_step_wall = (float)(-1.0 * (step_cost_for_tagger) + gwr_pen),                                   \
              gwr_runner_tag_wall = (float)(-1.0 * (tag_penalty_for_runner) + gwr_pen),                                  \
              gwr_runner_step_wall = (float)(1.0 * (step_cost_for_tagger) + gwr_pen)

// the reward of one agent: `tagger` = it is one, `tag` = the runner was tagged on this tick, `hit` = it walked into a wall
#define GW_REWARD(tagger, tag, hit)                                                                                     \
  ((hit) ? ((tagger) ? ((tag) ? gwr_tagger_tag_wall : gwr_tagger_step_wall)                                             \
                     : ((tag) ? gwr_runner_tag_wall : gwr_runner_step_wall))                                            \
         : ((tagger) ? ((tag) ? gwr_tagger_tag : gwr_tagger_step) : ((tag) ? gwr_runner_tag : gwr_runner_step)))
