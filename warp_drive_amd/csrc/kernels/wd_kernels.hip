// wd_kernels.hip -- unity translation unit for the rollout code object
// (wd_kernels.hsaco).  Plays the role of the reference's generated env_runner.cu +
// core_service.h (warp_drive/cuda_includes/template_env_runner.cu:7-10,
// core_service.h:10-15) but is compiled ONCE, offline, for gfx950: sizes are runtime
// kernel arguments, so there is no per-run source templating or JIT.
#include "wd_core.hip"
#include "dummy_env.hip"
#include "tag_gridworld.hip"
#include "tag_continuous.hip"
#include "cartpole.hip"
#include "policy_mlp.hip"
