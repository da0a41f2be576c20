// wd_kernels.hip -- unity translation unit of the MAIN rollout code object (wd_kernels.hsaco): the
// core services (sampler, reset, logger) and the small environments.  Plays the role of the
// reference's generated env_runner.cu + core_service.h (warp_drive/cuda_includes/
// template_env_runner.cu:7-10, core_service.h:10-15) but is compiled ONCE, offline, for gfx950:
// sizes are runtime kernel arguments, so there is no per-run source templating or JIT.
// The TagContinuous kernels, the trainer's policy forward and the shape-specialised kernels live in
// their own code objects (warp_drive_amd/build.py UNITS), loaded on demand through the manifest.
#include "wd_core.hip"
#include "dummy_env.hip"
#include "tag_gridworld.hip"
#include "cartpole.hip"
