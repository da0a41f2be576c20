// tc_config.h -- build-unit switches of the TagContinuous translation unit: the block size as a compile-time constant in the
// shape-specialised unit, and the phase probes (compiled out of the product build).
#pragma once
#include "wd_common.h"

// threads per block: a launch-time value, except in the unit that is built for ONE shape (its host geometry is fixed:
// envs/tag_continuous.py::_geometry), where the number of wavefronts, the replicas per block and every loop over them fold
#if defined(WD_TC_SHAPE_THREADS)
#define WD_TC_BLOCKDIM WD_TC_SHAPE_THREADS
#else
#define WD_TC_BLOCKDIM ((int)blockDim.x)
#endif

// Phase probes (experiments/phase_profile.py): compiled out of the product build.  With -DWD_TC_PROBES (variant
// "prof" of experiments/variant_sets.py) lane 0 of every wavefront of the fast path stamps the shader clock at
// the phase boundaries into 24 slots per wavefront behind a __device__ pointer the harness sets.
#ifdef WD_TC_PROBES
extern "C" { __device__ unsigned long long *tc_prof_g = nullptr; }
#define WD_TC_SLOT(k) ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 24 + (k))
#define WD_TC_PROBE(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) tc_prof_g[WD_TC_SLOT(k)] = __builtin_readcyclecounter(); } while (0)
#define WD_TC_PROBE_RT(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) tc_prof_g[WD_TC_SLOT(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// a counter of the wavefront (callable inside divergent code: the first active lane adds)
#define WD_TC_PROBE_VAL(k, v) do { if (tc_prof_g) { const unsigned long long m_ = __ballot(1);                          \
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m_) - 1)) tc_prof_g[WD_TC_SLOT(k)] += (unsigned long long)(v); } } while (0)
// where the wavefront runs: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) | XCC_ID << 32
#define WD_TC_PROBE_HW(k) do { if ((threadIdx.x & 63) == 0 && tc_prof_g) { unsigned hw_, xcc_;                        \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                  \
    tc_prof_g[WD_TC_SLOT(k)] = (unsigned long long)hw_ | ((unsigned long long)(xcc_ & 15u) << 32); } } while (0)
#else
#define WD_TC_PROBE_HW(k)
#define WD_TC_PROBE(k)
#define WD_TC_PROBE_RT(k)
#define WD_TC_PROBE_VAL(k, v)
#endif
