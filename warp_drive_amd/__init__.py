"""warp_drive_amd -- MI355X-native rollout engine behind WarpDrive's
EnvWrapper / DataManager / FunctionManager API (salesforce/warp-drive).

Only the rollout hot path lives here: hand-written gfx950 kernels (csrc/kernels),
a C-ABI runtime (csrc/wd_runtime.cpp, include/wd_hip.h) and the host-side mirror
of the reference's manager interface (managers/, env_wrapper.py, envs/).
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in device memory (ROCm reads them from host-coherent memory otherwise: +2-4 us per
# launch on MI355X, i.e. +8 % on a 50 us rollout tick and +35 % on a TagGridWorld tick).  Only effective
# when this package is imported before the HIP runtime initialises (i.e. before `import torch`
# touches the GPU); an explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
