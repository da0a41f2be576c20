"""warp_drive_amd -- MI355X-native rollout engine behind WarpDrive's
EnvWrapper / DataManager / FunctionManager API (salesforce/warp-drive).

Only the rollout hot path lives here: hand-written gfx950 kernels (csrc/kernels),
a C-ABI runtime (csrc/wd_runtime.cpp, include/wd_hip.h) and the host-side mirror
of the reference's manager interface (managers/, env_wrapper.py, envs/).
"""
__version__ = "0.1.0"
