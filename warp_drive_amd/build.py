"""Build the native pieces of warp_drive_amd in-tree.

  csrc/libwdhip.so       C-ABI runtime (include/wd_hip.h), host-only C++; binds
                         libamdhip64 at run time, so it is NOT linked against it.
  csrc/wd_kernels.hsaco  gfx950 code object with every kernel of the rollout path,
                         loaded through wd_module_load().
  csrc/wd_kernels_gw5.hsaco  shape-specialised kernels in their own code object (EXTRA_UNITS), loaded on
                         demand: the TagGridWorld T-tick rollout for 5 agents / full observations.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container;
the built files are git-ignored but travel to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
KDIR = os.path.join(CSRC, "kernels")
LIB = os.path.join(CSRC, "libwdhip.so")
HSACO = os.path.join(CSRC, "wd_kernels.hsaco")
# translation units that are NOT part of wd_kernels.hip: each becomes its own code object, so that adding or
# tuning a shape-specialised kernel leaves the main code object (and the counters collected on it) untouched
EXTRA_UNITS = {"tag_gridworld_n5.hip": os.path.join(CSRC, "wd_kernels_gw5.hsaco")}
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

# -ffp-contract=off + correctly rounded div/sqrt are part of the parity contract
# (see csrc/kernels/wd_common.h); do not change them without re-running the parity suite.
KERNEL_FLAGS = [
    # -Os: the rollout kernels are issue-bound straight-line code; the size-optimised schedule measured
    # 1.4 % faster than -O3 on the TagContinuous tick (38.25 -> 37.7 us; -O2 equal to -O3, -Oz 7 % slower;
    # again on the round-2 final kernel: -Os 34.1, -O2 34.5, -O3 34.6 us)
    "--offload-arch=gfx950", "--genco", "-Os", "-std=c++17", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # the fused tick kernels restore registered arrays through the reset table's untyped 32-bit
    # pointers, which alias the typed kernel arguments: no type-based alias analysis
    "-fno-strict-aliasing",
    # no SLP vectorisation: it pairs float32 add/mul into v_pk_add_f32 / v_pk_mul_f32, which issue at
    # less than half the rate of the plain VOP2 forms on gfx950 (experiments/ubench: 2.9 vs 1.2 cycles
    # per wave-instruction per SIMD); measured 43.2 -> 41.0 us per TagContinuous tick
    "-fno-slp-vectorize",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _hipcc():
    return shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")


def build_runtime(force=False, verbose=False):
    srcs = [os.path.join(CSRC, "wd_runtime.cpp"), os.path.join(ROOT, "include", "wd_hip.h")]
    if not force and _newer(LIB, srcs):
        return LIB
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-result",
           "-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include", f"-I{ROOT}/include", srcs[0],
           "-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_kernels(force=False, verbose=False, extra_flags=()):
    srcs = [os.path.join(KDIR, f) for f in sorted(os.listdir(KDIR)) if f.endswith((".hip", ".h")) and f not in EXTRA_UNITS]
    srcs.append(os.path.abspath(__file__))  # the compiler flags live here
    for unit, out in EXTRA_UNITS.items():
        unit_srcs = [os.path.join(KDIR, unit), os.path.join(KDIR, "wd_common.h"), os.path.abspath(__file__)]
        if force or not _newer(out, unit_srcs):
            cmd = [_hipcc(), *KERNEL_FLAGS, *extra_flags, unit_srcs[0], "-o", out]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    if not force and _newer(HSACO, srcs):
        return HSACO
    cmd = [_hipcc(), *KERNEL_FLAGS, *extra_flags, os.path.join(KDIR, "wd_kernels.hip"), "-o", HSACO]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HSACO


def build_kernels_locked(force=False, verbose=False):
    """build_kernels() under an exclusive file lock: safe to call from every rank of a node at once
    (ranks started by torch.distributed.run share no Event); the first caller builds, the others
    block on the lock and then find the code object fresh."""
    import fcntl

    if not force and os.path.exists(HSACO) and not os.access(CSRC, os.W_OK):
        return HSACO  # read-only install: use what is there
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build_kernels(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_all(force=False, verbose=False):
    return build_runtime(force, verbose), build_kernels(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
