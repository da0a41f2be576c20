"""Build the native pieces of warp_drive_amd in-tree.

  csrc/libwdhip.so            C-ABI runtime (include/wd_hip.h), host-only C++; binds libamdhip64 at run time, so it is
                              NOT linked against it.
  csrc/wd_kernels.hsaco       MAIN gfx950 code object: core services (sampler, reset, logger) + TagGridWorld + Cartpole,
                              loaded through wd_module_load().
  csrc/wd_kernels_*.hsaco     everything else, one code object per UNIT below, loaded on demand: the TagContinuous
                              generic entries, one object per K specialisation of its fast path, the BASELINE shape with
                              its sizes folded, the trainer's policy kernels, the TagGridWorld 5-agent rollout, the
                              test-only kernels.  Units build in parallel (one hipcc per core) and a change to one
                              specialisation rebuilds one small object.
  csrc/wd_kernels.manifest.json   kernel name -> code object, written after every build (the host asks it which object
                              to load for a function: managers/hip_driver.py)

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the built files are git-ignored but
travel to the GPU box with the tree.
"""
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
KDIR = os.path.join(CSRC, "kernels")
LIB = os.path.join(CSRC, "libwdhip.so")
HSACO = os.path.join(CSRC, "wd_kernels.hsaco")
MANIFEST = os.path.join(CSRC, "wd_kernels.manifest.json")


def _tc_unit(k, waves, big):
    return ("tag_continuous.hip", [f"-DWD_TC_KM={k}", f"-DWD_TC_WAVES={waves}"] + (["-DWD_TC_BIG=1"] if big else []))


# code object (under csrc/) -> (translation unit under csrc/kernels/, extra compiler flags)
UNITS = {
    "wd_kernels.hsaco": ("wd_kernels.hip", []),
    "wd_kernels_tc.hsaco": ("tag_continuous.hip", []),
    # fast TagContinuous entries: the smallest K specialisation >= num_other_agents_observed is used; occupancy per SIMD
    # (launch bound) falls with K; 513 .. 1024 agents (WD_TC_BIG) for K <= 16
    **{f"wd_kernels_tc_k{k}.hsaco": _tc_unit(k, w, big)
       for k, w, big in ((2, 4, False), (4, 4, True), (6, 4, False), (8, 4, True), (10, 4, True), (12, 3, True),
                         (16, 3, True), (24, 2, False), (32, 2, False))},
    # the BASELINE shape (5 taggers + 100 runners, K = 10, 21-way heads) with its sizes as compile-time constants
    # (and its block size: 105 agents = one replica per 128-thread block, envs/tag_continuous.py::_geometry)
    "wd_kernels_tc_k10_n105a21.hsaco": ("tag_continuous.hip", ["-DWD_TC_KM=10", "-DWD_TC_SHAPE_N=105",
                                                                 "-DWD_TC_SHAPE_A=21", "-DWD_TC_SHAPE_THREADS=128"]),
    # the trainer's kernels, one source, two objects compiled side by side: the rollout's (policy forward, record) and the
    # update's (returns, objective, backward passes)
    "wd_kernels_mlp.hsaco": ("policy_mlp.hip", ["-DWD_MLP_PART=1"]),
    "wd_kernels_update.hsaco": ("policy_mlp.hip", ["-DWD_MLP_PART=2"]),
    "wd_kernels_gw5.hsaco": ("tag_gridworld_n5.hip", []),
    "wd_kernels_test.hsaco": ("wd_test_kernels.hip", []),
}
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

# -ffp-contract=off + correctly rounded div/sqrt are part of the parity contract
# (see csrc/kernels/wd_common.h); do not change them without re-running the parity suite.
KERNEL_FLAGS = [
    # -Os: the rollout kernels are issue-bound straight-line code; the size-optimised schedule measured
    # 1.4 % faster than -O3 on the TagContinuous tick (38.25 -> 37.7 us; -O2 equal to -O3, -Oz 7 % slower;
    # again on the round-2 final kernel: -Os 34.1, -O2 34.5, -O3 34.6 us)
    "--offload-arch=gfx950", "--genco", "-Os", "-std=c++17", "-ffp-contract=off", f"-I{KDIR}",
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


def _cuid(code_object_name):
    """hipcc derives a "compilation unit id" from the paths of the source and of the OUTPUT file and bakes it into the
    code object: the same source built into another directory -- or through a temporary file name -- is a different
    file, and the PMC records under profiles/ (keyed by the object's sha256) would go stale with every rebuild.  An
    explicit id makes the object a function of the source and the flags alone."""
    return [f"-cuid={os.path.splitext(code_object_name)[0]}"]


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


_INCLUDE = re.compile(r'^\s*#include\s+"([^"]+)"', re.M)


def unit_sources(unit):
    """the translation unit and every local file it includes (transitively)"""
    todo, seen = [os.path.join(KDIR, unit)], []
    while todo:
        path = todo.pop()
        if path in seen or not os.path.exists(path):
            continue
        seen.append(path)
        todo += [os.path.join(KDIR, inc) for inc in _INCLUDE.findall(open(path).read())]
    return seen


def kernels_in(hsaco):
    """names of the kernels of a code object (their `<name>.kd` descriptor symbols in its string table)"""
    blob = open(hsaco, "rb").read()
    return sorted({m.decode() for m in re.findall(rb"\x00([A-Za-z_][A-Za-z0-9_]*)\.kd\x00", blob)})


SHAPE_UNIT_PATTERN = re.compile(r"^wd_kernels_tc_k(\d+)_n(\d+)a(\d+)t(\d+)\.hsaco$")


def shape_unit_name(k, n_agents, n_actions, threads):
    return f"wd_kernels_tc_k{int(k)}_n{int(n_agents)}a{int(n_actions)}t{int(threads)}.hsaco"


def shape_units_on_disk():
    """shape-specialised TagContinuous objects built on demand (build_shape_unit): {file name: threads per block}"""
    found = {}
    for f in sorted(os.listdir(CSRC)):
        m = SHAPE_UNIT_PATTERN.match(f)
        if m:
            found[f] = int(m.group(4))
    return found


def build_shape_unit(k, n_agents, n_actions, threads, verbose=False):
    """Build (once, under the build lock) the TagContinuous entries for ONE shape with its sizes and block size as
    compile-time constants -- what the reference does for every run by templating wkNumberAgents into the source it
    hands to nvcc (pycuda_function_manager.py:133-232, template_env_config.h:19-21); here it is opt-in
    (envs/tag_continuous.py: WD_TC_JIT_SHAPES=1), takes one ~15 s hipcc of one unit, and the result is reused by every later
    run.  Returns the code object's file name; the manifest is rewritten to include its kernels."""
    import fcntl

    assert 1 <= int(n_agents) <= 128 and int(threads) % 64 == 0
    out = shape_unit_name(k, n_agents, n_actions, threads)
    target = os.path.join(CSRC, out)
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            srcs = unit_sources("tag_continuous.hip") + [os.path.abspath(__file__)]
            if not _newer(target, srcs):
                tmp = target + f".{os.getpid()}.tmp"
                cmd = [_hipcc(), *KERNEL_FLAGS, *_cuid(out), f"-DWD_TC_KM={int(k)}", f"-DWD_TC_SHAPE_N={int(n_agents)}",
                       f"-DWD_TC_SHAPE_A={int(n_actions)}", f"-DWD_TC_SHAPE_THREADS={int(threads)}",
                       os.path.join(KDIR, "tag_continuous.hip"), "-o", tmp]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                os.replace(tmp, target)
                write_manifest()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def write_manifest():
    manifest = {}
    for out in list(UNITS) + list(shape_units_on_disk()):
        path = os.path.join(CSRC, out)
        if os.path.exists(path):
            for name in kernels_in(path):
                manifest.setdefault(name, out)
    tmp = MANIFEST + f".{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    os.replace(tmp, MANIFEST)
    return manifest


def build_kernels(force=False, verbose=False, extra_flags=(), only=None):
    """(re)build every stale code object of UNITS (`only`: a subset of its keys), in parallel; returns the main one"""
    jobs = []
    for out, (unit, flags) in UNITS.items():
        if only is not None and out not in only:
            continue
        target = os.path.join(CSRC, out)
        if force or not _newer(target, unit_sources(unit) + [os.path.abspath(__file__)]):
            # compile next to the target and rename: a rank that only reads never sees a half-written object
            tmp = target + f".{os.getpid()}.tmp"
            jobs.append((target, tmp, [_hipcc(), *KERNEL_FLAGS, *_cuid(out), *flags, *extra_flags, os.path.join(KDIR, unit),
                                       "-o", tmp]))

    def run(job):
        target, tmp, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, target)

    if jobs:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
            list(pool.map(run, jobs))
    if jobs or not os.path.exists(MANIFEST):
        write_manifest()
    return HSACO


def build_kernels_locked(force=False, verbose=False):
    """build_kernels() under an exclusive file lock: safe to call from every rank of a node at once
    (ranks started by torch.distributed.run share no Event); the first caller builds, the others
    block on the lock and then find the code objects fresh."""
    import fcntl

    if not force and os.path.exists(HSACO) and os.path.exists(MANIFEST) and not os.access(CSRC, os.W_OK):
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
