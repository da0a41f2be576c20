#!/usr/bin/env python3
"""Per-wavefront phase timeline of the fused TagContinuous tick: the kernel source built with -DWD_TC_PROBES (shader-clock
stamps at the phase boundaries + counters of the search's fallbacks, written through a __device__ pointer the harness
sets).  Build the stamped code object here (hipcc cross-compiles), run on the GPU box:
    python experiments/phase_profile.py build
    python experiments/phase_profile.py [num_envs] [episode tick of the stamped launch] [runners per replica]
The stamped object replaces the product's through WD_HSACO_DIR (build/variants/prof/), everything else is the product."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# WD_PROF_DIR / WD_PROF_SRC: another variant directory / another kernel source directory (e.g. an earlier commit's
# csrc/kernels exported with `git archive`), for before / after profiles of one change
PROF_DIR = os.environ.get("WD_PROF_DIR", os.path.join(ROOT, "build", "variants", "prof"))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from warp_drive_amd import build as wb

    if os.environ.get("WD_PROF_SRC"):
        wb.KDIR = os.environ["WD_PROF_SRC"]
        wb.KERNEL_FLAGS = [f"-I{wb.KDIR}" if f.startswith("-I") else f for f in wb.KERNEL_FLAGS]

    os.makedirs(PROF_DIR, exist_ok=True)
    for out, (unit, flags) in wb.UNITS.items():
        if out.startswith("wd_kernels_tc_k10"):  # the headline entry and the runtime-size K = 10 entries (big replicas)
            subprocess.run([wb._hipcc(), *wb.KERNEL_FLAGS, *flags, "-DWD_TC_PROBES=1", os.path.join(wb.KDIR, unit), "-o",
                            os.path.join(PROF_DIR, out)], check=True)
            print("built", os.path.join(PROF_DIR, out))
    sys.exit(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T_STAMP = int(sys.argv[2]) if len(sys.argv) > 2 else 300  # (episodes are 500 ticks; the live-agent count falls along them)
N_RUNNERS = int(sys.argv[3]) if len(sys.argv) > 3 else 100  # 5 taggers + this many runners per replica
name = "prof"
os.environ["WD_HSACO_DIR"] = PROF_DIR
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import time

import numpy as np
import torch

import bench
from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.envs.tag_continuous import TagContinuous
from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.managers.function_manager import HIPSampler
from warp_drive_amd.rollout import RolloutEngine
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

w = EnvWrapper(env_obj=TagContinuous(**dict(bench.BENCH_CFG, num_runners=N_RUNNERS)), num_envs=E, env_backend="hip")
WPB = (w.n_agents + 63) // 64  # wavefronts per block (one replica per block beyond 64 agents)
w.reset_all_envs()
sampler = HIPSampler(w.cuda_function_manager)
sampler.init_random(seed=1)
create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                  push_data_batch_placeholders=False)
engine = RolloutEngine(w, sampler, fused=True)
SLOTS = 24
n_waves = WPB * E
buf = drv.mem_alloc(n_waves * SLOTS * 8)
drv.memset(buf, 0, n_waves * SLOTS * 8)
sym, nbytes = w.cuda_function_manager._module_of(engine.step_kernel_name).get_global("tc_prof_g")  # the stamped object
drv.memcpy_htod(sym, np.array([int(buf)], dtype=np.uint64))
engine.run(300)
torch.cuda.synchronize()
t0 = time.perf_counter()
engine.run(1000)
torch.cuda.synchronize()
print(f"=== {name}: wall per tick {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us (stamped build)")
engine.run((T_STAMP - 1300) % 500)
torch.cuda.synchronize()
live = w.cuda_data_manager.pull_data_from_device("still_in_the_game").sum(axis=1).mean()
print(f"stamped launch = tick {T_STAMP} of an episode, {live:.1f} agents in the game")
drv.memset(buf, 0, n_waves * SLOTS * 8)
torch.cuda.synchronize()
engine.run(1)
torch.cuda.synchronize()
raw = np.zeros(n_waves * SLOTS, dtype=np.uint64)
drv.memcpy_dtoh(raw, buf)
drv.synchronize()
st = raw.reshape(-1, SLOTS).astype(np.int64)
ran = st[:, 15] > 0
searched = st[:, 9] > 0
print(f"wavefronts: {ran.sum()}; ran the search: {searched.sum()}; with a lane in the two-pass fallback / whole-wavefront resolution: "
      f"{(st[:, 20] > 0).sum()}; with a lane whose close pair was settled by one exact compare: "
      f"{(st[:, 23] > 0).sum()}; with a lane in the full ranking: {(st[:, 22] > 0).sum()}")
pre = searched & (st[:, 19] > 0)   # (replicas of more than 128 agents: the prefiltered search; 1 = its radius held, 2 = repeated with the full chain)
if pre.any():
    tr = st[pre, 18]
    print(f"prefiltered wavefronts: {pre.sum()}; radius check failed: {(st[pre, 19] == 2).sum()}; chain trips (pass 2) mean {tr.mean():.1f} "
          f"p50 {np.percentile(tr, 50):.0f} p99 {np.percentile(tr, 99):.0f} max {tr.max()}")
# phases a wavefront skipped inherit the previous stamp (duration 0)
for k in range(1, 16):
    st[:, k] = np.where(st[:, k] > 0, st[:, k], st[:, k - 1])
names = ["start", "loads issued+tables", "sampled", "barrier1", "moved", "barrier2", "tags", "search: radius", "search: prefiltered chain",
         "search: full chain", "search: keys resolved", "search: ids, remember", "id rows + barrier + nearest ids flushed",
         "obs gathered+flushed", "barrier3", "rewards/end"]
if pre.any():  # big replicas: the wavefronts of a block differ (cell-sorted packing: border cells see fewer candidates)
    sp = st[pre]
    print(f"--- all {pre.sum()} prefiltered wavefronts: mean / p10 / p50 / p90 / max shader cycles")
    for k in (7, 8, 9, 10, 11):
        d = sp[:, k] - sp[:, k - 1]
        print(f"  {names[k]:<40} {d.mean():9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 50):9.0f} {np.percentile(d, 90):9.0f} {d.max():9.0f}")
    d = sp[:, 11] - sp[:, 6]
    print(f"  {'search, all of it':<40} {d.mean():9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 50):9.0f} {np.percentile(d, 90):9.0f} {d.max():9.0f}")
    blk = np.nonzero(pre)[0] // WPB
    slow = np.array([d[blk == b].max() for b in np.unique(blk)])
    print(f"  slowest searcher of each block: mean {slow.mean():.0f} p50 {np.percentile(slow, 50):.0f} p90 {np.percentile(slow, 90):.0f}")
for wv, label in ((0, "wave 0 of each block"), (WPB - 1, "the last wave of each block")):
    s = st[wv::WPB]
    ok = (s[:, 6] > 0) & (s[:, 15] > 0)
    s = s[ok]
    print(f"--- {label}: {ok.sum()} waves; mean / p10 / p90 shader cycles per phase")
    for k in range(1, 16):
        d = s[:, k] - s[:, k - 1]
        print(f"  {names[k]:<40} {d.mean():9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
    tot = s[:, 15] - s[:, 0]
    real = (s[:, 17] - s[:, 16]) * 10.0
    print(f"  total {tot.mean():.0f} cycles = {real.mean() / 1000:.2f} us -> {tot.mean() / real.mean():.3f} GHz")
s = st[(st[:, 16] > 0) & (st[:, 17] > 0)]
t0 = s[:, 16].min()
start, end = (s[:, 16] - t0) / 100.0, (s[:, 17] - t0) / 100.0
pc = lambda a: " ".join(f"{np.percentile(a, q):6.2f}" for q in (0, 10, 50, 90, 99, 100))
print("percentiles 0/10/50/90/99/100 (us): wave start", pc(start), "| wave end", pc(end), "| lifetime", pc(end - start))
# absolute timeline of the phase boundaries (us since the first wave started), wave 0 only
s0 = st[0::WPB]
s0 = s0[(s0[:, 6] > 0) & (s0[:, 15] > 0)]
ghz = ((s0[:, 15] - s0[:, 0]) / ((s0[:, 17] - s0[:, 16]) * 10.0)).mean()
base = (s0[:, 16] - t0) / 100.0
print("phase boundary, us since first wave start (p10 / p50 / p90), wave 0:")
for k in range(0, 16):
    tk = base + (s0[:, k] - s0[:, 0]) / (ghz * 1000.0)
    print(f"  {names[k]:<40} {np.percentile(tk, 10):7.2f} {np.percentile(tk, 50):7.2f} {np.percentile(tk, 90):7.2f}")

# ---- placement: which wavefronts share a SIMD (HW_ID / XCC_ID stamped at the start)
hw = st[:, 21]
okw = st[:, 15] > 0
simd_key = ((hw >> 32) & 15) * (1 << 20) + ((hw >> 13) & 7) * (1 << 16) + ((hw >> 12) & 1) * (1 << 12) + ((hw >> 8) & 15) * 16 + ((hw >> 4) & 3)
role = np.arange(len(st)) % WPB   # 0 = wave 0 of its block (the searcher when <= 64 agents are in the game)
endt = (st[:, 17] - t0) / 100.0
import collections
per = collections.defaultdict(list)
for i in np.nonzero(okw)[0]:
    per[int(simd_key[i])].append(i)
n_on = np.array([len(v) for v in per.values()])
n_w0 = np.array([int((role[v] == 0).sum()) for v in per.values()])
last = np.array([endt[v].max() for v in per.values()])
print(f"SIMDs in use: {len(per)}; wavefronts per SIMD: " + " ".join(f"{k}:{(n_on == k).sum()}" for k in sorted(set(n_on))))
print("wave-0 roles per SIMD (all its wavefronts counted): " + " ".join(f"{k}:{(n_w0 == k).sum()}" for k in sorted(set(n_w0))))
for k in sorted(set(n_w0)):
    m = n_w0 == k
    print(f"  SIMDs with {k} wave-0 roles ({int(n_on[m].mean() * 100) / 100} wavefronts): last wavefront ends at {last[m].mean():6.2f} us (p90 {np.percentile(last[m], 90):6.2f}, max {last[m].max():6.2f})")
cu_key = simd_key // 4
percu = collections.defaultdict(list)
for i in np.nonzero(okw)[0]:
    percu[int(cu_key[i])].append(i)
ncu = np.array([len(v) for v in percu.values()])
lastcu = np.array([endt[v].max() for v in percu.values()])
print(f"CUs in use: {len(percu)}; wavefronts per CU: " + " ".join(f"{k}:{(ncu == k).sum()}" for k in sorted(set(ncu))))
for k in sorted(set(ncu)):
    print(f"  CUs with {k} wavefronts: last wavefront ends at {lastcu[ncu == k].mean():6.2f} us (max {lastcu[ncu == k].max():6.2f})")

# ---- the tail: the launch ends with its last wavefront.  Which wavefronts are last, and where did they lose time?
order = np.argsort(-endt * okw)
med = np.median(np.diff(st[okw][:, 0:16], axis=1), axis=0)
print("latest wavefronts: end us | start us | block wave | xcc se cu simd slot | phase durations minus the median of all wavefronts (cycles), flags")
short = ["loads", "sample", "bar1", "move", "bar2", "tags", "bound", "pass1", "chain", "resolve", "ids", "idrows", "obs", "bar3", "end"]
for i in order[:24]:
    d = np.diff(st[i, 0:16]) - med
    big = " ".join(f"{short[k]}{int(d[k]):+d}" for k in np.argsort(-np.abs(d))[:5])
    h = int(hw[i])
    print(f"  {endt[i]:6.2f} | {(st[i, 16] - t0) / 100.0:5.2f} | {i // WPB:5d} {i % WPB} | {(h >> 32) & 15} {(h >> 13) & 7} {(h >> 8) & 15:2d} {(h >> 4) & 3} {h & 15:2d} | {big} | {int(st[i, 18]):x} {int(st[i, 20])}")
print("end time by XCD (mean / max):", " ".join(f"{x}:{endt[okw & (((hw >> 32) & 15) == x)].mean():.1f}/{endt[okw & (((hw >> 32) & 15) == x)].max():.1f}" for x in range(8)))
