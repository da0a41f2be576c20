#!/usr/bin/env python3
"""One-off soak of the headline kernel's parity beyond what the GPU suite runs: all 2000 replicas of the fused tick against the
C oracle for a whole episode and a bit (every tick: actions draw for draw, state, rewards, done, observation rows,
nearest_neighbor_ids; tolerance 0 up to the counted, classified near-tie rows), with several sampler seeds; optionally another replica size (the big-replica entries).
    python scripts/soak_parity.py [ticks] [seeds] [runners] [replicas]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_tag_continuous as t  # noqa: E402  (test infrastructure: the oracle is the checker)

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 520
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
runners = int(sys.argv[3]) if len(sys.argv) > 3 else 100
E = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
kernel = (t.HEADLINE_TICK if runners == 100 else
          "HipTagContinuousTick_K10_N1024" if runners + 5 > 512 else "HipTagContinuousTick_K10_N512")
for seed in range(11, 11 + seeds):
    t0 = time.time()
    live, id_rows, id_pads, rows = t._fused_ticks_vs_c_oracle(dict(t.BENCH_CFG, num_runners=runners), E, ticks, seed, kernel=kernel)
    print(f"seed {seed}: 5 x {runners} agents, {ticks} ticks x {E} replicas: {rows} observation rows, {id_rows} id rows compared; rows whose "
          f"neighbour order differs from the oracle's (each one checked to be a <= 2-ulp near-tie: powf(x, 2) against x * x): "
          f"{t._fused_ticks_vs_c_oracle.last_near_tie_rows} "
          f"({time.time() - t0:.0f} s); agents in the game {live[0]:.0f} -> {min(live):.0f}", flush=True)
