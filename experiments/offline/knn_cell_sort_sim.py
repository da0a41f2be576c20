#!/usr/bin/env python3
"""Offline sizing of the NEXT design for replicas of hundreds of agents: sort the agents in the game by grid cell every
tick (a counting sort in LDS) and let a wavefront of 64 cell-neighbours run the key chain over the UNION of their 3 x 3
cell neighbourhoods only -- no hint from the previous tick, no per-candidate compare pass, broadcast reads as now.
Exactness as for the prefilter: the K-th other agent found must lie inside what the neighbourhood provably covers
(the distance from the searcher to the border of its 3 x 3 block), else the wavefront repeats with the full chain.

    python experiments/offline/knn_cell_sort_sim.py [runners] [ticks]
Prints, per cell size: candidates per wavefront (the full chain has n), searchers whose check fails."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tag_continuous_c import TagContinuousCOracle  # noqa: E402

RUNNERS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K, E, L = 10, 2, 20.0
cfg = dict(num_taggers=5, num_runners=RUNNERS, grid_length=L, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880, max_speed=1.0,
           skill_level_runner=1.0, skill_level_tagger=1.0)

o = TagContinuousCOracle(E, n_threads=8, **cfg)
rng = np.random.RandomState(1)
stats = {}
for t in range(TICKS):
    act = np.stack([rng.randint(0, 21, (E, o.N)), rng.randint(0, 21, (E, o.N))], -1).astype(np.int32)
    o.step(act)
    if t % 5:
        continue
    for e in range(E):
        live = np.nonzero(o.sig_before[e] > 0)[0]
        n = len(live)
        x, y = o.loc_x[e][live].astype(np.float64), o.loc_y[e][live].astype(np.float64)
        d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2
        np.fill_diagonal(d2, np.inf)
        kth = np.sqrt(np.sort(d2, axis=1)[:, K - 1])          # distance to the K-th other agent
        for cells in (4, 5, 6, 8, 10):
            c = L / cells
            cx, cy = np.minimum((x / c).astype(int), cells - 1), np.minimum((y / c).astype(int), cells - 1)
            order = np.lexsort((cx, cy))                        # row-major cell order
            # what the 3 x 3 block around a searcher's cell covers for sure: the distance to the block's border
            # (the arena's own border does not count: nobody stands beyond it)
            lo_x, hi_x = (cx - 1) * c, (cx + 2) * c
            lo_y, hi_y = (cy - 1) * c, (cy + 2) * c
            cover = np.minimum.reduce([np.where(cx > 0, x - lo_x, np.inf), np.where(cx < cells - 1, hi_x - x, np.inf),
                                       np.where(cy > 0, y - lo_y, np.inf), np.where(cy < cells - 1, hi_y - y, np.inf)])
            fail = kth > cover
            cand, wave_fail = [], []
            for w0 in range(0, n, 64):
                idx = order[w0:w0 + 64]
                blocks = set()
                for i in idx:
                    for dx in (-1, 0, 1):
                        for dy in (-1, 0, 1):
                            if 0 <= cx[i] + dx < cells and 0 <= cy[i] + dy < cells:
                                blocks.add((cx[i] + dx, cy[i] + dy))
                cand.append(sum(int(((cx == bx) & (cy == by)).sum()) for bx, by in blocks))
                wave_fail.append(bool(fail[idx].any()))
            s = stats.setdefault(cells, [0, 0, 0, 0, 0.0])
            s[0] += len(cand); s[1] += sum(cand); s[2] += sum(wave_fail); s[3] += int(fail.sum()); s[4] += n
print(f"{RUNNERS + 5} agents, ticks 0..{TICKS - 1} sampled every 5th; full chain = every agent in the game")
for cells, (waves, cand, wfail, afail, agents) in sorted(stats.items()):
    print(f"cells {cells:2d} x {cells:<2d} ({L / cells:4.1f} units): candidates per wavefront {cand / waves:6.1f} "
          f"(agents in the game per replica {agents / (len(stats) and (TICKS + 4) // 5 * E):6.1f}); searchers whose check fails "
          f"{100.0 * afail / agents:5.2f} %, wavefronts with one {100.0 * wfail / waves:5.1f} %")
