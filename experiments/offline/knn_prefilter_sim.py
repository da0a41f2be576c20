#!/usr/bin/env python3
"""Offline study (CPU, oracle trajectories): how many candidates survive a per-searcher threshold taken
from the PREVIOUS tick's neighbours?

For searcher i the K+S nearest others of the previous tick are remembered; at the current tick the
threshold is the K-th smallest CURRENT squared distance to those of them that are still in the game (any K
distinct live others bound the K-th nearest from above).  A candidate has to go through the insertion
chain only when its squared distance is <= the threshold.  The searchers of a replica are packed into
wavefronts of 64 in ascending id order (as the kernel packs them); a wavefront's second pass is as long
as its worst lane, so the statistic that matters is the per-wavefront MAXIMUM of the survivor count.

    python experiments/offline/knn_prefilter_sim.py [replicas] [spares]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tag_continuous_c import TagContinuousCOracle  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = 10
cfg = dict(num_taggers=5, num_runners=100, grid_length=20.0, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880)
o = TagContinuousCOracle(E, n_threads=8, **cfg)
N = o.N
rng = np.random.RandomState(1)
prev = None  # [E, N, K+S] ids of the previous tick (-1 = none)
rows = []
for t in range(500):
    act = np.stack([rng.randint(0, len(o.acceleration_actions), (E, N)), rng.randint(0, len(o.turn_actions), (E, N))], -1)
    o.step(act)
    live = o.sig_before > 0  # the search runs over the agents in the game before this tick's tagging
    x, y = o.loc_x.astype(np.float64), o.loc_y.astype(np.float64)
    d2 = (x[:, :, None] - x[:, None, :]) ** 2 + (y[:, :, None] - y[:, None, :]) ** 2
    big = 1e30
    d2m = np.where(live[:, None, :], d2, big)
    ii = np.arange(N)
    d2m[:, ii, ii] = big
    order = np.argsort(d2m, axis=2, kind="stable")[:, :, : K + S]
    dsel = np.take_along_axis(d2m, order, 2)
    cur = np.where(dsel < big, order, -1)
    if prev is not None:
        pd = np.take_along_axis(d2m, np.maximum(prev, 0), 2)
        pd = np.where(prev >= 0, pd, big)
        pd.sort(axis=2)
        thr = pd[:, :, K - 1]  # K-th smallest current distance to remembered live neighbours (big = no bound)
        surv = (d2m <= thr[:, :, None]).sum(2)  # candidates that go through the chain (others only)
        n_live = live.sum(1)
        wmax, wfull, nob = [], [], 0
        for e in range(E):
            ids = np.nonzero(live[e])[0]
            s = surv[e, ids]
            nob += int((thr[e, ids] >= big).sum() if n_live[e] > K + 1 else 0)
            for w0 in range(0, len(ids), 64):
                wmax.append(s[w0:w0 + 64].max())
                wfull.append(n_live[e] - 1)
        rows.append((t, n_live.mean(), np.mean(wmax), np.percentile(wmax, 99), np.max(wmax), np.mean(wfull), nob / E))
    prev = cur
print("tick  live  per-wave max survivors: mean  p99  max | chain length today | lanes without a bound per replica")
for r in rows[::25] + rows[:5]:
    print("%4d %5.1f %28.1f %5.0f %5.0f | %6.1f | %.3f" % r)
a = np.array(rows)
print("episode mean of per-wave max survivors %.1f (chain today %.1f); lanes without a bound per replica-tick %.4f" %
      (a[:, 2].mean(), a[:, 5].mean(), a[:, 6].mean()))
