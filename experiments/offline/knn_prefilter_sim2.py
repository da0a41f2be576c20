#!/usr/bin/env python3
"""Offline study, second part: the remembered set as the KERNEL maintains it.

The kernel remembers what its chain holds: the (at most K + 3) nearest others AMONG THE LISTED candidates.  A
tight threshold lists about K candidates, so no spares are remembered and the first remembered agent that is
tagged out leaves a far stand-in as the bound.  Policies compared: threshold = f x the `rank`-th largest finite
current squared distance to the remembered agents, provided at least K of them are still in the game.

    python experiments/offline/knn_prefilter_sim2.py [replicas]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tag_continuous_c import TagContinuousCOracle  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K, M, CAP = 10, 13, 43
cfg = dict(num_taggers=5, num_runners=100, grid_length=20.0, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880)


def run(factor, take):
    """take = which finite remembered distance is the base: 0 = the largest, 1 = second largest ..."""
    o = TagContinuousCOracle(E, n_threads=8, **cfg)
    N = o.N
    rng = np.random.RandomState(1)
    prev = np.full((E, N, M), -1)
    trips, direct_waves, waves = [], 0, 0
    big = 1e30
    for t in range(500):
        act = np.stack([rng.randint(0, 21, (E, N)), rng.randint(0, 21, (E, N))], -1)
        o.step(act)
        live = o.sig_before > 0
        x, y = o.loc_x.astype(np.float64), o.loc_y.astype(np.float64)
        d2 = (x[:, :, None] - x[:, None, :]) ** 2 + (y[:, :, None] - y[:, None, :]) ** 2
        d2m = np.where(live[:, None, :], d2, big)
        ii = np.arange(N)
        d2m[:, ii, ii] = big
        pd = np.where(prev >= 0, np.take_along_axis(d2m, np.maximum(prev, 0), 2), big)
        pd.sort(axis=2)                      # ascending; big = dead / none
        nfin = (pd < big).sum(2)
        base_idx = np.clip(nfin - 1 - take, K - 1, M - 1)
        base = np.take_along_axis(pd, base_idx[..., None], 2)[..., 0]
        thr = np.where(nfin >= K, base * factor, big)
        listed = d2m <= thr[:, :, None]
        nlist = listed.sum(2)
        nobound = (thr >= big) | (nlist > CAP)
        # what the chain remembers: the M nearest others among the listed (everything when there was no bound)
        dl = np.where(listed | nobound[:, :, None], d2m, big)
        order = np.argsort(dl, axis=2, kind="stable")[:, :, :M]
        ds = np.take_along_axis(dl, order, 2)
        prev = np.where(ds < big, order, -1)
        for e in range(E):
            ids = np.nonzero(live[e])[0]
            for w0 in range(0, len(ids), 64):
                sel = ids[w0:w0 + 64]
                waves += 1
                if t > 0 and nobound[e, sel].any() and live[e].sum() > K + 1:
                    direct_waves += 1
                else:
                    trips.append(nlist[e, sel].max() + 1)
    return np.mean(trips), np.percentile(trips, 99), direct_waves / waves


for factor, take in ((1.0, 3), (1.0, 0), (1.15, 0), (1.25, 0), (1.4, 0), (1.25, 1), (1.5, 1)):
    m, p99, dw = run(factor, take)
    print(f"threshold = {factor:4.2f} x the {take + 1}. largest finite remembered d2: pass-2 trips mean {m:5.1f} p99 {p99:4.0f}; "
          f"wavefronts that fall back to the full chain {100 * dw:6.3f} %")
