#!/usr/bin/env python3
"""Offline study for replicas of more than 128 agents: how many chain insertions does the prefiltered search of
tc_chain_prefiltered need, by radius policy?  The state is advanced by the C oracle under the benchmark's uniform
policy; the kernel's bookkeeping is replayed: the remembered K + 3 nearest others AMONG THE LISTED candidates, the
radius derived from them on the next tick, the exactness check, candidates in packed (id) order, 64 consecutive
searchers per wavefront, 32 candidates per mask word, trips per word = the fullest lane's count.

    python experiments/offline/knn_prefilter_big_sim.py [runners] [ticks]
Policies: ("max", f) = f x (K + 3) / n x the largest current d2 to the n remembered agents still in the game (the
kernel's: f = 1.15); ("rank", r, f) = f x the r-th smallest current d2 to them (needs n >= r; r = K + 1 provably holds
the K nearest at f = 1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tag_continuous_c import TagContinuousCOracle  # noqa: E402

RUNNERS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
K, M, E = 10, 13, 2
cfg = dict(num_taggers=5, num_runners=RUNNERS, grid_length=20.0, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880, max_speed=1.0,
           skill_level_runner=1.0, skill_level_tagger=1.0)
BIG = 1e30


def run(policy):
    o = TagContinuousCOracle(E, n_threads=8, **cfg)
    N = o.N
    rng = np.random.RandomState(1)
    prev = np.full((E, N, M), -1)
    rows = []
    for t in range(TICKS):
        act = np.stack([rng.randint(0, 21, (E, N)), rng.randint(0, 21, (E, N))], -1).astype(np.int32)
        o.step(act)
        live = o.sig_before > 0
        x, y = o.loc_x.astype(np.float64), o.loc_y.astype(np.float64)
        d2 = (x[:, :, None] - x[:, None, :]) ** 2 + (y[:, :, None] - y[:, None, :]) ** 2
        d2m = np.where(live[:, None, :], d2, BIG)
        ii = np.arange(N)
        d2m[:, ii, ii] = BIG
        pd = np.where(prev >= 0, np.take_along_axis(d2m, np.maximum(prev, 0), 2), BIG)
        pd.sort(axis=2)
        nfin = (pd < BIG).sum(2)
        if policy[0] == "max":
            far = np.take_along_axis(pd, np.maximum(nfin - 1, 0)[..., None], 2)[..., 0]
            thr = np.where(nfin >= 5, far * policy[1] * M / np.maximum(nfin, 1), BIG)
        else:  # (fewer than r remembered agents still in the game: the "max" policy)
            r, f = policy[1], policy[2]
            far = np.take_along_axis(pd, np.maximum(nfin - 1, 0)[..., None], 2)[..., 0]
            thr = np.where(nfin >= r, pd[..., r - 1] * f, np.where(nfin >= 5, far * 1.15 * M / np.maximum(nfin, 1), BIG))
        listed = d2m <= thr[:, :, None]
        # exactness check: the K-th other found lies inside the radius with a margin (two key buckets ~ 2e-4 relative)
        dl = np.where(listed, d2m, BIG)
        order = np.argsort(dl, axis=2, kind="stable")[:, :, :M]
        ds = np.take_along_axis(dl, order, 2)
        held = (ds[..., K - 1] * (1 + 3e-4) <= thr) & (thr < BIG)
        full = np.argsort(d2m, axis=2, kind="stable")[:, :, :M]
        remember = np.where(held[..., None], np.where(ds < BIG, order, -1), full)
        prev = np.where(live[..., None], remember, -1)
        for e in range(E):
            ids = np.nonzero(live[e])[0]
            n = len(ids)
            if n < 200:
                continue
            L = listed[e][np.ix_(ids, ids)]          # [searcher, candidate] in packed order (self excluded: +1 per lane)
            L[np.arange(n), np.arange(n)] = True      # the agent's own entry is listed (d2 = 0)
            for w0 in range(0, n, 64):
                lanes = L[w0:w0 + 64]
                ok = held[e][ids[w0:w0 + 64]].all()
                trips = sum(int(lanes[:, c:c + 32].sum(1).max()) for c in range(0, n, 32))
                trips64 = sum(int(lanes[:, c:c + 64].sum(1).max()) for c in range(0, n, 64))
                trips256 = sum(int(lanes[:, c:c + 256].sum(1).max()) for c in range(0, n, 256))
                rows.append((t, n, ok, lanes.sum(1).mean(), lanes.sum(1).max(), trips, trips64, trips256))
    a = np.array(rows, dtype=np.float64)
    late = a[a[:, 0] >= 3]
    print(f"{str(policy):<22} wavefronts {len(late):5d}  radius held {late[:, 2].mean() * 100:5.1f} %  listed per lane mean "
          f"{late[:, 3].mean():5.1f} fullest lane {late[:, 4].mean():5.1f}  insertions per wavefront: words of 32 "
          f"{late[:, 5].mean():6.1f}  words of 64 {late[:, 6].mean():6.1f}  chunks of 256 {late[:, 7].mean():6.1f}  (full chain: {late[:, 1].mean():.0f})")


for pol in (("max", 1.15), ("rank", K + 1, 1.0), ("rank", K + 1, 1.1), ("rank", K + 2, 1.0), ("rank", K + 2, 1.1), ("rank", M, 1.0)):
    run(pol)
