import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle.tag_continuous_c import TagContinuousCOracle
RUNNERS = int(sys.argv[1]); TICKS = int(sys.argv[2]); EVERY = int(sys.argv[3]); RULE=sys.argv[4]; FAC=float(sys.argv[5])
K, E, L = 10, 1, 20.0
cfg = dict(num_taggers=5, num_runners=RUNNERS, grid_length=L, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880, max_speed=1.0,
           skill_level_runner=1.0, skill_level_tagger=1.0)
o = TagContinuousCOracle(E, n_threads=8, **cfg)
rng = np.random.RandomState(1)
prev = None  # [N, 13] ids of the previous tick's K+3 nearest (by id), -1 none
def grid(n): 
    c = int(np.sqrt(n * (np.pi / 4.84) / K)); return min(c, 8) if c >= 4 else 0
for t in range(TICKS):
    act = np.stack([rng.randint(0, 21, (E, o.N)), rng.randint(0, 21, (E, o.N))], -1).astype(np.int32)
    o.step(act)
    e = 0
    sig = o.sig_before[e] > 0
    X, Y = o.loc_x[e].astype(np.float64), o.loc_y[e].astype(np.float64)
    live = np.nonzero(sig)[0]; n = len(live)
    x, y = X[live], Y[live]
    d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2
    dd = d2.copy(); np.fill_diagonal(dd, np.inf)
    order_nn = np.argsort(dd, axis=1)[:, :K + 3]
    kth2 = np.take_along_axis(dd, order_nn, 1)[:, K - 1]
    Tfeed = None
    if prev is not None:
        # radius rules from the remembered ids
        rem = prev[live]                                  # [n, 13] agent ids
        ok = (rem >= 0) & sig[np.maximum(rem, 0)]
        rd2 = (X[np.maximum(rem, 0)] - x[:, None]) ** 2 + (Y[np.maximum(rem, 0)] - y[:, None]) ** 2
        rd2 = np.where(ok, rd2, np.nan)
        nrem = ok.sum(1)
        far = np.nanmax(np.where(ok, rd2, -1), axis=1)
        T_max = 1.15 * (K + 3) / np.maximum(nrem, 1) * far
        srt = np.sort(np.where(ok, rd2, np.inf), axis=1)
        idx3 = np.maximum(nrem - 3, 0)
        m3 = np.take_along_axis(srt, idx3[:, None], 1)[:, 0]
        m2 = np.take_along_axis(srt, np.maximum(nrem - 2, 0)[:, None], 1)[:, 0]
        T_m3 = np.where(nrem >= K + 2, FAC * m3, T_max)
        T_m2 = np.where(nrem >= K + 1, FAC * m2, T_max)
        T_max = np.where(nrem >= 5, T_max, np.inf)
        Tfeed = {"max": T_max, "m3": np.where(nrem >= 5, T_m3, np.inf), "m2": np.where(nrem >= 5, T_m2, np.inf)}[RULE]
        C = grid(n)
        res = {}
        for name, Th in ((RULE, Tfeed),):
            if C:
                c = L / C
                cx, cy = np.minimum((x / c).astype(int), C - 1), np.minimum((y / c).astype(int), C - 1)
                cl = cy * C + cx
                order = np.argsort(cl, kind="stable")
            else:
                order = np.arange(n); cl = np.zeros(n, int); c = L
            xs, ys, cls = x[order], y[order], cl[order]
            if C:
                cxs, cys = cls % C, cls // C
                cover = np.minimum.reduce([np.where(cxs > 0, xs - (cxs - 1) * c, np.inf), np.where(cxs < C - 1, (cxs + 2) * c - xs, np.inf),
                                           np.where(cys > 0, ys - (cys - 1) * c, np.inf), np.where(cys < C - 1, (cys + 2) * c - ys, np.inf)])
                T = np.minimum(Th[order], (cover - c / 1024) ** 2)
                start = np.searchsorted(cls, np.arange(C * C + 1))
            else:
                T = Th[order]
            fail = kth2[order] * 1.0003 > T
            d2s = d2[np.ix_(order, order)]
            lst = d2s <= T[:, None]
            trips, cands, wf, lmax = [], [], 0, []
            for w0 in range(0, n, 64):
                w1 = min(n, w0 + 64)
                if C:
                    cl0, cl1 = cls[w0], cls[w1 - 1]
                    cy0, cx0, cy1, cx1 = cl0 // C, cl0 % C, cl1 // C, cl1 % C
                    ranges, prev_e = [], 0
                    for r in range(max(0, cy0 - 1), min(C - 1, cy1 + 1) + 1):
                        lo, hi = C, -1
                        for yy in range(max(cy0, r - 1), min(cy1, r + 1) + 1):
                            lo, hi = min(lo, cx0 if yy == cy0 else 0), max(hi, cx1 if yy == cy1 else C - 1)
                        if hi < 0: continue
                        lo, hi = max(0, lo - 1), min(C - 1, hi + 1)
                        a, b = start[r * C + lo], start[r * C + hi + 1]
                        a = max(prev_e, a // 4 * 4); b = min(n, (b + 3) // 4 * 4)
                        if b > a:
                            if ranges and ranges[-1][1] == a: ranges[-1] = (ranges[-1][0], b)
                            else: ranges.append((a, b))
                            prev_e = b
                else:
                    ranges = [(0, n)]
                # words of 32 within runs; flush every 8 words
                words = []
                for a, b in ranges:
                    for j in range(a, b, 32): words.append((j, min(j + 32, b)))
                tr = 0
                for f in range(0, len(words), 8):
                    cols = np.concatenate([np.arange(a, b) for a, b in words[f:f + 8]])
                    tr += lst[w0:w1][:, cols].sum(1).max()
                allc = np.concatenate([np.arange(a, b) for a, b in ranges])
                lmax.append(lst[w0:w1][:, allc].sum(1).max())
                trips.append(tr); cands.append(len(allc)); wf += bool(fail[w0:w1].any())
            res[name] = (np.mean(cands), np.mean(trips), np.max(trips), np.mean(lmax), wf, len(trips), lst.sum(1).mean())
        Tlast = np.empty(n); Tlast[order] = T
        if t % EVERY == 0: print(f"t={t} live={n} C={C} " + " | ".join(f"{k}: cand {v[0]:.0f} trips {v[1]:.0f} (max {v[2]}) lane-max {v[3]:.0f} listed/lane {v[6]:.1f} wfail {v[4]}/{v[5]}" for k, v in res.items()), flush=True)
    p = np.full((o.N, K + 3), -1, int)
    p[live] = live[order_nn]
    dn = np.take_along_axis(dd, order_nn, 1)
    ok2 = dn < np.inf
    if Tfeed is not None:
        heldm = kth2 * 1.0003 <= Tlast
        ok2 &= (dn <= Tlast[:, None]) | ~heldm[:, None]   # not held -> the full chain remembers the 13 nearest
    p[live] = np.where(ok2, p[live], -1)
    prev = p
