import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle.tag_continuous_c import TagContinuousCOracle
K, E, L = 10, 4, 20.0
cfg = dict(num_taggers=5, num_runners=1000, grid_length=L, episode_length=500, max_acceleration=0.1,
           min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20, num_turn_levels=20,
           use_full_observation=False, num_other_agents_observed=K, tagging_distance=0.02, tag_reward_for_tagger=10.0,
           tag_penalty_for_runner=-10.0, end_of_game_reward_for_runner=1.0, seed=274880, max_speed=1.0,
           skill_level_runner=1.0, skill_level_tagger=1.0)
o = TagContinuousCOracle(E, n_threads=8, **cfg)
rng = np.random.RandomState(1)
def grid(n):
    c = int(np.sqrt(n * (np.pi / 4.84) / K)); return min(c, 8) if c >= 4 else 0
def runs(mode, C, cls_sorted, start, n, w0, w1, lin2xy):
    (x0, y0), (x1, y1) = lin2xy(cls_sorted[w0]), lin2xy(cls_sorted[w1 - 1])
    total = 0; prev_e = 0
    for r in range(max(0, y0 - 1), min(C - 1, y1 + 1) + 1):
        lo, hi = C, -1
        for yy in range(max(y0, r - 1), min(y1, r + 1) + 1):
            if mode == "row":
                a = x0 if yy == y0 else 0; b = x1 if yy == y1 else C - 1
            else:
                fwd = (yy % 2 == 0)
                if yy == y0 and yy == y1: a, b = min(x0, x1), max(x0, x1)
                elif yy == y0: a, b = (x0, C - 1) if fwd else (0, x0)
                elif yy == y1: a, b = (0, x1) if fwd else (x1, C - 1)
                else: a, b = 0, C - 1
            lo, hi = min(lo, a), max(hi, b)
        if hi < 0: continue
        lo, hi = max(0, lo - 1), min(C - 1, hi + 1)
        if mode == "row" or r % 2 == 0: c0, c1 = r * C + lo, r * C + hi
        else: c0, c1 = r * C + (C - 1 - hi), r * C + (C - 1 - lo)
        a, b = start[c0], start[c1 + 1]
        a = max(prev_e, a // 4 * 4); b = min(n, (b + 3) // 4 * 4)
        if b > a: total += b - a; prev_e = b
    return total
for t in range(201):
    act = np.stack([rng.randint(0, 21, (E, o.N)), rng.randint(0, 21, (E, o.N))], -1).astype(np.int32)
    o.step(act)
    if t % 40: continue
    out = []
    for mode in ("row", "serp"):
        tot = []; mx = []
        for e in range(E):
            live = np.nonzero(o.sig_before[e] > 0)[0]; n = len(live)
            x, y = o.loc_x[e][live].astype(np.float64), o.loc_y[e][live].astype(np.float64)
            C = grid(n); c = L / C
            cx, cy = np.minimum((x / c).astype(int), C - 1), np.minimum((y / c).astype(int), C - 1)
            if mode == "row":
                cl = cy * C + cx; lin2xy = lambda l: (l % C, l // C)
            else:
                cl = cy * C + np.where(cy % 2 == 0, cx, C - 1 - cx)
                lin2xy = lambda l: ((l % C) if (l // C) % 2 == 0 else C - 1 - (l % C), l // C)
            cls = np.sort(cl); start = np.searchsorted(cls, np.arange(C * C + 1))
            per = [runs(mode, C, cls, start, n, w0, min(n, w0 + 64), lin2xy) for w0 in range(0, n, 64)]
            tot += per; mx.append(max(per))
        out.append(f"{mode}: cand/wave mean {np.mean(tot):.0f} block-max {np.mean(mx):.0f}")
    print(f"t={t} live={n} C={C} " + " | ".join(out), flush=True)
