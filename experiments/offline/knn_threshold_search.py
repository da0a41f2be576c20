"""Offline study for the transposed neighbour search (lane = candidate, loop over agents): how many
count(s <= t) probes does it take to find a threshold t with exactly K candidates inside, starting
from the previous tick's threshold?  Uses the committed reference trajectories (tests/golden)."""
import sys
import numpy as np

f32 = np.float32


def dists(x, y, sig, i):
    dx = (x[i] - x).astype(f32); dy = (y[i] - y).astype(f32)
    d2 = (dx * dx).astype(f32) + (dy * dy).astype(f32)
    d2 = d2.astype(f32)  # the search runs on squared distances (no sqrt per candidate)
    d2[sig == 0] = np.inf
    d2[i] = np.inf
    return d2


def fbits(v):
    return int(np.array(v, f32).view(np.uint32))


def search(sbits, K, hint_bits, step, grid_length, others, max_iter=40, secant_iters=3):
    """exact restatement of the device loop (tc_knn_transposed): returns (threshold bits | None, probes)"""
    lo, hi = fbits(1.0e-8), fbits(4.0 * grid_length * grid_length)
    t = hint_bits
    if t <= lo or t >= hi:  # no usable hint: uniform-density guess
        t = fbits(grid_length * grid_length * (K + 0.5) / (np.pi * others))
        t = min(max(t, lo + 1), hi - 1)
    for it in range(max_iter):
        c = int((sbits <= t).sum())
        if c == K:
            return t, it + 1
        if c < K: lo = t
        else: hi = t
        if hi - lo <= 1:
            return None, it + 1
        m = c - K
        cand = t - m * step - (step // 2 if m > 0 else -(step // 2))
        if it >= secant_iters or cand <= lo or cand >= hi:
            cand = (lo + hi) >> 1
        t = cand
    return None, max_iter


def main(path, K=10, step=600_000, hint_kind='t', bump=0):
    z = np.load(path, allow_pickle=True)
    X, Y, S = z["loc_x"], z["loc_y"], z["still_in_the_game"]
    T, E, N = X.shape
    import json
    L = float(json.loads(str(z["config"]))["grid_length"])
    hint = np.zeros((E, N), np.int64)
    hist = {}
    fails = 0
    per_tick = []
    for t in range(T):
        probes_t = []
        for e in range(E):
            # the search sees the positions after the move and still_in_the_game BEFORE this tick's tags
            sig = S[t - 1, e] if t > 0 else np.ones(N, np.int32)
            alive = int(sig.sum())
            for i in range(N):
                if not sig[i]:
                    continue
                if alive - 1 <= K:
                    continue
                s = dists(X[t, e], Y[t, e], sig, i)
                tb, n = search(s.view(np.uint32).astype(np.int64), K, int(hint[e, i]), step, L, alive - 1)
                if tb is None:
                    fails += 1
                else:
                    # keep the midpoint between the K-th and (K+1)-th distance? the kernel only knows t
                    hint[e, i] = tb if hint_kind == 't' else int(np.sort(s.view(np.uint32).astype(np.int64))[K - 1]) + bump
                hist[n] = hist.get(n, 0) + 1
                probes_t.append(n)
        per_tick.append(np.mean(probes_t) if probes_t else 0)
    tot = sum(hist.values())
    mean = sum(k * v for k, v in hist.items()) / tot
    print(f"{path}: step={step} agents-ticks={tot} mean probes={mean:.2f} fails={fails}")
    print("  per tick:", " ".join(f"{p:.1f}" for p in per_tick[:12]), "...")
    print("  hist:", {k: hist[k] for k in sorted(hist)})


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/tc_traj_bench5x100_ep.npz"
    for step in (800_000, 1_150_000):
        main(path, step=step)
    for bump in (0, 200_000, 400_000, 600_000):
        print("hint = K-th squared distance, bits +", bump)
        main(path, step=1_150_000, hint_kind="kth", bump=bump)
