"""The exactness argument of the one-pass neighbour search (csrc/kernels/tag_continuous.hip,
tc_knn_packed), replayed on the host: keys = squared-distance bits with the low 7 bits replaced by
the candidate id, the K+3 smallest kept in order, and the three-way decision
  * first K+1 keys >= 383 apart            -> chain order is the answer,
  * else, (K+2)-th bucket >= cut            -> exact (sqrt, id) ranking of the first K (+ the (K+1)-th
                                               when it is inside the uncertain buckets),
  * else                                    -> the two-pass search (not modelled here: counted).
Whatever the decision returns must equal the reference's order -- float32 sqrt distance, ties by id
(tag_continuous.py:422-444) -- on random, clustered, lattice and ulp-perturbed configurations.  The
device code itself is checked in tests/test_gpu_tag_continuous.py; this test explores the decision
rule far more densely than a device run can."""
import numpy as np
import pytest

f32 = np.float32
INVALID = 0x7F800000


def _d2(x, y, i):
    dx = (x[i] - x).astype(f32)
    dy = (y[i] - y).astype(f32)
    return ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32)


def _reference(x, y, sig, i, K):
    d = np.sqrt(_d2(x, y, i)).astype(f32)
    cand = [j for j in range(len(x)) if j != i and sig[j]]
    cand.sort(key=lambda j: (d[j], j))
    return cand[:K]


def _packed(x, y, sig, i, K):
    """-> (list of ids in order, padded with -1) or None when the kernel would fall back"""
    N = len(x)
    d2 = _d2(x, y, i)
    d2[sig == 0] = np.inf  # agents out of the game sit at x = 1e30: the squared distance overflows
    keys = (d2.view(np.uint32) & ~np.uint32(127)) | np.arange(N, dtype=np.uint32)
    S = np.sort(keys)[:K + 3].astype(np.int64)
    S = np.concatenate([S, np.full(K + 3 - len(S), 0xFFFFFFFF, np.int64)])
    o = list(S)
    if i in o:  # the agent's own entry (d2 = 0 exactly, key == its id); K+3 or more twins with lower ids push it out
        o.remove(i)
    o = np.array(o[:K + 2], dtype=np.int64)
    gap = min(int(o[k + 1] - o[k]) for k in range(K))
    if gap >= 383 and o[K - 1] < INVALID:
        return [int(v & 127) for v in o[:K]]
    oK, oExtra, oLook = int(o[K - 1]), int(o[K]), int(o[K + 1])
    cut = (oK >> 7) + 2
    exact = oK >= INVALID or (oLook >> 7) >= cut
    if not exact:
        return None
    d = np.sqrt(d2).astype(f32)
    entries = [int(v & 127) for v in o[:K] if v < INVALID]
    if oK < INVALID and oExtra < INVALID and (oExtra >> 7) < cut:
        entries.append(int(oExtra & 127))
    entries.sort(key=lambda j: (d[j], j))
    entries = entries[:K]
    return entries + [-1] * (K - len(entries))


def _check(x, y, sig, K, stats):
    for i in range(len(x)):
        if not sig[i]:
            continue
        got = _packed(x, y, sig, i, K)
        stats["agents"] += 1
        if got is None:
            stats["fallback"] += 1
            continue
        want = _reference(x, y, sig, i, K)
        assert got == want + [-1] * (K - len(want)), (i, got, want)


@pytest.mark.parametrize("N,K", [(105, 10), (40, 6), (128, 16), (12, 10)])
def test_random_and_clustered_configurations(N, K):
    rng = np.random.default_rng(N * 100 + K)
    stats = {"agents": 0, "fallback": 0}
    for trial in range(30):
        L = f32(20.0)
        if trial % 3 == 0:    # uniform
            x, y = rng.random(N) * L, rng.random(N) * L
        elif trial % 3 == 1:  # tight clusters: many near-equal distances
            c = rng.integers(0, 4, N)
            x = (c % 2) * 10 + rng.normal(0, 1e-3, N)
            y = (c // 2) * 10 + rng.normal(0, 1e-3, N)
        else:                 # on the walls (clipped coordinates repeat exactly)
            x, y = rng.random(N) * L, rng.random(N) * L
            x[rng.random(N) < 0.4] = L
            y[rng.random(N) < 0.3] = 0
        sig = (rng.random(N) < 0.85).astype(np.int32)
        _check(x.astype(f32), y.astype(f32), sig, K, stats)
    # (the wall trials put ~10 agents on the same corner: those take the two-pass search)
    assert stats["fallback"] < 0.2 * stats["agents"]


def test_lattice_with_exact_ties():
    g = np.arange(8, dtype=f32)
    x, y = [a.ravel().astype(f32) for a in np.meshgrid(g, g)]
    stats = {"agents": 0, "fallback": 0}
    for K in (1, 3, 4, 7, 12):
        _check(x, y, np.ones(64, np.int32), K, stats)
    assert stats["fallback"] > 0  # four candidates at exactly the same distance at the cut: the two-pass search


@pytest.mark.parametrize("swap", [0, 1])
def test_pairs_a_few_ulps_apart_everywhere_in_the_list(swap):
    """two candidates `delta` ulps of squared distance apart, at every position of the list"""
    K, N = 6, 14
    ulp4 = np.spacing(f32(4.0))
    stats = {"agents": 0, "fallback": 0}
    for pos in range(0, K + 2):
        for delta in (0, 1, 2, 5, 64, 127, 128, 129, 255, 256, 257, 300, 382, 383, 384, 385, 511, 512, 1000):
            b = f32(np.sqrt(np.float64(delta) * np.float64(ulp4)))
            # agent 0 at (8, 8); `pos` candidates closer than 2, the pair at squared distance 4 / 4 + delta ulps,
            # the rest well beyond
            x = [8.0] + [8.0 + 0.2 * (j + 1) for j in range(pos)]
            y = [8.0] * (pos + 1)
            pair = [(10.0, 8.0), (6.0, float(f32(8.0) + b))]
            for px, py in (pair[::-1] if swap else pair):
                x.append(px); y.append(py)
            while len(x) < N:
                x.append(8.0 + 3.0 + 0.37 * len(x)); y.append(8.0)
            x, y = np.array(x, f32), np.array(y, f32)
            got = _packed(x, y, np.ones(N, np.int32), 0, K)
            stats["agents"] += 1
            if got is None:
                stats["fallback"] += 1
                continue
            assert got == _reference(x, y, np.ones(N, np.int32), 0, K), (pos, delta, swap)
    assert stats["fallback"] == 0  # a lone pair never needs the two-pass search


def test_many_candidates_within_a_few_hundred_ulps():
    """eight candidates whose squared distances to agent 0 lie within 600 ulps of each other, around
    the cut: whenever the rule does not hand over to the two-pass search its answer must be exact"""
    rng = np.random.default_rng(5)
    K, N = 6, 16
    ulp4 = np.spacing(f32(4.0))
    answered = fell_back = 0
    for trial in range(400):
        n_close = int(rng.integers(2, 5))           # candidates well inside
        m = np.sort(rng.integers(0, 600, size=8))    # ulps above squared distance 4
        x = [8.0] + [8.0 + 0.25 * (j + 1) for j in range(n_close)]
        y = [8.0] * (n_close + 1)
        order = rng.permutation(8)                   # ids in random order relative to distance
        for q in order:
            b = f32(np.sqrt(np.float64(m[q]) * np.float64(ulp4)))
            side = 10.0 if q % 2 else 6.0
            x.append(side); y.append(float(f32(8.0) + b))
        while len(x) < N:
            x.append(14.0 + 0.5 * len(x)); y.append(3.0)
        x, y = np.array(x, f32), np.array(y, f32)
        sig = np.ones(N, np.int32)
        got = _packed(x, y, sig, 0, K)
        if got is None:
            fell_back += 1
            continue
        answered += 1
        assert got == _reference(x, y, sig, 0, K), (trial, m.tolist(), order.tolist())
    assert answered > 50 and fell_back > 50
