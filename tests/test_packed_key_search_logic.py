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


# ---------------------------------------------------------------- round 4: the shortcut for an isolated close pair, the
# merge of two wavefronts' key lists, the zone the exact resolution scans (tc_resolve_keys, tc_merge_sorted,
# tc_zone_resolve in csrc/kernels/tag_continuous.hip), for 7 / 9 / 10 id bits
def _packed_r4(x, y, sig, i, K, idb=7):
    """as _packed, plus: a lane that is not `apart` whose close gaps (< 3 * 2^idb - 1) are ISOLATED pairs -- no two
    adjacent, and not the pair at the cut together with the pair behind it -- settles each pair with one exact
    (float32 distance, index) compare.  -> (ids, how) with how in {"chain", "pairs", "ranking"}; None = exact resolution."""
    N, idm = len(x), (1 << idb) - 1
    d2 = _d2(x, y, i)
    d2[sig == 0] = np.inf
    keys = (d2.view(np.uint32) & ~np.uint32(idm)) | np.arange(N, dtype=np.uint32)
    S = np.sort(keys)[:K + 3].astype(np.int64)
    S = np.concatenate([S, np.full(K + 3 - len(S), 0xFFFFFFFF, np.int64)])
    got = _decide_from_list_r4(S, d2, i, K, idb)
    if got is not None:
        return got
    got = _packed(x, y, sig, i, K) if idb == 7 else None
    return (got, "ranking") if got is not None else None


def _decide_from_list_r4(S, d2, i, K, idb):
    """what tc_resolve_keys reads off the key list without the ranking: (ids, "chain" | "pairs") or None"""
    idm = (1 << idb) - 1
    o = list(S)
    if i in o:
        o.remove(i)
    o = np.array(o[:K + 2], dtype=np.int64)
    thr = 3 * (idm + 1) - 1
    d = np.sqrt(d2).astype(f32)
    if min(int(o[k + 1] - o[k]) for k in range(K)) >= thr and o[K - 1] < INVALID:
        return [int(v & idm) for v in o[:K]], "chain"
    if o[K - 1] < INVALID:
        close = [int(o[k + 1] - o[k]) < thr for k in range(K + 1)]  # pair (k, k + 1); k == K: behind the cut
        rel = close[:K]
        runs = any(rel[k] and rel[k + 1] for k in range(K - 1)) or (rel[K - 1] and close[K])
        if not runs:
            ids = [int(v & idm) for v in o[:K + 1]]
            for q in range(K):
                if rel[q]:
                    a, b = ids[q], ids[q + 1]
                    if (d[b], b) < (d[a], a):
                        ids[q], ids[q + 1] = b, a
            return ids[:K], "pairs"
    return None


def _check_r4(x, y, sig, K, stats, idb=7):
    for i in range(len(x)):
        if not sig[i]:
            continue
        got = _packed_r4(x, y, sig, i, K, idb)
        if got is None:
            stats["resolve"] += 1
            # what the exact resolution scans: every candidate in the key buckets <= bucket(K-th other) + 1 -- the
            # reference's K nearest must all be among them
            idm = (1 << idb) - 1
            d2 = _d2(x, y, i)
            d2[sig == 0] = np.inf
            bucket = d2.view(np.uint32).astype(np.int64) >> idb
            others = [j for j in range(len(x)) if j != i and sig[j]]
            kth_key = sorted(((int(d2[j].view(np.uint32)) & ~idm) | j) for j in others)[K - 1]
            zone = {j for j in others if bucket[j] <= (kth_key >> idb) + 1}
            want = _reference(x, y, sig, i, K)
            assert set(want) <= zone and len(zone) >= K
            continue
        ids, how = got
        stats[how] += 1
        want = _reference(x, y, sig, i, K)
        assert ids == want + [-1] * (K - len(want)), (i, how, ids, want)


@pytest.mark.parametrize("N,K,idb", [(105, 10, 7), (40, 6, 7), (128, 16, 7), (300, 10, 9), (700, 8, 10)])
def test_isolated_close_pairs_are_settled_by_one_compare(N, K, idb):
    rng = np.random.default_rng(N * 31 + K)
    stats = {"chain": 0, "pairs": 0, "ranking": 0, "resolve": 0}
    for trial in range(12 if N <= 128 else 3):
        L = f32(20.0)
        if trial % 3 == 0:
            x, y = rng.random(N) * L, rng.random(N) * L
        elif trial % 3 == 1:
            c = rng.integers(0, 4, N)
            x = (c % 2) * 10 + rng.normal(0, 1e-3, N)
            y = (c // 2) * 10 + rng.normal(0, 1e-3, N)
        else:
            x, y = rng.random(N) * L, rng.random(N) * L
            x[rng.random(N) < 0.4] = L
            y[rng.random(N) < 0.3] = 0
        sig = (rng.random(N) < 0.85).astype(np.int32)
        _check_r4(x.astype(f32), y.astype(f32), sig, K, stats, idb)
    assert stats["pairs"] > 0 and stats["chain"] > 0, stats


@pytest.mark.parametrize("swap", [0, 1])
@pytest.mark.parametrize("idb", [7, 9, 10])
def test_a_lone_close_pair_anywhere_in_the_list_takes_the_shortcut(swap, idb):
    K, N = 6, 14
    ulp4 = np.spacing(f32(4.0))
    w = 1 << idb
    how_seen = set()
    for pos in range(0, K + 2):
        for delta in (0, 1, 2, w - 1, w, w + 1, 2 * w - 1, 2 * w, 2 * w + 1, 3 * w - 2, 3 * w - 1, 3 * w, 4 * w, 10 * w):
            b = f32(np.sqrt(np.float64(delta) * np.float64(ulp4)))
            x = [8.0] + [8.0 + 0.2 * (j + 1) for j in range(pos)]
            y = [8.0] * (pos + 1)
            pair = [(10.0, 8.0), (6.0, float(f32(8.0) + b))]
            for px, py in (pair[::-1] if swap else pair):
                x.append(px); y.append(py)
            while len(x) < N:
                x.append(8.0 + 3.0 + 0.37 * len(x)); y.append(8.0)
            x, y = np.array(x, f32), np.array(y, f32)
            got = _packed_r4(x, y, np.ones(N, np.int32), 0, K, idb)
            assert got is not None and got[1] in ("chain", "pairs"), (pos, delta, got)
            assert got[0] == _reference(x, y, np.ones(N, np.int32), 0, K), (pos, delta, swap, idb)
            how_seen.add(got[1])
    assert how_seen == {"chain", "pairs"}


def _merge_sorted(S, P):
    """tc_merge_sorted: both lists padded to a power of two with 0xffffffff, c[k] = min(S[k], P[W-1-k]), bitonic merge"""
    L = len(S)
    W = 8 if L <= 8 else 16 if L <= 16 else 32 if L <= 32 else 64
    pad = 0xFFFFFFFF
    s = list(S) + [pad] * (W - L)
    p = list(P) + [pad] * (W - L)
    c = [min(s[k], p[W - 1 - k]) for k in range(W)]
    stride = W // 2
    while stride >= 1:
        for k in range(W):
            if (k & stride) == 0:
                lo, hi = min(c[k], c[k + stride]), max(c[k], c[k + stride])
                c[k], c[k + stride] = lo, hi
        stride //= 2
    return c[:L]


@pytest.mark.parametrize("L", [5, 13, 15, 19, 27, 35])
def test_two_wavefronts_key_lists_merge_to_the_single_chain_result(L):
    """wavefront 0 chains the first half of the candidates, wavefront 1 the second half; the L smallest of the union
    of their L-entry lists are exactly the L smallest of all candidates, in order"""
    rng = np.random.default_rng(L)
    for trial in range(300):
        n = int(rng.integers(2, 64))
        keys = rng.choice(1 << 30, size=n, replace=False).astype(np.int64)
        half = ((n + 7) >> 3) << 2
        a, b = np.sort(keys[:half])[:L], np.sort(keys[half:])[:L]
        pad = lambda v: list(v) + [0xFFFFFFFF] * (L - len(v))
        want = pad(np.sort(keys)[:L])
        assert _merge_sorted(pad(a), pad(b)) == want


# ---------------------------------------------------------------------------------------------------------------
# Round 4: the PREFILTERED search of replicas of more than 128 agents (tc_knn_bound16 / tc_pre_pass1 / tc_pre_pass2).
# Only candidates inside a radius derived from the previous tick's neighbours reach the chain; the radius is a
# heuristic, what makes the result exact is the check that the K-th other agent found lies at least two key buckets
# inside it (then everything that was left out is past the buckets the decision rule above looks at).
def _prefiltered_keys(x, y, i, T, K, idb):
    """the chain's list when only candidates with d2 <= T are inserted; None when the radius check fails"""
    N = len(x)
    idm = np.uint32((1 << idb) - 1)
    d2 = _d2(x, y, i)
    keys = (d2.view(np.uint32) & ~idm) | np.arange(N, dtype=np.uint32)
    listed = np.sort(keys[d2 <= T])[:K + 3].astype(np.int64)
    S = np.concatenate([listed, np.full(K + 3 - len(listed), 0x7F800000 | N, np.int64)])  # (lanes that ran out insert the pad)
    sK = int(S[K])  # entry K with the agent's own entry in the list = the K-th other agent
    Tb = int(np.array([T], f32).view(np.uint32)[0])
    if (sK >> idb) + 2 > (Tb >> idb):
        return None
    return S


@pytest.mark.parametrize("N,K,idb", [(300, 10, 9), (1000, 10, 10), (600, 4, 10), (1000, 12, 10)])
def test_prefiltered_list_gives_the_full_chain_result_whenever_the_radius_check_holds(N, K, idb):
    rng = np.random.default_rng(N + K)
    held = failed = decided = 0
    for trial in range(6):
        x = rng.uniform(0, 30, N).astype(f32)
        y = rng.uniform(0, 30, N).astype(f32)
        if trial >= 3:  # clustered: many near-equal distances
            x = (np.round(x * 2) / 2).astype(f32)
            y = (np.round(y * 2) / 2).astype(f32)
        sig = np.ones(N, np.int32)
        for i in rng.choice(N, 12, replace=False):
            d2 = _d2(x, y, i)
            order = np.argsort(d2, kind="stable")
            for scale in (0.5, 0.9, 1.0, 1.0000001, 1.15, 2.0):  # radii below, at and above the K-th neighbour
                T = f32(d2[order[K]] * f32(scale))
                S = _prefiltered_keys(x, y, i, T, K, idb)
                if S is None:
                    failed += 1
                    continue
                held += 1
                # every key the decision rule can look at is the same as in the full chain's list ...
                idm = np.uint32((1 << idb) - 1)
                full = np.sort((d2.view(np.uint32) & ~idm) | np.arange(N, dtype=np.uint32))[:K + 3].astype(np.int64)
                cut = (int(S[K]) >> idb) + 2
                assert [k for k in full if (k >> idb) < cut] == [k for k in S if (k >> idb) < cut]
                # ... and what it decides from the prefiltered list is the reference's answer (or a fallback)
                got = _decide_from_list_r4(S, d2, i, K, idb)
                if got is not None:
                    assert got[0] == _reference(x, y, sig, i, K), (i, scale, got)
                    decided += 1
    assert held > 20 and failed > 10 and decided > 10, (held, failed, decided)
