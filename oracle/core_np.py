"""numpy oracle for the core service kernels (test infrastructure).

Restates reference warp_drive/cuda_includes/core/random.cu:33-85 (categorical sampler:
inclusive float32 prefix sum + binary search with kEps), numba_includes/core/random.py:66-105
(OU process), cuda_includes/core/reset.cu:9-75 (reset when done / undo).

The reference pins its RNG streams only statistically (tests/warp_drive/pycuda_tests/
test_action_sampler.py:90-156,253-257; numba_tests/test_ou_sampler.py:68-82), so these
functions take the uniform / normal draws as INPUTS: given the same draws the device must
return the same indices.  Bitwise RNG parity is unpinned by design.
"""
import numpy as np

K_EPS = np.float32(1.0e-8)  # random.cu:9


def search_index(cum, p):
    """Binary search of random.cu:33-49 on one inclusive prefix-sum row."""
    left, right, r = 0, len(cum) - 1, len(cum) - 1
    while left <= right:
        mid = left + (right - left) // 2
        if abs(np.float32(cum[mid] - p)) < K_EPS:
            return mid
        if cum[mid] < p:
            left = mid + 1
        else:
            right = mid - 1
    return r if left > r else left


def sample_actions(distr, u, use_argmax=False):
    """distr float32 [..., A]; u float32 [...] in (0, 1].  Returns int32 [...]."""
    distr = np.asarray(distr, dtype=np.float32)
    A = distr.shape[-1]
    flat = distr.reshape(-1, A)
    out = np.empty(flat.shape[0], dtype=np.int32)
    if use_argmax:  # first strict maximum, random.cu:58-68
        for r, row in enumerate(flat):
            best, idx = row[0], 0
            for i in range(1, A):
                if best < row[i]:
                    best, idx = row[i], i
            out[r] = idx
        return out.reshape(distr.shape[:-1])
    uu = np.asarray(u, dtype=np.float32).reshape(-1)
    for r, row in enumerate(flat):
        cum = np.empty(A, dtype=np.float32)
        cum[0] = row[0]
        for i in range(1, A):
            cum[i] = np.float32(row[i] + cum[i - 1])  # random.cu:76-81
        out[r] = search_index(cum, uu[r])
    return out.reshape(distr.shape[:-1])


def sample_actions_counting(distr, u):
    """The closed form the device uses: #{i : cum_i < u}, clamped to A-1.  Equal to
    search_index except when u ties a prefix sum to within kEps (measure zero)."""
    distr = np.asarray(distr, dtype=np.float32)
    cum = np.cumsum(distr, axis=-1, dtype=np.float32)  # sequential float32 adds
    cnt = (cum < np.asarray(u, dtype=np.float32)[..., None]).sum(axis=-1)
    return np.minimum(cnt, distr.shape[-1] - 1).astype(np.int32)


def ou_step(ou_state, distr, normal, damping=0.15, stddev=0.2, scale=1.0):
    """numba random.py:66-105 given the N(0,1) draws."""
    f = np.float32
    if f(scale) < f(1e-8):
        return ou_state, np.asarray(distr, dtype=f)
    ou = (f(1.0) - f(damping)) * np.asarray(ou_state, f) + f(stddev) * np.asarray(normal, f)
    return ou.astype(f), (np.asarray(distr, f) + f(scale) * ou).astype(f)


def reset_when_done(data, ref, done, force_reset=False):
    """reset.cu:9-63: rows of finished replicas are restored from the reference copy."""
    mask = np.ones(len(done), dtype=bool) if force_reset else (np.asarray(done) > 0)
    out = np.array(data, copy=True)
    out[mask] = np.asarray(ref)[mask]
    return out


def undo_done_flag_and_reset_timestep(done, timestep, force_reset=False):
    """reset.cu:65-75"""
    mask = np.ones(len(done), dtype=bool) if force_reset else (np.asarray(done) > 0)
    d, t = np.array(done, copy=True), np.array(timestep, copy=True)
    d[mask] = 0
    t[mask] = 0
    return d, t


# ---------------------------------------------------------------------------------------------
# Counter-based uniform draws of the HIP sampler (Philox4x32-10, Salmon et al. SC'11).  The
# reference draws from curand's per-thread XORWOW state (random.cu:14-23,72), which has no CPU
# counterpart in the reference; this restates OUR generator so the device sampler can be checked
# draw-for-draw: same (seed, row, epoch, stream tag) -> same u -> same action index.
# ---------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays c0..c3; k0, k1 scalars.  Returns 4 uint32 arrays."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = c0 * np.uint64(M0), c2 * np.uint64(M1)
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & 0xFFFFFFFF, p1 >> np.uint64(32), p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01_open_closed(bits):
    """uint32 -> float32 uniform in (0, 1] with 24 random bits (curand_uniform's range)."""
    return ((np.asarray(bits, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)


def fused_tick_uniforms(n_rows, epochs, seed_lo, seed_hi, stream_tag):
    """The two uniforms (head 0, head 1) the fused tick kernel draws for every agent row:
    one Philox call, counter (row, epoch, stream_tag, 3), key (seed_lo, seed_hi)."""
    rows = np.arange(n_rows, dtype=np.uint32)
    x, y, _, _ = philox4x32_10(rows, np.asarray(epochs, dtype=np.uint32), np.uint32(stream_tag), np.uint32(3),
                               seed_lo, seed_hi)
    return u01_open_closed(x), u01_open_closed(y)


def single_head_tick_uniform(n_rows, epochs, seed_lo, seed_hi, stream_tag):
    """The uniform the fused tick of a SINGLE-head env (TagGridWorld, Cartpole) draws for every agent row:
    word (epoch & 3) of the Philox block with counter (row, epoch >> 2, stream_tag, 4), key (seed_lo, seed_hi)
    (csrc/kernels/wd_common.h::wd_tick_draw: a T-tick launch needs one Philox call per four ticks)."""
    rows = np.arange(n_rows, dtype=np.uint32)
    ep = np.broadcast_to(np.asarray(epochs, dtype=np.uint32), rows.shape)
    words = philox4x32_10(rows, ep >> np.uint32(2), np.uint32(stream_tag), np.uint32(4), seed_lo, seed_hi)
    sel = (ep & np.uint32(3)).astype(np.int64)
    bits = np.choose(sel, [np.asarray(w, dtype=np.uint32) for w in words])
    return u01_open_closed(bits.astype(np.uint32))
