"""Host-side replay of the lane arithmetic of HipTagGridWorldRollout_N5 (csrc/kernels/tag_gridworld_n5.hip): the
pieces that replace the general kernel's LDS tables and barriers.  The kernel itself is checked on the device by
tests/test_gpu_gridworld.py::test_gridworld_rollout_records_every_tick; this file pins the index arithmetic."""
import numpy as np

N, F, EPB = 5, 21, 12


def test_lane_to_replica_without_a_division():
    # the restore loop finds the replica of the lowest finished lane as (lane * 13) >> 6
    for lane in range(EPB * N):
        assert (lane * 13) >> 6 == lane // N, lane


def test_tag_check_by_ballot_and_shift():
    """cell = x | y << 8; every lane compares its cell with its replica's runner's (lane 5 e + 4, through ds_bpermute);
    ballot over (active, tagger, equal); replica e is tagged iff bits 5 e .. 5 e + 3 of the ballot are not all zero"""
    rng = np.random.default_rng(0)
    for trial in range(200):
        envs_here = int(rng.integers(1, EPB + 1))
        x = rng.integers(0, 4, size=(EPB, N))
        y = rng.integers(0, 4, size=(EPB, N))
        cell = (x | (y << 8)).reshape(-1)
        lanes = np.arange(64)
        el, ag = lanes // N, lanes % N
        active = el < envs_here
        cell64 = np.zeros(64, np.int64)
        cell64[:EPB * N] = cell
        runner_lane = np.minimum(el * N + N - 1, 63)
        hit = active & (ag < N - 1) & (cell64 == cell64[runner_lane])
        ballot = int(sum(1 << int(l) for l in lanes[hit]))
        for e in range(envs_here):
            want = any(x[e, j] == x[e, N - 1] and y[e, j] == y[e, N - 1] for j in range(N - 1))  # tag_gridworld.py:175-178
            got = ((ballot >> (e * N)) & 0xF) != 0
            assert got == want, (trial, e)


def test_record_copy_covers_the_block_slice_once():
    # five 16-byte vectors per lane, lane + 64 i, predicated on < nvec; reads are clamped into the image
    for envs_here in (4, 8, 12):
        nvec = envs_here * N * F // 4
        seen = np.zeros(EPB * N * F // 4, int)
        for lane in range(64):
            for i in range(5):
                q = lane + 64 * i
                assert min(q, EPB * N * F // 4 - 1) < EPB * N * F // 4
                if q < nvec:
                    seen[q] += 1
        assert (seen[:nvec] == 1).all() and (seen[nvec:] == 0).all()


def test_quotient_tables_hold_the_reference_expressions():
    """x / L and t / episode_length (tag_gridworld.py:208-214, :273) take few values: the kernel tabulates them with the
    same float32 division once per launch, so a lookup returns what the division would"""
    f32 = np.float32
    for L in (6, 10, 63):
        table = (np.arange(L + 1, dtype=f32) / f32(L)).astype(f32)
        for c in range(L + 1):
            assert table[c] == f32(f32(c) / f32(L))
    for T in (23, 100, 4095):
        table = (np.arange(T + 1, dtype=f32) / f32(T)).astype(f32)
        assert all(table[t] == f32(f32(t) / f32(T)) for t in (0, 1, T // 2, T))


def test_lds_size_formula_matches_the_kernel_layout():
    # image [12][105], restore cache [12][CD], 64 coordinate quotients, episode_length + 1 time quotients
    CD, T = 5 + 5 + N * F, 100
    floats = EPB * N * F + EPB * CD + 64 + T + 1
    assert (EPB * N * F) % 4 == 0 and (EPB * CD) % 4 == 0  # the tables behind the image stay 16-byte aligned
    assert 4 * floats <= 60000
