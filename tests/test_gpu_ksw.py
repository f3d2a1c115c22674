"""GPU parity: the HIP banded-DP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import dpgen

pytestmark = pytest.mark.gpu


def _cmp_global(g, c, pairs, ws):
    sg, cg, ng, og = g
    sc, cc, nc, oc = c
    assert np.array_equal(sg, sc)
    for i, (q, t) in enumerate(pairs):
        if len(t) > len(q) + ws[i]:
            # reference indexes outside its backtrack matrix here (unspecified): the library reports n_cigar=-1
            assert ng[i] == -1
            continue
        assert ng[i] == nc[i], (i, len(q), len(t), ws[i])
        assert np.array_equal(cg[og[i]:og[i] + ng[i]], cc[oc[i]:oc[i] + nc[i]]), (i, len(q), len(t), ws[i])


@pytest.mark.parametrize("seed", [1, 2])
def test_random_bag(gpu, seed):
    pairs, ws, hs = dpgen.random_tasks(seed, 1500)
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    e = ffi.ksw_extend_batch(P, tasks)
    eo = O.orc_ksw_extend_batch(P, tasks)
    bad = np.nonzero(e != eo)[0]
    assert len(bad) == 0, [(int(i), len(pairs[i][0]), len(pairs[i][1]), ws[i], hs[i], e[i], eo[i]) for i in bad[:5]]
    n = ffi.ksw_nw_batch(P, tasks)
    no = O.orc_ksw_nw_batch(P, tasks)
    bad = np.nonzero(n != no)[0]
    assert len(bad) == 0, [(int(i), len(pairs[i][0]), len(pairs[i][1]), ws[i], n[i], no[i]) for i in bad[:5]]
    _cmp_global(ffi.ksw_global_batch(P, tasks), O.orc_ksw_global_batch(P, tasks), pairs, ws)


@pytest.mark.parametrize("band,rl,err", [(21, 250, 0.02), (100, 1500, 0.1), (60, 700, 0.05)])
def test_mapper_like(gpu, band, rl, err):
    pairs, ws, hs = dpgen.mapper_like_tasks(7, 600, rl, err, band)
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    assert np.array_equal(ffi.ksw_extend_batch(P, tasks), O.orc_ksw_extend_batch(P, tasks))
    assert np.array_equal(ffi.ksw_nw_batch(P, tasks), O.orc_ksw_nw_batch(P, tasks))
    _cmp_global(ffi.ksw_global_batch(P, tasks), O.orc_ksw_global_batch(P, tasks), pairs, ws)


def test_other_scoring(gpu):
    pairs, ws, hs = dpgen.random_tasks(5, 600)
    for (ma, mi, go, ge, zd) in [(2, -3, -5, -2, 50), (1, -1, -1, -1, 0), (3, -2, 0, -1, 20)]:
        P = ffi.default_params(ma, mi, go, ge, zd)
        tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
        assert np.array_equal(ffi.ksw_extend_batch(P, tasks), O.orc_ksw_extend_batch(P, tasks))
        assert np.array_equal(ffi.ksw_nw_batch(P, tasks), O.orc_ksw_nw_batch(P, tasks))
        _cmp_global(ffi.ksw_global_batch(P, tasks), O.orc_ksw_global_batch(P, tasks), pairs, ws)


def test_empty_batch(gpu):
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks([], 0, 0)
    assert len(ffi.ksw_extend_batch(P, tasks)) == 0
    assert len(ffi.ksw_nw_batch(P, tasks)) == 0


def test_extend_dead_boundary_cells(gpu):
    """ksw_extend with h0 too small for the band: row -1 holds zeros inside the band (dead cells, ksw.c:403 extends nothing out
    of them).  Low-complexity sequences make the diagonals that start in a dead cell score as well as the main one -- the
    band-in-registers kernel must not revive them (it gives such tasks to the moving-band code)."""
    rng = np.random.default_rng(5)
    pairs, ws, hs = [], [], []
    for _ in range(6000):
        h0, w, n = int(rng.integers(1, 30)), int(rng.integers(1, 24)), int(rng.integers(6, 60))
        a = int(rng.integers(0, 4))
        x = (a + 1 + int(rng.integers(0, 3))) % 4
        unit = rng.integers(0, 4, int(rng.integers(1, 4)))
        body = np.resize(unit, n)
        body = np.where(rng.random(n) < 0.1, rng.integers(0, 4, n), body)
        lead = int(rng.integers(1, w + 1))
        q = np.concatenate([[x], [a] * lead, body]).astype(np.uint8)
        sh = int(rng.integers(1, lead + 1))
        t = np.concatenate([[a], q[sh + 1:], rng.integers(0, 4, int(rng.integers(0, 6)))]).astype(np.uint8)
        pairs.append((q, t)); ws.append(w); hs.append(h0)
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    e, eo = ffi.ksw_extend_batch(P, tasks), O.orc_ksw_extend_batch(P, tasks)
    bad = np.nonzero(e != eo)[0]
    assert len(bad) == 0, [(int(i), ws[i], hs[i], e[i], eo[i]) for i in bad[:5]]


def test_lazy_join_overlapped_batches(gpu):
    """alfa_set_lazy_join: six resident batches (two per ksw kind, short and long tasks) launched back to back, twice, with
    no synchronisation in between -- their kernel bins run on side streams that are only joined at the first fetch -- and every
    result equals the oracle's."""
    P = ffi.default_params()
    sets = []
    for seed, (rl, err, band) in enumerate([(250, 0.02, 21), (3000, 0.1, 100)]):
        pairs, ws, hs = dpgen.mapper_like_tasks(40 + seed, 300, rl, err, band)
        tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
        sets.append((pairs, ws, tasks, keep))
    ffi.set_lazy_join(0, True)
    try:
        batches = [(mode, k, ffi.DpBatch(mode, P, sets[k][2], 0)) for k in range(2) for mode in (ffi.DP_EXTEND, ffi.DP_NW, ffi.DP_GLOBAL)]
        for _ in range(2):
            for _, _, b in batches:
                b.launch()
        got = [(mode, k, b.fetch()) for mode, k, b in batches]
    finally:
        ffi.set_lazy_join(0, False)
    for mode, k, res in got:
        pairs, ws, tasks, _ = sets[k]
        if mode == ffi.DP_EXTEND:
            assert np.array_equal(res, O.orc_ksw_extend_batch(P, tasks))
        elif mode == ffi.DP_NW:
            assert np.array_equal(res, O.orc_ksw_nw_batch(P, tasks))
        else:
            _cmp_global(res, O.orc_ksw_global_batch(P, tasks), pairs, ws)


@pytest.mark.parametrize("band", [40, 100])
def test_extend_large_scores_two_reduction_form(gpu, band):
    """ksw_extend tasks whose scores can pass 2^22 (here through h0) take the wave kernel's two-reduction form of the row
    maximum instead of the packed (H << 8 | diagonal) key; bands of 81 / 201 diagonals = the 128- and 256-diagonal kernels."""
    pairs, ws, hs = dpgen.mapper_like_tasks(90 + band, 200, 600, 0.08, band)
    hs = [5_000_000 + 7 * i for i in range(len(pairs))]
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    e = ffi.ksw_extend_batch(P, tasks)
    eo = O.orc_ksw_extend_batch(P, tasks)
    bad = np.nonzero(e != eo)[0]
    assert len(bad) == 0, [(int(i), len(pairs[i][0]), len(pairs[i][1]), ws[i], hs[i], e[i], eo[i]) for i in bad[:5]]


def test_kernel_ms_under_lazy_join_spans_the_batch(gpu):
    """alfa_dp_batch_kernel_ms under alfa_set_lazy_join: the closing event is recorded at the fetch-time join, so the figure
    covers the batch's kernels (not the near-zero interval between two records on the caller's stream) and is 0 before the fetch."""
    P = ffi.default_params()
    pairs, ws, hs = dpgen.mapper_like_tasks(77, 4000, 3000, 0.1, 100)
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    b = ffi.DpBatch(ffi.DP_NW, P, tasks, 0)
    b.launch(); b.fetch()                      # warm-up
    b.launch(); b.fetch()
    plain = b.kernel_ms
    assert plain > 0
    ffi.set_lazy_join(0, True)
    try:
        b.launch()
        assert b.kernel_ms == 0                # launched, side chains not joined yet: no closing event
        b.fetch()
        lazy = b.kernel_ms
    finally:
        ffi.set_lazy_join(0, False)
    assert 0.8 * plain <= lazy <= 1.5 * plain + 0.2, (plain, lazy)


def test_all_streams_of_a_context_share_one_priority(gpu):
    """csrc/device_ctx.hip: stream priorities are prohibited (cross-priority event waits can starve)"""
    import ctypes as C
    lo, hi = C.c_int(0), C.c_int(0)
    ffi.check(ffi.lib().alfa_stream_priority_span(0, C.byref(lo), C.byref(hi)))
    assert lo.value == hi.value
