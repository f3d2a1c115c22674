"""GPU parity for ksw_align (src/ksw.c:330): alfa_ksw_align_batch through the C ABI vs the oracle and vs the
reference's own outputs (tests/golden/ksw_align.json.gz)."""
import numpy as np
import pytest

from alfalfa_amd import ffi
import dpgen
import golden_util as G
import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_align_reference_goldens(gpu):
    assert G.check_align(ffi.ksw_align_batch) > 2000


@pytest.mark.parametrize("si", range(len(dpgen.ALIGN_SCORES)))
def test_align_random_vs_oracle(gpu, si):
    sc = dpgen.ALIGN_SCORES[si]
    p = dpgen.align_params(sc)
    pairs, xs = dpgen.align_cases(4000 + si, 1500, match=sc[0])
    tasks, keep = ffi.make_align_tasks(pairs, xs)
    got = ffi.ksw_align_batch(p, tasks)
    exp = O.orc_ksw_align_batch(p, tasks)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, (sc, int(bad[0]), len(pairs[bad[0]][0]), len(pairs[bad[0]][1]), hex(xs[bad[0]]), got[bad[0]], exp[bad[0]])


def test_align_long_queries_use_the_hbm_variant(gpu):
    """queries beyond the LDS classes (byte kernel > 1024 positions, 16-bit kernel > 512)"""
    sc = (1, -4, 6, 1)
    p = dpgen.align_params(sc)
    pairs, xs = dpgen.align_cases(4100, 60, max_q=3000, max_t=4000)
    xs = [x | (dpgen.XBYTE if i % 3 == 0 else 0) for i, x in enumerate(xs)]
    tasks, keep = ffi.make_align_tasks(pairs, xs)
    assert np.array_equal(ffi.ksw_align_batch(p, tasks), O.orc_ksw_align_batch(p, tasks))


def test_align_mate_rescue_shape(gpu):
    """the mapper's call: 100..250-bp mate against a window of a few hundred bases, negative gap numbers, own xtra"""
    rng = np.random.default_rng(77)
    sc = dpgen.ALIGN_SCORES[0]
    p = dpgen.align_params(sc)
    pairs, xs = [], []
    for _ in range(20000):
        ql = int(rng.choice([100, 150, 250]))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        w = rng.integers(0, 4, int(rng.integers(ql, 3 * ql + 200))).astype(np.uint8)
        if rng.random() < 0.8:
            c = dpgen.mutate(rng, q, 0.03)
            o = int(rng.integers(0, max(1, len(w) - len(c))))
            w[o:o + len(c)] = c[:len(w) - o]
        pairs.append((q, w))
        xs.append(dpgen.XSTART | (dpgen.XBYTE if ql * sc[0] < 250 else 0))
    tasks, keep = ffi.make_align_tasks(pairs, xs)
    got = ffi.ksw_align_batch(p, tasks)
    assert np.array_equal(got, O.orc_ksw_align_batch(p, tasks))
    assert (got["score"] > 80).mean() > 0.4 and (got["tb"] >= 0).mean() > 0.9


def test_align_rejects(gpu):
    p = dpgen.align_params((1, -4, 6, 1))
    q = np.zeros(10, np.uint8)
    with pytest.raises(ffi.AlfaError):
        ffi.ksw_align_batch(p, ffi.make_align_tasks([(np.full(4, 7, np.uint8), q)], 0)[0])        # residue >= m
    with pytest.raises(ffi.AlfaError):
        ffi.ksw_align_batch(p, ffi.make_align_tasks([(np.zeros(40000, np.uint8), q)], 0)[0])      # beyond the 16-bit score range
    assert len(ffi.ksw_align_batch(p, ffi.make_align_tasks([], 0)[0])) == 0
