"""CPU tests (no GPU): pin the oracle against the golden vectors generated from the compiled reference."""
import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import golden_util as G


def _seed_fn(hidx, mt, q):
    return O.orc_seed_batch(hidx.view, mt, q)


@pytest.mark.parametrize("K", [1, 12])
def test_oracle_seeds_all_types(K):
    assert G.check_seeds(_seed_fn, K) > 1000


def test_oracle_ksw_mapper_calls():
    assert G.check_ksw_calls(O.orc_ksw_extend_batch, O.orc_ksw_nw_batch, O.orc_ksw_global_batch) > 3000


def test_oracle_ksw_random_edge_cases():
    assert G.check_ksw_random(O.orc_ksw_extend_batch, O.orc_ksw_nw_batch, O.orc_ksw_global_batch) == 1500


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (compiled reference) only exists in the build container")
def test_oracle_vs_live_reference_nochild_and_suflink():
    """index variants the fixtures do not cover: --no-child (binary-search traverse) and --suflink"""
    import seqgen
    rng = np.random.default_rng(77)
    ref = seqgen.rand_ref(rng, 40000)
    ref = ref + ref[1000:3000] + seqgen.rand_ref(rng, 5000)
    reads = seqgen.sim_reads(rng, ref, 40, 80, 600, 0.03)
    # the reference cannot build --no-child together with the 10-mer table (computeKmer's binary-search branch,
    # src/sparseSA.cpp:295-309, never terminates -> std::bad_alloc), so no-child implies --no-kmer here
    for K, suf, child, kmer in ((1, False, False, False), (3, True, True, True), (4, True, False, False), (12, False, False, False),
                                (12, False, True, False)):
        mine = ffi.HostIndex.build_mem(["x"], [ref], K=K, suflink=suf, child=child, kmer=kmer)
        rix = O.RefIndex.build_mem(["x"], [ref], K=K, suflink=suf, child=child, kmer=kmer)
        for mt in ((0, 1) if not child else (0, 1, 2, 3)):
            q = ffi.make_seed_queries(reads, 30, 1000, 1, 10)
            m, off = O.orc_seed_batch(mine.view, mt, q)
            for i, r in enumerate(reads):
                assert np.array_equal(m[off[i]:off[i + 1]], rix.seed(mt, r, 30, 1000, 1, 10)), (K, suf, child, mt, i)


def test_oracle_dynprog_vs_reference_goldens():
    assert G.check_dpband(O.orc_dp_band_batch) > 1000


def test_oracle_ksw_align_vs_reference_goldens():
    assert G.check_align(O.orc_ksw_align_batch) > 2000


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference only exists in the build container")
def test_oracle_ksw_align_vs_live_reference():
    import dpgen
    n = 0
    for si, sc in enumerate(dpgen.ALIGN_SCORES):
        p = dpgen.align_params(sc)
        pairs, xs = dpgen.align_cases(900 + si, 150, match=sc[0])
        tasks, keep = ffi.make_align_tasks(pairs, xs)
        got = O.orc_ksw_align_batch(p, tasks)
        for i, ((q, t), x) in enumerate(zip(pairs, xs)):
            if got[i]["qe"] >= len(q):      # undefined in the reference (it would reverse memory behind the query)
                continue
            assert tuple(int(v) for v in got[i]) == O.ref_ksw_align(p, q, t, x), (sc, i)
            n += 1
    assert n > 1000


def _odd_byte_reads(rng, ref, n):
    """reads with NUL bytes (and other non-ACGT bytes) sprinkled in: a NUL matches no text character in the reference"""
    import seqgen
    out = []
    for r in seqgen.sim_reads(rng, ref, n, 60, 400, 0.02):
        a = bytearray(r)
        for p in rng.integers(0, len(a), int(rng.integers(1, 4))):
            a[int(p)] = int(rng.choice([0, 0, 0, ord("n"), ord("$"), 1]))
        out.append(bytes(a))
    return out


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (compiled reference) only exists in the build container")
def test_oracle_vs_live_reference_reads_with_nul_bytes():
    import seqgen
    rng = np.random.default_rng(5)
    ref = seqgen.rand_ref(rng, 30000)
    reads = _odd_byte_reads(rng, ref, 60)
    for K in (1, 12):
        mine = ffi.HostIndex.build_mem(["x"], [ref], K=K)
        rix = O.RefIndex.build_mem(["x"], [ref], K=K)
        for mt in (0, 1, 2):
            q = ffi.make_seed_queries(reads, 30, 1000, 1, 10)
            m, off = O.orc_seed_batch(mine.view, mt, q)
            for i, r in enumerate(reads):
                assert np.array_equal(m[off[i]:off[i + 1]], rix.seed(mt, r, 30, 1000, 1, 10)), (K, mt, i)
