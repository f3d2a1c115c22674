"""C5-shaped inputs (BASELINE configs[4]: 3 Gbp human-scale reference, paired 5 kbp long reads), at what one test can afford:

* a single text of 2.3 x 10^9 bytes -- text positions beyond 2^31, which a 32-bit signed offset anywhere on the device would
  corrupt -- at sparseness K = 256 (the coarsest index that still finds 2 kbp reads; the host build takes ~30 s), seeded
  with reads drawn from the upper half of the text and compared with the oracle on the same host arrays;
* paired 5 kbp reads (insert 12 000 +- 2 000, 10 % error, -I 8000 -X 16000 -e 0.12) through the product command line tool
  against the md5 of the compiled reference's SAM (tests/golden/c5_paired_md5.json, tools/make_c5_goldens.py).
"""
import json
import os

import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import mapper_util as M
import golden_util as G
import seqgen

pytestmark = pytest.mark.gpu

N_TEXT = 2300000000
K_TEXT = 256


def test_text_positions_beyond_2_31(gpu):
    rng = np.random.default_rng(3)
    ref = seqgen.ACGT[rng.integers(0, 4, N_TEXT, dtype=np.uint8)]
    reads = []
    for i in range(240):
        ln = int(rng.integers(1500, 3000))
        lo = (1 << 31) + (1 << 20) if i % 4 else 0               # three quarters of the reads from beyond 2^31
        p = int(rng.integers(lo, N_TEXT - ln - 1))
        r = seqgen.mutate(rng, ref[p:p + ln].tobytes(), 0.002)
        reads.append(seqgen.revcomp(r) if i % 3 == 0 else r)
    reads.append(ref[N_TEXT - 2000:].tobytes())                   # the very end of the text
    reads.append(ref[(1 << 31) - 1000:(1 << 31) + 1000].tobytes())   # straddles 2^31
    reads.append(ref[(1 << 32) - 2300000000 + 1000:][:10].tobytes() * 30)   # short period: no seed
    refb = ref.tobytes()
    del ref
    hidx = ffi.HostIndex.build_mem(["chr1"], [refb], K=K_TEXT)
    # the same index built on the GPU (index_build.hip): every array equal, with suffix positions beyond 2^31
    ffi.index_build_on_device(0)
    try:
        dev_built = ffi.HostIndex.build_mem(["chr1"], [refb], K=K_TEXT)
    finally:
        ffi.index_build_on_device(-1)
    a, b = hidx.arrays(), dev_built.arrays()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
    assert int(np.asarray(a["SA"]).max()) > (1 << 31)
    del dev_built, a, b
    del refb
    assert hidx.view.N > (1 << 31) and hidx.view.n == hidx.view.N // K_TEXT
    didx = ffi.DeviceIndex(hidx, 0)
    both = reads + [seqgen.revcomp(r) for r in reads]
    total = 0
    for mt, min_len, sq in ((ffi.SEED_SMAM, 300, 1), (ffi.SEED_MEM, 280, 1), (ffi.SEED_SMAM, 600, 2)):
        q = ffi.make_seed_queries(both, min_len, 10000, sq)
        m, off = didx.seed_batch(mt, q)
        mo, oo = O.orc_seed_batch(hidx.view, mt, q)
        assert np.array_equal(off, oo), (mt, min_len)
        assert np.array_equal(m, mo), (mt, min_len)
        total += len(m)
        assert (m["ref"] >= (1 << 31)).sum() > 100, "no seed beyond 2^31: the test does not test what it says"
    assert total > 1000


def _note(msg):
    """progress line past pytest's capture: a silent GPU job is taken for hung after three minutes"""
    os.write(2, ("[c5] " + msg + "\n").encode())


def test_c5_three_gbp_k12_index_built_on_device(gpu):
    """BASELINE configs[4] at size: 3 x 10^9 bp iid text (PRNG seed 3), sparseness K = 12 -- 2.5 x 10^8 sampled suffixes, the
    arrays built ON THE GPU (index_build.hip; the host builder would take minutes) and ~5.4 GB resident.  Checked: (i) the
    suffix array and LCP values of 4 000 random neighbouring ranks against the text itself, (ii) seeds of 5 kbp reads at 10 %
    error (`-e 0.12` parameters), three quarters of them drawn beyond 2^31, against the oracle on the same arrays, order
    included, (iii) every match exact / maximal / where the read was drawn from."""
    N, K = 3000000000, 12
    rng = np.random.default_rng(3)
    ref = seqgen.ACGT[rng.integers(0, 4, N, dtype=np.uint8)]
    _note("text generated")
    reads, origin = [], []
    for i in range(160):
        ln = 5000
        lo = (1 << 31) + (1 << 20) if i % 4 else 0
        p = int(rng.integers(lo, N - ln - 1))
        r = seqgen.mutate(rng, ref[p:p + ln].tobytes(), 0.10)
        reads.append(seqgen.revcomp(r) if i % 3 == 0 else r)
        origin.append(p)
    reads.append(ref[N - 5000:].tobytes())
    reads.append(ref[(1 << 31) - 2500:(1 << 31) + 2500].tobytes())
    refb = ref.tobytes()
    ffi.index_build_on_device(0)
    try:
        hidx = ffi.HostIndex.build_mem(["chr1"], [refb], K=K)
    finally:
        ffi.index_build_on_device(-1)
    del refb
    _note("index built on the device")
    v = hidx.view
    assert v.N >= N and v.K == K and v.n == v.N // K and v.n > 249000000
    arr = hidx.arrays()
    SA, LCP = np.asarray(arr["SA"]), np.asarray(arr["LCP"])
    assert int(SA.max()) > (1 << 31)
    # (i) neighbouring suffixes: sorted, and LCP[i] = length of their common prefix (capped by the byte's 255 overflow marker)
    for i in rng.integers(1, v.n, 4000):
        a, b = int(SA[i - 1]), int(SA[i])
        assert a % K == 0 and b % K == 0
        x, y = ref[a:a + 400], ref[b:b + 400]
        m = min(len(x), len(y))
        neq = np.nonzero(x[:m] != y[:m])[0]
        l = int(neq[0]) if len(neq) else m
        assert l < m and x[l] < y[l], int(i)
        assert int(LCP[i]) == min(l, 255), int(i)
    del SA, LCP, arr
    _note("suffix order and LCP verified on a sample")
    didx = ffi.DeviceIndex(hidx, 0)
    both = reads + [seqgen.revcomp(r) for r in reads]
    prm = [seqgen.seed_params(K, len(r), 0.13) for r in both]
    q = ffi.make_seed_queries(both, [a for a, _ in prm], 10000, [b for _, b in prm])
    m, off = didx.seed_batch(ffi.SEED_SMAM, q)
    _note("seeds on the device: %d matches" % len(m))
    assert (m["ref"] >= (1 << 31)).sum() > 200
    # (iii) exact and maximal
    for k in range(len(both)):
        r = np.frombuffer(both[k], dtype=np.uint8)
        for x in m[off[k]:off[k + 1]]:
            rp, qp, ln = int(x["ref"]), int(x["query"]), int(x["len"])
            assert ln >= prm[k][0] - K and np.array_equal(r[qp:qp + ln], ref[rp:rp + ln])
            assert qp + ln == len(r) or rp + ln >= N or r[qp + ln] != ref[rp + ln]
            assert qp == 0 or rp == 0 or r[qp - 1] != ref[rp - 1]
    # ... and at least one seed of (nearly) every simulated read lies on its diagonal of origin
    hit = 0
    for k in range(160):
        fw_strand = k if k % 3 else k + len(reads)          # the strand that reads like the text
        mm = m[off[fw_strand]:off[fw_strand + 1]]
        hit += bool(np.any(np.abs(mm["ref"] - mm["query"] - origin[k]) < 600))
    assert hit >= 150, hit
    # (ii) the oracle on the same arrays, order included
    mo, oo = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, q)
    _note("oracle done")
    assert np.array_equal(off, oo) and np.array_equal(m, mo)


def test_c5_paired_long_reads_vs_reference_md5(gpu):
    d = G.tmpdir()
    M.write_c5_paired(d)
    gold = json.load(open(os.path.join(G.GOLD, "c5_paired_md5.json")))
    body = M.run_c5_paired(M.GPU_EXE, d, threads="4")
    assert M.md5(body) == gold["md5"]
