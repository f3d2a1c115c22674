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


def test_c5_paired_long_reads_vs_reference_md5(gpu):
    d = G.tmpdir()
    M.write_c5_paired(d)
    gold = json.load(open(os.path.join(G.GOLD, "c5_paired_md5.json")))
    body = M.run_c5_paired(M.GPU_EXE, d, threads="4")
    assert M.md5(body) == gold["md5"]
