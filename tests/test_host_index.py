"""CPU tests: the host index builder / loader must reproduce the reference's index bit for bit."""
import hashlib
import os

import numpy as np
import pytest

from alfalfa_amd import ffi
import golden_util as G


@pytest.mark.parametrize("K", [1, 12])
def test_lambda_index_files_match_reference_md5(K):
    hidx = G.lambda_index(K)
    pre = os.path.join(G.tmpdir(), "mine_lam%d" % K)
    hidx.save(pre)
    gold = G.index_md5()["lambda_K%d" % K]
    for ext, want in gold.items():
        assert G.md5_file(pre + "." + ext) == want, ext
    # load what we wrote: identical arrays
    again = ffi.HostIndex.load(pre)
    a, b = hidx.arrays(), again.arrays()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert again.descr(0) == hidx.descr(0) == "gi|9626243|ref|NC_001416.1|"


@pytest.mark.parametrize("K", [1, 3, 12])
def test_synthetic_contigs_arrays_match_reference_md5(K):
    import seqgen
    rng = np.random.default_rng(4242)
    a = seqgen.rand_ref(rng, 50000)
    b = seqgen.rand_ref(rng, 20000)
    c = a[2000:6000] + b"n" * 20 + b[100:3100] + b"acgt" * 200 + a[2000:6000] + b"a" * 900
    hidx = ffi.HostIndex.build_mem(["ctgA", "ctgB", "ctgC"], [a, b, c], K=K)
    arr = hidx.arrays()
    gold = G.index_md5()["synthetic3_K%d" % K]
    for k, want in gold.items():
        assert hashlib.md5(np.ascontiguousarray(arr[k]).tobytes()).hexdigest() == want, k
    assert hidx.view.nM > 0  # LCP overflow table exercised
    assert hidx.chrom_bounds(10) == (0, 50000)
    assert hidx.chrom_bounds(50001 + 5) == (50001, 70001)


def test_load_missing_index_fails():
    with pytest.raises(ffi.AlfaError):
        ffi.HostIndex.load(os.path.join(G.tmpdir(), "does_not_exist"))
