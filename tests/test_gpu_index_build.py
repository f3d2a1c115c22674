"""GPU parity: the index arrays built on the device (alfa_index_build_on_device: suffix array by prefix doubling, LCP,
CHILD by nearest-smaller-value queries, 10-mer table) equal the host builder's bit for bit -- and the host builder's are
pinned to the reference's files by md5 (tests/test_host_index.py, tests/golden/index_md5.json)."""
import os

import numpy as np
import pytest

from alfalfa_amd import ffi
import golden_util as G
import seqgen

pytestmark = pytest.mark.gpu


def _arrays(h):
    a = h.arrays()
    return {k: (np.asarray(v).copy() if v is not None else None) for k, v in a.items()}


def _same(a, b, what):
    for k in a:
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, (what, k)
        elif isinstance(a[k], np.ndarray):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (what, k)
        else:
            assert a[k] == b[k], (what, k)


def _build_both(names, seqs, **kw):
    ffi.index_build_on_device(-1)
    host = ffi.HostIndex.build_mem(names, seqs, **kw)
    ffi.index_build_on_device(0)
    try:
        dev = ffi.HostIndex.build_mem(names, seqs, **kw)
    finally:
        ffi.index_build_on_device(-1)
    return host, dev


@pytest.mark.parametrize("K", [1, 2, 3, 12, 20])
def test_device_build_random_and_repeats(gpu, K):
    rng = np.random.default_rng(100 + K)
    a = seqgen.rand_ref(rng, 60000)
    rep = seqgen.rand_ref(rng, 700)
    # tandem and dispersed repeats (lcp values far beyond 255), a homopolymer run, n runs, two contigs
    s1 = a[:20000] + rep * 6 + a[20000:40000] + b"a" * 900 + a[40000:] + rep + b"n" * 300 + a[1000:3000]
    s2 = seqgen.rand_ref(rng, 5000) + a[500:2500]
    for kw in ({"K": K}, {"K": K, "suflink": True}, {"K": K, "child": False, "kmer": False}):
        host, dev = _build_both(["c1", "c2 x"], [s1, s2], **kw)
        _same(_arrays(host), _arrays(dev), (K, kw))


def test_device_build_lambda_files_match_reference_md5(gpu):
    """the files written from device-built arrays carry the md5 of the reference's own index files"""
    fa = G.input_path("reference.fasta")
    for K in (1, 12):
        ffi.index_build_on_device(0)
        try:
            h = ffi.HostIndex.build_fasta(fa, K=K)
        finally:
            ffi.index_build_on_device(-1)
        pre = os.path.join(G.tmpdir(), "dev_lam%d" % K)
        h.save(pre)
        for ext, want in G.index_md5()["lambda_K%d" % K].items():
            assert G.md5_file(pre + "." + ext) == want, (K, ext)


def test_device_build_c3_size_and_search(gpu):
    """5 Mbp / K = 12 (the C3 index): same arrays as the host builder, and seeds found on the device-built index equal the oracle's"""
    import oracle_ffi as O
    rng = np.random.default_rng(1)
    ref = seqgen.rand_ref(rng, 5000000)
    host, dev = _build_both(["chr1"], [ref], K=12)
    _same(_arrays(host), _arrays(dev), "c3")
    reads = seqgen.sim_reads(rng, ref, 200, 250, 250, 0.02)
    q = ffi.make_seed_queries(reads, 40, 10000, 2)
    m, off = ffi.DeviceIndex(dev, 0).seed_batch(ffi.SEED_SMAM, q)
    mo, oo = O.orc_seed_batch(host.view, ffi.SEED_SMAM, q)
    assert np.array_equal(off, oo) and np.array_equal(m, mo)


def test_cli_gpu_index_files(gpu):
    """`alfalfa_amd index --gpu-index`: the files written carry the md5 of the reference's own index files (lambda, K = 12)"""
    import subprocess
    import mapper_util as M
    fa = G.input_path("reference.fasta")
    pre = os.path.join(G.tmpdir(), "cli_gpu_lam12")
    p = subprocess.run([M.GPU_EXE, "index", "-r", fa, "-p", pre, "-s", "12", "--gpu-index"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    for ext, want in G.index_md5()["lambda_K12"].items():
        assert G.md5_file(pre + "." + ext) == want, ext


def test_index_files_straight_to_device(gpu, tmp_path):
    """alfa_index_load_to_device: the arrays go from the files through pinned staging to HBM, the 10-mer companions are derived on
    the device; the resident copy answers every kind of seed query, and maps whole reads, exactly as the copy alfa_index_upload
    makes from host arrays -- child-table, suffix-link and no-child / no-kmer indexes, two contigs"""
    import seqgen
    from alfalfa_amd import ffi
    rng = np.random.default_rng(31)
    seqs = [seqgen.rand_ref(rng, 700000), seqgen.rand_ref(rng, 300000)]
    text = b"`".join(seqs)
    reads = []
    for _ in range(400):
        p = int(rng.integers(0, len(text) - 400))
        reads.append(seqgen.mutate(rng, text[p:p + 300].replace(b"`", b"a"), 0.03))
    for k, (suflink, child, kmer) in enumerate(((False, True, True), (True, True, True), (False, False, False))):
        hidx = ffi.HostIndex.build_mem(["c1", "c2"], seqs, K=3 if suflink else 12, suflink=suflink, child=child, kmer=kmer)
        pre = str(tmp_path / ("idx%d" % k))
        hidx.save(pre)
        up = ffi.DeviceIndex(hidx, 0)
        direct = ffi.DeviceIndex.load(pre, 0)
        assert direct.host.view.N == hidx.view.N and direct.host.view.n_seq == 2 and direct.host.descr(1) == "c2"
        second = ffi.DeviceIndex.load(pre, 0, host_index=direct.host)          # (a second GPU: the host's share is reused)
        for mt in (ffi.SEED_SMAM, ffi.SEED_MEM) + ((ffi.SEED_SMEM,) if child else ()):
            q = ffi.make_seed_queries(reads, 25, 10000, 1)
            m0, o0 = up.seed_batch(mt, q)
            for other in (direct, second):
                m1, o1 = other.seed_batch(mt, q)
                assert np.array_equal(o0, o1) and np.array_equal(m0, m1), (k, mt)
        if child and kmer and not suflink:
            opt = ffi.default_map_options()
            P = ffi.default_params()
            mr = ffi.make_map_reads(reads, [seqgen.revcomp(r) for r in reads], 12, opt)
            for x, y in zip(ffi.map_reads_batch(up, opt, P, mr), ffi.map_reads_batch(direct, opt, P, mr)):
                assert np.array_equal(x, y)
    with pytest.raises(ffi.AlfaError):
        ffi.DeviceIndex.load(str(tmp_path / "nothing_here"), 0)
