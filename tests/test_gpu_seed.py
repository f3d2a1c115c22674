"""GPU parity: the HIP seed kernels (SMAM, MEM, SMEM; through the C ABI) vs the CPU oracle, bit-exact and in order."""
import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import seqgen

pytestmark = pytest.mark.gpu


def _check(hidx, didx, reads, min_len, max_count, sq, mt=ffi.SEED_SMAM, max_smem=10):
    q = ffi.make_seed_queries(reads, min_len, max_count, sq, max_smem)
    mg, og = didx.seed_batch(mt, q)
    mc, oc = O.orc_seed_batch(hidx.view, mt, q)
    assert np.array_equal(og, oc), "per-read offsets differ"
    assert np.array_equal(mg, mc), "matches differ"
    return len(mg)


@pytest.mark.parametrize("K", [1, 3, 12])
def test_smam_random_ref(gpu, K):
    rng = np.random.default_rng(100 + K)
    ref = seqgen.rand_ref(rng, 300000)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=K)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 400, 100, 400, 0.03)
    reads += [b"", b"acgt", b"n" * 60, reads[0][:50] + b"n" + reads[0][50:]]
    ml = [seqgen.seed_params(K, len(r))[0] for r in reads]
    sq = [seqgen.seed_params(K, len(r))[1] for r in reads]
    n = _check(hidx, didx, reads, ml, 10000, sq)
    assert n > 300
    # rescue parameters of the mapper (src/mapper.cpp:544-553) and a tiny maxCount
    _check(hidx, didx, reads, max(20, K + 10), 100000, 1)
    _check(hidx, didx, reads, 25, 3, 1)


def test_smam_long_reads(gpu):
    rng = np.random.default_rng(3)
    ref = seqgen.rand_ref(rng, 1000000)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=12)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 60, 2000, 10000, 0.10)
    ml = [seqgen.seed_params(12, len(r), 0.13)[0] for r in reads]
    sq = [seqgen.seed_params(12, len(r), 0.13)[1] for r in reads]
    assert _check(hidx, didx, reads, ml, 10000, sq) > 50


def test_smam_repeats_multi_contig(gpu):
    rng = np.random.default_rng(4)
    a = seqgen.rand_ref(rng, 50000)
    b = seqgen.rand_ref(rng, 20000)
    c = a[2000:6000] + b"n" * 20 + b[100:3100] + b"acgt" * 200 + a[2000:6000] + b"a" * 900
    names, seqs = ["a", "b", "c"], [a, b, c]
    for K in (1, 4, 12):
        hidx = ffi.HostIndex.build_mem(names, seqs, K=K)
        didx = ffi.DeviceIndex(hidx, gpu)
        cat = b"a".join(seqs)
        reads = seqgen.sim_reads(rng, cat, 300, 60, 900, 0.02)
        reads += [b"acgt" * 60, b"a" * 300, c[3990:4300]]
        ml, sq = 40, seqgen.seed_params(K, 250)[1]
        assert _check(hidx, didx, reads, ml, 10000, sq) > 100
        _check(hidx, didx, reads, 25, 2, 1)


def test_smam_index_without_kmer_table(gpu):
    rng = np.random.default_rng(8)
    ref = seqgen.rand_ref(rng, 60000)
    hidx = ffi.HostIndex.build_mem(["x"], [ref], K=3, kmer=False)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 100, 60, 400, 0.03)
    assert _check(hidx, didx, reads, 30, 1000, 2) > 50


@pytest.mark.parametrize("mt", [ffi.SEED_MEM, ffi.SEED_SMEM, ffi.SEED_SMEMP])
@pytest.mark.parametrize("K", [1, 3, 12])
def test_mem_smem_random_ref(gpu, K, mt):
    rng = np.random.default_rng(200 + K)
    ref = seqgen.rand_ref(rng, 300000)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=K)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 300, 100, 400, 0.03)
    reads += [b"", b"acgt", b"n" * 60, reads[0][:50] + b"n" + reads[0][50:]]
    ml = [seqgen.seed_params(K, len(r))[0] for r in reads]
    sq = [seqgen.seed_params(K, len(r))[1] for r in reads]
    assert _check(hidx, didx, reads, ml, 10000, sq, mt) > 200
    _check(hidx, didx, reads, max(20, K + 10), 100000, 1, mt)
    _check(hidx, didx, reads, 25, 3, 1, mt, max_smem=2)
    if K > 1:
        _check(hidx, didx, reads, 50, 100, 3, mt, max_smem=1)     # sparseQ > 1: data-dependent prefix skipping in SMEM


@pytest.mark.parametrize("mt", [ffi.SEED_MEM, ffi.SEED_SMEM, ffi.SEED_SMEMP])
def test_mem_smem_repeats_multi_contig(gpu, mt):
    rng = np.random.default_rng(14)
    a = seqgen.rand_ref(rng, 50000)
    b = seqgen.rand_ref(rng, 20000)
    c = a[2000:6000] + b"n" * 20 + b[100:3100] + b"acgt" * 200 + a[2000:6000] + b"a" * 900
    names, seqs = ["a", "b", "c"], [a, b, c]
    for K in (1, 4, 12):
        hidx = ffi.HostIndex.build_mem(names, seqs, K=K)
        didx = ffi.DeviceIndex(hidx, gpu)
        cat = b"a".join(seqs)
        reads = seqgen.sim_reads(rng, cat, 200, 60, 900, 0.02)
        reads += [b"acgt" * 60, b"a" * 300, c[3990:4300]]
        assert _check(hidx, didx, reads, 40, 10000, seqgen.seed_params(K, 250)[1], mt) > 100
        _check(hidx, didx, reads, 25, 50, 1, mt, max_smem=3)


def test_smem_long_reads_use_global_tables(gpu):
    rng = np.random.default_rng(13)
    ref = seqgen.rand_ref(rng, 500000)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=12)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 30, 2000, 6000, 0.08)
    ml = [seqgen.seed_params(12, len(r), 0.10)[0] for r in reads]
    sq = [seqgen.seed_params(12, len(r), 0.10)[1] for r in reads]
    for mt in (ffi.SEED_MEM, ffi.SEED_SMEM, ffi.SEED_SMEMP):
        assert _check(hidx, didx, reads, ml, 10000, sq, mt) > 30


@pytest.mark.parametrize("mt", [ffi.SEED_SMAM, ffi.SEED_MEM])
def test_index_without_child_table(gpu, mt):
    rng = np.random.default_rng(9)
    ref = seqgen.rand_ref(rng, 80000)
    for K, kmer in ((1, False), (4, False)):
        hidx = ffi.HostIndex.build_mem(["x"], [ref], K=K, child=False, kmer=kmer)
        didx = ffi.DeviceIndex(hidx, gpu)
        reads = seqgen.sim_reads(rng, ref, 100, 60, 400, 0.03)
        assert _check(hidx, didx, reads, 30, 1000, 1, mt) > 50


@pytest.mark.parametrize("K", [1, 3, 12])
def test_smam_suffix_link_index(gpu, K):
    """--suflink index: the prefixes of a phase form a chain through suffix links (findSMAM src/sparseSA.cpp:1085-1112)"""
    rng = np.random.default_rng(300 + K)
    ref = seqgen.rand_ref(rng, 200000)
    rep = ref[5000:9000]
    ref = ref + rep + b"acgt" * 100 + rep
    hidx = ffi.HostIndex.build_mem(["x"], [ref], K=K, suflink=True)
    didx = ffi.DeviceIndex(hidx, gpu)
    reads = seqgen.sim_reads(rng, ref, 300, 80, 600, 0.03)
    reads += [b"", b"acgt" * 30, b"n" * 50, rep[100:400]]
    ml = [seqgen.seed_params(K, len(r))[0] for r in reads]
    sq = [seqgen.seed_params(K, len(r))[1] for r in reads]
    assert _check(hidx, didx, reads, ml, 10000, sq) > 200
    _check(hidx, didx, reads, max(20, K + 10), 100000, 1)
    _check(hidx, didx, reads, 25, 3, 1)
    assert _check(hidx, didx, reads, ml, 10000, sq, ffi.SEED_SMEM, 10) > 100      # SMEM does not use the links
    assert _check(hidx, didx, reads, ml, 10000, sq, ffi.SEED_MEM) > 200           # findMEM carries mli and xmi through the links
    _check(hidx, didx, reads, max(20, K + 10), 100000, 1, ffi.SEED_MEM)


def test_unsupported_combinations_fail_loudly(gpu):
    rng = np.random.default_rng(5)
    ref = seqgen.rand_ref(rng, 20000)
    hidx2 = ffi.HostIndex.build_mem(["x"], [ref], K=3, child=False, kmer=False)
    didx2 = ffi.DeviceIndex(hidx2, gpu)
    for mt in (ffi.SEED_SMEM, ffi.SEED_SMEMP):          # the reference calls traverse_faster (child table) unconditionally there
        with pytest.raises(ffi.AlfaError):
            didx2.seed_batch(mt, ffi.make_seed_queries([ref[:100]], 40, 100, 1))
    with pytest.raises(ffi.AlfaError):
        didx2.seed_batch(7, ffi.make_seed_queries([ref[:100]], 40, 100, 1))


def test_smam_read_bytes_of_every_kind(gpu):
    """the persistent kernel derives 2-bit codes four read bytes at a time: upper / lower case, IUPAC letters, punctuation and
    bytes >= 0x80 anywhere in a read must behave as in bitadd (src/sparseSA.h:38-63): only acgtACGT are residues"""
    rng = np.random.default_rng(77)
    ref = seqgen.rand_ref(rng, 200000)
    hidx = ffi.HostIndex.build_mem(["x"], [ref], K=12)
    didx = ffi.DeviceIndex(hidx, gpu)
    base = seqgen.sim_reads(rng, ref, 300, 90, 300, 0.01)
    odd = b"nNrRyY-*.@`{AaCcGgTt\x01\x7f\x80\xc1\xe3\xff5 "
    reads = []
    for k, r in enumerate(base):
        b = bytearray(r.upper() if k % 3 == 0 else r)
        if k % 2:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = odd[int(rng.integers(0, len(odd)))]
        if k % 5 == 0:
            for i in range(0, len(b), 2):
                b[i] = bytes([b[i]]).swapcase()[0] if bytes([b[i]]).isalpha() else b[i]
        reads.append(bytes(b))
    for plen in (9, 10, 11, 15, 16, 17, 23, 24, 25, 63, 64, 65):      # lengths around the 8-byte staging words
        reads.append(bytes(ref[1000:1000 + plen]))
    assert _check(hidx, didx, reads, 20, 10000, 1) > 150
    _check(hidx, didx, reads, 40, 10000, 2)


def test_reads_with_nul_bytes(gpu):
    """a NUL inside a read matches no text character (reference semantics, pinned against the compiled reference in
    tests/test_oracle_vs_golden.py); the kernels' NUL-padded chunk compares must not mistake it for the end of the text"""
    import seqgen
    from test_oracle_vs_golden import _odd_byte_reads
    rng = np.random.default_rng(5)
    ref = seqgen.rand_ref(rng, 30000)
    reads = _odd_byte_reads(rng, ref, 200) + [b"\x00" * 50, b"acgt" * 10 + b"\x00" + b"acgt" * 10]
    for K in (1, 12):
        hidx = ffi.HostIndex.build_mem(["x"], [ref], K=K)
        didx = ffi.DeviceIndex(hidx, 0)
        for mt in (ffi.SEED_SMAM, ffi.SEED_MEM, ffi.SEED_SMEM):
            q = ffi.make_seed_queries(reads, 30, 1000, 1, 10)
            m, off = didx.seed_batch(mt, q)
            mo, oo = O.orc_seed_batch(hidx.view, mt, q)
            assert np.array_equal(off, oo) and np.array_equal(m, mo), (K, mt)


def test_resident_objects_belong_to_their_thread(gpu):
    """launch / fetch of a resident plan or batch from another host thread is refused (they are ordered on the creator's streams)"""
    import threading
    import seqgen
    rng = np.random.default_rng(9)
    ref = seqgen.rand_ref(rng, 20000)
    hidx = ffi.HostIndex.build_mem(["x"], [ref], K=12)
    didx = ffi.DeviceIndex(hidx, 0)
    reads = seqgen.sim_reads(rng, ref, 20, 100, 200, 0.02)
    plan = ffi.SeedPlan(didx, ffi.SEED_SMAM, ffi.make_seed_queries(reads, 30, 1000, 1))
    P = ffi.default_params()
    a = rng.integers(0, 4, 40).astype(np.uint8)
    tasks, keep = ffi.make_dp_tasks([(a, a.copy())] * 8, 5, 0)
    batch = ffi.DpBatch(ffi.DP_NW, P, tasks, 0)
    errs = []

    def other():
        for fn in (plan.launch, plan.fetch, batch.launch, batch.fetch):
            try:
                fn()
                errs.append("accepted")
            except ffi.AlfaError as e:
                errs.append(str(e))
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert len(errs) == 4 and all("thread that created it" in e for e in errs), errs
    plan.launch()
    batch.launch()
    m, off = plan.fetch()
    assert np.array_equal(batch.fetch(), np.full(8, 40, np.int32))
    mo, oo = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, ffi.make_seed_queries(reads, 30, 1000, 1))
    assert np.array_equal(m, mo)


def test_short_lived_threads_give_their_memory_back(gpu):
    """a host thread's context (streams, staging buffers, recycled plan and batches) dies with the thread: calling the batch
    API from many short-lived threads must not accumulate HBM"""
    import ctypes as C
    import threading
    import hip_rt
    import seqgen
    rng = np.random.default_rng(10)
    ref = seqgen.rand_ref(rng, 200000)
    hidx = ffi.HostIndex.build_mem(["x"], [ref], K=12)
    didx = ffi.DeviceIndex(hidx, 0)
    reads = seqgen.sim_reads(rng, ref, 3000, 200, 300, 0.02)
    q = ffi.make_seed_queries(reads, 30, 1000, 1)
    P = ffi.default_params()
    a = rng.integers(0, 4, 300).astype(np.uint8)
    tasks, keep = ffi.make_dp_tasks([(a, a.copy())] * 20000, 20, 0)
    r = hip_rt.rt()
    r.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert r.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def work():
        didx.seed_batch(ffi.SEED_SMAM, q)
        ffi.ksw_nw_batch(P, tasks)
        ffi.ksw_global_batch(P, tasks)

    def once():
        t = threading.Thread(target=work)
        t.start()
        t.join()
    once()
    once()
    ffi.sync(0)
    before = free_bytes()
    for _ in range(6):
        once()
    ffi.sync(0)                 # any library call on a live thread releases what the finished threads left behind
    after = free_bytes()
    assert before - after < (8 << 20), (before, after)      # each thread used ~100 MB while it lived


def test_smam_sorted_path_on_a_big_index(gpu):
    """an index whose 10-mer table is full and that is far beyond the caches (40 Mbp at K = 12: 3.3 M sampled suffixes) takes the
    sorted path -- queries evaluated in 10-mer order, offsets from one scan over the flat counts -- with identical results"""
    import seqgen
    import oracle_ffi as O
    rng = np.random.default_rng(31)
    ref = seqgen.rand_ref(rng, 40000000)
    ffi.index_build_on_device(0)
    try:
        hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=12)
    finally:
        ffi.index_build_on_device(-1)
    didx = ffi.DeviceIndex(hidx, 0)
    reads = seqgen.sim_reads(rng, ref, 400, 100, 400, 0.03) + seqgen.sim_reads(rng, ref, 30, 2000, 9000, 0.10)
    reads += [b"", b"acgt", ref[:9], b"n" * 50, ref[5000:5200].replace(b"a", b"n", 3)]
    reads = reads + [seqgen.revcomp(r) for r in reads]
    for min_len, sq in ((20, 1), (40, 2), (60, 4)):
        q = ffi.make_seed_queries(reads, min_len, 10000, sq)
        plan = ffi.SeedPlan(didx, ffi.SEED_SMAM, q)
        assert plan.sorted
        plan.launch()
        m, off = plan.fetch()
        mo, oo = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, q)
        assert np.array_equal(off, oo), (min_len, sq)
        assert np.array_equal(m, mo), (min_len, sq)
        assert len(m) > 50
    # small maxCount (interval size filter) and the synchronous entry point
    q = ffi.make_seed_queries(reads, 25, 2, 1)
    m, off = didx.seed_batch(ffi.SEED_SMAM, q)
    mo, oo = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, q)
    assert np.array_equal(off, oo) and np.array_equal(m, mo)
