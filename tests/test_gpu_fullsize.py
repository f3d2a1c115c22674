"""GPU checks at the benchmark's sizes (BASELINE configs[2] shapes), where the oracle is too slow to replay everything:
size-independent properties of the outputs + exact oracle parity on random samples + determinism + (when the compiled
reference travelled with the tree) byte-identical SAM of the command line tool on 200 000 reads."""
import os
import subprocess
import sys

import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import seqgen
import mapper_util as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


def test_seeds_full_size_properties(gpu):
    K, nreads, rlen = 12, 300000, 250
    ref = bench.make_reference(5000000, 1)
    refb = ref.tobytes()
    hidx = ffi.HostIndex.build_mem(["chr1"], [refb], K=K)
    didx = ffi.DeviceIndex(hidx, gpu)
    fw, rc, _, _ = bench.simulate_batch(np.random.default_rng(5), ref, nreads, rlen, 0.02)
    reads = fw + rc
    min_len, sq = seqgen.seed_params(K, rlen)
    q = ffi.make_seed_queries(reads, min_len, 10000, sq)
    m, off = didx.seed_batch(ffi.SEED_SMAM, q)
    # shape
    assert off[0] == 0 and off[-1] == len(m) and np.all(np.diff(off) >= 0)
    assert len(m) > nreads            # about 2.8 seeds per read
    # determinism: the same call again gives the same bytes
    m2, off2 = didx.seed_batch(ffi.SEED_SMAM, q)
    assert np.array_equal(off, off2) and np.array_equal(m, m2)
    # every match is an exact match of at least min_len - K residues (sparseQ > 1), inside read and text
    owner = np.repeat(np.arange(len(reads)), np.diff(off))
    plen = np.array([len(r) for r in reads], dtype=np.int64)
    ln, qs, rs = m["len"].astype(np.int64), m["query"].astype(np.int64), m["ref"].astype(np.int64)
    assert np.all(ln >= min_len - K) and np.all(qs >= 0) and np.all(qs + ln <= plen[owner]) and np.all(rs >= 0) and np.all(rs + ln <= len(refb))
    cat = np.frombuffer(b"".join(reads), dtype=np.uint8)
    rstart = np.concatenate([[0], np.cumsum(plen)[:-1]])
    tot = int(ln.sum())
    within = np.arange(tot) - np.repeat(np.cumsum(ln) - ln, ln)
    assert np.array_equal(cat[np.repeat(rstart[owner] + qs, ln) + within], ref[np.repeat(rs, ln) + within])
    # maximal: cannot be extended to the right, nor (strictly, within the sparseQ*K window the reference looks at) to the left
    rq, rr = qs + ln, rs + ln
    can = (rq < plen[owner]) & (rr < len(refb))
    assert not np.any(cat[(rstart[owner] + rq)[can]] == ref[rr[can]])
    can = (qs > 0) & (rs > 0)
    assert not np.any(cat[(rstart[owner] + qs - 1)[can]] == ref[(rs - 1)[can]])
    # exact parity with the oracle on a random sample of reads
    pick = np.random.default_rng(6).choice(len(reads), 3000, replace=False)
    qs2 = ffi.make_seed_queries([reads[i] for i in pick], min_len, 10000, sq)
    mo, oo = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, qs2)
    for k, i in enumerate(pick):
        assert np.array_equal(m[off[i]:off[i + 1]], mo[oo[k]:oo[k + 1]]), i


def _rescore(P, q, t, cig):
    """score of the alignment a BAM CIGAR describes under affine gaps (gapo + gape*len per gap run)"""
    i = j = 0
    s = 0
    for c in cig:
        op, n = int(c) & 15, int(c) >> 4
        if op == 0:
            for k in range(n):
                s += P.mat[int(t[i + k]) * P.m + int(q[j + k])]
            i += n; j += n
        elif op == 1:
            s -= P.gapo + P.gape * n; j += n
        else:
            s -= P.gapo + P.gape * n; i += n
    return s, j, i


def test_dp_full_size_properties(gpu):
    rng = np.random.default_rng(17)
    n = 300000
    P = ffi.default_params()
    # task shapes of the C3 workload: queries 5..120, targets within +-20 %, bands wide enough for a defined backtrack
    qlen = rng.integers(5, 120, n)
    pairs, ws = [], []
    base = rng.integers(0, 4, 4000000).astype(np.uint8)
    for k in range(n):
        p = int(rng.integers(0, len(base) - 400))
        a = base[p:p + int(qlen[k])].copy()
        nmut = int(rng.integers(0, 4))
        if nmut:
            a[rng.integers(0, len(a), nmut)] = rng.integers(0, 4, nmut)
        d = int(rng.integers(-3, 4))
        b = base[p:p + max(1, int(qlen[k]) + d)]
        pairs.append((a, b))
        ws.append(abs(len(a) - len(b)) + int(rng.integers(3, 25)))
    tasks, keep = ffi.make_dp_tasks(pairs, ws, 0)
    nw = ffi.ksw_nw_batch(P, tasks)
    sc, cig, nc, off = ffi.ksw_global_batch(P, tasks)
    assert np.array_equal(nw, sc)                                   # ksw_nw is the score of ksw_global (src/ksw.c:544 vs :468)
    assert np.all(nc > 0)
    # CIGARs consume both sequences exactly and re-score to the reported score (sample; python loop)
    for k in rng.choice(n, 20000, replace=False):
        s, qj, ti = _rescore(P, pairs[k][0], pairs[k][1], cig[off[k]:off[k] + nc[k]])
        assert qj == len(pairs[k][0]) and ti == len(pairs[k][1]) and s == sc[k], k
    tk30, keep30 = ffi.make_dp_tasks(pairs, ws, 30)
    ext = ffi.ksw_extend_batch(P, tk30)
    e = np.column_stack([ext[f].astype(np.int64) for f in ext.dtype.names])
    ql = np.array([len(a) for a, _ in pairs]); tl = np.array([len(b) for _, b in pairs])
    assert np.all(e[:, 0] >= 30) and np.all(e[:, 1] <= ql) and np.all(e[:, 2] <= tl) and np.all(e[:, 3] <= tl)
    assert np.all(e[:, 4] <= e[:, 0])                               # the to-the-end score never beats the best score
    # exact parity with the oracle on a sample
    pick = rng.choice(n, 4000, replace=False)
    tk, kp = ffi.make_dp_tasks([pairs[i] for i in pick], [ws[i] for i in pick], 30)
    eo = O.orc_ksw_extend_batch(P, tk)
    assert np.array_equal(np.column_stack([eo[f].astype(np.int64) for f in eo.dtype.names]), e[pick])
    tk0, kp0 = ffi.make_dp_tasks([pairs[i] for i in pick], [ws[i] for i in pick], 0)
    so, co, no, oo = O.orc_ksw_global_batch(P, tk0)
    assert np.array_equal(so, sc[pick])
    for k, i in enumerate(pick):
        assert np.array_equal(co[oo[k]:oo[k] + no[k]], cig[off[i]:off[i] + nc[i]])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "alfalfa")), reason="compiled reference not in the tree")
def test_cli_200k_reads_byte_identical_to_reference_binary(gpu):
    d = M.workdir()
    ref = bench.make_reference(5000000, 1)
    with open(os.path.join(d, "c3.fa"), "w") as f:
        s = ref.tobytes().decode().upper()
        f.write(">chr1\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    fw, _, _, _ = bench.simulate_batch(np.random.default_rng(9), ref, 200000, 250, 0.02)
    with open(os.path.join(d, "c3.fq"), "w") as f:
        for i, r in enumerate(fw):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r.decode().upper(), "I" * len(r)))
    subprocess.run([M.GPU_EXE, "index", "-r", "c3.fa", "-p", "c3", "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outs = []
    for exe in (M.GPU_EXE, os.path.join(ROOT, "oracle", "_ref", "alfalfa")):
        out = os.path.join(d, "c3_%s.sam" % os.path.basename(os.path.dirname(exe)))
        p = subprocess.run([exe, "align", "-i", "c3", "-0", "c3.fq", "-o", out, "-t", "16"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        outs.append(M.md5(M.sam_body(out)))
    assert outs[0] == outs[1]
