"""Whole reads on the device (alfa_map_reads_batch, csrc/map_kernels.hip + csrc/map_core.h): the reference's unpairedMatch --
seeds, candidate regions, the trial loop with its ksw_nw / ksw_extend calls, CIGAR recovery with ksw_global -- in one call.

Parity is anchored three ways:
* the same source (map_core.h) compiled with g++ and driven with the oracle's DP reproduces every SAM golden on the CPU
  (tests/test_mapper_cpu.py: --core-trials);
* the command line tool, whose default path this is, writes the compiled reference's SAM on every golden / synthetic
  configuration (tests/test_gpu_mapper.py, test_gpu_fullsize.py, test_gpu_c4.py, test_gpu_c5.py) -- here additionally with
  the share of reads the device finished itself;
* through the C ABI directly: size-independent properties of every alignment, chain and CIGAR of a 60 000-read batch, the
  scores against the oracle's ksw_global on the region each alignment covers, determinism, edge cases.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import mapper_util as M
import golden_util as G
import seqgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu

NT4 = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"acgt"):
    NT4[_c] = _i


@pytest.fixture(scope="module")
def c3(gpu):
    ref = bench.make_reference(5000000, 1)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref.tobytes()], K=12)
    return ref, hidx, ffi.DeviceIndex(hidx, 0)


def _check_batch(ref, fw, rc, opt, res, want_done=0.99):
    st, aoff, al, soff, sd, ooff, ops = res
    n = len(fw)
    assert len(st) == n and aoff[0] == 0 and soff[0] == 0 and ooff[0] == 0
    assert np.all(np.diff(aoff) >= 0) and np.all(np.diff(soff) >= 0) and np.all(np.diff(ooff) >= 0)
    assert aoff[-1] == len(al) and soff[-1] == len(sd) and ooff[-1] == len(ops)
    assert set(np.unique(st)) <= {0, 1, 2, 3}                       # 2, 3: the rescue rounds are further calls (alfa_hot.h)
    assert (st == 0).mean() >= want_done
    handed = np.nonzero(st != 0)[0]
    assert np.all(aoff[handed + 1] == aoff[handed])                 # a read that is handed back brings nothing
    P = ffi.default_params()
    refn = NT4[ref]
    checked = rescored_equal = 0
    for i in range(n):
        for a in al[aoff[i]:aoff[i + 1]]:
            read = fw[i] if a["forward"] else rc[i]
            plen = len(read)
            qb, qe, rb, re_ = int(a["q_begin"]), int(a["q_end"]), int(a["ref_begin"]), int(a["ref_end"])
            assert 0 <= qb <= qe < plen and 0 <= rb <= re_ < len(ref)
            assert (qb == 0 or rb == 0) and (qe == plen - 1 or re_ >= len(ref) - 1)     # global alignment (alnOptions.global): the whole read
            ch = sd[soff[i] + a["chain_first"]: soff[i] + a["chain_first"] + a["chain_count"]]
            assert len(ch) >= 1
            r8 = np.frombuffer(read, np.uint8)
            for s in ch:                                            # chain seeds: exact matches inside the alignment, collinear
                q0, r0, ln = int(s["query"]), int(s["ref"]), int(s["len"])
                assert qb <= q0 and q0 + ln - 1 <= qe and rb <= r0 and r0 + ln - 1 <= re_
                assert np.array_equal(r8[q0:q0 + ln], ref[r0:r0 + ln])
            assert np.all(np.diff(ch["query"]) > 0) and np.all(np.diff(ch["ref"]) > 0)
            if a["op_count"] < 0:
                continue
            op = ops[ooff[i] + a["op_first"]: ooff[i] + a["op_first"] + a["op_count"]]
            codes = op["code"].astype(np.uint8).tobytes()
            assert set(codes) <= set(b"=XIDS")
            ln = op["len"].astype(np.int64)
            qcons = int(ln[np.isin(op["code"], [ord("="), ord("X"), ord("I"), ord("S")])].sum())
            rcons = int(ln[np.isin(op["code"], [ord("="), ord("X"), ord("D")])].sum())
            assert qcons == plen and rcons == re_ - rb + 1
            assert int(a["edit_dist"]) == int(ln[np.isin(op["code"], [ord("X"), ord("I"), ord("D")])].sum())
            # the operations, replayed: '=' over equal residues, 'X' over unequal ones; re-scored = the alignment's score
            x, y, score = 0, rb, 0
            rn = NT4[r8]
            for c, l in zip(codes, ln):
                l = int(l)
                if c in b"=X":
                    eq = rn[x:x + l] == refn[y:y + l]
                    assert eq.all() if c == ord("=") else (~eq).all()
                    score += int(np.where(eq, opt.match, opt.mismatch).sum())
                    x += l; y += l
                elif c == ord("I"):
                    score += opt.open_gap + opt.extend_gap * l; x += l
                elif c == ord("D"):
                    score += opt.open_gap + opt.extend_gap * l; y += l
                else:
                    x += l
            rescored_equal += score == int(a["score"])
            checked += 1
    # the operations re-score to the alignment's score (an extension's ksw_extend and the ksw_global that recovers its CIGAR
    # can differ when the band estimate of the latter is narrower: rare)
    assert rescored_equal >= 0.99 * checked, (rescored_equal, checked)
    return checked


def test_map_batch_properties_and_determinism(c3):
    ref, hidx, didx = c3
    n = 60000
    fw, rc, pos, strand = bench.simulate_batch(np.random.default_rng(11), ref, n, 250, 0.02)
    opt = ffi.default_map_options()
    P = ffi.default_params()
    reads = ffi.make_map_reads(fw, rc, 12, opt)
    res = ffi.map_reads_batch(didx, opt, P, reads)
    res2 = ffi.map_reads_batch(didx, opt, P, reads)
    for x, y in zip(res, res2):
        assert np.array_equal(x, y)
    os.environ["ALFA_MAP_SCAN_LIB"] = "1"           # offsets by the library scan (the path of calls beyond 2^20 reads) instead of map_offsets_kernel
    try:
        res3 = ffi.map_reads_batch(didx, opt, P, reads)
    finally:
        del os.environ["ALFA_MAP_SCAN_LIB"]
    for x, y in zip(res, res3):
        assert np.array_equal(x, y)
    st, aoff, al, soff, sd, ooff, ops = res
    sub = slice(0, 4000)
    checked = _check_batch(ref, fw[sub], rc[sub], opt, ffi.map_reads_batch(didx, opt, P, ffi.make_map_reads(fw[sub], rc[sub], 12, opt)))
    assert checked > 3900
    # nearly every read has exactly one alignment, at the place it was drawn from, on the strand it was drawn from
    one = np.nonzero((st == 0) & (np.diff(aoff) == 1))[0]
    assert len(one) > 0.98 * n
    a = al[aoff[one]]
    assert np.mean(np.abs(a["ref_begin"] - pos[one]) <= 30) > 0.995
    assert np.mean(a["forward"].astype(bool) == strand[one]) > 0.995
    # the score of an alignment = ksw_global's score of the read against the text it covers (band wide enough for its indels)
    pick = one[:3000]
    pairs = []
    for i in pick:
        x = al[aoff[i]]
        read = fw[i] if x["forward"] else rc[i]
        pairs.append((NT4[np.frombuffer(read, np.uint8)], NT4[ref[int(x["ref_begin"]):int(x["ref_end"]) + 1]]))
    tasks, keep = ffi.make_dp_tasks(pairs, 40, 0)
    so = O.orc_ksw_nw_batch(P, tasks)
    got = al[aoff[pick]]["score"]
    assert np.all(so >= got)                                        # the optimum is never below what the chain found
    assert np.mean(so == got) > 0.9                                 # ... and the chain finds the optimum nearly always


def test_map_batch_edge_cases(c3):
    ref, hidx, didx = c3
    opt = ffi.default_map_options()
    P = ffi.default_params()
    rng = np.random.default_rng(5)
    good = ref[1000000:1000250].tobytes()
    reads_fw = [good, b"n" * 250, seqgen.rand_ref(rng, 250), good[:60], b"acgt" * 60, good[:120] + b"n" * 10 + good[130:],
                seqgen.mutate(rng, ref[2000000:2000250].tobytes(), 0.07)]
    reads_rc = [seqgen.revcomp(r) for r in reads_fw]
    res = ffi.map_reads_batch(didx, opt, P, ffi.make_map_reads(reads_fw, reads_rc, 12, opt))
    st, aoff, al, soff, sd, ooff, ops = res
    assert st[0] == 0 and aoff[1] - aoff[0] == 1 and al[0]["ref_begin"] == 1000000 and al[0]["score"] == 250 and al[0]["forward"] == 1
    assert st[1] == 2 and st[2] == 2                                # no seed: the seed rescue rounds are further calls
    # ... the last of which (min_len = max(20, K + 10), every position, 100000 candidates) still finds none for the n read
    optm = ffi.default_map_options()
    optm.max_seed_candidates = 100000
    rm = ffi.make_map_reads(reads_fw[1:2], reads_rc[1:2], 12, optm)
    rm[0].min_len, rm[0].sparse_q = max(20, 12 + 10), 1
    assert ffi.map_reads_batch(didx, optm, P, rm)[0][0] == 2
    # RESCUE2's call on a read that maps anyway: only the looser pass runs, the same alignment comes out; a read without
    # alignment is done there (status 0, nothing)
    optr = ffi.default_map_options()
    optr.rescue_only = 1
    rr = ffi.make_map_reads(reads_fw, reads_rc, 12, optr)
    for x in rr:
        x.region_edit_dist = int(2 * optr.error_percent * x.plen) + 1
    r2 = ffi.map_reads_batch(didx, optr, P, rr)
    assert r2[0][0] == 0 and r2[2][0]["ref_begin"] == 1000000 and r2[2][0]["score"] == 250
    assert set(np.unique(r2[0])) <= {0, 1, 2}
    assert st[5] == 0 and aoff[6] - aoff[5] == 1 and al[aoff[5]]["score"] == 250 - 10 * 2          # n against the text: -1 each instead of +1
    # without rescue a seedless read is simply unmapped
    opt2 = ffi.default_map_options()
    opt2.rescue = 0
    st2, aoff2 = ffi.map_reads_batch(didx, opt2, P, ffi.make_map_reads(reads_fw, reads_rc, 12, opt2))[:2]
    assert st2[1] == 0 and st2[2] == 0 and aoff2[2] == aoff2[1] and aoff2[3] == aoff2[2]
    # empty batch, and no CIGARs on request
    e = ffi.map_reads_batch(didx, opt, P, ffi.make_map_reads([], [], 12, opt))
    assert len(e[0]) == 0 and len(e[2]) == 0
    opt3 = ffi.default_map_options()
    opt3.cigar_alignments = 0
    r3 = ffi.map_reads_batch(didx, opt3, P, ffi.make_map_reads(reads_fw, reads_rc, 12, opt3))
    assert len(r3[6]) == 0 and np.all(r3[2]["op_count"] == -1) and np.array_equal(r3[2]["score"], al["score"])


def test_cli_device_path_equals_host_trials_and_counts(gpu):
    """short reads with repeats and two contigs, long reads, paired reads: the default path (whole reads on the device) and
    --host-trials (the trial loop on the host through the batch entry points) write the same SAM; the device finishes nearly
    all short reads itself"""
    d = M.synth_single()
    pre = os.path.join(d, "devmap_idx")
    subprocess.run([M.GPU_EXE, "index", "-r", "synse.fa", "-p", pre, "-s", "8"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for extra in ([], ["--local"], ["-a", "3"], ["-e", "0.1", "--seed", "MEM"], ["--no-rescue"], ["-M", "2", "-U", "-3", "-O", "-4", "-E", "-2"]):
        outs = []
        for mode in ([], ["--host-trials"]):
            out = os.path.join(d, "devmap_%d.sam" % len(mode))
            p = subprocess.run([M.GPU_EXE, "align", "-i", pre, "-0", "synse.fq", "-o", out, "-t", "4", "--stats"] + extra + mode, cwd=d,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-1000:]
            outs.append(M.sam_body(out))
            if not mode:
                m = re.search(r"#CORE reads_done (\d+) handed_back (\d+)", p.stderr.decode())
                assert m and int(m.group(1)) > 2000, (extra, m and m.group(0))
        assert outs[0] == outs[1], extra
    # the same reads through the general formatter (read / alignment objects for every read), with one device call at a time, with
    # every slice making its own call instead of rendering from a group's call, and with groups of 64 reads (many per batch):
    # the direct renderer, the barrier-free slices and the group calls change nothing in the output
    base = None
    for env in ({}, {"ALFA_NO_DIRECT_RENDER": "1"}, {"ALFA_MAP_LANES": "1"}, {"ALFA_NO_GROUP_CALLS": "1"}, {"ALFA_GROUP_READS": "64"},
                {"ALFA_GROUP_READS": "64", "ALFA_MIN_SLICE": "1"}, {"ALFA_GROUP_READS": "64", "ALFA_ALIAS_DEVICES": "1"},
                {"ALFA_WORKERS": "1"}):
        out = os.path.join(d, "devmap_env.sam")
        more = ["--gpus", "2"] if "ALFA_ALIAS_DEVICES" in env else []          # (two worker groups, each with its own index copy, on the one GPU)
        p = subprocess.run([M.GPU_EXE, "align", "-i", pre, "-0", "synse.fq", "-o", out, "-t", "5", "--map-batch", "300"] + more, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        body = M.sam_body(out)
        base = body if base is None else base
        assert body == base, env
    # paired mode 1: each mate is a read of the device batch
    body, err = M.run_synth(M.GPU_EXE, "synpe_strict_pairs", ["-t", "4", "--stats"])
    assert M.md5(body) == M.synth_md5("synpe_strict_pairs")
    m = re.search(r"#CORE reads_done (\d+) handed_back (\d+)", err)
    assert m and int(m.group(1)) > 3500          # (a third of the second mates has no seed: the re-seeding rescue is handed back)
