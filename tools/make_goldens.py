#!/usr/bin/env python3
"""tools/make_goldens.py -- generate the parity fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and oracle/_ref built by `make -C oracle ref`).
Everything written is data (inputs + expected outputs); no reference source is copied.

  inputs/            the reference's own example inputs (example/reference.fasta, 454reads.fq, mate1/2.fq), gzip
  index_md5.json     md5 of the index files the reference writes for lambda phage (K=1, K=12) and of the
                     arrays for a synthetic 3-contig reference with repeats / N runs (K=1,3,12)
  sam_*.sam.gz       whole-SAM goldens (all lines except @PG) for the example runs of example/EXAMPLE:46,94,102
                     plus -s 12, --seed MEM, --seed PSMEM, --full-dp variants, and sam_md5.json
  seeds_lambda.npz   sparseSA::SMAM/MEM/SMEM/SMEMP outputs for every example read, fw + rc, K=1 and K=12,
                     with the parameters the mapper derives (min_len, sparseSkip, maxSeedCandidates)
  ksw_calls.npz      every ksw_extend / ksw_nw / ksw_global call the reference mapper makes while mapping the
                     examples (captured by oracle/ref_ksw_logger.c), de-duplicated, args + outputs
  ksw_random.npz     synthetic edge cases (0/1-length, band < |qlen-tlen|, h0=0, Z-drop, all-N ...) answered by
                     the reference's ksw.c through oracle/_ref/libalfalfa_ref.so
"""
import gzip
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
EXE = os.path.join(ROOT, "oracle", "_ref", "alfalfa")
EXE_LOG = os.path.join(ROOT, "oracle", "_ref", "alfalfa_kswlog")

from alfalfa_amd import ffi  # noqa: E402
import oracle_ffi as O  # noqa: E402
import seqgen  # noqa: E402
import dpgen  # noqa: E402


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def run(cmd, env=None):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)


def read_fastq(path):
    names, seqs = [], []
    L = open(path, "rb").read().split(b"\n")
    for i in range(0, len(L) - 3, 4):
        names.append(L[i][1:].split()[0])
        seqs.append(L[i + 1].lower())
    return names, seqs


def synthetic_contigs():
    rng = np.random.default_rng(4242)
    a = seqgen.rand_ref(rng, 50000)
    b = seqgen.rand_ref(rng, 20000)
    c = a[2000:6000] + b"n" * 20 + b[100:3100] + b"acgt" * 200 + a[2000:6000] + b"a" * 900
    return ["ctgA", "ctgB", "ctgC"], [a, b, c]


def arrays_md5(arr):
    return {k: hashlib.md5(np.ascontiguousarray(arr[k]).tobytes()).hexdigest()
            for k in ("S", "SA", "LCP", "LCP_M_idx", "LCP_M_val", "CHILD", "KMR")}


def main():
    os.makedirs(os.path.join(GOLD, "inputs"), exist_ok=True)
    ex = os.path.join(REF, "example")
    for f in ("reference.fasta", "454reads.fq", "mate1.fq", "mate2.fq"):
        with open(os.path.join(ex, f), "rb") as fi, gzip.GzipFile(os.path.join(GOLD, "inputs", f + ".gz"), "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    tmp = tempfile.mkdtemp(prefix="alfa_gold_")
    fasta = os.path.join(ex, "reference.fasta")

    # ---------------- index files
    index_md5 = {}
    for K in (1, 12):
        pre = os.path.join(tmp, "lam%d" % K)
        run([EXE, "index", "-r", fasta, "-p", pre, "-s", str(K)])
        index_md5["lambda_K%d" % K] = {e: md5(pre + "." + e) for e in ("aux", "ref", "sa", "lcp", "child", "kmer")}
    names, seqs = synthetic_contigs()
    for K in (1, 3, 12):
        r = O.RefIndex.build_mem(names, seqs, K=K)
        index_md5["synthetic3_K%d" % K] = arrays_md5(r.arrays())
    json.dump(index_md5, open(os.path.join(GOLD, "index_md5.json"), "w"), indent=1, sort_keys=True)

    # ---------------- whole-SAM goldens (+ ksw call capture)
    runs = {
        "454_s1": (1, ["-0", ex + "/454reads.fq"]),
        "pe_s1": (1, ["-1", ex + "/mate1.fq", "-2", ex + "/mate2.fq", "-I", "3600", "-X", "4400"]),
        "pe_local_s1": (1, ["-1", ex + "/mate1.fq", "-2", ex + "/mate2.fq", "-I", "3600", "-X", "4400", "--local"]),
        "454_s12": (12, ["-0", ex + "/454reads.fq"]),
        "pe_s12": (12, ["-1", ex + "/mate1.fq", "-2", ex + "/mate2.fq", "-I", "3600", "-X", "4400"]),
        "454_s1_mem": (1, ["-0", ex + "/454reads.fq", "--seed", "MEM"]),
        "454_s1_psmem": (1, ["-0", ex + "/454reads.fq", "--seed", "PSMEM"]),
        "454_s12_mem": (12, ["-0", ex + "/454reads.fq", "--seed", "MEM"]),
        "454_s12_psmem": (12, ["-0", ex + "/454reads.fq", "--seed", "PSMEM"]),
        "454_s1_fulldp": (1, ["-0", ex + "/454reads.fq", "--full-dp"]),
        "454_s1_local": (1, ["-0", ex + "/454reads.fq", "--local"]),
        "454_s12_e12": (12, ["-0", ex + "/454reads.fq", "-e", "0.12"]),
        "454_s1_a5": (1, ["-0", ex + "/454reads.fq", "-a", "5"]),
    }
    sam_md5 = {}
    logs = []
    for name, (K, extra) in runs.items():
        out = os.path.join(tmp, name + ".sam")
        log = os.path.join(tmp, name + ".kswlog")
        env = dict(os.environ, ALFA_KSW_LOG=log)
        run([EXE_LOG, "align", "-i", os.path.join(tmp, "lam%d" % K), "-o", out] + extra, env=env)
        lines = [l for l in open(out, "rb").read().split(b"\n") if not l.startswith(b"@PG")]
        body = b"\n".join(lines)
        sam_md5[name] = {"md5": hashlib.md5(body).hexdigest(), "args": ["-s%d" % K] + [a.replace(ex + "/", "") for a in extra]}
        with gzip.GzipFile(os.path.join(GOLD, "sam_%s.sam.gz" % name), "wb", mtime=0) as fo:
            fo.write(body)
        if os.path.exists(log):
            logs.append(log)
        # the un-instrumented executable must give the same records
        out2 = os.path.join(tmp, name + "_plain.sam")
        run([EXE, "align", "-i", os.path.join(tmp, "lam%d" % K), "-o", out2] + extra)
        l2 = [l for l in open(out2, "rb").read().split(b"\n") if not l.startswith(b"@PG")]
        assert l2 == lines, name
    json.dump(sam_md5, open(os.path.join(GOLD, "sam_md5.json"), "w"), indent=1, sort_keys=True)

    # ---------------- ksw call vectors captured from the real mapper
    seen = set()
    recs = []
    for log in logs:
        buf = open(log, "rb").read()
        p = 0
        while p < len(buf):
            hdr = struct.unpack_from("<10i", buf, p)
            p += 40
            kind, qlen, tlen, m, gapo, gape, w, zdrop, h0, nout = hdr
            mat = buf[p:p + m * m]; p += m * m
            q = buf[p:p + qlen]; p += qlen
            t = buf[p:p + tlen]; p += tlen
            out = buf[p:p + 4 * nout]; p += 4 * nout
            key = (hdr, mat, q, t)
            if key in seen:
                continue
            seen.add(key)
            recs.append((hdr, mat, q, t, out))
    # keep a bounded, deterministic subset: all extend/nw calls + every 2nd global
    recs.sort(key=lambda r: (r[0][0], r[0][1], r[0][2], r[2], r[3], r[0]))
    sel = [r for i, r in enumerate(recs) if r[0][0] != 1 or i % 2 == 0]
    hdrs = np.array([r[0] for r in sel], dtype=np.int32)
    mats = np.frombuffer(b"".join(r[1].ljust(25, b"\0") for r in sel), dtype=np.int8).reshape(-1, 25)
    qcat = np.frombuffer(b"".join(r[2] for r in sel), dtype=np.uint8)
    tcat = np.frombuffer(b"".join(r[3] for r in sel), dtype=np.uint8)
    ocat = np.frombuffer(b"".join(r[4] for r in sel), dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "ksw_calls.npz"), hdr=hdrs, mat=mats, q=qcat, t=tcat, out=ocat)
    print("ksw calls captured: %d unique, %d kept (nw %d, global %d, extend %d)" %
          (len(recs), len(sel), sum(h[0] == 0 for h in hdrs), sum(h[0] == 1 for h in hdrs), sum(h[0] == 2 for h in hdrs)))

    # ---------------- synthetic DP edge cases answered by the reference
    P = ffi.default_params()
    pairs, ws, hs = dpgen.random_tasks(20261110, 1500)
    ext = np.array([O.ref_ksw_extend(P, q, t, w, h) for (q, t), w, h in zip(pairs, ws, hs)], dtype=np.int32)
    nw = np.array([O.ref_ksw_nw(P, q, t, w) for (q, t), w in zip(pairs, ws)], dtype=np.int32)
    gs, gc, gn = [], [], []
    for (q, t), w in zip(pairs, ws):
        s, c = O.ref_ksw_global(P, q, t, w)
        gs.append(s)
        if len(t) > len(q) + w:      # reference reads outside its matrix: CIGAR unspecified, not a golden
            c = np.zeros(0, np.uint32)
            gn.append(-1)
        else:
            gn.append(len(c))
        gc.append(c)
    np.savez_compressed(os.path.join(GOLD, "ksw_random.npz"),
                        qlen=np.array([len(q) for q, _ in pairs], np.int32), tlen=np.array([len(t) for _, t in pairs], np.int32),
                        q=np.concatenate([q for q, _ in pairs]), t=np.concatenate([t for _, t in pairs]),
                        w=np.array(ws, np.int32), h0=np.array(hs, np.int32), extend=ext, nw=nw,
                        gscore=np.array(gs, np.int32), gncig=np.array(gn, np.int32),
                        gcig=np.concatenate(gc) if gc else np.zeros(0, np.uint32))

    # ---------------- seeds for every example read
    _, r454 = read_fastq(ex + "/454reads.fq")
    _, m1 = read_fastq(ex + "/mate1.fq")
    _, m2 = read_fastq(ex + "/mate2.fq")
    reads = []
    for r in r454 + m1 + m2:
        reads.append(r)
        reads.append(seqgen.revcomp(r))
    # adversarial extras: N runs, too short, homopolymer
    reads += [b"n" * 80, r454[0][:30], b"a" * 200, r454[1][:100] + b"nnnn" + r454[1][100:300], b""]
    save = {"reads": np.frombuffer(b"".join(reads), dtype=np.uint8), "read_len": np.array([len(r) for r in reads], np.int32)}
    for K in (1, 12):
        ridx = O.RefIndex.build_fasta(fasta, K=K)
        for mt in (0, 1, 2, 3):
            for tag, rescue in (("main", False), ("rescue", True)):
                allm, offs, prm = [], [0], []
                for r in reads:
                    if rescue:
                        ml, sq, mc = max(20, K + 10), 1, 100000       # src/mapper.cpp:544-553
                    else:
                        ml, sq = seqgen.seed_params(K, len(r))
                        mc = 10000
                    if mt == 2 and len(r) < ml - (sq * K - 1):
                        # sparseSA::SMEM reserves P.length()-min_lenK elements (src/sparseSA.cpp:1134): a shorter
                        # read makes the reference throw std::length_error -> not a golden; expected output: none
                        m = np.zeros(0, ffi.MATCH_DTYPE)
                    else:
                        m = ridx.seed(mt, r, ml, mc, sq, 10)
                    allm.append(m)
                    offs.append(offs[-1] + len(m))
                    prm.append((ml, mc, sq, 10))
                key = "K%d_t%d_%s" % (K, mt, tag)
                save[key + "_m"] = np.concatenate(allm) if allm else np.zeros(0, ffi.MATCH_DTYPE)
                save[key + "_off"] = np.array(offs, np.int64)
                save[key + "_prm"] = np.array(prm, np.int32)
    np.savez_compressed(os.path.join(GOLD, "seeds_lambda.npz"), **save)
    shutil.rmtree(tmp, ignore_errors=True)
    for f in sorted(os.listdir(GOLD)):
        p = os.path.join(GOLD, f)
        if os.path.isfile(p):
            print("%8d  %s" % (os.path.getsize(p), f))


if __name__ == "__main__":
    main()
