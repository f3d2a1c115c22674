#!/usr/bin/env python3
"""End-to-end wall-clock of the command line tools on a synthetic workload (FASTQ in -> SAM out, index load included).
usage: e2e_bench.py [--reads N] [--threads T] [--ref-len L] [--read-len R | --long] [--err E] [--exe gpu|cpu|ref] ..."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import seqgen  # noqa: E402

EXES = {"gpu": os.path.join(ROOT, "alfalfa_amd", "alfalfa_amd"), "cpu": os.path.join(ROOT, "tests", "_cpu_mapper"),
        "ref": os.path.join(ROOT, "oracle", "_ref", "alfalfa")}


def write_workload(d, ref_len, n, read_len, err, long_reads, seed=1):
    ref = bench.make_reference(ref_len, seed)
    fa = os.path.join(d, "ref.fa")
    with open(fa, "w") as f:
        f.write(">chr1\n")
        s = ref.tobytes().decode().upper()
        f.write("\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    rng = np.random.default_rng(seed + 100)
    fq = os.path.join(d, "reads.fq")
    with open(fq, "w") as f:
        if long_reads:
            reads = seqgen.sim_reads(rng, ref.tobytes(), n, 2000, 10000, err)
        else:
            reads, _, _, _ = bench.simulate_batch(rng, ref, n, read_len, err)
        for i, r in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r.decode().upper(), "I" * len(r)))
    return fa, fq


def write_pairs(d, ref_len, n, seed=1):
    """n FR pairs of 150-bp mates, fragments of 300-600 bp; a third of the second mates carry a substitution every 14-18
    bases (no seed of 20+, but alignable without gaps: what --paired-rescue recovers), the others 1-12 % error"""
    ref = bench.make_reference(ref_len, seed).tobytes()
    rng = np.random.default_rng(seed + 200)

    def noisy(r):
        return "periodic" if r.random() < 0.35 else float(r.choice([0.01, 0.03, 0.06, 0.12]))
    m1, m2 = seqgen.sim_pairs(rng, ref, n, 150, 300, 600, 0.01, noisy, 3000)
    paths = []
    for tag, reads in ((1, m1), (2, m2)):
        path = os.path.join(d, "mates%d.fq" % tag)
        with open(path, "w") as f:
            for i, r in enumerate(reads):
                f.write("@p%d/%d\n%s\n+\n%s\n" % (i, tag, r.decode().upper(), "I" * len(r)))
        paths.append(path)
    return paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--ref-len", type=int, default=5000000)
    ap.add_argument("--read-len", type=int, default=250)
    ap.add_argument("--err", type=float, default=0.02)
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--paired", action="store_true", help="--reads pairs of 150-bp mates (-1/-2, -I 250 -X 650); add e.g. --extra=--paired-rescue")
    ap.add_argument("--exe", default="gpu,cpu")
    ap.add_argument("--extra", default="")
    ap.add_argument("--dir", default="/tmp/alfa_e2e")
    ap.add_argument("--index-with", default="cpu", choices=["cpu", "gpu"], help="which tool builds the index (gpu: suffix array built on the device)")
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    fa, fq = write_workload(args.dir, args.ref_len, args.reads, args.read_len, args.err, args.long)
    mates = write_pairs(args.dir, args.ref_len, args.reads) if args.paired else None
    idx = os.path.join(args.dir, "idx12")
    subprocess.run([EXES["cpu"] if os.path.exists(EXES["cpu"]) and args.index_with == "cpu" else EXES["gpu"], "index", "-r", fa, "-p", idx, "-s", "12"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    md5s = {}
    for name in args.exe.split(","):
        exe = EXES.get(name, os.path.join(ROOT, name))   # anything else: a path below the repository (an older build, for A/B runs)
        if not os.path.exists(exe):
            print(name, "missing")
            continue
        out = os.path.join(args.dir, "out_%s.sam" % name.replace("/", "_"))
        inputs = ["-1", mates[0], "-2", mates[1], "-I", "250", "-X", "650"] if args.paired else ["-0", fq]
        cmd = [exe, "align", "-i", idx] + inputs + ["-o", out, "-t", str(args.threads)] + args.extra.split()
        if name != "ref":
            cmd.append("--stats")
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
        stats = [l for l in p.stderr.decode().split("\n") if l.startswith(("#STATS", "#PHASES", "#WORKERS", "#STARTUP", "#CORE"))]
        calls = [l for l in p.stderr.decode().split("\n") if l.startswith("#MAPCALL")]
        if calls:      # ALFA_CALL_LOG=1: where the whole-read device calls spend their time (sums over all calls, milliseconds)
            import re
            keys = ["fill_ms", "seeds_ms", "enqueue_ms", "regions_map_wait_ms", "total_ms", "map_kernel_ms", "seed_kernel_ms"]
            tot = {k: sum(float(re.search(k + r" ([0-9.]+)", l).group(1)) for l in calls) for k in keys}
            stats.append("#MAPCALLS n=%d " % len(calls) + " ".join("%s %.1f" % (k, tot[k]) for k in keys))
        import hashlib
        body = b"\n".join(l for l in open(out, "rb").read().split(b"\n") if not l.startswith(b"@PG"))
        md5s[name] = hashlib.md5(body).hexdigest()
        print("%s rc=%d %.2fs %.0f reads/s md5=%s %s" % (name, p.returncode, dt, args.reads * (2 if args.paired else 1) / dt, md5s[name], " ".join(stats)), flush=True)
    print("identical:", len(set(md5s.values())) == 1)


if __name__ == "__main__":
    main()
