#!/usr/bin/env python3
"""tools/task_census.py -- the DP task stream of a synthetic workload, captured on the CPU (development aid).

Runs tests/_cpu_mapper (the host mapper bound to the oracle: test infrastructure) with ALFA_TASK_LOG on simulated reads
and prints the distribution of (kind, qlen, tlen, band) of every ksw task the mapper issues, plus the algorithmic cells.
The .hdr/.seq files it leaves under --out can be replayed by tools/fuzz_dp.py style scripts on the GPU box.

  python tools/task_census.py --config c3 --reads 20000 --out /tmp/census_c3
"""
import argparse
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c4"])
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--ref-len", type=int, default=0)
    ap.add_argument("--out", default="/tmp/census")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    import bench
    from alfalfa_amd import ffi
    os.makedirs(args.out, exist_ok=True)
    if args.config == "c3":
        ref_len, seed, err, extra = args.ref_len or 5000000, 1, 0.02, []
    else:
        ref_len, seed, err, extra = args.ref_len or 20000000, 2, 0.10, ["-e", "0.12"]
    ref = bench.make_reference(ref_len, seed)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref.tobytes()], K=12)
    prefix = os.path.join(args.out, "idx")
    hidx.save(prefix)
    rng = np.random.default_rng(5)
    if args.config == "c3":
        fw, _, _, _ = bench.simulate_batch(rng, ref, args.reads, 250, err)
    else:
        fw = []
        for _ in range(args.reads):
            L = int(rng.integers(2000, 10001))
            a, _, _, _ = bench.simulate_batch(rng, ref, 1, L, err)
            fw.append(a[0])
    fq = os.path.join(args.out, "reads.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(fw):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)))
    for old in glob.glob(os.path.join(args.out, "tasks.*")):
        os.remove(old)
    env = dict(os.environ, ALFA_TASK_LOG=os.path.join(args.out, "tasks"))
    subprocess.run([os.path.join(ROOT, "tests", "_cpu_mapper"), "align", "-i", prefix, "-0", fq, "-o", os.path.join(args.out, "out.sam"),
                    "-t", str(args.threads)] + extra, check=True, env=env, stdout=subprocess.DEVNULL)
    hdr = np.concatenate([np.fromfile(h, np.int32).reshape(-1, 5) for h in sorted(glob.glob(os.path.join(args.out, "tasks.*.hdr")))])
    for name, kind in (("nw", 0), ("global", 1), ("extend", 2)):
        h = hdr[hdr[:, 0] == kind]
        if not len(h):
            continue
        q, t, w = h[:, 1].astype(np.int64), h[:, 2].astype(np.int64), h[:, 3].astype(np.int64)
        weff = np.minimum(w, np.maximum(q, t))
        W = 2 * weff + 1
        cells = np.zeros(len(h), np.int64)
        for i in range(len(h)):
            ii = np.arange(t[i])
            cells[i] = np.maximum(np.minimum(ii + weff[i] + 1, q[i]) - np.maximum(ii - weff[i], 0), 0).sum()
        print("== %s: %d tasks (%.2f per read), %d cells (%.0f per task)" % (name, len(h), len(h) / args.reads, cells.sum(), cells.mean()))
        for lab, v in (("qlen", q), ("tlen", t), ("w", w), ("W=2w+1 (eff)", W), ("cells", cells)):
            pc = np.percentile(v, [0, 10, 25, 50, 75, 90, 99, 100]).astype(np.int64)
            print("   %-14s min/10/25/50/75/90/99/max = %s" % (lab, " ".join(map(str, pc))))
        # share of cells by band class
        for lo, hi in ((0, 8), (9, 16), (17, 24), (25, 32), (33, 48), (49, 64), (65, 128), (129, 256), (257, 1 << 30)):
            m = (W >= lo) & (W <= hi)
            if m.any():
                print("   W in [%d,%d]: %6.2f%% of tasks, %6.2f%% of cells, median qlen %d tlen %d" %
                      (lo, hi, 100 * m.mean(), 100 * cells[m].sum() / cells.sum(), np.median(q[m]), np.median(t[m])))
        # cells by min(qlen, W)
        big = cells > 4000
        print("   tasks with > 4000 cells: %.2f%% of tasks, %.2f%% of cells" % (100 * big.mean(), 100 * cells[big].sum() / cells.sum()))


if __name__ == "__main__":
    main()
