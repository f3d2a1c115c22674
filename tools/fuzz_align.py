#!/usr/bin/env python3
"""Randomised parity sweep for ksw_align beyond the fixed score sets: random alphabets (m = 2..8), random scoring matrices,
penalties of either sign, random flags.   fuzz_align.py ref [n]  : oracle vs the compiled reference (build container)
                                           fuzz_align.py gpu [n]  : alfa_ksw_align_batch vs the oracle (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dpgen             # noqa: E402
import oracle_ffi as O   # noqa: E402
from alfalfa_amd import ffi  # noqa: E402


def random_setup(rng):
    m = int(rng.integers(2, 9))
    p = ffi.DpParams()
    p.m = m
    style = rng.random()
    for i in range(m):
        for j in range(m):
            if style < 0.5:
                v = int(rng.integers(1, 6)) if i == j else -int(rng.integers(0, 7))
            else:
                v = int(rng.integers(-6, 7))
            p.mat[i * m + j] = v
    if max(p.mat[k] for k in range(m * m)) <= 0:
        p.mat[0] = 2
    p.gapo = int(rng.integers(-8, 12))
    p.gape = int(rng.integers(-3, 5))
    return p


def cases(rng, p, n):
    m = p.m
    mx = max(p.mat[k] for k in range(m * m))
    pairs, xs = [], []
    for _ in range(n):
        ql = int(rng.integers(0, 200)) if rng.random() < 0.8 else int(rng.integers(200, 700))
        q = rng.integers(0, m, ql).astype(np.uint8)
        parts = [rng.integers(0, m, int(rng.integers(0, 150))).astype(np.uint8)]
        for _c in range(int(rng.integers(0, 3))):
            c = q.copy()
            if len(c):
                k = rng.random(len(c)) < 0.06
                c[k] = rng.integers(0, m, int(k.sum()))
                if rng.random() < 0.4 and len(c) > 10:
                    cut = int(rng.integers(1, len(c) - 1))
                    c = np.concatenate([c[:cut], rng.integers(0, m, int(rng.integers(0, 4))).astype(np.uint8), c[cut + int(rng.integers(0, 3)):]])
            parts += [c, rng.integers(0, m, int(rng.integers(0, 150))).astype(np.uint8)]
        t = np.concatenate(parts)
        x = 0
        if rng.random() < 0.7:
            x |= dpgen.XSTART
        if rng.random() < 0.5 or ql * mx < 250:
            x |= dpgen.XBYTE if rng.random() < 0.8 else 0
        if rng.random() < 0.25:
            x |= dpgen.XSUBO | int(rng.integers(0, 80))
        elif rng.random() < 0.2:
            x |= dpgen.XSTOP | int(rng.integers(1, 120))
        pairs.append((q, t))
        xs.append(x)
    return pairs, xs


def main():
    mode = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 99)
    total = bad = skipped = 0
    for r in range(rounds):
        p = random_setup(rng)
        pairs, xs = cases(rng, p, 300 if mode == "gpu" else 120)
        tasks, keep = ffi.make_align_tasks(pairs, xs)
        exp = O.orc_ksw_align_batch(p, tasks)
        if mode == "gpu":
            got = ffi.ksw_align_batch(p, tasks)
            diff = np.nonzero(got != exp)[0]
            total += len(pairs)
            bad += len(diff)
            if len(diff):
                k = int(diff[0])
                print("MISMATCH m=%d gapo=%d gape=%d xtra=%#x qlen=%d tlen=%d gpu=%s oracle=%s mat=%s" % (
                    p.m, p.gapo, p.gape, xs[k], len(pairs[k][0]), len(pairs[k][1]), got[k], exp[k], [p.mat[i] for i in range(p.m * p.m)]))
        else:
            for k, ((q, t), x) in enumerate(zip(pairs, xs)):
                if exp[k]["qe"] >= len(q) and (x & dpgen.XSTART):
                    skipped += 1
                    continue
                ref = O.ref_ksw_align(p, q, t, x)
                total += 1
                if tuple(int(v) for v in exp[k]) != ref:
                    bad += 1
                    if bad < 6:
                        print("MISMATCH m=%d gapo=%d gape=%d xtra=%#x qlen=%d tlen=%d oracle=%s ref=%s mat=%s" % (
                            p.m, p.gapo, p.gape, x, len(q), len(t), exp[k], ref, [p.mat[i] for i in range(p.m * p.m)]))
    print("fuzz %s: %d cases, %d mismatches, %d skipped (undefined in the reference)" % (mode, total, bad, skipped))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
