#!/usr/bin/env python3
"""Randomised parity sweep for the banded-DP entry points beyond the seeds the test suite fixes: alfa_ksw_extend_batch,
alfa_ksw_nw_batch and alfa_ksw_global_batch against the oracle (GPU box), over random task bags, mapper-like tasks and
random scoring.     fuzz_dp.py [rounds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dpgen             # noqa: E402
import oracle_ffi as O   # noqa: E402
from alfalfa_amd import ffi  # noqa: E402


def check(P, pairs, ws, hs, tag):
    tasks, keep = ffi.make_dp_tasks(pairs, ws, hs)
    bad = 0
    e, eo = ffi.ksw_extend_batch(P, tasks), O.orc_ksw_extend_batch(P, tasks)
    bad += int((e != eo).sum())
    n, no = ffi.ksw_nw_batch(P, tasks), O.orc_ksw_nw_batch(P, tasks)
    bad += int((n != no).sum())
    (sg, cg, ng, og), (sc, cc, nc, oc) = ffi.ksw_global_batch(P, tasks), O.orc_ksw_global_batch(P, tasks)
    bad += int((sg != sc).sum())
    for i, (q, t) in enumerate(pairs):
        if len(t) > len(q) + ws[i]:          # unspecified in the reference: the library reports -1
            bad += int(ng[i] != -1)
        elif ng[i] != nc[i] or not np.array_equal(cg[og[i]:og[i] + ng[i]], cc[oc[i]:oc[i] + nc[i]]):
            bad += 1
    if bad:
        print(f"MISMATCH {tag}: {bad}", flush=True)
    return 3 * len(pairs), bad


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    total = wrong = 0
    t0 = time.time()
    for r in range(rounds):
        seed = seed0 + r
        rng = np.random.default_rng(seed)
        ma, mi = int(rng.integers(1, 5)), -int(rng.integers(1, 6))
        go, ge, zd = -int(rng.integers(0, 8)), -int(rng.integers(1, 4)), int(rng.integers(0, 120))
        for P, tag in ((ffi.default_params(), "default"), (ffi.default_params(ma, mi, go, ge, zd), f"{ma},{mi},{go},{ge},{zd}")):
            pairs, ws, hs = dpgen.random_tasks(seed, 1200)
            n, b = check(P, pairs, ws, hs, f"bag seed {seed} {tag}")
            total += n; wrong += b
            rl, err, band = int(rng.integers(60, 1200)), float(rng.uniform(0.0, 0.12)), int(rng.integers(5, 110))
            pairs, ws, hs = dpgen.mapper_like_tasks(seed, 400, rl, err, band)
            n, b = check(P, pairs, ws, hs, f"mapper-like seed {seed} rl {rl} err {err:.3f} band {band} {tag}")
            total += n; wrong += b
        print(f"round {r + 1}/{rounds}: {total} results compared, {wrong} differ, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_dp: {total} results, {wrong} mismatches")
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
