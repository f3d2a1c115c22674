#!/usr/bin/env python3
"""ksw_align throughput on mate-rescue shaped tasks: alfa_ksw_align_batch (kernel time from alfa_last_timing) beside the
compiled reference / the oracle on one host core.  usage: align_bench.py [ntasks] [qlen]"""
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


def tasks_for(n, ql, seed=5):
    rng = np.random.default_rng(seed)
    pairs = []
    for _ in range(n):
        q = rng.integers(0, 4, ql).astype(np.uint8)
        w = rng.integers(0, 4, int(rng.integers(2 * ql, 3 * ql + 200))).astype(np.uint8)
        c = dpgen.mutate(rng, q, 0.03)
        o = int(rng.integers(0, max(1, len(w) - len(c))))
        w[o:o + len(c)] = c[:len(w) - o]
        pairs.append((q, w))
    return pairs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    for ql, sc in ((100, dpgen.ALIGN_SCORES[0]), (250, dpgen.ALIGN_SCORES[0]), (100, dpgen.ALIGN_SCORES[1]), (250, dpgen.ALIGN_SCORES[1])):
        p = dpgen.align_params(sc)
        pairs = tasks_for(n, ql)
        x = dpgen.XSTART | (dpgen.XBYTE if ql * sc[0] < 250 else 0)
        tasks, keep = ffi.make_align_tasks(pairs, x)
        cells = sum(len(q) * len(t) for q, t in pairs)
        ffi.ksw_align_batch(p, tasks)
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            got = ffi.ksw_align_batch(p, tasks)
            wall = time.time() - t0
            best = min(best, ffi.last_timing().kernel_ms)
        m = min(n, 2000)
        sub, _k = ffi.make_align_tasks(pairs[:m], x)
        t0 = time.time()
        exp = O.orc_ksw_align_batch(p, sub)
        t_orc = time.time() - t0
        assert np.array_equal(got[:m], exp)
        line = "qlen %d scores %s %s: %d tasks, %.1f Mcells/task fwd; GPU kernels %.2f ms = %.1f GCUPS (forward cells; wall %.0f ms)" % (
            ql, sc, "u8" if x & dpgen.XBYTE else "i16", n, cells / n / 1e6, best, cells / best / 1e6, wall * 1e3)
        sub_cells = sum(len(q) * len(t) for q, t in pairs[:m])
        line += "; oracle 1 core %.3f GCUPS" % (sub_cells / t_orc / 1e9)
        if O.have_ref():
            t0 = time.time()
            for q, t in pairs[:m]:
                O.ref_ksw_align(p, q, t, x)
            line += "; reference SSE2 1 core %.2f GCUPS" % (sub_cells / (time.time() - t0) / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
