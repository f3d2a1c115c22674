#!/usr/bin/env python3
"""tools/dp_long_probe.py -- row latency of the wave-per-task DP kernels (development aid, GPU box).

One launch of N identical long tasks (query = target with 10 % substitutions, band 100 => 201 diagonals, the C4 shape):
with N = 1 the launch time is one wavefront's serial chain (rows x instructions per row); N = 1024 / 4096 put one / four
such wavefronts on every SIMD.  Prints ns per row for ksw_extend, ksw_nw and ksw_global."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alfalfa_amd import ffi  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    rng = np.random.default_rng(3)
    P = ffi.default_params()
    t = rng.integers(0, 4, rows).astype(np.uint8)
    q = t.copy()
    flip = rng.random(rows) < 0.1
    q[flip] = (q[flip] + 1) & 3
    counts = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 256, 1024, 4096)
    for n in counts:
        tasks, keep = ffi.make_dp_tasks([(q, t)] * n, 100, 200)
        line = []
        for name, mode in (("extend", ffi.DP_EXTEND), ("nw", ffi.DP_NW), ("global", ffi.DP_GLOBAL)):
            b = ffi.DpBatch(mode, P, tasks, 0)
            ms = []
            for _ in range(3):
                b.launch()
                ffi.sync(0)
                ms.append(b.kernel_ms)
            line.append("%s %.3f ms = %.0f ns/row" % (name, min(ms), min(ms) * 1e6 / rows))
            del b
        print("tasks %5d rows %d: %s" % (n, rows, "; ".join(line)), flush=True)


if __name__ == "__main__":
    main()
