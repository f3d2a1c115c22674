#!/usr/bin/env python3
"""GPU box: one alfa_map_reads_batch call on a C3-shaped batch, with the call's internal phase times (ALFA_CALL_LOG)."""
import os
import sys
import time

os.environ["ALFA_CALL_LOG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ref = bench.make_reference(5000000, 1)
hidx = ffi.HostIndex.build_mem(["chr1"], [ref.tobytes()], K=12)
didx = ffi.DeviceIndex(hidx, 0)
fw, rc, _, _ = bench.simulate_batch(np.random.default_rng(5), ref, n, 250, 0.02)
opt = ffi.default_map_options()
P = ffi.default_params()
reads = ffi.make_map_reads(fw, rc, 12, opt)
for rep in range(4):
    t0 = time.perf_counter()
    st, aoff, al, soff, sd, ooff, ops = ffi.map_reads_batch(didx, opt, P, reads)
    dt = time.perf_counter() - t0
    print("rep %d: %.1f ms, %.2f M reads/s; done %d handed back %d alignments %d (with CIGAR %d) chain seeds %d ops %d" %
          (rep, dt * 1e3, n / dt / 1e6, int((st == 0).sum()), int((st != 0).sum()), len(al), int((al["op_count"] >= 0).sum()), len(sd), len(ops)), flush=True)
for m in (16384, 4096):
    sub = (ffi.MapRead * m)(*[reads[i] for i in range(m)])
    for rep in range(2):
        t0 = time.perf_counter()
        ffi.map_reads_batch(didx, opt, P, sub)
        print("n=%d: %.2f ms" % (m, (time.perf_counter() - t0) * 1e3), flush=True)
