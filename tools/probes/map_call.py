#!/usr/bin/env python3
"""GPU box: ONE process, three alfa_map_reads_batch calls on a C3 / C4 batch -- the program to put behind `rocprofv3 --pmc ... --`
(tools/probes/map_pmc.sh).  usage: map_call.py c3|c4 [reads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (100000 if cfg == "c3" else 16384)
if cfg == "c3":
    ref = bench.make_reference(5000000, 1)
    fw, rc, _, _ = bench.simulate_batch(np.random.default_rng(5), ref, n, 250, 0.02)
    opt = ffi.default_map_options()
else:
    ref = bench.make_reference(100000000, 2)
    fw, rc = bench.simulate_long(np.random.default_rng(5), ref.tobytes(), n, 2000, 10000, 0.10)
    opt = ffi.default_map_options(0.13)
ffi.index_build_on_device(0)
hidx = ffi.HostIndex.build_mem(["chr1"], [ref.tobytes()], K=12)
ffi.index_build_on_device(-1)
didx = ffi.DeviceIndex(hidx, 0)
P = ffi.default_params()
reads = ffi.make_map_reads(fw, rc, 12, opt)
bases = sum(len(r) for r in fw)
for rep in range(3):
    res = ffi.map_reads_batch(didx, opt, P, reads)
print("%s: %d reads, %d bases, done %d, map_kernel %.2f ms" % (cfg, n, bases, int((res[0] == 0).sum()), ffi.last_timing().kernel_ms))
