#!/usr/bin/env python3
"""GPU box: where map_kernel's cycles go, per phase (library built by tools/probes/map_profile.sh with -DALFA_MAP_PROFILE).
usage: map_profile.py c3|c4 [reads]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402
ffi.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libalfa_hot_prof.so")
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (100000 if cfg == "c3" else 4000)
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
out = (C.c_ulonglong * 16)()
for rep in range(2):
    ffi.lib().alfa_map_profile(out, 1)
    res = ffi.map_reads_batch(didx, opt, P, reads)
    ffi.lib().alfa_map_profile(out, 0)
    v = list(out)
    tot = max(v[5], 1)
    names = ["stage", "ksw_nw", "ksw_extend", "ksw_global fill", "walk", "whole read"]
    print("%s rep %d: %d reads, map_kernel %.2f ms; cycles per read %.0f; share of a read's cycles: %s; rest (trial loop, operations, chains) %.1f %%" %
          (cfg, rep, v[6], ffi.last_timing().kernel_ms, tot / max(v[6], 1),
           ", ".join("%s %.1f %%" % (names[k], 100.0 * v[k] / tot) for k in range(5)), 100.0 * (tot - sum(v[:5])) / tot), flush=True)
    print("   inclusive: load_region %.1f %%, covers_new_bases %.1f %%, try_region %.1f %% (anchor ranks %.1f %%, grow_chain %.1f %%, env.extend %.1f %%), build_cigars %.1f %% (dp_piece %.1f %%, env.global_cigar %.1f %%)" %
          tuple(100.0 * v[k] / tot for k in (8, 9, 10, 11, 12, 15, 14, 13, 7)), flush=True)
