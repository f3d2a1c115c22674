#!/usr/bin/env python3
"""GPU box: map_kernel's time under several builds of the library, same batch, with a digest of everything the call returns.
usage: map_ab.py <library.so> c3|c4 [reads]     (prints one line; tools/probes/map_ab.sh runs it over tools/probes/ab/*.so)"""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("MAP_AB_CHILD") != "1":
    env = dict(os.environ, MAP_AB_CHILD="1", ALFA_CALL_LOG="1")
    p = subprocess.run([sys.executable, "-u"] + sys.argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-2000:])
        sys.exit(p.returncode)
    ms = [float(x) for x in re.findall(r"map_kernel_ms ([0-9.]+)", p.stderr)]
    print("%-28s %s  map_kernel_ms %s  %s" % (os.path.basename(sys.argv[1]), sys.argv[2], " ".join("%.2f" % x for x in ms), p.stdout.strip()))
    if os.environ.get("MAP_AB_VERBOSE"):
        print("   " + [l for l in p.stderr.splitlines() if l.startswith("#MAPCALL")][-1])
    sys.exit(0)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402
ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402

cfg = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else (100000 if cfg == "c3" else 4000)
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
h = hashlib.sha1()
for rep in range(3):
    res = ffi.map_reads_batch(didx, opt, P, reads)
for a in res:
    h.update(np.ascontiguousarray(a).tobytes())
print("done %d of %d digest %s" % (int((res[0] == 0).sum()), n, h.hexdigest()[:12]))
