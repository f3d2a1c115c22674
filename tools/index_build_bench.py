#!/usr/bin/env python3
"""tools/index_build_bench.py -- time of the index construction: GPU builder (index_build.hip) vs the host builder, same text.

  python tools/index_build_bench.py [--mbp 5,50] [--K 12,1]
Prints one JSON line per (size, K): seconds of alfa_host_index_build_mem with the arrays built on the host core and on the
device (PCIe copies of the text and of the arrays included), and whether all arrays are equal."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", default="5,50")
    ap.add_argument("--K", default="12,1")
    ap.add_argument("--no-host", action="store_true")
    args = ap.parse_args()
    from alfalfa_amd import ffi
    import seqgen
    for mbp in [float(x) for x in args.mbp.split(",")]:
        rng = np.random.default_rng(1)
        ref = seqgen.rand_ref(rng, int(mbp * 1e6))
        for K in [int(x) for x in args.K.split(",")]:
            ffi.index_build_on_device(0)
            ffi.HostIndex.build_mem(["w"], [ref[:100000]], K=K)          # context + code objects
            t = time.time()
            dev = ffi.HostIndex.build_mem(["chr1"], [ref], K=K)
            t_dev = time.time() - t
            ffi.index_build_on_device(-1)
            line = {"mbp": mbp, "K": K, "entries": int(dev.view.N // K), "device_s": round(t_dev, 3)}
            if not args.no_host:
                t = time.time()
                host = ffi.HostIndex.build_mem(["chr1"], [ref], K=K)
                line["host_s"] = round(time.time() - t, 3)
                a, b = host.arrays(), dev.arrays()
                line["equal"] = all((a[k] is None and b[k] is None) or np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)
                line["speedup"] = round(line["host_s"] / t_dev, 1)
                del host
            del dev
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
