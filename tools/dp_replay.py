#!/usr/bin/env python3
"""tools/dp_replay.py -- replay a captured DP task stream (tools/task_census.py -> tools/data/*.npz) on the GPU.

Development aid: GCUPS per ksw kind with the batches resident (create once, launch several times, HIP-event time of the
kernels), plus a bit-exact check of every result against the oracle (checker only).  `--rep R` tiles the stream R times
so that a batch is as large as one bench step (the census holds 20 000 reads; a bench step is 100 000).

  python tools/dp_replay.py tools/data/c3_tasks_20k.npz --rep 5 [--no-check]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz")
    ap.add_argument("--rep", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--kinds", default="nw,global,extend")
    args = ap.parse_args()
    from alfalfa_amd import ffi
    import bench
    z = np.load(args.npz)
    hdr, seq = z["hdr"], z["seq"]
    off = np.concatenate([[0], np.cumsum(hdr[:, 1].astype(np.int64) + hdr[:, 2])[:-1]])
    base = seq.ctypes.data
    P = ffi.default_params()
    out = {}
    for name, kind, mode in (("nw", 0, ffi.DP_NW), ("global", 1, ffi.DP_GLOBAL), ("extend", 2, ffi.DP_EXTEND)):
        if name not in args.kinds.split(","):
            continue
        rows = np.nonzero(hdr[:, 0] == kind)[0]
        rows = np.tile(rows, args.rep)
        arr = np.zeros(len(rows), dtype=bench.TASK_DTYPE)
        arr["query"] = base + off[rows]
        arr["target"] = base + off[rows] + hdr[rows, 1]
        arr["qlen"], arr["tlen"], arr["w"], arr["h0"] = hdr[rows, 1], hdr[rows, 2], hdr[rows, 3], hdr[rows, 4]
        tasks = (ffi.DpTask * len(rows)).from_buffer(arr)
        b = ffi.DpBatch(mode, P, tasks, 0)
        ms = []
        for _ in range(args.iters):
            b.launch()
            ffi.sync(0)
            ms.append(b.kernel_ms)
        res = b.fetch()
        cells = b.cells
        best = min(ms)
        out[name] = {"tasks": len(rows), "cells": int(cells), "ms_best": best, "ms_all": [round(x, 4) for x in ms],
                     "gcups": cells / best / 1e6, "fast_tasks_giveups": b.giveups()}
        if not args.no_check:
            import oracle_ffi as O
            n1 = len(rows) // args.rep
            t1 = (ffi.DpTask * n1).from_buffer(arr)
            t0 = time.time()
            if name == "extend":
                exp = O.orc_ksw_extend_batch(P, t1)
                ok = all(np.array_equal(res[k * n1:(k + 1) * n1], exp) for k in range(args.rep))
            elif name == "nw":
                exp = O.orc_ksw_nw_batch(P, t1)
                ok = all(np.array_equal(res[k * n1:(k + 1) * n1], exp) for k in range(args.rep))
            else:
                sc, cg, ng, og = res
                se, ce, ne, oe = O.orc_ksw_global_batch(P, t1)
                ok = np.array_equal(sc[:n1], se) and np.array_equal(ng[:n1], ne)
                if ok:
                    for i in range(n1):
                        if ng[i] > 0 and not np.array_equal(cg[og[i]:og[i] + ng[i]], ce[oe[i]:oe[i] + ne[i]]):
                            ok = False
                            break
                    # the tiled copies must equal the first copy
                    for k in range(1, args.rep):
                        ok = ok and np.array_equal(sc[k * n1:(k + 1) * n1], se) and np.array_equal(ng[k * n1:(k + 1) * n1], ne)
            out[name]["parity_vs_oracle"] = bool(ok)
            out[name]["oracle_s"] = round(time.time() - t0, 2)
        del b
    import json
    print(json.dumps(out))
    if not args.no_check and not all(v.get("parity_vs_oracle", True) for v in out.values()):
        sys.exit(1)


if __name__ == "__main__":
    main()
