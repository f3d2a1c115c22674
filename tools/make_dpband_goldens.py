#!/usr/bin/env python3
"""Golden vectors for the dynProg rows: a grid over the four variants x the dp_type flags
x affine/linear scores x band sizes on random related pairs (<= 200 bp), with the outputs of the REAL reference class
(dynProg compiled from /root/reference/src/dp.cpp into oracle/_ref/libalfalfa_ref.so, called through oracle/ref_shim.cpp).

Every reference call runs in a forked child because dynProg calls exit(1) when its traceback finds no valid move; such
cases are recorded as expected ret 3.  Cases in which the reference reads matrix cells it never wrote (detected by the
restatement, oracle/dp_oracle.cpp) have no defined answer and are recorded as expected ret 2 without a reference call.
Writes tests/golden/dpband.json.gz.  Runs only where /root/reference exists (this container)."""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from alfalfa_amd import ffi  # noqa: E402
import dpgen  # noqa: E402
import oracle_ffi as O  # noqa: E402


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/libalfalfa_ref.so is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20240611)
    groups = []
    n_ref = n_exit = n_undef = 0
    for variant in range(4):
        for (m, mm, o, e) in dpgen.DPBAND_SCORES:
            if variant == 2 and o == 0:
                continue          # dpBandScore's traceback reads UP/LEFT, which only exist with affine gaps
            sc = ffi.DpbandScores(m, mm, o, e, int(o != 0))
            specs = dpgen.dpband_cases(rng, 60, variant)
            orc = O.orc_dp_band_batch(variant, sc, ffi.make_dpband_tasks(specs))
            cases = []
            for sp, r in zip(specs, orc):
                if r[0] == 2:
                    exp = [2, 0, 0, 0, 0, 0, 0, ""]
                    n_undef += 1
                else:
                    ref = O.ref_dynprog_forked(variant, sc, sp)
                    if ref[0] == "exit":
                        exp = [3, 0, 0, 0, 0, 0, 0, ""]
                        n_exit += 1
                    else:
                        exp = list(ref)
                        n_ref += 1
                c = {k: (v.decode("ascii") if isinstance(v, bytes) else (list(v) if isinstance(v, tuple) else v)) for k, v in sp.items()}
                c["expect"] = exp
                cases.append(c)
            groups.append({"variant": variant, "scores": [m, mm, o, e], "cases": cases})
    out = os.path.join(ROOT, "tests", "golden", "dpband.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps({"groups": groups}, separators=(",", ":")).encode())
    print("wrote %s: %d reference results, %d reference exits, %d undefined" % (out, n_ref, n_exit, n_undef), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
