#!/usr/bin/env python3
"""Where the seed kernel's instructions go: launches smam_count_kernel on the C3 workload with reads that stop at different
stages (run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY`, one variant per invocation).
usage: seed_stage_probe.py normal|allN|random"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench   # noqa: E402
import seqgen  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402


def main():
    kind = sys.argv[1]
    ref = bench.make_reference(5000000, 1)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref.tobytes()], K=12)
    didx = ffi.DeviceIndex(hidx, 0)
    rng = np.random.default_rng(5)
    n = 100000
    if kind == "normal":
        fw, rc, _, _ = bench.simulate_batch(rng, ref, n, 250, 0.02)
        reads = fw + rc
    elif kind == "allN":
        reads = [b"n" * 250] * (2 * n)
    else:
        reads = [bench.ACGT[rng.integers(0, 4, 250)].tobytes() for _ in range(2 * n)]
    min_len, sq = seqgen.seed_params(12, 250)
    q = ffi.make_seed_queries(reads, min_len, 10000, sq)
    plan = ffi.SeedPlan(didx, ffi.SEED_SMAM, q)
    for _ in range(3):
        plan.launch()
        ffi.sync(0)
    print(kind, "kernel_ms", plan.kernel_ms)


if __name__ == "__main__":
    main()
