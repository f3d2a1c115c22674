#!/usr/bin/env python3
"""Randomised parity sweep for alfa_seed_batch beyond the seeds the test suite fixes: random references (iid, with planted
repeats, several contigs), random sparseness K, read lengths, error rates and seed parameters, all four seed types, against
the oracle (GPU box): matches and per-read offsets, in order.     fuzz_seed.py [rounds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import seqgen            # noqa: E402
import oracle_ffi as O   # noqa: E402
from alfalfa_amd import ffi  # noqa: E402

TYPES = [(ffi.SEED_SMAM, "SMAM"), (ffi.SEED_MEM, "MEM"), (ffi.SEED_SMEM, "SMEM"), (ffi.SEED_SMEMP, "SMEMP")]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = 0
    total = wrong = calls = 0
    t0 = time.time()
    for r in range(rounds):
        seed = seed0 + r
        rng = np.random.default_rng(seed)
        K = int(rng.choice([1, 2, 3, 4, 6, 12]))
        a = seqgen.rand_ref(rng, int(rng.integers(20000, 400000)))
        names, seqs = ["a"], [a]
        if rng.random() < 0.5:          # a second contig made of copies, a gap of n and low-complexity runs
            lo = int(rng.integers(0, len(a) - 5000))
            piece = a[lo:lo + int(rng.integers(500, 5000))]
            seqs.append(piece + b"n" * int(rng.integers(1, 40)) + b"acgt" * int(rng.integers(10, 300)) + piece +
                        b"a" * int(rng.integers(10, 1000)) + seqgen.rand_ref(rng, 3000))
            names.append("b")
        hidx = ffi.HostIndex.build_mem(names, seqs, K=K)
        didx = ffi.DeviceIndex(hidx, dev)
        cat = b"a".join(seqs)
        lo_len = int(rng.integers(30, 200))
        reads = seqgen.sim_reads(rng, cat, 250, lo_len, lo_len + int(rng.integers(1, 1500)), float(rng.uniform(0.0, 0.12)))
        reads += [b"", b"acgt" * int(rng.integers(1, 80)), b"n" * 40, b"a" * int(rng.integers(1, 400))]
        sq3 = int(rng.integers(1, 4)) if K > 1 else 1
        settings = [([seqgen.seed_params(K, len(x))[0] for x in reads], 10000, [seqgen.seed_params(K, len(x))[1] for x in reads], 10),
                    (max(20, K + 10), 100000, 1, 10),
                    (sq3 * K + int(rng.integers(8, 60)), int(rng.integers(1, 50)), sq3, int(rng.integers(1, 12)))]
        for mt, name in TYPES:
            for (ml, mc, sq, msm) in settings:
                q = ffi.make_seed_queries(reads, ml, mc, sq, msm)
                mg, og = didx.seed_batch(mt, q)
                mo, oo = O.orc_seed_batch(hidx.view, mt, q)
                calls += 1
                total += len(mo)
                if not (np.array_equal(og, oo) and np.array_equal(mg, mo)):
                    wrong += 1
                    print(f"MISMATCH seed {seed} K {K} {name} min_len {ml if isinstance(ml, int) else 'per read'} "
                          f"max_count {mc} sparse_q {sq if isinstance(sq, int) else 'per read'} max_smem {msm}", flush=True)
        print(f"round {r + 1}/{rounds}: {calls} batches, {total} matches compared, {wrong} batches differ, {time.time() - t0:.0f} s",
              flush=True)
    print(f"fuzz_seed: {calls} batches, {total} matches, {wrong} mismatching batches")
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
