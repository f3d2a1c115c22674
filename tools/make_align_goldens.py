#!/usr/bin/env python3
"""Golden vectors for ksw_align (src/ksw.c:330) from the compiled reference (oracle/_ref/libalfalfa_ref.so).

Writes tests/golden/ksw_align.json.gz: {"scores": [(match, mismatch, gapo, gape) as handed to ksw_align],
"cases": [[score set, query digits, target digits, xtra, [score, te, qe, score2, te2, tb, qb] | null]]}.
null = the reference's behaviour is undefined for the case (its first pass names a padded lane as qe and the second
pass would reverse memory behind the query; only possible when a gap costs nothing) -- the reference is not run on it.
Runs only in the build container (needs oracle/_ref)."""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dpgen            # noqa: E402
import oracle_ffi as O  # noqa: E402
from alfalfa_amd import ffi  # noqa: E402


def main():
    cases = []
    undef = 0
    for si, sc in enumerate(dpgen.ALIGN_SCORES):
        p = dpgen.align_params(sc)
        for seed, n, mq, mt in ((7000 + si, 220, 300, 700), (7100 + si, 12, 1200, 1500)):
            pairs, xs = dpgen.align_cases(seed, n, max_q=mq, max_t=mt, match=sc[0])
            tasks, keep = ffi.make_align_tasks(pairs, [x & ~dpgen.XSTART for x in xs])
            first = O.orc_ksw_align_batch(p, tasks)         # only to recognise the undefined cases before running the reference
            for i, ((q, t), x) in enumerate(zip(pairs, xs)):
                if (x & dpgen.XSTART) and first[i]["qe"] >= len(q) and not ((x & dpgen.XSUBO) and first[i]["score"] < (x & 0xffff)):
                    exp = None
                    undef += 1
                else:
                    exp = list(O.ref_ksw_align(p, q, t, x))
                cases.append([si, "".join(map(str, q.tolist())), "".join(map(str, t.tolist())), int(x), exp])
    out = os.path.join(ROOT, "tests", "golden", "ksw_align.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps({"scores": dpgen.ALIGN_SCORES, "cases": cases}, separators=(",", ":")).encode())
    print("wrote %s: %d cases (%d undefined in the reference), %d bytes" % (out, len(cases), undef, os.path.getsize(out)))


if __name__ == "__main__":
    main()
