#!/usr/bin/env python3
"""md5 of the compiled reference's SAM records for the C5-shaped paired long-read set of tests/mapper_util.py
(oracle/_ref/alfalfa, built from /root/reference by oracle/Makefile) -> tests/golden/c5_paired_md5.json"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mapper_util as M  # noqa: E402

d = tempfile.mkdtemp(prefix="c5gold_")
M.write_c5_paired(d)
ref = os.path.join(ROOT, "oracle", "_ref", "alfalfa")
body = M.run_c5_paired(ref, d, threads="8")
recs = [l for l in body.split(b"\n") if l and not l.startswith(b"@")]
mapped = sum(1 for l in recs if not (int(l.split(b"\t")[1]) & 4))
out = {"md5": M.md5(body), "records": len(recs), "mapped": mapped, "args": M.C5_ARGS,
       "made_by": "tools/make_c5_goldens.py with oracle/_ref/alfalfa (the reference compiled from /root/reference)"}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c5_paired_md5.json"), "w"), indent=1)
print(out)
