#!/usr/bin/env python3
"""md5 of the compiled reference's SAM records for the C4 set of tests/mapper_util.py (100 Mbp reference, 20 000 reads of
2-10 kbp at 10 % error, -e 0.12, index -s 12) -> tests/golden/c4_md5.json.  The reference is oracle/_ref/alfalfa, built from
/root/reference by oracle/Makefile.  Takes a few minutes and ~3 GB of disk in a temporary directory."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mapper_util as M  # noqa: E402

d = tempfile.mkdtemp(prefix="c4gold_")
M.write_c4(d)
ref = os.path.join(ROOT, "oracle", "_ref", "alfalfa")
t0 = time.time()
body = M.run_c4(ref, d, threads="8")
recs = [l for l in body.split(b"\n") if l and not l.startswith(b"@")]
mapped = sum(1 for l in recs if not (int(l.split(b"\t")[1]) & 4))
out = {"md5": M.md5(body), "records": len(recs), "mapped": mapped, "args": M.C4_ARGS, "reads": M.C4_READS, "ref_len": M.C4_REF_LEN,
       "reference_seconds_index_plus_align_8_threads": round(time.time() - t0, 1),
       "made_by": "tools/make_c4_goldens.py with oracle/_ref/alfalfa (the reference compiled from /root/reference)"}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c4_md5.json"), "w"), indent=1)
print(out)
shutil.rmtree(d, ignore_errors=True)
