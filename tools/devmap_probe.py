#!/usr/bin/env python3
"""GPU box: the command line tool with whole reads mapped on the device (default) against the SAM goldens, with the
#CORE / #WORKERS stats lines (how many reads the device finished, how many it handed back)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mapper_util as M  # noqa: E402

names = sys.argv[1:] or ["454_s12", "454_s1", "pe_s12", "454_s1_local", "454_s12_e12", "454_s1_a5", "454_s12_mem", "454_s12_psmem", "pe_s12_rescue"]
bad = 0
for name in names:
    d = M.workdir()
    K, args = M.CONFIGS[name]
    exe = M.GPU_EXE
    pre = os.path.join(d, "idx_%s_%d" % (os.path.basename(exe), K))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([exe, "index", "-r", "reference.fasta", "-p", pre, "-s", str(K)], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = os.path.join(d, "dev_%s.sam" % name)
    p = subprocess.run([exe, "align", "-i", pre, "-o", out] + args + ["--stats", "-t", "2"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    err = p.stderr.decode()
    ok = p.returncode == 0 and M.md5(M.sam_body(out)) == M.golden_md5(name)
    core = re.search(r"#CORE.*", err)
    print(name, "OK" if ok else "MISMATCH rc=%d" % p.returncode, core.group(0) if core else err[-300:], flush=True)
    if not ok:
        bad += 1
        host = os.path.join(d, "host_%s.sam" % name)
        subprocess.run([exe, "align", "-i", pre, "-o", host] + args + ["--host-trials", "-t", "2"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        a = M.sam_body(out).split(b"\n")
        b = M.sam_body(host).split(b"\n")
        nd = 0
        for x, y in zip(a, b):
            if x != y:
                print("  dev :", x[:200].decode(errors="replace"))
                print("  host:", y[:200].decode(errors="replace"))
                nd += 1
                if nd >= 3:
                    break
sys.exit(1 if bad else 0)
