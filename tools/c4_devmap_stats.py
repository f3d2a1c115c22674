#!/usr/bin/env python3
"""GPU box: C4-shaped long reads through the tool's default path (whole reads on the device): how many the device finishes,
how many it hands back, SAM against --host-trials."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mapper_util as M  # noqa: E402

d = tempfile.mkdtemp(prefix="c4stats_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M.write_c4(d, n)
pre = os.path.join(d, "idx")
subprocess.run([M.GPU_EXE, "index", "-r", "c4.fa", "-p", pre, "-s", "12", "--gpu-index"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
outs = []
for mode in ([], ["--host-trials"]):
    out = os.path.join(d, "o%d.sam" % len(mode))
    t0 = time.time()
    p = subprocess.run([M.GPU_EXE, "align", "-i", pre, "-0", "c4.fq", "-o", out, "-t", "16", "-e", "0.12", "--stats"] + mode, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, ALFA_CALL_LOG="1"))
    dt = time.time() - t0
    err = p.stderr.decode()
    print(mode, "rc", p.returncode, "%.2f s" % dt, "%.0f reads/s" % (n / dt))
    for l in err.split("\n"):
        if l.startswith(("#CORE", "#WORKERS", "#PHASES", "#STATS", "#STARTUP")):
            print("  ", l[:400])
    calls = [l for l in err.split("\n") if l.startswith("#MAPCALL")]
    for l in calls[:3] + [l for l in calls if int(l.split()[2]) < 200][:12]:
        print("  ", l)
    why = [l.rsplit(":", 1)[1].split() for l in err.split("\n") if l.startswith("#MAPWHY")]
    try:
        if why:
            print("   handed back by cause 1..11:", [sum(int(w[k]) for w in why if len(w) > k) for k in range(11)])
    except ValueError as e:
        print("   (cause lines not parsed: %s)" % e)
    outs.append(M.md5(M.sam_body(out)))
print("identical:", outs[0] == outs[1])
