#!/bin/bash
# GPU box: the C4 end-to-end run with ALFA_CALL_LOG=1, its #MAPCALL lines (one per alfa_map_reads_batch call) summed per field
cd ${GRAFT_REPO_ROOT:-.}
D=${D:-/tmp/e2e_c4b}
[ -f $D/reads.fq ] || timeout -k 10 500 python3 tools/e2e_bench.py --reads ${NREADS:-50000} --threads 16 --exe gpu --ref-len 100000000 --long --err 0.1 --extra=-e\ 0.12 --index-with gpu --dir $D > /dev/null 2>&1 || exit 1
ALFA_CALL_LOG=1 timeout -k 10 100 alfalfa_amd/alfalfa_amd align -i $D/idx12 -0 $D/reads.fq -o $D/o.sam -t 16 -e 0.12 --stats "$@" 2> $D/err.txt > /dev/null || { tail -5 $D/err.txt; exit 1; }
python3 - $D/err.txt <<'PY'
import re, sys, collections
tot = collections.OrderedDict(); n = 0
rows = []
for l in open(sys.argv[1]):
    if not l.startswith("#MAPCALL"): continue
    f = l.split()[1:]
    d = {f[i]: float(f[i + 1]) for i in range(0, len(f) - 1, 2)}
    rows.append(d)
    for k, v in d.items(): tot[k] = max(tot.get(k, 0), v) if k.endswith("_mb") else tot.get(k, 0) + v
    n += 1
print("calls", n, " sums:", "  ".join("%s %.1f" % kv for kv in tot.items()))
rows.sort(key=lambda d: -d["total_ms"])
for d in rows[:3] + rows[-2:]: print("   ", "  ".join("%s %.1f" % kv for kv in d.items()))
for l in open(sys.argv[1]):
    if l.startswith(("#PHASES", "#WORKERS")): print(l.strip()[:300])
PY
