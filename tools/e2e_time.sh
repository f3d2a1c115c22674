#!/bin/bash
# GPU box: end-to-end C3 run (4 000 000 reads, 16 workers) three times, with the shell's clock around it
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/e2e_c3b/reads.fq ] || timeout -k 10 300 python3 tools/e2e_bench.py --reads 4000000 --threads 16 --exe gpu --dir /tmp/e2e_c3b > /dev/null 2>&1 || exit 1
EXE=${EXE:-alfalfa_amd/alfalfa_amd}
export ALFA_TEARDOWN_CLOCK=1
for rep in 1 2 3; do
  rm -f /tmp/e2e_c3b/o.sam
  s=$(date +%s.%N)
  timeout -k 10 60 $EXE align -i /tmp/e2e_c3b/idx12 -0 /tmp/e2e_c3b/reads.fq -o /tmp/e2e_c3b/o.sam -t 16 --stats "$@" 2> /tmp/e2e_c3b/err.txt > /dev/null || { echo FAILED; tail -5 /tmp/e2e_c3b/err.txt; exit 1; }
  e=$(date +%s.%N)
  python3 - "$s" "$e" <<'PY'
import re, sys
s, e = float(sys.argv[1]), float(sys.argv[2])
txt = open("/tmp/e2e_c3b/err.txt").read()
m = re.search(r"#CLOCK mapper_start ([0-9.]+) last_line ([0-9.]+)", txt)
a, b = float(m.group(1)), float(m.group(2))
m2 = re.search(r"#CLOCK main_returns ([0-9.]+)", txt)
ph = re.search(r"#PHASES.*", txt).group(0)
wk = re.search(r"#WORKERS.*", txt).group(0)
st = re.search(r"device_ready ([0-9.]+)", txt).group(1)
print("wall %.3f = before_main %.3f + run %.3f (device_ready %s) + teardown %.3f | %s | %s" % (e - s, a - s, b - a, st, e - b, ph, wk[:170]))
PY
done
md5sum /tmp/e2e_c3b/o.sam | cut -c1-32
grep -v "^@PG" /tmp/e2e_c3b/o.sam | md5sum | cut -c1-32
