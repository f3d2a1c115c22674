#!/bin/bash
# GPU box: the bench's C4 end-to-end size (250 000 reads of 2-10 kbp at 10 %, 100 Mbp, -e 0.12, 16 workers) twice, with the shell's clock around it
cd ${GRAFT_REPO_ROOT:-.}
D=${D:-/tmp/e2e_c4d}
[ -f $D/reads.fq ] || timeout -k 10 1000 python3 tools/e2e_bench.py --reads ${NREADS:-250000} --threads 16 --exe gpu --ref-len 100000000 --long --err 0.1 --extra=-e\ 0.12 --index-with gpu --dir $D > /dev/null 2>&1 || exit 1
EXE=alfalfa_amd/alfalfa_amd
export ALFA_TEARDOWN_CLOCK=1
for rep in 1 2; do
  rm -f $D/o.sam
  s=$(date +%s.%N)
  timeout -k 10 100 $EXE align -i $D/idx12 -0 $D/reads.fq -o $D/o.sam -t 16 -e 0.12 --stats "$@" 2> $D/err.txt > /dev/null || { echo FAILED; tail -5 $D/err.txt; exit 1; }
  e=$(date +%s.%N)
  python3 - "$s" "$e" $D/err.txt <<'PY'
import re, sys
s, e = float(sys.argv[1]), float(sys.argv[2])
txt = open(sys.argv[3]).read()
m = re.search(r"#CLOCK mapper_start ([0-9.]+) last_line ([0-9.]+)", txt)
a, b = float(m.group(1)), float(m.group(2))
ph = re.search(r"#PHASES.*", txt).group(0)
wk = re.search(r"#WORKERS.*", txt).group(0)
st = re.search(r"device_ready ([0-9.]+)", txt).group(1)
print("wall %.3f = before_main %.3f + run %.3f (device_ready %s) + teardown %.3f | %s | %s" % (e - s, a - s, b - a, st, e - b, ph[:400], wk[:300]))
PY
  sleep 3
done
grep -v "^@PG" $D/o.sam | md5sum | cut -c1-32
