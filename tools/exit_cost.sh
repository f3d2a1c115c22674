#!/bin/bash
# GPU box: what ending the process costs after its last line (kernel-side teardown), for a tiny and a full-size input
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/e2e_c3/reads.fq ] || timeout -k 10 200 python3 tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --dir /tmp/e2e_c3 > /dev/null 2>&1 || exit 1
head -400 /tmp/e2e_c3/reads.fq > /tmp/e2e_c3/tiny.fq
head -800000 /tmp/e2e_c3/reads.fq > /tmp/e2e_c3/mid.fq
EXE=alfalfa_amd/alfalfa_amd
run() {   # fastq threads extra...
  local fq=$1 t=$2; shift 2
  rm -f /tmp/e2e_c3/o.sam
  local s=$(date +%s.%N)
  timeout -k 10 30 $EXE align -i /tmp/e2e_c3/idx12 -0 $fq -o /tmp/e2e_c3/o.sam -t $t --stats "$@" 2> /tmp/e2e_c3/err.txt > /dev/null || { echo FAILED; tail -3 /tmp/e2e_c3/err.txt; exit 1; }
  local e=$(date +%s.%N)
  python3 - "$s" "$e" "$fq" "$t" "$*" <<'PY'
import re, sys
s, e = float(sys.argv[1]), float(sys.argv[2])
txt = open("/tmp/e2e_c3/err.txt").read()
m = re.search(r"#CLOCK mapper_start ([0-9.]+) last_line ([0-9.]+)", txt)
a, b = float(m.group(1)), float(m.group(2))
m2 = re.search(r"#CLOCK main_returns ([0-9.]+)", txt)
if m2:
    print("   after_last_line = mapper teardown %.3f + after main() %.3f" % (float(m2.group(1)) - b, e - float(m2.group(1))))
st = re.search(r"#STARTUP.*", txt).group(0)
ph = re.search(r"#PHASES.*", txt).group(0)
print("%s t=%s %s: wall %.3f = before_main %.3f + run %.3f + after_last_line %.3f | %s | %s" % (sys.argv[3].split("/")[-1], sys.argv[4], sys.argv[5], e - s, a - s, b - a, e - b, st[:110], ph))
PY
}
export ALFA_TEARDOWN_CLOCK=1
run /tmp/e2e_c3/tiny.fq 16
run /tmp/e2e_c3/tiny.fq 1
run /tmp/e2e_c3/reads.fq 16
run /tmp/e2e_c3/reads.fq 8
