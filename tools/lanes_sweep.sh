#!/bin/bash
# GPU box: end-to-end C3 run (4 000 000 reads): A/B of environment settings
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/e2e_c3b/reads.fq ] || timeout -k 10 300 python3 tools/e2e_bench.py --reads 4000000 --threads 16 --exe gpu --dir /tmp/e2e_c3b > /dev/null 2>&1 || exit 1
EXE=alfalfa_amd/alfalfa_amd
run() {   # batch threads
  rm -f /tmp/e2e_c3b/o.sam
  local s=$(date +%s.%N)
  timeout -k 10 60 $EXE align -i /tmp/e2e_c3b/idx12 -0 /tmp/e2e_c3b/reads.fq -o /tmp/e2e_c3b/o.sam -t $2 --map-batch $1 --stats 2> /tmp/e2e_c3b/err.txt > /dev/null || { echo "FAILED batch=$1 t=$2"; tail -5 /tmp/e2e_c3b/err.txt; exit 1; }
  local e=$(date +%s.%N)
  echo "pack=$ALFA_PACK_THREADS batch=$1 t=$2 wall=$(python3 -c "print(round($e - $s, 3))") $(grep -E '#CORE|#STARTUP|#PHASES|#WORKERS' /tmp/e2e_c3b/err.txt | cut -c1-400 | tr '\n' ' ' | sed 's/reads_done.*device_call/device_call/; s/index_loaded [0-9.]* //; s/(seconds since start)//')"
}
run 16384 16
export ALFA_PACK_THREADS=1
run 16384 16
run 16384 16
unset ALFA_PACK_THREADS
run 16384 16
export ALFA_PACK_THREADS=1
run 16000 16
run 12288 16
run 24576 16
