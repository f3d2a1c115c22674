#!/bin/bash
# GPU box: end-to-end C3 run (2 000 000 reads) for several --map-batch / -t / ALFA_MAP_LANES settings
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/e2e_c3/reads.fq ] || timeout -k 10 200 python3 tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --dir /tmp/e2e_c3 > /dev/null 2>&1 || exit 1
EXE=alfalfa_amd/alfalfa_amd
run() {   # lanes batch threads
  rm -f /tmp/e2e_c3/o.sam
  local s=$(date +%s.%N)
  ALFA_MAP_LANES=$1 timeout -k 10 30 $EXE align -i /tmp/e2e_c3/idx12 -0 /tmp/e2e_c3/reads.fq -o /tmp/e2e_c3/o.sam -t $3 --map-batch $2 --stats 2> /tmp/e2e_c3/err.txt > /dev/null || { echo "FAILED lanes=$1 batch=$2 t=$3"; tail -5 /tmp/e2e_c3/err.txt; exit 1; }
  local e=$(date +%s.%N)
  echo "lanes=$1 batch=$2 t=$3 wall=$(python3 -c "print(round($e - $s, 3))") $(grep -E '#CORE|#STARTUP|#PHASES|#WORKERS' /tmp/e2e_c3/err.txt | cut -c1-400 | tr '\n' ' ' | sed 's/reads_done.*device_call/device_call/; s/index_loaded [0-9.]* //; s/(seconds since start)//')"
}
for b in 4096 8192 16384 32768 65536; do run 0 $b 16; done
for b in 8192 16384 32768 65536; do run 0 $b 8; done
run 0 16384 4
run 0 32768 4
run 2 16384 16
run 0 16384 24
