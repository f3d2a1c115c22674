#!/bin/bash
# A/B of the hardware-queue count on the command line tool (end to end) and a higher count on the C4 line
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for q in 24 32; do
  ALFA_HW_QUEUES=$q timeout -k 10 300 python -u bench.py --config c4 --steps 10 --e2e-reads 0 --no-cpu-baseline --align-tasks 1000 > gpurun_out/q_c4_$q.json 2>gpurun_out/q_c4_$q.err || { tail -n 20 gpurun_out/q_c4_$q.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/q_c4_$q.json').read().strip().split('\n')[-1]);print('c4 q=$q', round(d['value']), round(d['ms_per_step'],3))"
done
for q in 4 16; do
  ALFA_HW_QUEUES=$q timeout -k 10 300 python tools/e2e_bench.py --long --reads 50000 --ref-len 100000000 --err 0.10 --threads 16 --exe gpu,gpu --extra="-e 0.12" 2>&1 | cut -c1-330 | sed "s/^/long q=$q /"
done
for q in 4 16; do
  ALFA_HW_QUEUES=$q timeout -k 10 300 python tools/e2e_bench.py --reads 1000000 --threads 16 --exe gpu,gpu 2>&1 | cut -c1-330 | sed "s/^/short q=$q /"
done
