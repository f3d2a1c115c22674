#!/bin/bash
# A/B of the hardware-queue count (GPU_MAX_HW_QUEUES) on the C4 and C3 bench lines
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seed/dp", round(d["step_split"]["seed_only_ms_per_step"],3), round(d["step_split"]["dp_only_ms_per_step"],3), "gcups", [round(v["gcups"]) for v in d["dp_detail"].values()])
PY
}
for q in 4 8 16; do
  ALFA_HW_QUEUES=$q timeout -k 10 300 python -u bench.py --config c4 --steps 10 --e2e-reads 0 --no-cpu-baseline --align-tasks 1000 > gpurun_out/q_c4_$q.json 2>gpurun_out/q_c4_$q.err || { tail -n 20 gpurun_out/q_c4_$q.err; exit 1; }
  show "c4 q=$q" gpurun_out/q_c4_$q.json
done
for q in 4 8 16; do
  ALFA_HW_QUEUES=$q timeout -k 10 300 python -u bench.py --steps 20 --e2e-reads 0 --no-cpu-baseline --align-tasks 1000 > gpurun_out/q_c3_$q.json 2>gpurun_out/q_c3_$q.err || { tail -n 20 gpurun_out/q_c3_$q.err; exit 1; }
  show "c3 q=$q" gpurun_out/q_c3_$q.json
done
