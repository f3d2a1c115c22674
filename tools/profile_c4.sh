#!/bin/bash
# kernel trace of the C4 bench line (BASELINE configs[3]) on the GPU box; summary -> gpurun_out/c4_kernel_trace_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c4
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$R/bench.py" --config c4 --steps 3 --warmup 1 --e2e-reads 0 --no-cpu-baseline --align-tasks 1000 > "$OUT/kt.log" 2>&1 || { tail -n 20 "$OUT/kt.log"; exit 1; }
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/kt" gpurun_out/c4_kernel_trace_stats.csv --only bench.py > /dev/null
python3 tools/rocprof_summary.py dispatches "$OUT/kt" gpurun_out/c4_dispatches.csv --only bench.py
rm -rf "$OUT"/*/
cut -c1-140 gpurun_out/c4_kernel_trace_stats.csv | head -40
