#!/bin/bash
# SQ and L2 counters of the C4 bench line's kernels (BASELINE configs[3]); summaries -> gpurun_out/c4_pmc_{sq,l2}.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c4p
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--config c4 --steps 2 --warmup 1 --e2e-reads 0 --no-cpu-baseline --align-tasks 1000"
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/sq" -- python3 "$R/bench.py" $ARGS > "$OUT/sq.log" 2>&1 || { tail -n 20 "$OUT/sq.log"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/l2" -- python3 "$R/bench.py" $ARGS > "$OUT/l2.log" 2>&1 || { tail -n 20 "$OUT/l2.log"; exit 1; }
cd "$R"
python3 tools/rocprof_summary.py pmc "$OUT/sq" gpurun_out/c4_pmc_sq.csv --only bench.py > /dev/null
python3 tools/rocprof_summary.py pmc "$OUT/l2" gpurun_out/c4_pmc_l2.csv --only bench.py > /dev/null
rm -rf "$OUT"/*/
grep -E "kernel,|smam|wave_kernel|group_kernel|backtrack" gpurun_out/c4_pmc_sq.csv | cut -c1-200
grep -E "kernel,|smam" gpurun_out/c4_pmc_l2.csv | cut -c1-200
