#!/bin/bash
# dynamic instruction counts of one long wave-per-task DP launch (one task): rocprofv3 --pmc over tools/dp_long_probe.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_probe
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -d "$OUT/sq" -- python3 "$R/tools/dp_long_probe.py" 8000 1 > "$OUT/sq.log" 2>&1 || { tail -n 20 "$OUT/sq.log"; exit 1; }
cd "$R"
python3 tools/rocprof_summary.py pmc "$OUT/sq" gpurun_out/probe_pmc_sq.csv | grep -i "ksw_wave\|backtrack\|kernel" | cut -c1-250
rm -rf "$OUT"/*/
