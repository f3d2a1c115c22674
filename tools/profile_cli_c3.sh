#!/bin/bash
# GPU box: kernel trace of the product command line on C3 (2 000 000 reads), with how its kernels share the device
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cli
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
EXE=$R/alfalfa_amd/alfalfa_amd
cd "$R"
[ -f /tmp/e2e_c3/reads.fq ] || timeout -k 10 300 python3 tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --dir /tmp/e2e_c3 > "$OUT/c3_plain.log" 2>&1 || exit 1
cd /tmp
export ALFA_FULL_TEARDOWN=1      # the tool otherwise ends with _exit, before the profiler has written its database
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/c3" -- "$EXE" align -i /tmp/e2e_c3/idx12 -0 /tmp/e2e_c3/reads.fq -o /tmp/e2e_c3/prof.sam "$@" --stats > "$OUT/c3.log" 2>&1 || { tail -n 20 "$OUT/c3.log"; exit 1; }
cd "$R"
grep -E "#PHASES|#WORKERS|#CORE" "$OUT/c3.log" | cut -c1-300
python3 tools/rocprof_summary.py kernels "$OUT/c3" gpurun_out/cli_c3_kernel_trace_stats.csv > /dev/null
python3 tools/rocprof_summary.py timeline "$OUT/c3" gpurun_out/cli_c3_timeline.csv
python3 tools/rocprof_summary.py dispatches "$OUT/c3" gpurun_out/cli_c3_dispatches.csv
rm -rf "$OUT"/c3/
cut -c1-150 gpurun_out/cli_c3_kernel_trace_stats.csv | head -16
