#!/bin/bash
# Kernel trace of the PRODUCT command line (alfalfa_amd align, whole reads on the device) on the GPU box:
#   C3 (5 Mbp, 2 000 000 reads of 250 bp at 2 %) and C4 (100 Mbp, 20 000 reads of 2-10 kbp at 10 %, -e 0.12).
# Summaries -> gpurun_out/cli_c3_kernel_trace_stats.csv, gpurun_out/cli_c4_kernel_trace_stats.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cli
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
EXE=$R/alfalfa_amd/alfalfa_amd
cd "$R"
timeout -k 10 300 python3 tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --dir /tmp/e2e_c3 > "$OUT/c3_plain.log" 2>&1 || { tail -n 5 "$OUT/c3_plain.log"; exit 1; }
cd /tmp
export ALFA_FULL_TEARDOWN=1      # the tool otherwise ends with _exit, before the profiler has written its database
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/c3" -- "$EXE" align -i /tmp/e2e_c3/idx12 -0 /tmp/e2e_c3/reads.fq -o /tmp/e2e_c3/prof.sam -t 16 --stats > "$OUT/c3.log" 2>&1 || { tail -n 20 "$OUT/c3.log"; exit 1; }
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/c3" gpurun_out/cli_c3_kernel_trace_stats.csv > /dev/null
rm -rf "$OUT"/c3/
timeout -k 10 500 python3 tools/e2e_bench.py --reads 20000 --threads 16 --exe gpu --ref-len 100000000 --long --err 0.1 --extra=-e\ 0.12 --index-with gpu --dir /tmp/e2e_c4 > "$OUT/c4_plain.log" 2>&1 || { tail -n 5 "$OUT/c4_plain.log"; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/c4" -- "$EXE" align -i /tmp/e2e_c4/idx12 -0 /tmp/e2e_c4/reads.fq -o /tmp/e2e_c4/prof.sam -t 16 -e 0.12 --stats > "$OUT/c4.log" 2>&1 || { tail -n 20 "$OUT/c4.log"; exit 1; }
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/c4" gpurun_out/cli_c4_kernel_trace_stats.csv > /dev/null
rm -rf "$OUT"/c4/
tail -n 2 "$OUT/c3_plain.log" | cut -c1-600
cut -c1-150 gpurun_out/cli_c3_kernel_trace_stats.csv | head -16
tail -n 2 "$OUT/c4_plain.log" | cut -c1-600
cut -c1-150 gpurun_out/cli_c4_kernel_trace_stats.csv | head -24
