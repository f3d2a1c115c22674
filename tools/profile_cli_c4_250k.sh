#!/bin/bash
# GPU box: kernel trace of the product command line on 250 000 long reads (the C4 block's end-to-end leg), with how its kernels share the device
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cli4
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/e2e_c4d
cd "$R"
[ -f $D/reads.fq ] || timeout -k 10 800 python3 tools/e2e_bench.py --reads 250000 --threads 16 --exe gpu --ref-len 100000000 --long --err 0.1 --extra=-e\ 0.12 --index-with gpu --dir $D > "$OUT/plain.log" 2>&1 || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/c4" -- "$R/alfalfa_amd/alfalfa_amd" align -i $D/idx12 -0 $D/reads.fq -o $D/prof.sam -t 16 -e 0.12 --stats > "$OUT/c4.log" 2>&1 || { tail -n 20 "$OUT/c4.log"; exit 1; }
cd "$R"
grep -E "#PHASES|#WORKERS" "$OUT/c4.log" | cut -c1-300
python3 tools/rocprof_summary.py kernels "$OUT/c4" gpurun_out/cli_c4_250k_kernel_trace_stats.csv > /dev/null
python3 tools/rocprof_summary.py timeline "$OUT/c4" gpurun_out/cli_c4_250k_timeline.csv
rm -rf "$OUT"/c4/
cut -c1-150 gpurun_out/cli_c4_250k_kernel_trace_stats.csv | head -12
