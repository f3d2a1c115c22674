set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/proftl
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/kt" -- python3 "$R/bench.py" --steps 30 --warmup 3 --e2e-reads 0 --no-cpu-baseline --align-tasks 100 > "$OUT/kt.log" 2>&1
cd "$R"
python3 tools/rocprof_summary.py timeline "$OUT/kt" gpurun_out/kernel_timeline.csv --only bench.py
rm -rf "$OUT"/*/
