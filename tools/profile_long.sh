set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/proflong
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$R/bench.py" --steps 4 --warmup 1 --e2e-reads 0 --no-cpu-baseline --align-tasks 100 --ref-len 20000000 --read-len 5000 --err 0.10 --batch 5000 > "$OUT/kt.log" 2>&1
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/kt" gpurun_out/long_kernel_trace.csv --only bench.py
rm -rf "$OUT"/*/
tail -c 1500 "$OUT/kt.log"
