#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root through gpurun):
#   one kernel-trace pass and three counter passes (each within the per-block counter limits) over the same bench command,
#   then the default bench line.  Outputs go to gpurun_out/; tools/rocprof_summary.py turns them into profiles/rNN_*.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 6 --warmup 2 --e2e-reads 0 --no-cpu-baseline --no-c4 --align-tasks 20000"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1
echo "kernel trace done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
echo "fetch done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/write" -- python3 "$R/bench.py" $ARGS > "$OUT/write.log" 2>&1
echo "write done"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d "$OUT/sq" -- python3 "$R/bench.py" $ARGS > "$OUT/sq.log" 2>&1
echo "sq done"
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/kt" gpurun_out/kernel_trace_stats.csv --only bench.py

python3 tools/rocprof_summary.py pmc "$OUT/fetch" gpurun_out/pmc_fetch_size.csv --json gpurun_out/pmc_traffic.json --batch 100000 --only bench.py \
  --note "bytes per dispatch at 100000 reads/step (C3 workload): raw FETCH_SIZE x 1024 (rocprofv3 reports KiB), rocprofv3 --pmc FETCH_SIZE pass; calibration for this access pattern in r01_fetch_calibration.csv (an isolated 4-16 B gather tallies 64 B; streaming reads tally half their bytes)" > /dev/null
python3 tools/rocprof_summary.py pmc "$OUT/write" gpurun_out/pmc_write_l2.csv --only bench.py > /dev/null
python3 tools/rocprof_summary.py pmc "$OUT/sq" gpurun_out/pmc_sq.csv --only bench.py > /dev/null
python3 tools/valu_busy.py gpurun_out/pmc_sq.csv gpurun_out/kernel_trace_stats.csv gpurun_out/valu_busy.json > /dev/null
python3 tools/seed_limiter.py gpurun_out/pmc_sq.csv gpurun_out/pmc_write_l2.csv gpurun_out/pmc_traffic.json gpurun_out/kernel_trace_stats.csv gpurun_out/seed_limiter.json > /dev/null
rm -rf "$OUT"/*/      # the raw databases are large; the summaries are what travels back
# the DP kernels by themselves: the captured C3 task stream (tools/data) x5 = one bench step, every result checked
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/dp" -- python3 "$R/tools/dp_replay.py" "$R/tools/data/c3_tasks_20k.npz" --rep 5 > "$R/gpurun_out/dp_replay.json" 2> "$OUT/dp.log"
cd "$R"
python3 tools/rocprof_summary.py kernels "$OUT/dp" gpurun_out/dp_replay_kernel_trace.csv --only dp_replay > /dev/null
rm -rf "$OUT"/*/
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
