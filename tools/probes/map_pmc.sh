#!/bin/bash
# GPU box: counters of map_kernel on one C4 call of 16 384 long reads (and one C3 call of 100 000): HBM bytes, instructions, waiting
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/map_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for cfg in c4 c3; do
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/${cfg}_mem" -- python3 "$R/tools/probes/map_call.py" $cfg > "$OUT/${cfg}_mem.log" 2>&1 || { tail -5 "$OUT/${cfg}_mem.log"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/${cfg}_wr" -- python3 "$R/tools/probes/map_call.py" $cfg > "$OUT/${cfg}_wr.log" 2>&1 || { tail -5 "$OUT/${cfg}_wr.log"; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d "$OUT/${cfg}_sq" -- python3 "$R/tools/probes/map_call.py" $cfg > "$OUT/${cfg}_sq.log" 2>&1 || { tail -5 "$OUT/${cfg}_sq.log"; exit 1; }
  cd "$R"
  python3 tools/rocprof_summary.py pmc "$OUT/${cfg}_mem" gpurun_out/map_pmc_${cfg}_mem.csv > /dev/null
  python3 tools/rocprof_summary.py pmc "$OUT/${cfg}_sq" gpurun_out/map_pmc_${cfg}_sq.csv > /dev/null
  python3 tools/rocprof_summary.py pmc "$OUT/${cfg}_wr" gpurun_out/map_pmc_${cfg}_wr.csv > /dev/null
  grep -i "map_kernel\|^kernel" gpurun_out/map_pmc_${cfg}_wr.csv | cut -c1-300
  tail -1 "$OUT/${cfg}_mem.log"
  grep -i "map_kernel\|^kernel" gpurun_out/map_pmc_${cfg}_mem.csv | cut -c1-300
  grep -i "map_kernel\|^kernel" gpurun_out/map_pmc_${cfg}_sq.csv | cut -c1-400
  cd /tmp
done
rm -rf "$OUT"/*/
