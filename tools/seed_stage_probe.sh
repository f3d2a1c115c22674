set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for k in normal allN random; do
  OUT=$R/gpurun_out/probe_$k; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d "$OUT" -- python3 "$R/tools/seed_stage_probe.py" $k > "$OUT.log" 2>&1
  (cd "$R" && python3 tools/rocprof_summary.py pmc "$OUT" gpurun_out/probe_$k.csv | grep -i "kernel,\|smam_count")
  rm -rf "$OUT"
  grep kernel_ms "$OUT.log" || true
done
