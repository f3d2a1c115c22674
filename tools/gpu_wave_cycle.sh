#!/bin/bash
# development cycle for the long-band DP kernels on the GPU box: parity tests, randomised sweep against the oracle, C4 line
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python -u -m pytest tests/test_gpu_ksw.py tests/test_gpu_golden.py -x -q --timeout 200 --timeout-method thread > gpurun_out/w_ksw.log 2>&1 || { tail -n 30 gpurun_out/w_ksw.log; exit 1; }
echo KSW_OK
timeout -k 10 400 python -u tools/fuzz_dp.py ${FUZZ_ROUNDS:-6} ${FUZZ_SEED:-7000} > gpurun_out/w_fuzz.log 2>&1 || { tail -n 20 gpurun_out/w_fuzz.log; exit 1; }
tail -n 1 gpurun_out/w_fuzz.log
timeout -k 10 400 python -u bench.py --config c4 --steps 10 --e2e-reads ${E2E:-0} > gpurun_out/w_c4.json 2>gpurun_out/w_c4.err || { tail -n 20 gpurun_out/w_c4.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/w_c4.json").read().strip().split("\n")[-1])
print("c4 value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "split", d["step_split"]["seed_only_ms_per_step"], d["step_split"]["dp_only_ms_per_step"])
for k,v in d["dp_detail"].items(): print(k, round(v["gcups"],1), "GCUPS", round(v["ms"]/d["steps"],2), "ms")
print("parity", d["timed_step_parity_vs_oracle"])
PY
