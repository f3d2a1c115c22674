#!/bin/bash
# development cycle for the DP kernels on the GPU box: parity tests, replay of the C3 task stream with the oracle check, kernel trace
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python -u -m pytest tests/test_gpu_ksw.py tests/test_gpu_golden.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_ksw.log 2>&1 || { tail -n 30 gpurun_out/r2_ksw.log; exit 1; }
echo KSW_OK
timeout -k 10 300 python -u tools/dp_replay.py tools/data/c3_tasks_20k.npz --rep 5 > gpurun_out/r2_replay_new.json 2>gpurun_out/r2_replay_new.err || { cat gpurun_out/r2_replay_new.json; tail -n 20 gpurun_out/r2_replay_new.err; exit 1; }
cat gpurun_out/r2_replay_new.json
export TMPDIR=/tmp
rm -rf gpurun_out/prof_replay
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_replay -o replay -- python3 tools/dp_replay.py tools/data/c3_tasks_20k.npz --rep 5 --no-check --iters 10 > gpurun_out/r2_replay_prof.json 2>gpurun_out/r2_replay_prof.err || exit 1
python3 tools/rocprof_summary.py kernels gpurun_out/prof_replay gpurun_out/r2_replay_kt.csv > /dev/null 2>&1
cut -c1-120 gpurun_out/r2_replay_kt.csv
rm -rf gpurun_out/prof_replay
