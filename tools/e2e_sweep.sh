cd ${GRAFT_REPO_ROOT:-.}
for mb in 4096 1024 16384; do
  echo "== map-batch $mb"
  timeout -k 10 200 python tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --extra="--map-batch $mb" 2>&1 | cut -c1-1100 | tail -2
done
echo "== host-trials"
timeout -k 10 200 python tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu --extra="--host-trials" 2>&1 | cut -c1-1100 | tail -2
