cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -u tools/devmap_probe.py 454_s12 454_s1 pe_s12 454_s1_local 454_s12_e12 454_s1_a5 454_s12_mem 454_s1_fulldp pe_s12_rescue_a3 pe_s12_mode3 || exit 1
timeout -k 10 200 python -u tools/devmap_time.py 100000 2>&1 | tail -8 || exit 1
timeout -k 10 200 python tools/e2e_bench.py --reads 2000000 --threads 16 --exe gpu 2>&1 | cut -c1-1300 | tail -2
