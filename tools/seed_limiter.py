#!/usr/bin/env python3
"""tools/seed_limiter.py -- what limits the seed kernel, from the counter summaries of tools/profile_round.sh.

  python tools/seed_limiter.py gpurun_out/pmc_sq.csv gpurun_out/pmc_write_l2.csv gpurun_out/pmc_traffic.json gpurun_out/kernel_trace_stats.csv out.json
One JSON object for smam_count_kernel: L2 hit rate, fraction of wave cycles spent waiting, instruction mix per wave and
HBM bytes per dispatch against the kernel's duration -- the figures bench.py's roofline.limiter_counters quotes."""
import csv
import json
import sys

KERNEL = "smam_count_kernel"


def row(path, kernel):
    for r in csv.DictReader(open(path)):
        if r.get("kernel", r.get("Name", "")).strip('"').startswith(kernel):
            return r
    return None


def main():
    sq, l2, traffic, kt, out = sys.argv[1:6]
    s, w = row(sq, KERNEL), row(l2, KERNEL)
    j = {"kernel": KERNEL, "source": "rocprofv3 --pmc passes of tools/profile_round.sh (means per dispatch)"}
    if s:
        f = lambda k: float(s["mean_" + k])
        waves, cyc = f("SQ_WAVES"), f("SQ_WAVE_CYCLES")
        j.update({
            "waves_per_dispatch": waves,
            "wave_cycles_waiting_frac": f("SQ_WAIT_ANY") / cyc,
            "valu_insts_per_wave": f("SQ_INSTS_VALU") / waves, "salu_insts_per_wave": f("SQ_INSTS_SALU") / waves,
            "vmem_rd_insts_per_wave": f("SQ_INSTS_VMEM_RD") / waves, "lds_insts_per_wave": f("SQ_INSTS_LDS") / waves,
            "wave_cycles_per_vmem_rd": cyc / max(f("SQ_INSTS_VMEM_RD"), 1.0),
        })
    if w:
        hit, miss = float(w["mean_TCC_HIT_sum"]), float(w["mean_TCC_MISS_sum"])
        j["l2_hit_rate"] = hit / max(hit + miss, 1.0)
        j["l2_requests_per_dispatch"] = hit + miss
    try:
        t = json.load(open(traffic))
        j["hbm_fetch_bytes_per_dispatch"] = t["bytes_per_dispatch"].get(KERNEL)
        j["batch"] = t.get("batch")
    except Exception:
        pass
    k = None
    for r in csv.reader(open(kt)):
        if r and r[0].strip('"').startswith(KERNEL):
            k = r
    if k:
        j["kernel_trace_row"] = k
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
