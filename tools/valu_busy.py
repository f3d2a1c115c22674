#!/usr/bin/env python3
"""Per kernel, from the committed rocprofv3 summaries of one bench command: wavefronts per dispatch, VALU / SALU wave-instructions,
the share of wave cycles spent waiting, and VALU busy = VALU wave-instructions x 2 cycles / (SIMDs x duration x clock)
(a wave64 VALU instruction occupies its SIMD-32 for 2 cycles: MI355X_MICROARCH.md).
usage: valu_busy.py <pmc_sq.csv> <kernel_trace_stats.csv> <out.json>"""
import csv
import json
import sys

SIMDS, CLOCK_HZ = 256 * 4, 2.4e9

sq = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[1]))}
kt = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
out = {"formula": "valu_busy = mean_SQ_INSTS_VALU * 2 / (1024 SIMDs * avg_ns * 1e-9 * 2.4e9 Hz); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; resident waves per SIMD = SQ_WAVE_CYCLES x 4 / (1024 x avg_ns x clock): the counter ticks once per 4 cycles (map_kernel holds 3 per SIMD by construction and reads 2.5 with its tail)",
       "sources": [sys.argv[1], sys.argv[2]], "kernels": {}}
for k, r in sq.items():
    if k not in kt:
        continue
    dur = float(kt[k]["avg_ns"]) * 1e-9
    valu = float(r["mean_SQ_INSTS_VALU"])
    wc = float(r["mean_SQ_WAVE_CYCLES"])
    out["kernels"][k] = {
        "avg_us": round(dur * 1e6, 1), "waves_per_dispatch": float(r["mean_SQ_WAVES"]),
        "valu_wave_instructions": valu, "salu_wave_instructions": float(r["mean_SQ_INSTS_SALU"]),
        "valu_busy": round(valu * 2 / (SIMDS * dur * CLOCK_HZ), 4) if dur > 0 else None,
        "wave_cycles_waiting": round(float(r["mean_SQ_WAIT_ANY"]) / wc, 4) if wc > 0 else None,
        "resident_waves_per_simd_mean": round(wc * 4 / (SIMDS * dur * CLOCK_HZ), 2) if dur > 0 else None,
    }
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_us"])[:12]:
    print("%-34s %9.1f us  waves %7.0f  VALU busy %5.1f %%  waiting %5.1f %%  resident waves/SIMD %.2f" %
          (k[:34], v["avg_us"], v["waves_per_dispatch"], 100 * (v["valu_busy"] or 0), 100 * (v["wave_cycles_waiting"] or 0), v["resident_waves_per_simd_mean"] or 0))
