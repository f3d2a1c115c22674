#!/usr/bin/env python3
"""Turn rocprofv3 output (rocpd sqlite .db files) into the small text summaries committed under profiles/.

usage: rocprof_summary.py kernels <dir-or-db> <out.csv>          per-kernel calls / total / average / min / max / %
       rocprof_summary.py pmc     <dir-or-db> <out.csv> [--json out.json --batch N --note "..."]
                                                                 per-kernel mean of every collected counter per dispatch;
                                                                 the JSON form keeps FETCH_SIZE / WRITE_SIZE as bytes per dispatch
                                                                 (raw counter x 1024: rocprofv3 reports KiB)
Kernel names are shortened (template arguments kept, parameter lists dropped)."""
import argparse
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def dbs(path):
    if os.path.isfile(path):
        return [path]
    return sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))


def process_command(con):
    """command line of the profiled process of one rocpd database ('' when the schema does not tell)"""
    try:
        names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        for t in names:
            if t == "processes" or t.startswith("rocpd_info_process"):
                cols = [r[1] for r in con.execute("pragma table_info('%s')" % t)]
                for c in ("command", "cmdline", "exe", "name"):
                    if c in cols:
                        row = con.execute("select %s from '%s' limit 1" % (c, t)).fetchone()
                        if row and row[0]:
                            return str(row[0])
    except sqlite3.Error:
        pass
    return ""


def select_dbs(path, only):
    """rocprofv3 also profiles child processes (bench.py runs the command line tool once to capture its DP tasks): keep the
    databases whose process command contains `only`, all of them when none does or the schema does not tell"""
    all_dbs = dbs(path)
    if not only:
        return all_dbs
    keep = [d for d in all_dbs if only in process_command(sqlite3.connect(d))]
    if not keep:
        print("note: no database matched --only %r (%d databases used)" % (only, len(all_dbs)), file=sys.stderr)
        return all_dbs
    return keep


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in name:
        if "init_lookback" in name or "init_kernel" in name:
            return "rocprim scan (init)"
        return "rocprim scan (lookback)"
    # drop the parameter list: the first '(' outside template brackets
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def kernels(path, out, only=""):
    agg = defaultdict(list)
    for d in select_dbs(path, only):
        con = sqlite3.connect(d)
        for name, dur in con.execute("select name, duration from kernels"):
            agg[short(name)].append(dur)
    tot = sum(sum(v) for v in agg.values()) or 1
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for k, v in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
    return rows


def pmc(path, out, jout=None, batch=None, note="", only=""):
    agg = defaultdict(lambda: defaultdict(list))
    for d in select_dbs(path, only):
        con = sqlite3.connect(d)
        per = defaultdict(float)
        names = {}
        # one row per (dispatch, counter instance): sum the instances of one dispatch
        for disp, kname, cname, val in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            per[(disp, cname)] += val
            names[disp] = short(kname)
        for (disp, cname), v in per.items():
            agg[names[disp]][cname].append(v)
    counters = sorted({c for k in agg.values() for c in k})
    with open(out, "w") as f:
        f.write("kernel,dispatches," + ",".join("mean_" + c for c in counters) + "\n")
        for k in sorted(agg):
            n = max(len(v) for v in agg[k].values())
            f.write('"%s",%d,%s\n' % (k, n, ",".join("%.1f" % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else "" for c in counters)))
    if jout:
        j = {"note": note, "batch": batch, "bytes_per_dispatch": {}, "write_bytes_per_dispatch": {}}
        for k in agg:
            if agg[k].get("FETCH_SIZE"):
                j["bytes_per_dispatch"][k] = 1024.0 * sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"])
            if agg[k].get("WRITE_SIZE"):
                j["write_bytes_per_dispatch"][k] = 1024.0 * sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"])
        json.dump(j, open(jout, "w"), indent=1)
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["kernels", "pmc"])
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--json")
    ap.add_argument("--batch", type=int)
    ap.add_argument("--note", default="")
    ap.add_argument("--only", default="", help="keep only the databases of processes whose command contains this text")
    a = ap.parse_args()
    if not dbs(a.src):
        sys.exit("no rocprofv3 .db under " + a.src)
    if a.mode == "kernels":
        for k, v in kernels(a.src, a.out, a.only)[:12]:
            print("%-60s calls %5d avg %10.1f us" % (k[:60], len(v), sum(v) / len(v) / 1e3))
    else:
        pmc(a.src, a.out, a.json, a.batch, a.note, a.only)
        print(open(a.out).read())


if __name__ == "__main__":
    main()
