#!/usr/bin/env python3
"""Turn rocprofv3 output (rocpd sqlite .db files) into the small text summaries committed under profiles/.

usage: rocprof_summary.py kernels <dir-or-db> <out.csv>          per-kernel calls / total / average / min / max / %
       rocprof_summary.py timeline <dir-or-db> <out.csv>         how the kernels overlap: busy fraction, time alone / shared
       rocprof_summary.py dispatches <dir-or-db> <out.csv>       every dispatch in start order (start, duration, queue/stream ids)
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


def timeline(path, out, only=""):
    """how the kernels of the busiest process share the device: wall span, time with at least one kernel running, and per
    kernel the time it runs alone / together with others (from the start and end stamps of the kernel trace)"""
    ev = []
    for d in select_dbs(path, only):
        con = sqlite3.connect(d)
        for name, st, en in con.execute("select name, start, end from kernels"):
            ev.append((st, en, short(name)))
    ev.sort()
    if not ev:
        return
    # keep the densest stretch: kernels separated by less than 2 ms of idle device form a cluster (the timed region of the
    # bench is one: its steps are issued back to back); of the clusters with at least 100 kernels the one with the most kernels per unit of time is analysed
    clusters, cur, cur_end = [], [], None
    for e in ev:
        if cur and e[0] - cur_end > 2000000:
            clusters.append(cur)
            cur = []
        cur.append(e)
        cur_end = e[1] if cur_end is None or not cur[:-1] else max(cur_end, e[1])
    clusters.append(cur)
    big = [c for c in clusters if len(c) >= 100] or clusters
    # the bench also issues seed-only and DP-only passes: the timed region holds both kinds of kernel
    both = [c for c in big if any("smam_count" in e[2] for e in c) and any("ksw_" in e[2] for e in c)]
    big = both or big
    best = max(big, key=lambda c: len(c) / max(1, max(e[1] for e in c) - c[0][0]))
    lo = best[0][0]
    pts = []
    for st, en, nm in best:
        pts.append((st, 1, nm))
        pts.append((en, -1, nm))
    pts.sort()
    active = defaultdict(int)
    alone = defaultdict(int)
    shared = defaultdict(int)
    busy = 0
    conc_time = defaultdict(int)
    prev = pts[0][0]
    for t, dlt, nm in pts:
        dt = t - prev
        if dt > 0:
            live = [k for k, v in active.items() if v > 0]
            n = sum(active[k] for k in live)
            if n:
                busy += dt
                conc_time[min(n, 8)] += dt
                for k in live:
                    (alone if len(live) == 1 else shared)[k] += dt
        active[nm] += dlt
        prev = t
    span = pts[-1][0] - lo
    with open(out, "w") as f:
        f.write("# densest stretch of the trace (%d kernels): span %.3f ms, some kernel running %.1f %%\n" % (len(best), span / 1e6, 100.0 * busy / span))
        f.write("# kernels in flight: " + ", ".join("%d%s: %.1f %%" % (k, "+" if k == 8 else "", 100.0 * v / span) for k, v in sorted(conc_time.items())) + "\n")
        f.write("kernel,alone_ms,with_others_ms,alone_percent_of_span\n")
        for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k])):
            f.write('"%s",%.3f,%.3f,%.1f\n' % (k, alone[k] / 1e6, shared[k] / 1e6, 100.0 * alone[k] / span))
    print(open(out).read())


def dispatches(path, out, only=""):
    """every kernel dispatch in start order: start (us since the first), duration (us), queue / stream ids when the
    database has them, name -- to read how the launches of one step overlap"""
    rows = []
    for d in select_dbs(path, only):
        con = sqlite3.connect(d)
        cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
        extra = [c for c in cols if ("queue" in c or "stream" in c) and c.endswith("id")]
        for r in con.execute("select name, start, end%s from kernels" % "".join(", " + c for c in extra)):
            rows.append((r[1], r[2], short(r[0])) + tuple(r[3:]))
    rows.sort()
    t0 = rows[0][0] if rows else 0
    with open(out, "w") as f:
        f.write("start_us,dur_us,%skernel\n" % "".join(c + "," for c in (extra if rows else [])))
        for r in rows:
            f.write("%.1f,%.1f,%s\"%s\"\n" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, "".join(str(x) + "," for x in r[3:]), r[2]))
    print("%d dispatches -> %s" % (len(rows), out))


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
    ap.add_argument("mode", choices=["kernels", "pmc", "timeline", "dispatches"])
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--json")
    ap.add_argument("--batch", type=int)
    ap.add_argument("--note", default="")
    ap.add_argument("--only", default="", help="keep only the databases of processes whose command contains this text")
    a = ap.parse_args()
    if not dbs(a.src):
        sys.exit("no rocprofv3 .db under " + a.src)
    if a.mode == "timeline":
        timeline(a.src, a.out, a.only)
    elif a.mode == "dispatches":
        dispatches(a.src, a.out, a.only)
    elif a.mode == "kernels":
        for k, v in kernels(a.src, a.out, a.only)[:12]:
            print("%-60s calls %5d avg %10.1f us" % (k[:60], len(v), sum(v) / len(v) / 1e3))
    else:
        pmc(a.src, a.out, a.json, a.batch, a.note, a.only)
        print(open(a.out).read())


if __name__ == "__main__":
    main()
