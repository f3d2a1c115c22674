#!/usr/bin/env python3
"""tools/call_log_summary.py <stderr log> -- sums the `#CALL` lines a run with ALFA_CALL_LOG=1 prints (one per synchronous
DP call: mode 0 nw / 1 global / 2 extend, tasks, sequence and direction bytes, packing ms, run+fetch ms, kernel ms) and
repeats the tool's #STATS / #PHASES lines.  Development aid for the end-to-end tool."""
import collections,sys
agg=collections.defaultdict(lambda:[0,0,0,0,0,0,0])
for l in open(sys.argv[1]):
    if not l.startswith("#CALL"): continue
    f=l.split()
    mode=int(f[2]); a=agg[mode]
    a[0]+=1; a[1]+=int(f[4]); a[2]+=int(f[6]); a[3]+=int(f[8]); a[4]+=float(f[10]); a[5]+=float(f[12]); a[6]+=float(f[14])
for m,a in sorted(agg.items()): print("mode",m,"calls",a[0],"tasks",a[1],"seqMB",round(a[2]/1e6),"zGB",round(a[3]/1e9,1),"fill_ms",round(a[4]),"run_fetch_ms",round(a[5]),"kernel_ms",round(a[6]))
for l in open(sys.argv[1]):
    if l.startswith(("#STATS","#PHASES")): print(l.strip()[:300])
