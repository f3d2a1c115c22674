#!/usr/bin/env python3
"""tools/isa_census.py -- instruction mix of the largest basic block (the row loop) of each kernel matching a pattern in a
hipcc -S listing (development aid).   hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o k.s file.hip; isa_census.py k.s lane2"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
i = 0
while i < len(txt):
    m = re.match(r"^(_Z\w+):", txt[i])
    if m and pat in m.group(1):
        name = m.group(1)
        blocks, cur, j = [], [], i + 1
        while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
            l = txt[j].strip()
            if re.match(r"^\.LBB\w+:", l):
                blocks.append(cur); cur = []
            elif l and not l.startswith((".", ";")):
                cur.append(l.split()[0])
            j += 1
        blocks.append(cur)
        big = max(blocks, key=len)
        tot = sum(len(b) for b in blocks)
        c = Counter(big)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("%s\n   total %d instr, row-loop block %d (VALU %d, SALU %d, DS %d, VMEM %d)" %
              (name, tot, len(big), valu, sum(v for k, v in c.items() if k.startswith("s_")),
               sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_")))))
        print("   " + ", ".join("%s %d" % kv for kv in c.most_common(24)))
        i = j
    i += 1
