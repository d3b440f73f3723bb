#!/usr/bin/env python3
"""Static instruction mix per kernel from a hipcc --save-temps gfx950 .s file:  tools/isa_count.py file.s [name-substring ...]"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
cur, body = None, {}
for l in txt:
    m = re.match(r"^(_Z\w+):", l)
    if m and "@" in l:
        cur = m.group(1); body[cur] = []
    elif l.startswith(".Lfunc_end"):
        cur = None
    elif cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        body[cur].append(l.strip().split()[0])
for name, ops in body.items():
    if want and not any(w in name for w in want):
        continue
    c = Counter()
    for op in ops:
        k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") \
            else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        c[k] += 1
    print(name[:60], dict(c), "total", len(ops))
    print("    ", Counter(ops).most_common(16))
