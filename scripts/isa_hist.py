#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -S device assembly file (measurement aid).
usage: isa_hist.py file.s substring-of-mangled-name [--loop]  (--loop: only the largest basic block)"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9$.]+:", l) and pat in l and not l.startswith(".L"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start + 1:end]
blocks, cur = [], []
for l in body:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB", s):
            blocks.append(cur)
            cur = []
        continue
    cur.append(s.split()[0])
    if s.startswith("s_cbranch") or s.startswith("s_branch"):
        blocks.append(cur)
        cur = []
blocks.append(cur)
sel = max(blocks, key=len) if "--loop" in sys.argv else [x for b in blocks for x in b]
h = collections.Counter(sel)
print(f"{pat}: {len(sel)} instructions ({'largest block' if '--loop' in sys.argv else 'whole function'}; {len(blocks)} blocks)")
for k, v in h.most_common(40):
    print(f"  {k:28s} {v}")
for l in lines[end:end + 400]:
    if any(t in l for t in (".vgpr_count", ".agpr_count", ".sgpr_count", "ScratchSize", "NumVgprs", "NumAgprs", "Occupancy", "LDSByteSize")):
        print(l.strip())
        if "LDSByteSize" in l:
            break
