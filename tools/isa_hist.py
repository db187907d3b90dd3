#!/usr/bin/env python3
"""Opcode histogram of the straight-line blocks of a kernel in hipcc's -S output (profiling aid).
usage: isa_hist.py file.s kernel_substring [min_ds_reads]  -> prints, for every basic block with at least
min_ds_reads LDS reads (the unrolled frame_pass bodies of the query kernel), its opcode counts."""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
min_ds = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lines = open(path).read().split("\n")
inside, blocks, cur, label = False, [], [], None
for ln in lines:
    if re.match(r"^[A-Za-z_][\w$.]*:", ln):
        name = ln.split(":")[0]
        if kern in name and not name.startswith(".L"):
            inside = True
        elif not name.startswith(".L") and inside and kern not in name:
            inside = False
    if not inside:
        continue
    t = ln.strip()
    if re.match(r"^\.LBB[\w]*:", t):
        if cur:
            blocks.append((label, cur))
        cur, label = [], t
        continue
    m = re.match(r"^([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+)\b", t)
    if m:
        cur.append(m.group(1))
        if m.group(1).startswith("s_cbranch") or m.group(1) in ("s_branch", "s_endpgm"):
            blocks.append((label, cur))
            cur, label = [], (label or "") + "+"
if cur:
    blocks.append((label, cur))
for label, ops in blocks:
    nds = sum(1 for o in ops if o.startswith("ds_read"))
    if nds < min_ds:
        continue
    h = collections.Counter(re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", o) for o in ops)
    valu = sum(c for o, c in h.items() if o.startswith("v_"))
    print("block %s: %d instructions, %d VALU, %d SALU, %d ds_read, %d s_waitcnt" % (
        label, len(ops), valu, sum(c for o, c in h.items() if o.startswith("s_") and o != "s_waitcnt"), nds, h.get("s_waitcnt", 0)))
    for o, c in sorted(h.items(), key=lambda x: -x[1]):
        print("    %-24s %4d" % (o, c))
