#!/usr/bin/env python3
"""One-line picture of how hipcc scheduled a kernel's hot block (profiling aid): usage sched_view.py file.s nreads
prints, for every basic block with exactly `nreads` ds_read_b32, '.' per VALU, 'L' per LDS read, 's' per SALU,
'wN' per s_waitcnt lgkmcnt(N), 'wv' per vmcnt wait (a scratch reload if the kernel spills)."""
import re
import sys
txt = open(sys.argv[1]).read()
want = int(sys.argv[2])
for m in re.finditer(r"; NumVgprs: \d+|ScratchSize: \d+", txt):
    pass
print(re.findall(r"; NumVgprs: \d+|ScratchSize: \d+", txt)[-4:])
for b in re.split(r"\n(?=\.LBB\d+_\d+:)", txt):
    if b.count("ds_read_b32") != want:
        continue
    seq = []
    for ln in b.split("\n"):
        t = ln.strip()
        if t.startswith("ds_read"):
            seq.append("L")
        elif t.startswith("s_waitcnt"):
            seq.append("w" + re.sub(r".*lgkmcnt\((\d+)\).*", r"\1", t) if "lgkmcnt" in t else "wv")
        elif t.startswith("v_"):
            seq.append(".")
        elif t.startswith("s_"):
            seq.append("s")
    print("".join(seq), len(seq))
    break
