#!/usr/bin/env python3
"""Per kernel of a gfx950 .s file (hipcc -S --cuda-device-only): every loop (backward branch) with its instruction, MFMA,
scratch (spill) and waitcnt counts - shows whether spills sit inside a hot loop or in once-per-tile edge code.
    python tools/asm_loops.py file.s [name-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(src) if re.match(r"_Z\w+:", l)]
for si, a in enumerate(starts):
    b = starts[si + 1] if si + 1 < len(starts) else len(src)
    name = src[a].split(":")[0]
    if want not in name:
        continue
    lines = src[a:b]
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    is_ins = lambda l: l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")
    print(name[:70], "instructions", sum(map(is_ins, lines)), "scratch ops", sum("scratch_" in l for l in lines))
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            print("   loop %6d-%6d: %5d instr %4d mfma %4d scratch %4d waitcnt %3d barriers" % (
                labels[m.group(1)], i, sum(map(is_ins, body)), sum("v_mfma" in x for x in body), sum("scratch_" in x for x in body),
                sum("s_waitcnt" in x for x in body), sum("s_barrier" in x for x in body)))
