#!/usr/bin/env python3
"""For every kernel of a gfx950 .s file: each barrier-to-barrier segment that contains MFMAs - the instruction mix inside
[first MFMA, last MFMA] (what runs in the MFMA shadow) and how many packed / plain VALU sit outside it.
    python tools/asm_mfma_regions.py file.s [name-substring]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(src) if re.match(r"_Z\w+:", l)]
for si, a in enumerate(starts):
    b = starts[si + 1] if si + 1 < len(starts) else len(src)
    if want not in src[a]:
        continue
    lines = src[a:b]
    bars = [0] + [i for i, l in enumerate(lines) if "s_barrier" in l] + [len(lines)]
    print(src[a].split(":")[0][:60], "| pk ops", sum("v_pk_" in l for l in lines), "| mfma", sum("v_mfma" in l for l in lines),
          "| scratch", sum("scratch_" in l for l in lines), "| branches", sum("s_cbranch" in l for l in lines))
    for st, en in zip(bars[:-1], bars[1:]):
        mf = [i for i in range(st, en) if "v_mfma" in lines[i]]
        if not mf:
            continue
        seg = lines[mf[0]:mf[-1] + 1]
        h = collections.Counter(m.group(1) for l in seg if (m := re.match(r"\t([a-z_0-9]+)", l)))
        out_pk = sum("v_pk_" in l for l in lines[st:mf[0]]) + sum("v_pk_" in l for l in lines[mf[-1]:en])
        print(f"  segment {st}-{en}: {len(mf)} mfma, {sum(h.values())} instr in the MFMA region, branches {h['s_cbranch_vccnz'] + h['s_cbranch_vccz'] + h['s_cbranch_scc1'] + h['s_cbranch_scc0'] + h['s_cbranch_execz']}, "
              f"pk in {sum(v for k, v in h.items() if k.startswith('v_pk_'))} / pk outside {out_pk}, ds_read {h['ds_read_b128']}, waitcnt {h['s_waitcnt']}, s_nop {h['s_nop']}, dma {h['global_load_lds_dwordx4']}")
