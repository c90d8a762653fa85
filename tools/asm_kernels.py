#!/usr/bin/env python3
"""Per-kernel summary of a hipcc `-save-temps` .s file: store / load widths, registers, scratch, LDS.
usage: tools/asm_kernels.py file.s [name-filter]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    parts = re.split(r"\n(?=_Z\w+: +; @_Z)", s)
    for f in parts:
        m = re.match(r"(_Z\w+):", f)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if flt and flt not in name:
            continue
        cnt = lambda pat: len(re.findall(pat, f))
        g = lambda pat: (re.search(pat, f) or [None, "?"])[1]
        nv, na, sc, oc = g(r"; NumVgprs: (\d+)"), g(r"; NumAgprs: (\d+)"), g(r"; ScratchSize: (\d+)"), g(r"; Occupancy: (\d+)")
        st16, st8, st4 = cnt(r"global_store_dwordx4"), cnt(r"global_store_dwordx2"), cnt(r"global_store_dword ")
        ld16, ld8 = cnt(r"global_load_dwordx4"), cnt(r"global_load_dwordx2")
        print(f"{name[:90]:90s} st16 {st16:3d} st8 {st8:3d} st4 {st4:3d} ld16 {ld16:3d} ld8 {ld8:3d} mfma {cnt('v_mfma'):4d}"
              f" vgpr {nv} agpr {na} scratch {sc} occ {oc}")


if __name__ == "__main__":
    main()
