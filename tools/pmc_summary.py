#!/usr/bin/env python3
"""Merge the rocpd databases of tools/run_pmc.sh into one per-kernel-CLASS summary (markdown on stdout, JSON beside it).

    python tools/pmc_summary.py r02 gpurun_out/r02_trace gpurun_out/r02_fetch ... > profiles/r02_pmc_summary.md

Kernels are grouped by class (the classes of `fvhd_profile_read` / bench.py's `kernels` table, plus the channel count for the
fused ConvFFN and the depthwise kernels) with a regular expression on the demangled or mangled name, NOT by exact instantiation,
so a re-tuned template parameter does not orphan the evidence ("What's weak" 7).
HBM bytes: read = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md "HBM"; calibrated here with
tools/ubench/fetch_calib.hip, profiles/r04_fetch_calib.log: ratio 0.500 for 16-B AND for lane-linear 2-B loads) - EXCEPT the stem,
whose 70-B runs of 2-byte NCHW loads go out as genuine 64-B requests (33 B of the buffer per request in the same calibration): factor 1
there (round 3 doubled it: "594 MB read for a 201-MB image" was 297 MB) -, write = WRITE_SIZE x 1024.
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE-derived cycles of the dispatch) when GRBM_GUI_ACTIVE is present
(the counter is summed over the 8 XCD instances: / 8), else / (1024 x duration x 2.0 GHz)."""
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

CLASSES = [  # (class, regex on kernel name); first match wins
    ("ffn_fused_c384", r"ffn_fused_kernelILi384E|ffn_fused_kernel<384"),
    ("ffn_fused_c192", r"ffn_fused_kernelILi192E|ffn_fused_kernel<192"),
    ("ffn_fused_c96", r"ffn_fused_kernelILi96E|ffn_fused_kernel<96"),
    ("dw7_mfma_c64 (C = 192, 384)", r"dw7_mfma_kernelILi4E|dw7_mfma_kernel<4"),
    ("dw7_mfma_c96 (C = 96)", r"dw7_mfma_kernelILi6E|dw7_mfma_kernel<6"),
    ("dw7_s1", r"dwconv_tiled_kernelILi7ELi1ELi1E|dwconv_tiled_kernel<7, 1, 1"),
    ("dw3_s1", r"dwconv_tiled_kernelILi3ELi1ELi1E|dwconv_tiled_kernel<3, 1, 1"),
    ("dw_mixer_fused", r"dw3_dw7_kernel"),
    ("dw_down", r"dwconv_tiled_kernelILi7ELi2ELi2E|dwconv_tiled_kernel<7, 2, 2|dw7s2_mfma_kernel"),
    ("dw_head", r"dwconv_tiled_kernelILi3ELi1ELi2E|dwconv_tiled_kernel<3, 1, 2"),
    ("stem", r"stem_(fused|conv)_kernel|dwconv_tiled_kernelILi3ELi2ELi1E"),
    ("attention", r"attention_kernel"),
    ("layernorm", r"layernorm_kernel"),
    ("gemm_gelu (fc1 / 1x1 / proj0)", r"gemm(256|_pp)_kernel<2,|gemm(256|_pp)_kernelILi2E"),          # streaming / ping-pong kernels: template <EPI, ODT, ...>
    ("gemm_resid (fc2 / proj)", r"gemm(256|_pp)_kernel<3,|gemm(256|_pp)_kernelILi3E"),
    ("gemm_plain (qkv)", r"gemm(256|_pp)_kernel<0,|gemm(256|_pp)_kernelILi0E"),
    ("gemm_bias (proj2)", r"gemm(256|_pp)_kernel<1,|gemm(256|_pp)_kernelILi1E"),
    ("gemm_gelu (fc1 / 1x1 / proj0)", r"gemm\w*_kernel<\d+, \d+, 2,|gemm\w*_kernelILi\d+ELi\d+ELi2E"),
    ("gemm_resid (fc2 / proj)", r"gemm\w*_kernel<\d+, \d+, 3,|gemm\w*_kernelILi\d+ELi\d+ELi3E"),
    ("gemm_plain (qkv)", r"gemm\w*_kernel<\d+, \d+, 0,|gemm\w*_kernelILi\d+ELi\d+ELi0E"),
    ("gemm_bias (proj2)", r"gemm\w*_kernel<\d+, \d+, 1,|gemm\w*_kernelILi\d+ELi\d+ELi1E"),
    ("se_head", r"se_\w+_kernel"),
    ("splice", r"splice_\w*kernel"),
    ("preprocess", r"pre_[hv]pass_kernel"),
]


READ_FACTOR = {"stem": 1.0}          # FETCH_SIZE -> bytes, per class (default 2: wide / lane-linear loads); see the module docstring


def classify(name):
    for cls, rx in CLASSES:
        if re.search(rx, name):
            return cls
    return None


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))   # class -> counter -> [sum, dispatches, sum duration ns]
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            cur = sqlite3.connect(path).cursor()
            try:
                rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value), max(duration) from pmc_events "
                                   "group by name, counter_name, dispatch_id").fetchall()
            except sqlite3.Error:
                rows = []
            for name, cn, _, v, dur in rows:
                cls = classify(name)
                if cls:
                    a = acc[cls][cn]
                    a[0] += v; a[1] += 1; a[2] += dur or 0
            if not rows:          # plain kernel trace: durations only
                for name, dur in cur.execute("select name, end - start from kernels").fetchall():
                    cls = classify(name)
                    if cls:
                        a = acc[cls]["_trace"]
                        a[0] += dur; a[1] += 1; a[2] += dur
    out = {}
    for cls, ctrs in acc.items():
        out[cls] = {cn: {"per_dispatch": s / n, "dispatches": n, "avg_us": dsum / n / 1e3} for cn, (s, n, dsum) in ctrs.items() if n}
    # the workload the passes were taken on (tools/run_pmc.sh runs `bench.py` with its defaults): bench.py attaches these counters to a
    # run only when its own workload is the same
    out["_meta"] = json.loads(os.environ.get("FVHD_PMC_META", '{"res": 1024, "batch": 32, "hidden": 896}'))
    out["_meta"]["read_factor"] = dict(READ_FACTOR, default=2.0)
    json_path = os.path.join("profiles", f"{tag}_pmc_summary.json")
    json.dump(out, open(json_path, "w"), indent=1, sort_keys=True)
    print(f"# PMC summary {tag} (rocprofv3 --pmc, one pass per counter group, B = 32 @1024^2; tools/run_pmc.sh + tools/pmc_summary.py)\n")
    print("Per kernel class, per dispatch.  HBM read = 2 x FETCH_SIZE KiB (gfx950 correction; 1 x for the stem's 64-B gather requests, calibrated), write = WRITE_SIZE KiB; MFMA busy = "
          "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch cycles), dispatch cycles from GRBM_GUI_ACTIVE when collected (else 2.0 GHz x duration).\n")
    print("| class | dispatches | avg us (trace) | HBM read MB | HBM write MB | HBM GB/s | MFMA busy % | clock GHz | LDS bank-conflict % | wave-cycles: active / issue-stall / waitcnt % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    for cls in dict.fromkeys(c for c, _ in CLASSES):       # a class may have several patterns: one row
        c = out.get(cls)
        if not c:
            continue
        g = lambda k: c[k]["per_dispatch"] if k in c else None
        us = c["_trace"]["avg_us"] if "_trace" in c else next(iter(c.values()))["avg_us"]
        rd = READ_FACTOR.get(cls, 2.0) * g("FETCH_SIZE") * 1024 / 1e6 if g("FETCH_SIZE") is not None else None
        wr = g("WRITE_SIZE") * 1024 / 1e6 if g("WRITE_SIZE") is not None else None
        gbs = (rd + wr) / us * 1e3 if rd is not None and wr is not None else None      # MB / us = TB/s
        gui = g("GRBM_GUI_ACTIVE")                                                         # summed over the 8 XCD instances
        clk = gui / 8.0 / (c["GRBM_GUI_ACTIVE"]["avg_us"] * 1e3) if gui else None         # cycles per ns
        mf = g("SQ_VALU_MFMA_BUSY_CYCLES")
        busy = None
        if mf is not None:
            cyc = (clk or 2.0) * c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_us"] * 1e3
            busy = 100 * mf / (1024 * cyc)
        bc = 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None
        wc = g("SQ_WAVE_CYCLES")
        mix = "-" if not wc else "%.0f / %.0f / %.0f" % (100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc)
        f = lambda v, p="%.1f": "-" if v is None else p % v
        n = max(v["dispatches"] for v in c.values())
        print(f"| {cls} | {n} | {us:.1f} | {f(rd)} | {f(wr)} | {f(gbs, '%.0f')} | {f(busy)} | {f(clk, '%.2f')} | {f(bc)} | {mix} |")
    print(f"\n(JSON: `{json_path}`; `bench.py` reads `roofline.traffic` from the newest `profiles/*_pmc_summary.json` by class.)")


if __name__ == "__main__":
    main()
