"""Builds `ml_fastvlm_amd/libfvhd.so` (the C-ABI library of include/fvhd.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
`.so` is git-ignored but travels with the tree to the GPU box.  Objects are rebuilt only when a
source (or a header) is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libfvhd.so")
SOURCES = ["dwconv.hip", "dwconv_mfma.hip", "dwconv_fused.hip", "dwconv_down.hip", "gemm.hip", "attention.hip", "stem_head.hip", "ffn_fused.hip", "splice.hip", "preprocess.hip", "llm.hip", "llm_api.hip", "fvhd_api.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "-fno-gpu-rdc"]
# Per-file extras.  ffn_fused.hip: packed fp32 VALU (v_pk_*) beside MFMAs is slower than the scalar forms and loses the
# |x| / -x operand modifiers (MI355X_MICROARCH "price of one filler beside MFMAs"), so the SLP vectoriser stays off there.
# Everywhere else it stays ON: a wave64 VALU instruction issues every ~4 cycles packed or not, so v_pk_fma_f32 halves the
# instruction count of the VALU-bound depthwise kernels (measured: 588 -> 380 instructions per tap row of the dw7x7).
# -pragma-unroll-threshold: the 48-slot software pipeline of ffn_fused.hip must be FULLY unrolled (all register-array
# indices compile-time); above the default 16 K-instruction threshold LLVM silently keeps a loop and the accumulators
# land in scratch (2.7 KB/lane).
# -packed-fp32-ops (round 3): no v_pk_*_f32 at all in the kernels whose VALU work runs beside MFMAs - the explicit f32x2 code of the
# GELU / softmax still produced 50-90 of them per loop body, and two behind one MFMA turn a 33-cycle slot into 60 cycles
# (tools/ubench/f16_rate.hip "shadow"); bit-identical results, -1 % on the fused FFN, -2 % on attention (profiles/r03_nopk_ab.log).
# (the host pass prints "'-packed-fp32-ops' is not a recognized feature for this target": it is a device feature, harmless)
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# -amdgpu-mfma-vgpr-form (round 3): MFMA accumulators in VGPRs.  By default hipcc puts them in AGPRs and shuttles every value the VALU
# touches through v_accvgpr_read / _write (8 cycles each, tools/ubench/f16_rate.hip): 144 of ~450 VALU instructions per key tile of
# the attention kernel.  Bit-identical results; attention 1.23 -> 1.02 ms per step, stem -2 %, the prefill attention -0.1 ms of TTFT
# (profiles/r03_vgpr_form_ab.log).  Not for ffn_fused.hip (C = 384 needs all 512 registers) nor gemm.hip (neutral).
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
EXTRA_FLAGS = {"ffn_fused.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=200000"] + NOPK,
               "attention.hip": NOPK + VGPR_FORM, "llm.hip": NOPK + VGPR_FORM, "stem_head.hip": VGPR_FORM,
               # dwconv_mfma.hip: 7 x 84 hand-placed MFMA slots, every register-array index compile-time (768 B/lane of scratch otherwise)
               "dwconv_mfma.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               # dwconv_fused.hip: the same row loops (3 x 60 producer slots, 7 x 84 consumer slots)
               "dwconv_fused.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               # dwconv_down.hip: 4 x 56 slots; NOPK: its GELU runs beside the MFMAs
               "dwconv_down.hip": ["-mllvm", "-pragma-unroll-threshold=200000"] + NOPK}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build libfvhd.so)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, ablate: bool = False, tag: str = "", defs=()) -> str:
    """ablate=True (or FVHD_FFN_ABLATE=1 on the command line): the experiment library `libfvhd_ablate.so` with the fused-FFN
    ablation variants and the kernel-selection knobs (fvhd_debug_set_*, -DFVHD_DEBUG_KNOBS) compiled in; tools/bench_ops.py picks
    it with FVHD_LIB for A/B runs.  Never what the package loads by default: the shipped library has no such switches."""
    # tag / defs (experiments only: FVHD_VARIANT_TAG=g5 FVHD_EXTRA_DEFS="-DFVHD_GELU_DEG=5"): a separately named library
    # `libfvhd_<tag>.so` built with extra preprocessor definitions, for same-box A/B runs through FVHD_LIB
    suffix = ("_ablate" if ablate else "") + (f"_{tag}" if tag else "")
    build_dir = BUILD + suffix
    lib = LIB.replace("libfvhd.so", f"libfvhd{suffix}.so")
    os.makedirs(build_dir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "fvhd.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            extra = EXTRA_FLAGS.get(src, []) + list(defs) + (["-DFVHD_DEBUG_KNOBS"] if ablate else []) + (["-DFVHD_FFN_ABLATE"] if ablate and src == "ffn_fused.hip" else [])
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    if force or jobs or _newer(lib, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, ablate=os.environ.get("FVHD_FFN_ABLATE") == "1",
                        tag=os.environ.get("FVHD_VARIANT_TAG", ""), defs=os.environ.get("FVHD_EXTRA_DEFS", "").split()))
