#!/usr/bin/env python3
"""Bit-level fingerprint of fvhd_op_gemm outputs over the shapes / epilogues the tower and the prefill launch - to compare two builds of
the library (FVHD_LIB=... selects one; e.g. libfvhd.so against libfvhd_g1.so = the round-4 GEMM layout, -DFVHD_GEMM_GRP1):

    python tools/gemm_bits.py > a.json ; FVHD_LIB=ml_fastvlm_amd/libfvhd_g1.so python tools/gemm_bits.py > b.json ; python tools/gemm_bits.py --diff a.json b.json

Every kernel variant computes each output element from the same products in the same K order, so the fingerprints must be IDENTICAL."""
import ctypes as C
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (M, N, K, epilogue): 0 none, 1 bias, 2 bias+gelu, 3 bias+ls+resid, 4 resid, 5 swiglu
    (32768, 2304, 768, 0), (32768, 3072, 768, 2), (32768, 768, 3072, 3), (8192, 4608, 1536, 0), (8192, 6144, 1536, 2), (8192, 1536, 6144, 3),
    (32768, 768, 768, 3), (8192, 896, 3072, 2), (8192, 896, 896, 1), (8192, 3584, 3072, 2),          # stage 3 / 4, projectors
    (131072, 384, 384, 2), (32768, 768, 768, 2),                                                        # PatchEmbed 1x1
    (2304, 1152, 896, 1), (2304, 9728, 896, 5), (2304, 896, 896, 4), (2304, 896, 4864, 4), (256, 151936, 896, 0),   # Qwen2-0.5B prefill (B = 8)
    (2304, 4608, 3584, 1), (2304, 37888, 3584, 5), (2304, 3584, 18944, 4),                              # Qwen2-7B prefill
    (300, 384, 96, 2), (513, 192, 384, 3), (77, 768, 3072, 3), (1024, 768, 768, 3), (256, 4608, 1536, 0), (640, 384, 320, 3), (1, 96, 96, 1),
]


QUICK = [  # --quick: shapes with more tiles than CUs on the streaming 256-row kernels (what FVHD_GEMM_PERSIST=1 turns into a tile loop), every epilogue
    (32768, 2304, 768, 0), (16384, 3072, 768, 2), (32768, 768, 768, 3), (32768, 896, 896, 1), (32768, 768, 768, 2), (2304, 9728, 896, 5), (32768, 1024, 512, 4),
]


def main():
    global SHAPES
    if "--quick" in sys.argv:
        sys.argv.remove("--quick")
        SHAPES = QUICK
    if len(sys.argv) > 1 and sys.argv[1] == "--diff":
        a, b = (json.load(open(f)) for f in sys.argv[2:4])
        bad = [k for k in a["out"] if a["out"][k] != b["out"].get(k)]
        print(f"{a['lib']} vs {b['lib']}: {len(a['out']) - len(bad)} / {len(a['out'])} shapes bit-identical" + (f"; DIFFERENT: {bad}" if bad else ""))
        sys.exit(1 if bad else 0)
    from ml_fastvlm_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    out = {}
    for (M, N, K, epi) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + epi)
        A = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16)
        bias = (torch.randn(N, generator=g) * 0.1).to(dev) if epi in (1, 2, 3) else None
        ls = torch.rand(N, generator=g).to(dev) if epi == 3 else None
        NO = N // 2 if epi == 5 else N
        resid = torch.randn(M, NO, generator=g).to(dev, torch.bfloat16) if epi in (3, 4) else None
        odt = torch.float32 if (N == 151936) else torch.bfloat16
        o = torch.empty(M, NO, dtype=odt, device=dev)
        _lib.check(lib.fvhd_op_gemm(st, p(A), p(W), p(bias), p(ls), p(resid), p(o), M, N, K, epi, _lib.dtype_code(odt)), f"gemm {M}x{N}x{K} epi {epi}")
        torch.cuda.synchronize()
        out[f"{M}x{N}x{K}e{epi}"] = hashlib.sha1(o.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
        del A, W, o
    print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "out": out}))


if __name__ == "__main__":
    main()
