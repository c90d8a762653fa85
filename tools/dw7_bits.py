#!/usr/bin/env python3
"""Output digests + launch times of the matrix-core dw7x7 (fvhd_op_dw7_mfma) over fixed shapes: run it under two libraries
(FVHD_LIB=...) and diff the digest lines - a change that only re-orders instructions must leave every digest unchanged.
    python tools/dw7_bits.py [--time]"""
import ctypes as C
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ml_fastvlm_amd import _lib  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
stream = lambda: C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
SHAPES = [(96, 37, 70, 3), (96, 256, 256, 2), (192, 40, 128, 2), (192, 128, 128, 4), (384, 64, 64, 8), (64, 33, 67, 2), (288, 21, 150, 1),
          (768, 32, 32, 8), (128, 5, 16, 2), (64, 3, 64, 1)]


def run(C_, H, W, B, amax=False):
    g = torch.Generator().manual_seed(C_ * 1000 + H * 10 + W)
    x = torch.randn(B, H, W, C_, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(49, C_, generator=g) / 7).to(DEV)
    b = torch.randn(C_, generator=g).to(DEV)
    y = torch.full((B, H, W, C_), 7.0, device=DEV, dtype=torch.bfloat16)
    if amax:
        am = torch.zeros(64, dtype=torch.int32, device=DEV)                  # FVHD_AMAX_SLOTS
        _lib.check(lib.fvhd_op_dw7_amax(stream(), p(x), p(y), p(w), p(b), B, H, W, C_, 1, p(am)), "dw7 amax")
        torch.cuda.synchronize()
        return hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16] + " amax " + str(int(am.max().item()))
    _lib.check(lib.fvhd_op_dw7_mfma(stream(), p(x), p(y), p(w), p(b), B, H, W, C_), "dw7 mfma")
    torch.cuda.synchronize()
    return hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for s in SHAPES:
    print("digest", s, run(*s), run(*s, amax=True))
if "--time" in sys.argv:
    for B in (32, 8, 1):
        for C_, H in ((96, 256), (192, 128), (384, 64), (768, 32)):
            if raw.fvhd_dw7_mfma_supported(B, H, H, C_, 0) == 0:
                continue
            x = torch.randn(B, H, H, C_).to(DEV, torch.bfloat16)
            y = torch.empty_like(x)
            w, b = torch.randn(49, C_, device=DEV) / 7, torch.randn(C_, device=DEV)
            t = timeit(lambda: lib.fvhd_op_dw7_mfma(stream(), p(x), p(y), p(w), p(b), B, H, H, C_))
            print(f"time B={B:2d} C={C_:4d} H={H:3d}: {t:8.1f} us")
