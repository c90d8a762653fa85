#!/usr/bin/env python3
"""Cycle stamps of the fused dw3 -> dw7 kernel's row loop (library built with -DFZ_TRACE): per iteration of one producer and one consumer wave,
cycles from the barrier release to the first MFMA's issue, through the thirds of the MFMA stream, to the end-of-iteration wait and the barrier."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ml_fastvlm_amd import _lib
DEV = "cuda:0"
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
for Cc, H, B in ((384, 64, 32), (192, 128, 32)):
    x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
    y, a = torch.empty_like(x), torch.empty_like(x)
    w3, b3 = torch.randn(9, Cc, device=DEV) * 0.15, torch.randn(Cc, device=DEV) * 0.2
    w7, b7 = torch.randn(49, Cc, device=DEV) / 7, torch.randn(Cc, device=DEV) * 0.2
    for _ in range(5):
        _lib.check(lib.fvhd_op_dw3_dw7(st(), p(x), p(y), p(a), p(w3), p(b3), p(w7), p(b7), B, H, H, Cc, None))
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 1024)()
    raw.fvhd_debug_fz_trace(buf, 1024)
    v = list(buf)
    # record dumped at the head of iteration it: ts[0..4] of iteration it - 1 (after its barrier, MFMA 0, thirds, last MFMA),
    # ts[5] / ts[6] = before / after the tail wait at the head of iteration it (= the end of iteration it - 1)
    print(f"--- C={Cc} {H}x{H} B={B}, block 7: [role][wave][iteration]: barrier->mfma0, thirds..., ->tail, tail wait | busy (release -> arrival) | barrier wait | total")
    for role in range(2):
        for wq in range(4):
            recs = [v[(((role * 4 + wq) * 16 + it) * 8):][:8] for it in range(16)]
            for it in range(2, 8):
                t, nxt = recs[it], recs[it + 1]
                if not t[0] or not nxt[0]:
                    continue
                print(f"  {'producer' if role == 0 else 'consumer'} {wq} it {it + 8:3d}: {t[1] - t[0]:5d} {t[2] - t[1]:5d} {t[3] - t[2]:5d} {t[4] - t[3]:5d} {t[5] - t[4]:5d} {t[6] - t[5]:4d} | "
                      f"{t[6] - t[0]:5d} | {nxt[0] - t[6]:5d} | {nxt[0] - t[0]:6d}")
