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
    buf = (C.c_uint64 * 512)()
    raw.fvhd_debug_fz_trace(buf, 512)
    v = list(buf)
    print(f"--- C={Cc} {H}x{H} B={B}: [block][role][iteration]: barrier->mfma0, ->third1, ->third2, ->last mfma, ->tail wait start, wait, barrier+, iteration total")
    for blk in range(2):
        for role in range(2):
            rows = []
            for it in range(16):
                t = v[((blk * 2 + role) * 16 + it) * 8:][:8]
                rows.append(t)
            for it in range(1, 15):
                t, nxt = rows[it], rows[it + 1]
                if not t[0] or not nxt[0]:
                    continue
                # ts[5], ts[6] stamped in iteration it + 1's head belong to the END of iteration it (they are dumped with iteration it + 1)
                print(f"  block {'0' if blk == 0 else '77'} {'producer' if role == 0 else 'consumer'} it {it + 9:3d}: "
                      f"{t[1] - t[0]:5d} {t[2] - t[1]:5d} {t[3] - t[2]:5d} {t[4] - t[3]:5d} | {nxt[5] - t[4]:5d} {nxt[6] - nxt[5]:5d} {nxt[0] - nxt[6]:5d} | {nxt[0] - t[0]:6d}")
