#!/usr/bin/env python3
"""Per-op microbenchmarks through the C ABI at the BASELINE.json config-2 shapes (B=32, 1024x1024):
HIP-event timing on the launch stream, algorithmic FLOP/s and bytes/s.  GPU only.
    python tools/bench_ops.py [ffn] [dw] [gemm] [attn] [stem]
"""
import ctypes as C
import os
import sys

import torch

FFN_PREC = int(__import__('os').environ.get('FVHD_FFN_PREC', '0'))      # 0 = FFN_HALF (default), 1 = FFN_BF16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ml_fastvlm_amd import _lib  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()


def _knobs():
    """the kernel-selection setters exist in the debug build only"""
    raw = C.CDLL(_lib.LIB_PATH)
    if not hasattr(raw, "fvhd_debug_set_dw7_cfg"):
        raise SystemExit("this benchmark switches kernels: build the debug library (FVHD_FFN_ABLATE=1 python -m ml_fastvlm_amd.build) "
                         "and run with FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so")
    return raw


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def timeit(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def bench_ffn(B=32):
    """fused ConvFFN at the B = 32 stage shapes"""
    for Cc, H in ((96, 256), (192, 128), (384, 64)):
        M, HID = B * H * H, 4 * Cc
        g = torch.Generator().manual_seed(0)
        A = torch.randn(M, Cc, generator=g).to(DEV, torch.bfloat16)
        X = torch.randn(M, Cc, generator=g).to(DEV, torch.bfloat16)
        W1 = (torch.randn(HID, Cc, generator=g) * Cc ** -0.5).to(torch.bfloat16).float().contiguous()
        W2 = (torch.randn(Cc, HID, generator=g) * HID ** -0.5).to(torch.bfloat16).float().contiguous()
        nch, che = HID // 32, 32 * Cc
        i1 = torch.empty((nch + 1) * che, dtype=torch.bfloat16)
        i2 = torch.empty(nch * che, dtype=torch.bfloat16)
        _lib.check(lib.fvhd_ffn_pack(Cc, p(W1), p(W2), p(i1), p(i2), FFN_PREC))
        i1, i2 = i1.to(DEV), i2.to(DEV)
        b1 = torch.randn(HID, generator=g).to(DEV) * 0.1
        b2 = torch.randn(Cc, generator=g).to(DEV) * 0.1
        ls = torch.full((Cc,), 0.01, device=DEV)
        t = timeit(lambda: _lib.check(lib.fvhd_op_ffn_fused(stream(), p(A), p(i1), p(b1), p(i2), p(b2), p(ls), p(X), M, Cc, FFN_PREC)))
        fl, by = 16.0 * M * Cc * Cc, 6.0 * M * Cc
        print(f"ffn_fused C={Cc:4d} M={M:8d}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.1f} GB/s (algorithmic)")


def bench_stem(B=32, R=1024):
    img = torch.rand(B, 3, R, R).to(DEV, torch.bfloat16)
    w0, b0 = (torch.randn(27, 96) * 0.3).to(DEV), (torch.randn(96) * 0.1).to(DEV)
    w1, b1 = (torch.randn(9, 96) * 0.3).to(DEV), (torch.randn(96) * 0.1).to(DEV)
    mid = torch.empty(B, R // 2, R // 2, 96, dtype=torch.bfloat16, device=DEV)
    out = torch.empty(B, R // 4, R // 4, 96, dtype=torch.bfloat16, device=DEV)
    t0 = timeit(lambda: _lib.check(lib.fvhd_op_stem_conv(stream(), p(img), 2, p(mid), p(w0), p(b0), B, R)))
    t1 = timeit(lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(mid), p(out), p(w1), p(b1), B, R // 2, R // 2, 96, 3, 2, 1, 1)))
    t2 = timeit(lambda: _lib.check(lib.fvhd_op_stem_fused(stream(), p(img), 2, p(out), p(w0), p(b0), p(w1), p(b1), p(None), p(None), B, R)))
    by = 2.0 * (img.numel() + out.numel())
    print(f"stem[0] {t0*1e6:8.1f} us + stem[1] {t1*1e6:8.1f} us = {(t0+t1)*1e6:8.1f} us;  fused {t2*1e6:8.1f} us ({by/t2/1e9:6.1f} GB/s algorithmic)")


def bench_dw(B=32, modes=(0,)):
    raw = _knobs()
    for mode in modes:
        raw.fvhd_debug_set_dw_mode(mode)
        if mode:
            print(f"--- dw debug mode {mode} ({'stage + store only' if mode == 1 else 'no staging loads'})")
        _bench_dw(B)
    raw.fvhd_debug_set_dw_mode(0)


def bench_dw3cfg():
    raw = _knobs()
    for B in (32, 16):                         # 16 = the half-batch one stream sees inside fvhd_encode at B = 32
        for cfg in (0, 2):
            raw.fvhd_debug_set_dw3_cfg(cfg)
            print(f"--- dw3 config {cfg}, B = {B}")
            _bench_dw(B, only_k3=True)
    for Cc, H, B in ((96, 37, 3), (192, 20, 2), (384, 64, 2), (768, 9, 5), (96, 256, 2)):
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        w = torch.randn(9, Cc, device=DEV)
        bias = torch.randn(Cc, device=DEV)
        outs = []
        for cfg in (0, 1, 2):
            raw.fvhd_debug_set_dw3_cfg(cfg)
            y = torch.full((B, H, H, Cc), 7.0, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 3, 1, 1, 0))
            torch.cuda.synchronize()
            outs.append(y.clone())
        print(f"dw3 cfg agreement C={Cc} H={H} B={B}:", [bool(torch.equal(outs[0], o)) for o in outs[1:]])
    raw.fvhd_debug_set_dw3_cfg(-1)


def bench_dw7cfg():
    """dw7x7 stride 1: 0 = VALU kernel (dwconv.hip), 1 = default dispatch (matrix-core kernel dwconv_mfma.hip where it applies)"""
    raw = _knobs()
    for cfg in (0, 1):
        raw.fvhd_debug_set_dw7_cfg(cfg)
        print(f"--- dw7 config {cfg} ({'VALU kernel' if cfg == 0 else 'default dispatch'})")
        _bench_dw(32, only_k7=True)
    # the two kernels round the taps differently (fp32 vs bf16 operands): agreement to bf16 resolution, ragged sizes included
    for Cc, H, B in ((96, 70, 3), (192, 64, 2), (384, 64, 2), (768, 32, 5), (96, 256, 2)):
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        w = torch.randn(49, Cc, device=DEV) / 7
        bias = torch.randn(Cc, device=DEV)
        outs = []
        for cfg in (0, 5):
            raw.fvhd_debug_set_dw7_cfg(cfg)
            y = torch.full((B, H, H, Cc), 7.0, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 7, 1, 1, 0))
            torch.cuda.synchronize()
            outs.append(y.float())
        d = (outs[0] - outs[1]).abs()
        print(f"dw7 VALU vs MFMA C={Cc} H={H} B={B}: max |diff| {d.max().item():.4f}, rel-L2 {(d.norm() / outs[0].norm()).item():.2e}")
    raw.fvhd_debug_set_dw7_cfg(1)


def bench_dw7nw():
    """matrix-core dw7x7: the 64-channel workgroup (4 waves, 2 per CU) against the 96-channel one (6 waves, 1 per CU) at C = 192 / 384"""
    raw = _knobs()
    for nw in (0, 6, 0, 6):
        raw.fvhd_debug_set_dwm_nw(nw)
        print(f"--- dw7 matrix-core kernel, {'default workgroup' if nw == 0 else '96-channel workgroup wherever C % 96 == 0'}")
        _bench_dw(32, only_k7=True)
    for Cc, H, B in ((192, 64, 2), (384, 70, 2), (384, 64, 32)):
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        w = torch.randn(49, Cc, device=DEV) / 7
        bias = torch.randn(Cc, device=DEV)
        outs = []
        for nw in (0, 6):
            raw.fvhd_debug_set_dwm_nw(nw)
            y = torch.full((B, H, H, Cc), 7.0, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 7, 1, 1, 0))
            torch.cuda.synchronize()
            outs.append(y.clone())
        print(f"dw7 64- vs 96-channel workgroup C={Cc} H={H} B={B}: equal {bool(torch.equal(outs[0], outs[1]))}")
    raw.fvhd_debug_set_dwm_nw(0)


def bench_dw7small():
    """dw7x7 stride 1 at the TTFT batch sizes: VALU kernel, default dispatch, and the matrix-core kernel with forced rows per chunk"""
    raw = _knobs()
    for B in (8, 1):
        for cfg, rc in ((0, 0), (1, 0), (5, 4), (5, 8), (5, 16), (5, 32), (5, 64)):
            raw.fvhd_debug_set_dw7_cfg(cfg)
            raw.fvhd_debug_set_dwm_rc(rc)
            print(f"--- B = {B}: dw7 config {cfg} ({'VALU kernel' if cfg == 0 else 'default dispatch' if cfg == 1 else f'matrix-core kernel, {rc} rows per chunk'})")
            _bench_dw(B, only_k7=True)
    raw.fvhd_debug_set_dw7_cfg(1)
    raw.fvhd_debug_set_dwm_rc(0)


def _bench_dw(B=32, only_k7=False, only_k3=False):
    for K, S, mult, gelu, Cc, H in ((3, 1, 1, 0, 96, 256), (3, 1, 1, 0, 192, 128), (3, 1, 1, 0, 384, 64),
                                    (7, 1, 1, 0, 96, 256), (7, 1, 1, 0, 192, 128), (7, 1, 1, 0, 384, 64),
                                    (7, 1, 1, 0, 768, 32), (7, 1, 1, 0, 1536, 16),
                                    (7, 2, 2, 1, 96, 256), (7, 2, 2, 1, 192, 128), (7, 2, 2, 1, 384, 64), (7, 2, 2, 1, 768, 32),
                                    (3, 2, 1, 1, 96, 512)):
        if only_k7 and not (K == 7 and S == 1):
            continue
        if only_k3 and not (K == 3 and S == 1):
            continue
        OH = H // S
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        y = torch.empty(B, OH, OH, Cc * mult, device=DEV, dtype=torch.bfloat16)
        w = torch.randn(K * K, Cc * mult, device=DEV)
        bias = torch.randn(Cc * mult, device=DEV)
        t = timeit(lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, K, S, mult, gelu)))
        by = 2.0 * (x.numel() + y.numel())
        fl = 2.0 * y.numel() * K * K
        print(f"dwconv K={K} S={S} mult={mult} C={Cc:4d} H={H:3d}: {t*1e6:9.1f} us  {by/t/1e9:7.1f} GB/s  {fl/t/1e12:6.2f} TF/s")


def bench_dwdown(B=int(os.environ.get("BENCH_B", "32"))):
    """PatchEmbed dw7x7 / stride 2 / multiplier 2 + GELU at the four stage boundaries (FVHD_DWDOWN_MFMA=0: the VALU kernel; debug library:
    a sweep of the matrix-core kernel's rows per chunk)"""
    raw = C.CDLL(_lib.LIB_PATH)
    rcs = (0, 8, 16, 32, 64) if hasattr(raw, "fvhd_debug_set_dd_rc") else (0,)
    for Cc, H in ((96, 256), (192, 128), (384, 64), (768, 32)):
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        y = torch.empty(B, H // 2, H // 2, 2 * Cc, device=DEV, dtype=torch.bfloat16)
        w, bias = torch.randn(49, 2 * Cc, device=DEV) / 7, torch.randn(2 * Cc, device=DEV) * 0.2
        by = 2.0 * (x.numel() + y.numel())
        for rc in rcs:
            if rc:
                raw.fvhd_debug_set_dd_rc(rc)
            t = timeit(lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 7, 2, 2, 1)))
            print(f"dw_down Cin={Cc:4d} H={H:3d} B={B} mfma={lib.fvhd_dw7s2_mfma_supported(B, H, H, Cc, 0)} rows/chunk={rc or 'auto'}: {t*1e6:8.1f} us  {by/t/1e9:7.1f} GB/s")
        if rcs != (0,):
            raw.fvhd_debug_set_dd_rc(0)


def bench_dw7s34(B=int(os.environ.get("BENCH_B", "32"))):
    """the stride-1 dw7x7 launches of stages 3 and 4 (RepCPE, ConvFFN.conv): 32-px strips (round 6) against the 64-px strips / the VALU kernel
    (debug library: fvhd_debug_set_dwm_nt(4) restores the round-5 dispatch)"""
    raw = C.CDLL(_lib.LIB_PATH)
    nts = (0, 4, 0, 4) if hasattr(raw, "fvhd_debug_set_dwm_nt") else (0,)
    for Cc, H in ((768, 32), (1536, 16), (384, 32), (768, 16)):
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        y = torch.empty_like(x)
        w, bias = torch.randn(49, Cc, device=DEV) / 7, torch.randn(Cc, device=DEV) * 0.2
        by = 4.0 * x.numel()
        for nt in nts:
            if len(nts) > 1:
                raw.fvhd_debug_set_dwm_nt(nt)
            t = timeit(lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 7, 1, 1, 0)))
            print(f"dw7 C={Cc:4d} H={H:3d} B={B} strips={'round-5 dispatch' if nt == 4 else 'default (32-px strips for maps <= 32 px)'}: {t*1e6:8.1f} us  {by/t/1e9:7.1f} GB/s")
    if len(nts) > 1:
        raw.fvhd_debug_set_dwm_nt(0)


def bench_gemm(B=32):
    shapes = [("stem 1x1", B * 65536, 96, 96, 2), ("down0 1x1", B * 16384, 192, 192, 2), ("down1 1x1", B * 4096, 384, 384, 2),
              ("down2 1x1", B * 1024, 768, 768, 2), ("down3 1x1", B * 256, 1536, 1536, 2),
              ("s3 qkv", B * 1024, 2304, 768, 0), ("s3 proj", B * 1024, 768, 768, 3), ("s3 fc1", B * 1024, 3072, 768, 2),
              ("s3 fc2", B * 1024, 768, 3072, 3), ("s4 qkv", B * 256, 4608, 1536, 0), ("s4 proj", B * 256, 1536, 1536, 3),
              ("s4 fc1", B * 256, 6144, 1536, 2), ("s4 fc2", B * 256, 1536, 6144, 3),
              ("proj0 H896", B * 256, 896, 3072, 2), ("proj2 H896", B * 256, 896, 896, 1),
              ("proj0 H3584", B * 256, 3584, 3072, 2), ("proj2 H3584", B * 256, 3584, 3584, 1),
              # Qwen2-0.5B prefill at B = 8 x 285 tokens (padded to 2304 rows): qkv, o_proj, gate|up (SwiGLU epilogue), down
              ("llm qkv", 2304, 1152, 896, 1), ("llm o_proj", 2304, 896, 896, 4), ("llm gate_up", 2304, 9728, 896, 5), ("llm down", 2304, 896, 4864, 4)]
    raw = _knobs()
    names = ('v1 only (128x128, register prefetch)', 'default dispatch', '256x128 LDS-DMA ring wherever legal', '256x256 tile / 2-stage LDS-DMA wherever legal',
             '256x256 ping-pong (two wave groups one phase apart) wherever legal', '256x128 / 4 waves / BK 32 / two workgroups per CU wherever legal',
             '256x256 / 8 waves / BK 32 / 4-stage ring wherever legal (256x128 / 6 stages otherwise)', '256x128 / 8 waves / BK 32 / 6-stage ring wherever legal',
             'ABLATION 256x256: no LDS-DMA (stale operands, wrong results)', 'ABLATION 256x256: DMA + barriers only (no fragment reads, no MFMAs)')
    variants = tuple(int(v) for v in os.environ.get("BENCH_GEMM_VARIANTS", "0,2,3,4,5,1").split(","))
    for v2 in variants:
      raw.fvhd_debug_set_gemm_v2(v2)
      print(f"--- gemm: {names[v2]}")
      for name, M, N, K, epi in shapes:
        A = torch.randn(M, K).to(DEV, torch.bfloat16)
        W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
        bias, ls = torch.randn(N, device=DEV), torch.rand(N, device=DEV)
        out = torch.randn(M, N).to(DEV, torch.bfloat16)      # (SwiGLU writes the first M * N / 2 elements of it)
        t = timeit(lambda: _lib.check(lib.fvhd_op_gemm(stream(), p(A), p(W), p(bias), p(ls), p(out), p(out), M, N, K, epi, 2)))
        print(f"gemm {name:12s} M={M:8d} N={N:5d} K={K:5d} epi={epi}: {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s")
    for name, M, N, K, epi in [("check fc1-like", 1024, 3072, 768, 2), ("check fc2-like", 1024, 768, 3072, 3), ("check swiglu", 512, 1024, 256, 5)]:
        A = torch.randn(M, K).to(DEV, torch.bfloat16)
        W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
        bias, ls = torch.randn(N, device=DEV), torch.rand(N, device=DEV)
        res = torch.randn(M, N).to(DEV, torch.bfloat16)
        outs = []
        for v2 in (0, 2, 3, 4, 5, 6, 7):
            raw.fvhd_debug_set_gemm_v2(v2)
            out = res.clone() if epi != 5 else torch.zeros(M, N // 2, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fvhd_op_gemm(stream(), p(A), p(W), p(bias), p(ls), p(res), p(out), M, N, K, epi, 2))
            torch.cuda.synchronize()
            outs.append(out.float())
        print(f"gemm {name}: v1 vs 256x128 equal {bool(torch.equal(outs[0], outs[1]))}, v1 vs 256x256 equal {bool(torch.equal(outs[0], outs[2]))}, "
              f"v1 vs ping-pong equal {bool(torch.equal(outs[0], outs[3]))} (max diff {float((outs[0] - outs[3]).abs().max()):.3g}), "
              f"v1 vs two-workgroup BK 32 equal {bool(torch.equal(outs[0], outs[4]))} (max diff {float((outs[0] - outs[4]).abs().max()):.3g}), "
              f"v1 vs BK 32 rings equal {bool(torch.equal(outs[0], outs[5]))} / {bool(torch.equal(outs[0], outs[6]))}")
    raw.fvhd_debug_set_gemm_v2(1)


def bench_gemmsmall():
    """the tower's GEMM route at B = 1 and B = 8 (stages 1-4 fc1 / fc2 below the fused-FFN threshold, qkv, proj): v1 / 256x128 / 256x256 / default"""
    raw = _knobs()
    for B in (1, 8):
        shapes = [("s3 qkv", B * 1024, 2304, 768, 0), ("s3 proj", B * 1024, 768, 768, 3), ("s3 fc1", B * 1024, 3072, 768, 2), ("s3 fc2", B * 1024, 768, 3072, 3),
                  ("s4 qkv", B * 256, 4608, 1536, 0), ("s4 fc1", B * 256, 6144, 1536, 2), ("s4 fc2", B * 256, 1536, 6144, 3)]
        if B == 1:
            shapes = [("s1 fc1", 16384, 768, 192, 2), ("s1 fc2", 16384, 192, 768, 3), ("s2 fc1", 4096, 1536, 384, 2), ("s2 fc2", 4096, 384, 1536, 3)] + shapes
        shapes += [("llm qkv", 2304, 1152, 896, 1), ("llm oprj", 2304, 896, 896, 4), ("llm gtup", 2304, 9728, 896, 5), ("llm down", 2304, 896, 4864, 4)] if B == 8 else \
                  [("llm qkv", 512, 1152, 896, 1), ("llm oprj", 512, 896, 896, 4), ("llm gtup", 512, 9728, 896, 5), ("llm down", 512, 896, 4864, 4)]
        for v2, nm in ((0, "v1"), (10, "v1s"), (2, "256x128"), (3, "256x256"), (1, "default")):
            raw.fvhd_debug_set_gemm_v2(v2)
            print(f"--- B = {B}, gemm: {nm}")
            for name, M, N, K, epi in shapes:
                A = torch.randn(M, K).to(DEV, torch.bfloat16)
                W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
                bias, ls = torch.randn(N, device=DEV), torch.rand(N, device=DEV)
                out = torch.randn(M, N).to(DEV, torch.bfloat16)
                t = timeit(lambda: _lib.check(lib.fvhd_op_gemm(stream(), p(A), p(W), p(bias), p(ls), p(out), p(out), M, N, K, epi, 2)))
                print(f"gemm {name:8s} M={M:6d} N={N:5d} K={K:5d} epi={epi}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s")
    # split-K partial GEMMs + reduce (the prefill's o_proj / down_proj, the tower's fc2 at B = 1): v1 vs v1s slices
    for v2, nm in ((0, "v1"), (10, "v1s")):
        raw.fvhd_debug_set_gemm_v2(v2)
        print(f"--- split-K, slices on {nm}")
        for name, M, N, K, sp in (("llm o_proj", 2304, 896, 896, 2), ("llm down", 2304, 896, 4864, 4), ("s2 fc2 B1", 4096, 384, 1536, 4), ("s3 fc2 B1", 1024, 768, 3072, 8),
                                  ("s4 fc2 B1", 256, 1536, 6144, 16), ("s4 fc2 B8", 2048, 1536, 6144, 2)):
            A = torch.randn(M, K).to(DEV, torch.bfloat16)
            W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
            out = torch.randn(M, N).to(DEV, torch.bfloat16)
            part = torch.empty(sp * M * N, device=DEV, dtype=torch.float32)
            t = timeit(lambda: _lib.check(lib.fvhd_op_gemm_splitk(stream(), p(A), p(W), p(out), p(out), p(part), M, N, K, sp)))
            print(f"split-K {name:10s} M={M:6d} N={N:5d} K={K:5d} / {sp:2d}: {t*1e6:8.1f} us")
    # identical bits: v1 vs v1s, plain and split
    for name, M, N, K, epi in (("proj-like", 1024, 768, 768, 3), ("qkv-like", 256, 4608, 1536, 0), ("llm qkv", 2304, 1152, 896, 1), ("swiglu", 512, 1024, 256, 5),
                               ("one K tile", 128, 128, 128, 2), ("three K tiles", 384, 256, 192, 2)):
        A = torch.randn(M, K).to(DEV, torch.bfloat16)
        W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
        bias, ls = torch.randn(N, device=DEV), torch.rand(N, device=DEV)
        res = torch.randn(M, N).to(DEV, torch.bfloat16)
        outs = []
        for v2 in (0, 10):
            raw.fvhd_debug_set_gemm_v2(v2)
            out = res.clone() if epi != 5 else torch.zeros(M, N // 2, device=DEV, dtype=torch.bfloat16)
            _lib.check(lib.fvhd_op_gemm(stream(), p(A), p(W), p(bias), p(ls), p(res), p(out), M, N, K, epi, 2))
            torch.cuda.synchronize()
            outs.append(out.float())
        parts = []
        for v2 in (0, 10):
            raw.fvhd_debug_set_gemm_v2(v2)
            o2 = res.clone()
            part = torch.zeros(2 * M * N, device=DEV, dtype=torch.float32)
            if K % 128 == 0:
                _lib.check(lib.fvhd_op_gemm_splitk(stream(), p(A), p(W), p(res), p(o2), p(part), M, N, K, 2))
            torch.cuda.synchronize()
            parts.append((o2.float(), part.clone()))
        print(f"v1 vs v1s {name}: plain equal {bool(torch.equal(outs[0], outs[1]))} (max diff {float((outs[0] - outs[1]).abs().max()):.3g}), "
              f"split-K equal {bool(torch.equal(parts[0][0], parts[1][0]) and torch.equal(parts[0][1], parts[1][1]))}")
    raw.fvhd_debug_set_gemm_v2(1)


def bench_dw37(B=int(os.environ.get("BENCH_B", "32"))):
    """RepMixer dw3x3 -> ConvFFN dw7x7: the two-kernel route against the fused launch (csrc/dwconv_fused.hip), interleaved rounds on one box;
    with the debug library also a sweep of the fused kernel's output rows per workgroup (fvhd_debug_set_fz_rc)"""
    raw = C.CDLL(_lib.LIB_PATH)
    shapes = [(Cc, H) for Cc, H in ((192, 128), (384, 64), (64, 256), (96, 256)) if lib.fvhd_dw3_dw7_supported(B, H, H, Cc, 1)]
    for Cc, H in shapes:
        x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
        y, a = torch.empty_like(x), torch.empty_like(x)
        w3, b3 = torch.randn(9, Cc, device=DEV) * 0.15, torch.randn(Cc, device=DEV) * 0.2
        w3[4] += 1.0
        w7, b7 = torch.randn(49, Cc, device=DEV) / 7, torch.randn(Cc, device=DEV) * 0.2
        two = lambda: (_lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w3), p(b3), B, H, H, Cc, 3, 1, 1, 0)),
                       _lib.check(lib.fvhd_op_dwconv(stream(), p(y), p(a), p(w7), p(b7), B, H, H, Cc, 7, 1, 1, 0)))
        one = lambda: _lib.check(lib.fvhd_op_dw3_dw7(stream(), p(x), p(y), p(a), p(w3), p(b3), p(w7), p(b7), B, H, H, Cc, None))
        by = 2.0 * x.numel()
        rounds = [(timeit(two, iters=30), timeit(one, iters=30)) for _ in range(3)]
        t2, t1 = min(r[0] for r in rounds), min(r[1] for r in rounds)
        print(f"dw3+dw7 C={Cc:4d} {H}x{H} B={B}: two launches {t2*1e6:7.1f} us ({4*by/t2/1e12:.2f} TB/s of 4 passes)   fused {t1*1e6:7.1f} us "
              f"({3*by/t1/1e12:.2f} TB/s of 3 passes)   rounds {[(round(r[0]*1e6, 1), round(r[1]*1e6, 1)) for r in rounds]}")
        if hasattr(raw, "fvhd_debug_set_fz_rc"):
            for rc in ((8, 12, 16, 24, 32) if B < 16 else (16, 24, 32, 48, 64, 96, 128)):
                raw.fvhd_debug_set_fz_rc(rc)
                print(f"    output rows per workgroup {rc:3d}: {timeit(one, iters=30)*1e6:7.1f} us")
            raw.fvhd_debug_set_fz_rc(0)


def bench_attn(B=32):
    for N, Cc in ((1024, 768), (256, 1536), (2304, 768), (576, 1536)):
        qkv = torch.randn(B * N, 3 * Cc).to(DEV, torch.bfloat16)
        out = torch.empty(B * N, Cc, device=DEV, dtype=torch.bfloat16)
        t = timeit(lambda: _lib.check(lib.fvhd_op_attention(stream(), p(qkv), p(out), B, N, Cc)))
        fl = 4.0 * B * (Cc // 32) * N * N * 32
        print(f"attention N={N:5d} C={Cc:5d}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a != "all"] or ["ffn", "dw", "gemm", "attn"]
    for w in which:
        {"ffn": bench_ffn, "dw": bench_dw, "dw_ablate": lambda: bench_dw(modes=(0, 1, 2)), "stem": bench_stem, "dwraw": _bench_dw, "dw7cfg": bench_dw7cfg, "dw7small": bench_dw7small, "dw7nw": bench_dw7nw, "dw3cfg": bench_dw3cfg, "gemm": bench_gemm, "gemmsmall": bench_gemmsmall, "attn": bench_attn, "dw37": bench_dw37, "dwdown": bench_dwdown, "dw7s34": bench_dw7s34}[w]()
