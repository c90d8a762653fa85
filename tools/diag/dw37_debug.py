#!/usr/bin/env python3
"""Where does fvhd_op_dw3_dw7 differ from the two-kernel route?  Prints error histograms by row / column / channel (debug aid)."""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ml_fastvlm_amd import _lib
DEV = "cuda:0"
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def run(Cc, H, W, B):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cc, H, W, generator=g).to(torch.bfloat16)
    w3 = torch.randn(Cc, 1, 3, 3, generator=g) * 0.15
    w3[:, 0, 1, 1] += 1.0
    b3 = torch.randn(Cc, generator=g) * 0.2
    w7 = torch.randn(Cc, 1, 7, 7, generator=g) / 7
    b7 = torch.randn(Cc, generator=g) * 0.2
    pk = lambda w: w.reshape(w.shape[0], -1).t().contiguous().to(DEV)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y, a, y2, a2 = (torch.full((B, H, W, Cc), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(4))
    w3d, b3d, w7d, b7d = pk(w3), b3.to(DEV), pk(w7), b7.to(DEV)
    _lib.check(lib.fvhd_op_dw3_dw7(st(), p(xn), p(y), p(a), p(w3d), p(b3d), p(w7d), p(b7d), B, H, W, Cc, None))
    _lib.check(lib.fvhd_op_dwconv(st(), p(xn), p(y2), p(w3d), p(b3d), B, H, W, Cc, 3, 1, 1, 0))
    _lib.check(lib.fvhd_op_dw7_mfma(st(), p(y), p(a2), p(w7d), p(b7d), B, H, W, Cc))
    torch.cuda.synchronize()
    want = F.conv2d(x.float(), w3, b3, padding=1, groups=Cc).permute(0, 2, 3, 1)
    for name, got, ref in (("y vs fp32 conv", y.float().cpu(), want), ("y vs VALU kernel", y.float().cpu(), y2.float().cpu()), ("a vs dw7_mfma(y)", a.float().cpu(), a2.float().cpu())):
        nan = torch.isnan(got)
        d = (got - ref).abs()
        d[nan] = 1e9
        tol = 2.0 ** -7 * torch.maximum(got.abs(), ref.abs()) + 1e-30
        print(f"   differing elements {int((got != ref).sum())}, rel-L2 {float((got - ref).norm() / ref.norm()):.3e}")
        bad = d > tol
        print(f"[C={Cc} H={H} W={W} B={B}] {name}: NaN {int(nan.sum())}, beyond one ulp {int(bad.sum())} of {bad.numel()}, max |diff| {float(d[~nan].max()) if (~nan).any() else -1:.4g}")
        if bad.any():
            idx = bad.nonzero()
            print("   rows   ", torch.bincount(idx[:, 1], minlength=H).tolist())
            print("   cols   ", torch.bincount(idx[:, 2], minlength=W).tolist())
            print("   ch % 16", torch.bincount(idx[:, 3] % 16, minlength=16).tolist())
            print("   images ", torch.bincount(idx[:, 0], minlength=B).tolist())
            b_, r_, c_, ch_ = idx[0].tolist()
            print("   first:", idx[0].tolist(), "got", float(got[b_, r_, c_, ch_]), "ref", float(ref[b_, r_, c_, ch_]))


if __name__ == "__main__":
    shapes = [(64, 40, 64, 1), (64, 5, 64, 1), (64, 2, 64, 1), (64, 24, 20, 1)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for s in shapes:
        run(*s)
