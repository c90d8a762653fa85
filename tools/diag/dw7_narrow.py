import ctypes as C, sys, torch, time
sys.path.insert(0, ".")
from ml_fastvlm_amd import _lib
lib = _lib.load(); DEV = torch.device("cuda", 0)
p = lambda t: C.c_void_p(t.data_ptr()); st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
import torch.nn.functional as F
for (B, H, W, Cc) in ((32, 32, 32, 768), (32, 16, 16, 1536), (16, 48, 48, 768), (16, 24, 24, 1536)):
    x = torch.randn(B, H, W, Cc).to(DEV, torch.bfloat16); w = (torch.randn(49, Cc, device=DEV) / 7); b = torch.randn(Cc, device=DEV)
    y0 = torch.empty_like(x); y1 = torch.empty_like(x)
    def run(fn, y):
        for _ in range(3): fn(y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn(y)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 30 * 1e6
    tv = run(lambda y: _lib.check(lib.fvhd_op_dwconv(st(), p(x), p(y), p(w), p(b), B, H, W, Cc, 7, 1, 1, 0)), y0)
    tm = run(lambda y: _lib.check(lib.fvhd_op_dw7_mfma(st(), p(x), p(y), p(w), p(b), B, H, W, Cc)), y1)
    d = (y0.float() - y1.float()).abs()
    print(f"B={B} {H}x{W} C={Cc}: VALU {tv:.1f} us, MFMA {tm:.1f} us, max diff {d.max().item():.3f} rel {(d.norm()/y0.float().norm()).item():.2e}")
