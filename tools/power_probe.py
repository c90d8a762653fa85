#!/usr/bin/env python3
"""Is a kernel power-limited?  Loops ONE op for a few seconds while sampling the GPU's socket power and shader clock
(sysfs hwmon `power1_average` / `freq1_input`, falling back to `rocm-smi`), and prints mean / max power, mean clock and the op's rate.

    python tools/power_probe.py ffn384 ffn192 ffn96 gemm dw7 idle          (FVHD_LIB / FVHD_FFN_VARIANT as for tools/bench_ops.py)
"""
import ctypes as C
import glob
import os
import subprocess
import sys
import threading
import time

import torch

FFN_PREC = int(__import__('os').environ.get('FVHD_FFN_PREC', '0'))      # 0 = FFN_HALF (default), 1 = FFN_BF16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ml_fastvlm_amd import _lib  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
stream = lambda: C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def _pci_bus_id():
    """PCI address of HIP device 0 (the box exposes several cards in sysfs; only one is ours)"""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def _sysfs():
    bus = _pci_bus_id()
    cands = glob.glob(f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*") if bus else []
    cands += glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
    for hw in cands:
        pw = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f))]
        if pw:
            print(f"[power_probe] HIP device 0 = PCI {bus}; reading {hw}/{pw[0]}", file=sys.stderr)
            return os.path.join(hw, pw[0]), (os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None), \
                (os.path.join(hw, "power1_cap") if os.path.exists(os.path.join(hw, "power1_cap")) else None)
    return None, None, None


class Sampler(threading.Thread):
    def __init__(self, period=0.1):
        super().__init__(daemon=True)
        self.period, self.stop, self.pw, self.clk = period, False, [], []
        self.pfile, self.cfile, self.capfile = _sysfs()

    def run(self):
        while not self.stop:
            try:
                if self.pfile:
                    self.pw.append(int(open(self.pfile).read()) / 1e6)
                    if self.cfile:
                        self.clk.append(int(open(self.cfile).read()) / 1e6)
                else:
                    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                    for line in out.splitlines():
                        if "Power (W)" in line or "Socket Power" in line:
                            self.pw.append(float(line.split(":")[-1].strip().split()[0]))
                        if "sclk" in line and "Mhz" in line:
                            self.clk.append(float(line.split("(")[-1].split("Mhz")[0]))
            except Exception:
                pass
            time.sleep(self.period)


def ffn(Cc):
    H = {96: 256, 192: 128, 384: 64}[Cc]
    M, HID = 32 * H * H, 4 * Cc
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
    keep = (A, X, i1, i2, b1, b2, ls)
    return (lambda: _lib.check(lib.fvhd_op_ffn_fused(stream(), p(A), p(i1), p(b1), p(i2), p(b2), p(ls), p(X), M, Cc, FFN_PREC))), 16.0 * M * Cc * Cc, keep


def gemm():
    M, N, K = 32768, 3072, 768
    A = torch.randn(M, K).to(DEV, torch.bfloat16)
    W = (torch.randn(N, K) * K ** -0.5).to(DEV, torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    return (lambda: _lib.check(lib.fvhd_op_gemm(stream(), p(A), p(W), p(bias), p(None), p(None), p(out), M, N, K, 2, 2))), 2.0 * M * N * K, (A, W, bias, out)


def dw7(Cc=192):
    B, H = int(os.environ.get("FVHD_PROBE_B", "32")), {96: 256, 192: 128, 384: 64}[Cc]
    x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(49, Cc, device=DEV)
    bias = torch.randn(Cc, device=DEV)
    return (lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 7, 1, 1, 0))), 2.0 * y.numel() * 49, (x, y, w, bias)


def dwmix(Cc=192):
    """RepMixer dw3x3 -> ConvFFN dw7x7 in one launch (csrc/dwconv_fused.hip, round 6)"""
    B, H = int(os.environ.get("FVHD_PROBE_B", "32")), {192: 128, 384: 64}[Cc]
    x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
    y, a = torch.empty_like(x), torch.empty_like(x)
    w3, b3 = torch.randn(9, Cc, device=DEV) * 0.15, torch.randn(Cc, device=DEV) * 0.2
    w3[4] += 1.0
    w7, b7 = torch.randn(49, Cc, device=DEV) / 7, torch.randn(Cc, device=DEV) * 0.2
    return (lambda: _lib.check(lib.fvhd_op_dw3_dw7(stream(), p(x), p(y), p(a), p(w3), p(b3), p(w7), p(b7), B, H, H, Cc, None))), 2.0 * y.numel() * 58, (x, y, a, w3, b3, w7, b7)


def attn():
    B, N, Cc = 32, 1024, 768
    qkv = torch.randn(B * N, 3 * Cc).to(DEV, torch.bfloat16)
    out = torch.empty(B * N, Cc, device=DEV, dtype=torch.bfloat16)
    return (lambda: _lib.check(lib.fvhd_op_attention(stream(), p(qkv), p(out), B, N, Cc))), 4.0 * B * (Cc // 32) * N * N * 32, (qkv, out)


def dw3():
    B, H, Cc = 32, 128, 192
    x = torch.randn(B, H, H, Cc).to(DEV, torch.bfloat16)
    y = torch.empty_like(x)
    w = torch.randn(9, Cc, device=DEV)
    bias = torch.randn(Cc, device=DEV)
    return (lambda: _lib.check(lib.fvhd_op_dwconv(stream(), p(x), p(y), p(w), p(bias), B, H, H, Cc, 3, 1, 1, 0))), 2.0 * y.numel() * 9, (x, y, w, bias)


def stem():
    B, R = 32, 1024
    img = torch.rand(B, 3, R, R).to(DEV, torch.bfloat16)
    w0, b0 = (torch.randn(27, 96) * 0.3).to(DEV), (torch.randn(96) * 0.1).to(DEV)
    w1, b1 = (torch.randn(9, 96) * 0.3).to(DEV), (torch.randn(96) * 0.1).to(DEV)
    out = torch.empty(B, R // 4, R // 4, 96, dtype=torch.bfloat16, device=DEV)
    return (lambda: _lib.check(lib.fvhd_op_stem_fused(stream(), p(img), 2, p(out), p(w0), p(b0), p(w1), p(b1), p(None), p(None), B, R))), 2.0 * B * (R // 2) ** 2 * 96 * 27, (img, w0, b0, w1, b1, out)


def run(name, seconds=3.0):
    if name == "idle":
        fn, flops, keep = (lambda: None), 0.0, None
    elif name.startswith("ffn"):
        fn, flops, keep = ffn(int(name[3:]))
    elif name == "gemm":
        fn, flops, keep = gemm()
    elif name.startswith("dwmix"):
        fn, flops, keep = dwmix(int(name[5:]))
    elif name in ("attn", "dw3", "stem"):
        fn, flops, keep = {"attn": attn, "dw3": dw3, "stem": stem}[name]()
    else:
        fn, flops, keep = dw7(int(name[4:]) if name.startswith("dw7c") else 192)       # dw7, dw7c96, dw7c384
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
        if name == "idle":
            time.sleep(0.05)
    dt = time.perf_counter() - t0
    s.stop = True
    s.join(timeout=2)
    pw, clk = s.pw[2:] or s.pw or [0.0], s.clk[2:] or s.clk or [0.0]
    cap = None
    try:
        cap = int(open(s.capfile).read()) / 1e6 if s.capfile else None
    except Exception:
        pass
    us = 1e6 * dt / max(n, 1)
    print(f"{name:8s} lib {os.path.basename(_lib.LIB_PATH)} variant {os.environ.get('FVHD_FFN_VARIANT', '0')} prec {FFN_PREC}: {us:8.1f} us/launch  {flops * n / dt / 1e12:7.1f} TF/s | "
          f"energy {sum(pw) / len(pw) * us * 1e-3:7.1f} mJ/launch | power mean {sum(pw) / len(pw):6.0f} W max {max(pw):6.0f} W"
          f" (cap {cap}) | sclk mean {sum(clk) / len(clk):5.0f} MHz | {len(pw)} samples via {'sysfs' if s.pfile else 'rocm-smi'}")


if __name__ == "__main__":
    for w in sys.argv[1:] or ["idle", "ffn384", "ffn192", "ffn96", "gemm", "dw7"]:
        run(w)
