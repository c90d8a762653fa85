"""The half-precision GELU of the fused ConvFFN (csrc/ffn_fused.hip: gelu16_stage), restated in numpy float16 with the kernel's own
coefficient bit patterns (parsed from the source, so the two cannot drift apart), against the exact erf GELU the reference uses
(nn.GELU() default, mci.py:870): the bounds DESIGN.md / include/fvhd.h / INTEGRATION.md state.  CPU only."""
import os
import re

import numpy as np
import pytest
from scipy.special import erf

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml_fastvlm_amd", "csrc", "ffn_fused.hip")
h = np.float16


def _kernel_constants():
    src = open(SRC).read()
    body = src[src.index("void gelu16_stage("):src.index("void gelu16_dispatch(")]
    bits = [int(b, 16) for b in re.findall(r"FFN_H2\(0x([0-9a-fA-F]{4})\)", body)]
    # order of appearance: UMAX, c5, c4, c3, c2, c1, c0, 0.5
    assert len(bits) == 8, bits
    vals = [np.array([b], dtype=np.uint16).view(h)[0] for b in bits]
    return vals[0], vals[1:7], vals[7]


def _fma16(a, b, c):       # one rounding, like v_pk_fma_f16 (the product of two halves is exact in float64)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(h)


def _rtz16(x):             # v_cvt_pkrtz_f16_f32: round toward zero, i.e. saturating at the largest finite half
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        y = x.astype(h)
    y = np.where(np.isinf(y), np.sign(x).astype(h) * h(65504), y).astype(h)
    too_big = np.abs(y.astype(np.float32)) > np.abs(x)
    return np.where(too_big, np.nextafter(y, h(0)), y).astype(h)


def gelu_half16(x_over_4):
    """x' = x / 4 (what GEMM1 delivers) -> (y' = gelu(x) / 4 as half, Phi as half); the instruction sequence of the kernel"""
    umax, (c5, c4, c3, c2, c1, c0), half = _kernel_constants()
    x = _rtz16(x_over_4)
    u = np.minimum((x.astype(np.float64) ** 2).astype(h), umax)
    q = _fma16(np.full_like(u, c5), u, np.full_like(u, c4))
    for c in (c3, c2, c1, c0):
        q = _fma16(q, u, np.full_like(u, c))
    phi = np.clip(_fma16(x, q, np.full_like(u, half)).astype(np.float32), 0.0, 1.0).astype(h)      # the clamp modifier
    with np.errstate(over="ignore"):
        y = (x.astype(np.float32) * phi.astype(np.float32)).astype(h)
    return y, phi


def test_constants_are_the_documented_ones():
    umax, coef, half = _kernel_constants()
    assert float(umax) == (3.5 / 4.0) ** 2 and float(half) == 0.5
    # c_k' = 4 * 16^k * FVHD_GELU5_Ck rounded to half, c0' one ulp up (exact saturation at +-3.5)
    c = [3.980601132e-01, -6.438287348e-02, 8.499878459e-03, -7.195603685e-04, 3.409395140e-05, -6.780236390e-07]
    want = [h(ck * 4 * 16.0 ** k) for k, ck in enumerate(c)]
    want[0] = np.nextafter(want[0], h(10))
    assert [float(v) for v in coef] == [float(v) for v in want[::-1]]


def test_phi_error_and_exact_tails():
    x = np.linspace(-8.0, 8.0, 400001).astype(np.float32)
    y, phi = gelu_half16(x / 4.0)
    ref = 0.5 * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    assert np.abs(phi.astype(np.float64) - ref).max() <= 1.4e-3                      # include/fvhd.h, DESIGN.md
    assert np.sqrt(np.mean((phi.astype(np.float64) - ref) ** 2)) <= 3e-4
    assert (phi[x >= 3.5] == 1).all() and (phi[x <= -3.51] == 0).all()               # gelu(x) = x from 3.5 on, 0 below -3.51, exactly
    assert (phi[(x <= -3.5) & (x > -3.51)] <= h(2.0 ** -13)).all()                   # (in between the half-precision sum leaves one ulp of 2^-13)
    big = np.array([10.0, 1e3, 2.6e5, 3e5, 1e30, -10.0, -1e3, -1e30], np.float32)
    yb, _ = gelu_half16(big / 4.0)
    assert np.isfinite(yb.astype(np.float32)).all()                                  # saturates, never inf / nan
    assert float(yb[3]) == 65504.0 and float(yb[4]) == 65504.0 and float(yb[2]) * 4 <= 2.6e5
    assert (yb[5:] == 0).all()


@pytest.mark.parametrize("sigma", [0.3, 1.0, 2.0, 4.0])
def test_hidden_activation_is_more_accurate_than_a_bf16_hidden_tensor(sigma):
    """the reason the change is a parity improvement and not a trade: rms error of the hidden activation over N(0, sigma) inputs"""
    rng = np.random.default_rng(int(sigma * 10))
    x = rng.normal(0.0, sigma, 200000).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    y, _ = gelu_half16(x / 4.0)
    err16 = np.sqrt(np.mean((4.0 * y.astype(np.float64) - ref) ** 2))
    u = ref.astype(np.float32).view(np.uint32)                                        # exact GELU rounded to bf16 (RNE): the best a bf16 tensor can do
    bf = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).view(np.float32)
    errbf = np.sqrt(np.mean((bf.astype(np.float64) - ref) ** 2))
    rms = np.sqrt(np.mean(ref ** 2))
    assert err16 / rms <= 1.1e-3
    assert err16 <= errbf, (err16, errbf)
