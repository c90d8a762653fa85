"""The identity behind csrc/dwconv_mfma.hip, checked in numpy (no GPU): a 7-tap row convolution of 4-px output tiles equals the
sum over three aligned 4-px input segments s of (segment) x (4x4 Toeplitz block), with exactly the operand formulas the kernel
comments state - A (lane 4b + i): pixels 16t + 4(i + s - 1) + k of channel b; B (lane 4b + j)[k] = tap(kx = 4(s - 1) + k - j + 3);
D (lane 4b + j, register i) = output pixel 16t + 4i + j - and 28 of the 48 products per (tile, tap row) are taps."""
import numpy as np


def _toeplitz_operand(taps_row, s, j):
    """B operand of lane (b, j) for segment s in {0, 1, 2}: 4 values over k"""
    out = np.zeros(4)
    for k in range(4):
        kx = 4 * (s - 1) + k - j + 3
        if 0 <= kx < 7:
            out[k] = taps_row[kx]
    return out


def test_row_convolution_as_toeplitz_blocks():
    rng = np.random.default_rng(0)
    W = 64
    row = rng.standard_normal(W + 8)            # transposed-image row: column 4 + x holds pixel x, 4 zero-able halo columns each side
    row[:1] = 0
    taps = rng.standard_normal(7)
    want = np.array([sum(taps[kx] * row[4 + x + kx - 3] for kx in range(7)) for x in range(W)])
    got = np.zeros(W)
    nonzero = 0
    for t in range(4):                          # 16-px groups
        for s in range(3):                      # segments
            for i in range(4):                  # MFMA row i = 4-px tile i of the group
                a = row[16 * t + 4 * s + 4 * i: 16 * t + 4 * s + 4 * i + 4]          # the kernel's A read: rd[16 t + 4 s] at lane offset 4 i
                for j in range(4):
                    b = _toeplitz_operand(taps, s, j)
                    nonzero += int(np.count_nonzero(b)) if (t == 0 and i == 0) else 0
                    got[16 * t + 4 * i + j] += float(a @ b)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert nonzero == 28                         # of 3 segments x 4 x 4 = 48 products per tile and tap row


def test_seven_live_rows_cover_every_tap_row_once():
    """input row r feeds output rows r + 3 - ky; with slots (yo - r_lo + 3) % 7 the seven rows in flight never share a slot, and
    output row r - 3 is complete once input row r has been applied"""
    H, r_lo = 40, 5
    contrib = {}
    for r in range(r_lo, H):
        u = (r - r_lo) % 7
        slots = set()
        for ky in range(7):
            yo = r + 3 - ky
            slot = (u + 6 - ky) % 7
            assert slot == (yo - r_lo + 3) % 7
            slots.add(slot)
            contrib.setdefault(yo, []).append(ky)
        assert len(slots) == 7
        done = r - 3
        if done - 3 >= r_lo:                    # all of its input rows were inside the processed range
            assert sorted(contrib[done]) == list(range(7))
