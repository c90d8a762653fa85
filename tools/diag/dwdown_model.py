"""numpy model of csrc/dwconv_down.hip's index maps (one wave, one strip): transposition destinations, operand reads, Toeplitz^T operands,
the iteration / slot schedule - checked against a direct stride-2 conv.  CPU only; a design aid, not a test of the kernel."""
import numpy as np

PP, CHE = 40, 80
rng = np.random.default_rng(0)


def dd_step(k):
    if k < 24:
        op, rem = divmod(k, 6)
        grp, tile = rem >> 1, rem & 1
        return (0, 5 - 2 * grp, grp, op, tile)
    j, late = (k - 24) & 15, (k - 24) >> 4
    which, jj = j & 1, j >> 1
    tile, op = jj & 1, jj >> 1
    if late == 0:
        return (1, 6, 0, op, tile) if which == 0 else (1, 4, 1, op, tile)
    return (1, 2, 2, op, tile) if which == 0 else (1, 0, 3, op, tile)


def run(H, W, strip, ylo, yhi):
    OH, OW = (H + 1) // 2, (W + 1) // 2
    xi0, xo0 = 64 * strip, 32 * strip
    x = rng.standard_normal((H, W)).astype(np.float32)          # one input channel
    taps = rng.standard_normal((7, 7)).astype(np.float32)        # one output channel
    bias = 0.37
    # reference
    xp = np.zeros((H + 6, W + 6), np.float32); xp[3:3 + H, 3:3 + W] = x
    ref = np.zeros((OH, OW), np.float32)
    for o in range(OH):
        for j in range(OW):
            ref[o, j] = bias + (xp[2 * o:2 * o + 7, 2 * j:2 * j + 7] * taps).sum()
    # transposed image of one input row (channel 0): plane 0 = E', 1 = O
    def image(r):
        T = np.zeros(CHE, np.float32)
        if not (0 <= r < H):
            return T
        for lane in range(64):
            c0 = lane + 4
            if 0 <= xi0 - 4 + c0 < W:
                T[(c0 & 1) * PP + ((c0 - 1) // 2 if c0 & 1 else c0 // 2 - 1)] = x[r, xi0 - 4 + c0]
        for lane in range(8):
            c1 = lane if lane < 4 else 64 + lane
            if 0 <= xi0 - 4 + c1 < W and c1 != 0:
                T[(c1 & 1) * PP + ((c1 - 1) // 2 if c1 & 1 else c1 // 2 - 1)] = x[r, xi0 - 4 + c1]
        return T
    def bop(ky, op, q):                                          # A-lane (b, i = q): 4 values over k
        v = np.zeros(4, np.float32)
        for k in range(4):
            d = k - q
            if op == 0 and d >= 0: v[k] = taps[ky, 2 * d]
            if op == 1 and d <= -1: v[k] = taps[ky, 2 * (4 + d)]
            if op == 2 and 0 <= d <= 2: v[k] = taps[ky, 2 * d + 1]
            if op == 3 and d <= -2: v[k] = taps[ky, 2 * (4 + d) + 1]
        return v
    out = np.full((OH, 32), np.nan, np.float32)
    acc = np.full((4, 2, 4, 4), bias, np.float32)                # [slot][tile][lane j][reg i] -> pixel 16 tile + 4 j + i
    t0, t1 = ylo - 2, yhi
    u = 0
    for t in range(t0, t1 + 1):
        imgs = [image(2 * t), image(2 * t + 1)]
        for k in range(56):
            row, ky, slot, op, tile = dd_step(k)
            sl = (u + slot) & 3
            if ky == 0 and op == 0:
                acc[sl][tile][:] = bias
            for j in range(4):                                   # B-lane (b, j): pixels
                base = (PP if op < 2 else 0) + 16 * tile + 4 * j + 4 * (op & 1)
                px = imgs[row][base:base + 4]
                for i in range(4):
                    acc[sl][tile][j][i] += (bop(ky, op, i) * px).sum()
        fin = acc[u].copy()
        o = t - 1
        if ylo <= o < yhi:
            out[o] = fin.reshape(2, 16).reshape(32)              # [tile][4 j + i]
        u = (u + 1) & 3
    for o in range(ylo, yhi):
        n = min(32, OW - xo0)
        assert np.allclose(out[o, :n], ref[o, xo0:xo0 + n], atol=1e-4), (H, W, strip, o, np.abs(out[o, :n] - ref[o, xo0:xo0 + n]).max())
    return True


for H, W, strip, ylo, yhi in ((16, 16, 0, 0, 8), (33, 67, 1, 0, 17), (33, 67, 0, 8, 16), (10, 130, 2, 0, 5), (7, 5, 0, 0, 4), (2, 2, 0, 0, 1), (64, 128, 1, 16, 32)):
    run(H, W, strip, ylo, yhi)
print("dwdown model: ok")


def store_map():
    """the transposing read of the output row images: every (pixel, channel) of the 32 x 64 row must land in the lane / element the store expects"""
    YP, YSK = 48, 64
    YE = 16 * YP + YSK
    WSYe = (2 * YE * 2 + 8) // 2                                 # wave stride in elements
    img = np.zeros(4 * WSYe, np.int64)                           # element = 1000 px + channel (of the 64)
    for wv in range(4):
        for blk in range(16):
            for px in range(32):
                img[wv * WSYe + blk * YP + YSK * (blk >> 3) + px] = 1000 * px + 16 * wv + blk
    seen = set()
    for sw in range(2):                                          # store waves 2, 3
      for hh in range(2):                                        # the two store instructions
        for h in range(2):                                       # lo / hi read
            src = {}
            for lane in range(64):
                ss = 4 * (lane >> 4) + (lane & 3); so = ss & 7; sj = (lane >> 2) & 3; sr = 8 * (so & 1) + sj
                src[lane] = (so >> 1) * WSYe + (sr + 4 * h) * YP + YSK * (sr >> 3) + 16 * sw + 8 * hh + 4 * (ss >> 3)
            for lane in range(64):
                g = lane >> 4
                ds_ = 4 * g + ((lane >> 2) & 3)
                px = 16 * sw + 8 * hh + 4 * (ds_ >> 3) + (lane & 3)
                for j in range(4):
                    sl = 16 * g + 4 * j + ((lane >> 2) & 3)
                    got = img[src[sl] + (lane & 3)]
                    want = 1000 * px + 8 * (ds_ & 7) + 4 * h + j
                    assert got == want, (sw, hh, h, lane, j, got, want)
                    seen.add(got)
    assert len(seen) == 32 * 64
    print("store map: ok")


store_map()
