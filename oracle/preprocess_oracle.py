"""TEST INFRASTRUCTURE ONLY - CPU restatement of the image preprocessing in front of encode_images():
`process_images` / `expand2square` (`llava/mm_utils.py:154-184`) around the HF `CLIPImageProcessor` the tower builds
(`mobileclip_encoder.py:45-49`: bicubic resize of the shortest edge to R, centre crop R x R, rescale 1/255, mean 0 / std 1).

The arithmetic lives in third-party code that is not under /root/reference: Pillow's 8-bit separable resampler
(`src/libImaging/Resample.c`: `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`; the reference's pyproject does not pin Pillow - 12.2.0 is what this image has) and
`transformers`' image-processor glue (resize -> center_crop -> rescale -> normalize).  The published algorithm is restated here
in numpy (integer arithmetic, so parity is bit-exact) and pinned by tests/test_preprocess.py against Pillow itself, against
`CLIPImageProcessor.preprocess`, and against the reference's `process_images(..., image_aspect_ratio='pad')`.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits of pixel, 2 bits of filter overshoot


def _bicubic(x: float) -> float:
    """Resample.c `bicubic_filter`, a = -0.5 (Keys), same operation order"""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """`precompute_coeffs` + `normalize_coeffs_8bpc` for the full box (0, in_size): bounds [out][2] = (first input index,
    count), integer coefficients [out][ksize] (fixed point, PRECISION_BITS fractional bits)."""
    support_f = 2.0                                   # bicubic support
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0      # antialias when shrinking: the filter widens with the scale
    support = support_f * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)            # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)      # arithmetic shift (floor), then the clip8 table


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`Image.resize((out_w, out_h), BICUBIC)` of an HWC uint8 image: horizontal pass (skipped when the width is unchanged),
    rounding to uint8, then the vertical pass (`ImagingResample`)."""
    h, w, _ = img.shape
    cur = img
    if out_w != w:
        bounds, kk, _ = precompute_coeffs(w, out_w)
        acc = np.full((h, out_w, 3), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc[:, xx, :] += np.tensordot(cur[:, x0:x0 + n, :].astype(np.int64), kk[xx, :n].astype(np.int64), axes=([1], [0]))
        cur = _clip8(acc)
    if out_h != h:
        bounds, kk, _ = precompute_coeffs(h, out_h)
        acc = np.full((out_h, cur.shape[1], 3), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc[yy] += np.tensordot(kk[yy, :n].astype(np.int64), cur[y0:y0 + n].astype(np.int64), axes=([0], [0]))
        cur = _clip8(acc)
    return cur


def expand2square(img: np.ndarray, background=(0, 0, 0)) -> np.ndarray:
    """`mm_utils.py:154-165` on an HWC array: paste centred on a square canvas of the background colour"""
    h, w, _ = img.shape
    if w == h:
        return img
    s = max(w, h)
    out = np.empty((s, s, 3), dtype=np.uint8)
    out[:] = np.asarray(background, dtype=np.uint8)
    if w > h:
        t = (w - h) // 2
        out[t:t + h] = img
    else:
        l = (h - w) // 2
        out[:, l:l + w] = img
    return out


def shortest_edge_size(h: int, w: int, r: int) -> Tuple[int, int]:
    """transformers `get_resize_output_image_size(size=r, default_to_square=False)`: shortest edge -> r, the other int(r * long / short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = r, int(r * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)        # (height, width)


def preprocess(img: np.ndarray, r: int, pad: bool) -> np.ndarray:
    """HWC uint8 RGB -> [3, r, r] float32 in [0, 1]: (expand2square when `pad`) -> bicubic resize of the shortest edge to r ->
    centre crop -> x * (1 / 255) evaluated in float64 and rounded to float32 (np_rescale) -> (x - 0) / 1"""
    if pad:
        img = expand2square(img)
    h, w, _ = img.shape
    nh, nw = shortest_edge_size(h, w, r)
    res = resize_bicubic_u8(img, nh, nw)
    top, left = (nh - r) // 2, (nw - r) // 2
    crop = res[top:top + r, left:left + r]
    return (crop.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1)


# ---- anyres (`llava/mm_utils.py:14-147`): best grid resolution, aspect-preserving resize pasted on a black canvas, S x S patches,
# ---- plus the whole image squeezed to S x S; every patch then goes through the processor (a no-op resize / crop at S x S)
def select_best_resolution(size_wh, candidates):
    w, h = size_wh
    best, best_eff, best_waste = None, 0, float("inf")
    for cw, ch in candidates:                                   # mm_utils.py:31-40
        s = min(cw / w, ch / h)
        eff = min(int(w * s) * int(h * s), w * h)
        waste = cw * ch - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (cw, ch), eff, waste
    return best


def resize_and_pad(img: np.ndarray, target_wh) -> np.ndarray:
    h, w, _ = img.shape
    tw, th = target_wh
    sw, sh = tw / w, th / h                                     # mm_utils.py:57-66
    if sw < sh:
        nw, nh = tw, min(math.ceil(h * sw), th)
    else:
        nh, nw = th, min(math.ceil(w * sh), tw)
    res = resize_bicubic_u8(img, nh, nw)                        # Image.resize default filter for RGB: BICUBIC
    out = np.zeros((th, tw, 3), dtype=np.uint8)
    px, py = (tw - nw) // 2, (th - nh) // 2
    out[py:py + nh, px:px + nw] = res
    return out


def preprocess_anyres(img: np.ndarray, r: int, grid_pinpoints) -> np.ndarray:
    """HWC uint8 -> [1 + patches, 3, r, r] float32 (`process_anyres_image`, mm_utils.py:121-147)"""
    h, w, _ = img.shape
    padded = resize_and_pad(img, select_best_resolution((w, h), [tuple(p) for p in grid_pinpoints]))
    tiles = [resize_bicubic_u8(img, r, r)]
    for i in range(0, padded.shape[0], r):                      # divide_to_patches: row-major
        for j in range(0, padded.shape[1], r):
            tiles.append(padded[i:i + r, j:j + r])
    return np.stack([(t.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1) for t in tiles])
