"""Image preprocessing on the GPU - the drop-in for `process_images` (`llava/mm_utils.py:168-184`) with the tower's
`CLIPImageProcessor` (`mobileclip_encoder.py:45-49`) for the aspect-ratio modes `'pad'` (expand2square, `mm_utils.py:154-165`)
the default (plain processor) and `'anyres'` (`process_anyres_image`, `mm_utils.py:121-147`: best grid resolution, aspect-preserving
resize on a black canvas, S x S patches + the whole image squeezed to S x S), SURVEY.md 8f-3.  The feature-side re-layout of
anyres patches (`llava_arch.py:165-206`) is `ml_fastvlm_amd.splice.merge_patch_features`.

Host side (this file): the geometry (square canvas, shortest edge -> R, centre crop) and Pillow's coefficient tables
(`precompute_coeffs` + `normalize_coeffs_8bpc` of Resample.c, restated in numpy float64 in the same operation order and cached per
(input size, output size)), restricted to the R x R crop window.  Device side: `fvhd_op_preprocess` (csrc/preprocess.hip) runs
Pillow's two 8-bit passes and the 1/255 rescale.  The result is bit-identical to the reference's CPU pipeline (tests/
test_preprocess.py pins the numpy oracle to Pillow / transformers / the reference function, and the kernels to the oracle).
Images are uint8 HWC RGB tensors ALREADY on the device (a decoded camera frame or a batch staged by the data loader): there is
no CPU path here.
"""
from __future__ import annotations

import collections
import ctypes as C
import functools
import math
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=64)
def _coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's bounds [out][2] and fixed-point taps [out][ksize] for resampling in_size -> out_size (identity when equal: Pillow
    skips the pass, and a single tap of 1.0 reproduces that exactly)."""
    if in_size == out_size:
        b = np.stack([np.arange(out_size), np.ones(out_size, dtype=np.int64)], 1).astype(np.int32)
        return b, np.full((out_size, 1), 1 << PRECISION_BITS, dtype=np.int32)
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast of a non-negative-or-clamped value
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    ss = 1.0 / filterscale
    x = np.arange(ksize, dtype=np.float64)[None, :]
    w = _bicubic((x + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for j in range(ksize):                                                   # the C loop's left-to-right accumulation order
        ww = ww + w[:, j]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    v = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, (-0.5 + v), (0.5 + v)).astype(np.int64).astype(np.int32)   # (int) truncates toward zero
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


def _shortest_edge(h: int, w: int, r: int) -> Tuple[int, int]:
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = r, int(r * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


@functools.lru_cache(maxsize=8)
def _lut(scale: float) -> np.ndarray:
    return (np.arange(256, dtype=np.float64) * scale).astype(np.float32)     # np_rescale: float64 product rounded to float32


class _Plan:
    """device-resident tables of one (source size, mode, R) geometry"""

    def __init__(self, h: int, w: int, r: int, pad: bool, device):
        s = max(h, w) if pad else None
        self.canvas_h, self.canvas_w = (s, s) if pad else (h, w)
        self.pad_top = (s - h) // 2 if pad and w > h else 0
        self.pad_left = (s - w) // 2 if pad and h > w else 0
        nh, nw = _shortest_edge(self.canvas_h, self.canvas_w, r)
        top, left = (nh - r) // 2, (nw - r) // 2
        hb, hc = _coeffs(self.canvas_w, nw)
        vb, vc = _coeffs(self.canvas_h, nh)
        hb, hc, vb, vc = hb[left:left + r], hc[left:left + r], vb[top:top + r], vc[top:top + r]
        self.row0 = int(vb[:, 0].min())
        self.nrows = int((vb[:, 0] + vb[:, 1]).max()) - self.row0
        self.hk, self.vk = hc.shape[1], vc.shape[1]
        dev = lambda a: _upload(a, device)
        self.hb, self.hc, self.vb, self.vc = dev(hb), dev(hc), dev(vb), dev(vc)
        self.lut = dev(_lut(1 / 255))


class _WindowPlan:
    """anyres geometry: the image resampled to (res_h, res_w), placed at (off_y, off_x) of a black canvas, window [win_y, +r) x
    [win_x, +r) of that canvas.  Columns / rows of the window outside the placed image get ZERO taps: the fixed-point sum is then
    the rounding constant alone and clips to 0 = black, exactly the pasted canvas."""

    def __init__(self, h, w, res_h, res_w, off_y, off_x, win_y, win_x, r, device):
        def axis(n_in, n_res, off, win):
            b, k = _coeffs(n_in, n_res)
            idx = np.arange(win, win + r) - off
            ok = (idx >= 0) & (idx < n_res)
            bb = np.zeros((r, 2), dtype=np.int32)
            kk = np.zeros((r, k.shape[1]), dtype=np.int32)
            bb[ok], kk[ok] = b[idx[ok]], k[idx[ok]]
            return bb, kk, ok
        hb, hc, okx = axis(w, res_w, off_x, win_x)
        vb, vc, oky = axis(h, res_h, off_y, win_y)
        self.empty = not (okx.any() and oky.any())
        if self.empty:
            return
        self.pad_top = self.pad_left = 0
        self.row0 = int(vb[oky, 0].min())
        self.nrows = int((vb[oky, 0] + vb[oky, 1]).max()) - self.row0
        vb = vb.copy()
        vb[~oky, 0] = self.row0                               # empty rows: zero taps, any valid first row
        self.hk, self.vk = hc.shape[1], vc.shape[1]
        dev = lambda a: _upload(a, device)
        self.hb, self.hc, self.vb, self.vc = dev(hb), dev(hc), dev(vb), dev(vc)
        self.lut = dev(_lut(1 / 255))


def _upload(a: np.ndarray, device) -> torch.Tensor:
    """tap tables -> device: through pinned memory without a host sync per table (5 tables per plan, 1 + patches plans per anyres image)"""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


class _PlanCache:
    """LRU over the tap tables (a few KB of device memory each): arbitrary-size serving traffic evicts the oldest plan instead of
    wiping the cache (an anyres image alone makes 1 + patches plans)."""

    def __init__(self, capacity: int = 512):
        self.capacity = capacity
        self._d = collections.OrderedDict()

    def get(self, key, make):
        plan = self._d.get(key)
        if plan is None:
            plan = self._d[key] = make()
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)      # its tensors carry record_stream marks (see _run): the allocator reuses them safely
        else:
            self._d.move_to_end(key)
        return plan

    def __len__(self):
        return len(self._d)

    def clear(self):
        self._d.clear()


_plans = _PlanCache()


def _run(plan, image, h, w, r, out):
    p = lambda t: C.c_void_p(t.data_ptr())
    dev = image.device
    # The op-level C entry point takes a stream, not a device: launch with the image's device current so that the stream handle
    # (and the kernel) belong to the device of the pointers - a tower on cuda:1 while cuda:0 is current (device_map) otherwise
    # launches on the wrong GPU.
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        # scratch of the horizontal pass: per call from the stream-ordered caching allocator (a buffer cached in the plan would be
        # shared by concurrent calls on different streams)
        tmp = torch.empty((plan.nrows, r, 3), dtype=torch.uint8, device=dev)
        for t in (plan.hb, plan.hc, plan.vb, plan.vc, plan.lut):
            t.record_stream(stream)              # plans are created on one stream and used on others; eviction must not recycle them early
        _lib.check(_lib.load().fvhd_op_preprocess(C.c_void_p(stream.cuda_stream), p(image), h, w, image.stride(0), plan.pad_top, plan.pad_left, 0,
                                                  p(plan.hb), p(plan.hc), plan.hk, p(plan.vb), p(plan.vc), plan.vk, plan.row0, plan.nrows,
                                                  p(tmp), p(plan.lut), r, p(out), _lib.dtype_code(out.dtype)), "fvhd_op_preprocess")


def _check_image(image):
    if image.device.type != "cuda":
        raise RuntimeError("preprocess (MI355X): the image must be on a HIP device - this path has no CPU implementation")
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise ValueError(f"expected a uint8 [H, W, 3] RGB image, got {image.dtype} {tuple(image.shape)}")
    return image if image.stride(2) == 1 and image.stride(1) == 3 else image.contiguous()


def _best_resolution(w: int, h: int, candidates) -> Tuple[int, int]:
    """the grid resolution that keeps most of the image's pixels after an aspect-preserving fit, least padding on ties, first
    candidate on full ties (`select_best_resolution`, mm_utils.py:14-41)"""
    best, best_key = None, None
    for cw, ch in candidates:
        s = min(cw / w, ch / h)
        eff = min(int(w * s) * int(h * s), w * h)
        key = (eff, eff - cw * ch)
        if best_key is None or key > best_key:
            best, best_key = (int(cw), int(ch)), key
    return best


def process_anyres_image(image: torch.Tensor, image_size: int, grid_pinpoints, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """uint8 [H, W, 3] on the device -> [1 + patches, 3, S, S]: the whole image squeezed to S x S, then the S x S patches (row-major)
    of the image fitted into the best grid resolution on a black canvas (`process_anyres_image`, mm_utils.py:121-147)."""
    image = _check_image(image)
    if isinstance(grid_pinpoints, str):
        import ast
        grid_pinpoints = ast.literal_eval(grid_pinpoints)
    h, w, s = int(image.shape[0]), int(image.shape[1]), int(image_size)
    tw, th = _best_resolution(w, h, [tuple(p) for p in grid_pinpoints])
    sw, sh = tw / w, th / h
    if sw < sh:
        nw, nh = tw, min(math.ceil(h * sw), th)
    else:
        nh, nw = th, min(math.ceil(w * sh), tw)
    off_x, off_y = (tw - nw) // 2, (th - nh) // 2
    windows = [(s, s, 0, 0, 0, 0)] + [(nh, nw, off_y, off_x, i, j) for i in range(0, th, s) for j in range(0, tw, s)]
    out = torch.empty((len(windows), 3, s, s), dtype=dtype, device=image.device)
    for n, (rh, rw, oy, ox, wy, wx) in enumerate(windows):
        key = ("win", h, w, rh, rw, oy, ox, wy, wx, s, image.device.index)
        plan = _plans.get(key, lambda: _WindowPlan(h, w, rh, rw, oy, ox, wy, wx, s, image.device))
        if plan.empty:
            out[n].zero_()
        else:
            _run(plan, image, h, w, s, out[n])
    return out


def preprocess_image(image: torch.Tensor, image_size: int, pad: bool = True, dtype: torch.dtype = torch.float32,
                     out: torch.Tensor = None) -> torch.Tensor:
    """uint8 [H, W, 3] RGB on a HIP device -> [3, R, R] `dtype` in [0, 1]; `pad` = image_aspect_ratio 'pad' (expand2square with the
    processor's mean * 255 = black background)."""
    image = _check_image(image)
    h, w, r = int(image.shape[0]), int(image.shape[1]), int(image_size)
    key = (h, w, r, bool(pad), image.device.index)
    plan = _plans.get(key, lambda: _Plan(h, w, r, bool(pad), image.device))
    if out is None:
        out = torch.empty((3, r, r), dtype=dtype, device=image.device)
    _run(plan, image, h, w, r, out)
    return out


def process_images(images: Sequence[torch.Tensor], image_size: int, image_aspect_ratio: Union[str, None] = "pad",
                   dtype: torch.dtype = torch.float32, grid_pinpoints=None):
    """`mm_utils.process_images` for device-resident uint8 images: the stacked [B, 3, R, R] batch ('pad' / default) or, for
    'anyres', the per-image [1 + patches, 3, R, R] tensors stacked when their shapes agree and returned as a list otherwise."""
    if not images:
        raise ValueError("no images")
    if image_aspect_ratio == "anyres":
        if grid_pinpoints is None:
            raise ValueError("image_aspect_ratio='anyres' needs grid_pinpoints (model_cfg.image_grid_pinpoints)")
        outs = [process_anyres_image(im, image_size, grid_pinpoints, dtype) for im in images]
        return torch.stack(outs, 0) if all(o.shape == outs[0].shape for o in outs) else outs
    batch = torch.empty((len(images), 3, image_size, image_size), dtype=dtype, device=images[0].device)
    for i, im in enumerate(images):
        preprocess_image(im, image_size, pad=image_aspect_ratio == "pad", dtype=dtype, out=batch[i])
    return batch
