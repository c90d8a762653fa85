"""Multimodal embedding splice on the GPU - the drop-in for the tensor work of
`LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` (`llava/model/llava_arch.py:233-332`).

`splice_plan` turns the reference's per-sample Python walk into a few batched integer tensor ops (device-agnostic torch: they
run wherever `input_ids` lives) that say where every kept input token starts in its output sequence; `multimodal_splice` then
calls ONE HIP gather kernel (`fvhd_op_splice`, csrc/splice.hip) that writes inputs_embeds, attention mask, position ids and
labels.  Same return convention as the reference (`llava_arch.py:317-332`): labels / attention_mask / position_ids come back as
None when they were passed as None.

Quirks of the reference that are kept (tests/test_splice.py pins them against the reference itself):
* padding is removed first using the attention mask (`:228-231`); a sample WITHOUT an image token still consumes one entry of
  the image-feature list (`:248-256`, `cur_image_idx += 1`);
* image positions get IGNORE_INDEX labels (`:283`); sequences are truncated to `tokenizer_model_max_length` AFTER the splice (`:292-296`);
* padding side "left" right-aligns every sequence (`:306-314`); position ids restart at 0 at the first kept token, padding gets 0.
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _lib

IGNORE_INDEX = -100          # llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # llava/constants.py:8


def flatten_features(image_features: Union[torch.Tensor, Sequence[torch.Tensor]]) -> Tuple[torch.Tensor, torch.Tensor]:
    """[n, T, H] tensor or list of [T_i, H] tensors (anyres gives different T_i, llava_arch.py:161-208) -> ([rows, H], lens [n] int64)"""
    if isinstance(image_features, torch.Tensor):
        if image_features.dim() != 3:
            raise ValueError(f"image_features must be [n_images, tokens, hidden], got {tuple(image_features.shape)}")
        n, t, h = image_features.shape
        return image_features.reshape(n * t, h), torch.full((n,), t, dtype=torch.int64, device=image_features.device)
    feats = [f for f in image_features]
    if not feats or any(f.dim() != 2 for f in feats):
        raise ValueError("image_features must be a non-empty list of [tokens, hidden] tensors")
    lens = torch.tensor([f.shape[0] for f in feats], dtype=torch.int64, device=feats[0].device)
    return torch.cat(feats, 0), lens


def splice_plan(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], feat_lens: torch.Tensor,
                max_length: Optional[int] = None):
    """-> start [B, L] int32, seqlen [B] int32, feat_row0 [B, L] int64, keep [B, L] bool, max_len (int; the ONE host sync).
    start[b, j] = output position of input token j inside sample b's spliced sequence (exclusive prefix sum of the token
    lengths: 0 for positions the attention mask drops, the image's row count for a -200 token, 1 otherwise)."""
    if input_ids.dim() != 2:
        raise ValueError(f"input_ids must be [batch, length], got {tuple(input_ids.shape)}")
    keep = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    is_img = (input_ids == IMAGE_TOKEN_INDEX) & keep
    n_img = is_img.sum(1)
    slots = torch.clamp(n_img, min=1)                         # llava_arch.py:248-256: no image token still uses one feature entry
    base = torch.cumsum(slots, 0) - slots
    rank = torch.cumsum(is_img.to(torch.int64), 1) - is_img.to(torch.int64)
    n_feat = feat_lens.shape[0]
    img_idx = torch.clamp(base[:, None] + rank, max=n_feat - 1)
    feat_off = torch.cumsum(feat_lens, 0) - feat_lens
    tok_len = torch.where(is_img, feat_lens[img_idx], torch.ones_like(input_ids)) * keep
    start = torch.cumsum(tok_len, 1) - tok_len
    seqlen = tok_len.sum(1)
    if max_length is not None:
        seqlen = torch.clamp(seqlen, max=int(max_length))      # llava_arch.py:292-296
    feat_row0 = torch.where(is_img, feat_off[img_idx], torch.full_like(input_ids, -1))
    need, max_len = torch.stack([slots.sum(), seqlen.max()]).tolist()      # one device-to-host transfer for both scalars
    if need > n_feat:
        raise ValueError(f"the batch needs {need} image-feature entries, {n_feat} given")
    return start.to(torch.int32), seqlen.to(torch.int32), feat_row0.to(torch.int64), keep, int(max_len)


def splice_sources(ids: torch.Tensor, start: torch.Tensor, seqlen: torch.Tensor, feat_row0: torch.Tensor, max_len: int, left_pad: bool):
    """The source of every output row [B, max_len], as the kernel's binary search finds it (csrc/splice.hip) - in torch, for the
    BACKWARD pass only: (table_row, feat_row), each -1 where the row does not come from that source (padding rows: both -1)."""
    B, L = ids.shape
    t = torch.arange(max_len, device=ids.device, dtype=torch.int64)[None, :]
    ln = seqlen.to(torch.int64)[:, None]
    u = t - ((max_len - ln) if left_pad else torch.zeros_like(ln))
    valid = (u >= 0) & (u < ln)
    uc = torch.clamp(u, min=0)
    st = start.to(torch.int64)
    j = torch.clamp(torch.searchsorted(st, uc.contiguous(), right=True) - 1, min=0, max=L - 1)    # last j with start[b, j] <= u
    f0 = feat_row0.gather(1, j)
    is_img = valid & (f0 >= 0)
    feat_row = torch.where(is_img, f0 + (uc - st.gather(1, j)), torch.full_like(f0, -1))
    table_row = torch.where(valid & (f0 < 0), ids.gather(1, j), torch.full_like(f0, -1))
    return table_row, feat_row


def splice_backward(grad_out: torch.Tensor, table_row: torch.Tensor, feat_row: torch.Tensor, vocab: int, n_feat_rows: int):
    """d(inputs_embeds) -> (d embed_tokens.weight [vocab, H], d image features [n_feat_rows, H]): the transpose of the gather, i.e. a
    scatter-add of the output-row gradients onto their source rows (a token id may occur many times; a feature row occurs at most once)."""
    H = grad_out.shape[-1]
    g = grad_out.reshape(-1, H)
    tr, fr = table_row.reshape(-1), feat_row.reshape(-1)
    gt = torch.zeros((vocab, H), dtype=grad_out.dtype, device=grad_out.device)
    gf = torch.zeros((n_feat_rows, H), dtype=grad_out.dtype, device=grad_out.device)
    mt = (tr >= 0) & (tr < vocab)
    mf = (fr >= 0) & (fr < n_feat_rows)
    gt.index_add_(0, tr[mt], g[mt])
    gf.index_add_(0, fr[mf], g[mf])
    return gt, gf


def _launch(ids, start, seqlen, row0, lab_in, table, feats, max_len: int, left_pad: bool):
    """ONE kernel: inputs_embeds + attention mask + position ids (+ labels).  Launched with the tensors' device current
    (the op-level C entry points take a stream, not a device: the stream handle must belong to the device of the pointers)."""
    dev = table.device
    B, L = ids.shape
    V, H = table.shape
    out = torch.empty((B, max_len, H), device=dev, dtype=table.dtype)
    mask_out = torch.empty((B, max_len), device=dev, dtype=torch.uint8)
    pos_out = torch.empty((B, max_len), device=dev, dtype=torch.int64)
    lab_out = None if lab_in is None else torch.empty((B, max_len), device=dev, dtype=torch.int64)
    p = lambda t: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fvhd_op_splice(_lib.stream_ptr(dev), p(ids), p(start), p(seqlen), p(row0), p(lab_in), p(table), p(feats),
                                              p(out), p(mask_out), p(pos_out), p(lab_out), B, L, H, max_len, V, feats.shape[0],
                                              int(left_pad), _lib.dtype_code(table.dtype)), "fvhd_op_splice")
    return out, mask_out, pos_out, lab_out


class _SpliceFn(torch.autograd.Function):
    """The splice kernel with a backward: training (`labels` given, a trainable `mm_projector` and / or `embed_tokens`,
    llava/train/train.py) needs d(inputs_embeds) to reach the image features and the embedding table.  Forward = the same HIP
    kernel as inference; backward = `splice_backward` (index_add_ on the device)."""

    @staticmethod
    def forward(ctx, table, feats, ids, start, seqlen, row0, lab_in, max_len, left_pad):
        out, mask_out, pos_out, lab_out = _launch(ids, start, seqlen, row0, lab_in, table, feats, max_len, left_pad)
        ctx.save_for_backward(ids, start, seqlen, row0)
        ctx.meta = (max_len, left_pad, table.shape[0], feats.shape[0])
        ctx.mark_non_differentiable(mask_out, pos_out)
        if lab_out is not None:
            ctx.mark_non_differentiable(lab_out)
        return out, mask_out, pos_out, lab_out

    @staticmethod
    def backward(ctx, g_out, *_):
        ids, start, seqlen, row0 = ctx.saved_tensors
        max_len, left_pad, vocab, n_rows = ctx.meta
        tr, fr = splice_sources(ids, start, seqlen, row0, max_len, left_pad)
        gt, gf = splice_backward(g_out.contiguous(), tr, fr, vocab, n_rows)
        return (gt if ctx.needs_input_grad[0] else None, gf if ctx.needs_input_grad[1] else None) + (None,) * 7


def multimodal_splice(input_ids: torch.Tensor, position_ids: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor],
                      labels: Optional[torch.Tensor], image_features, embed_weight: torch.Tensor,
                      padding_side: str = "right", max_length: Optional[int] = None):
    """Returns (None, position_ids, attention_mask, None, inputs_embeds, labels) exactly like the reference
    (`llava_arch.py:332`; the fourth entry is past_key_values, passed through by the caller).  Differentiable with respect to
    `embed_weight` and the image features when gradients are enabled (scatter-add backward), so the drop-in also serves the
    reference's training contract (`labels` given, trainable projector / embeddings)."""
    if embed_weight.device.type != "cuda":
        raise RuntimeError("multimodal_splice (MI355X): tensors must be on a HIP device - this path has no CPU implementation")
    dev = embed_weight.device
    feats, lens = flatten_features(image_features)
    feats = feats.to(device=dev, dtype=embed_weight.dtype).contiguous()
    ids = input_ids.to(dev).contiguous()
    am = None if attention_mask is None else attention_mask.to(dev)
    start, seqlen, row0, _, max_len = splice_plan(ids, am, lens.to(dev), max_length)
    B, L = ids.shape
    H = embed_weight.shape[1]
    lab_in = None if labels is None else labels.to(device=dev, dtype=torch.int64).contiguous()
    if max_len == 0:          # every position masked out: the reference returns empty [B, 0, ...] tensors (llava_arch.py:297-322)
        out = embed_weight.new_zeros((B, 0, H))
        mask_out = torch.zeros((B, 0), device=dev, dtype=torch.uint8)
        pos_out = torch.zeros((B, 0), device=dev, dtype=torch.int64)
        lab_out = None if labels is None else torch.zeros((B, 0), device=dev, dtype=torch.int64)
    else:
        args = (embed_weight.contiguous(), feats, ids, start.contiguous(), seqlen.contiguous(), row0.contiguous(), lab_in, max_len,
                padding_side == "left")
        if torch.is_grad_enabled() and (embed_weight.requires_grad or feats.requires_grad):
            out, mask_out, pos_out, lab_out = _SpliceFn.apply(*args)
        else:
            out, mask_out, pos_out, lab_out = _launch(args[2], args[3], args[4], args[5], lab_in, args[0], feats, max_len, args[8])
    new_mask = None if attention_mask is None else mask_out.to(attention_mask.dtype)      # llava_arch.py:325-328
    new_pos = None if position_ids is None else pos_out.to(position_ids.dtype)            # :330-331
    new_lab = None if labels is None else lab_out.to(labels.dtype)                        # the reference builds them with labels.dtype (:299)
    return None, new_pos, new_mask, None, out, new_lab


def _unpad_window(cur_h: int, cur_w: int, orig_w: int, orig_h: int):
    """rows / columns of a [cur_h, cur_w] feature map that hold the image when an orig_w x orig_h picture was fitted into it with
    its aspect ratio kept and centred (`unpad_image`, llava_arch.py:100-128): (row slice, column slice)"""
    if orig_w / orig_h > cur_w / cur_h:                 # the picture is wider than the map: bars above and below
        new_h = int(orig_h * (cur_w / orig_w))
        pad = (cur_h - new_h) // 2
        return slice(pad, cur_h - pad), slice(0, cur_w)
    new_w = int(orig_w * (cur_h / orig_h))
    pad = (cur_w - new_w) // 2
    return slice(0, cur_h), slice(pad, cur_w - pad)


def merge_patch_features(features: Sequence[torch.Tensor], image_sizes, merge_type: str = "flat", grid_pinpoints=None,
                         image_size: int = 1024, image_newline: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """Feature-side half of anyres (`llava_arch.py:165-206`): per image, the [1 + patches, T, H] tokens of the squeezed picture and
    of its grid patches become ONE token sequence - 'flat': plain concatenation; 'spatial': the patch maps are laid out as one
    big [rows, cols] map after the base image's tokens; 'spatial_unpad': the big map is cropped to the picture (the bars the
    black canvas added carry no image) and every row gets the learned `image_newline` token appended.  Plain tensor views and
    copies on whatever device the features live on - no kernel: this is re-layout, not arithmetic.
    image_sizes: (width, height) of every original picture."""
    if merge_type == "flat":
        return [f.flatten(0, 1) for f in features]
    if not merge_type.startswith("spatial"):
        raise ValueError(f"Unexpected mm_patch_merge_type: {merge_type}")
    unpad = "unpad" in merge_type
    if unpad and image_newline is None:
        raise ValueError("mm_patch_merge_type with 'unpad' needs the model's image_newline parameter")
    from .preprocess import _best_resolution
    out = []
    for f, size in zip(features, image_sizes):
        if f.shape[0] == 1:                              # a single tile: its tokens (+ one newline token)
            g = f[0]
            out.append(torch.cat([g, image_newline[None].to(g.device, g.dtype)], 0) if unpad else g)
            continue
        if grid_pinpoints is None:
            raise NotImplementedError("spatial merge of several tiles is defined for image_aspect_ratio='anyres' only")
        if isinstance(grid_pinpoints, str):
            import ast
            grid_pinpoints = ast.literal_eval(grid_pinpoints)
        ow, oh = size
        base, tiles = f[0], f[1:]
        side, hid = int(round(base.shape[0] ** 0.5)), base.shape[-1]
        if side * side != base.shape[0]:
            raise ValueError(f"{base.shape[0]} tokens per tile is not a square map")
        bw, bh = _best_resolution(int(ow), int(oh), [tuple(p) for p in grid_pinpoints])
        gw, gh = bw // image_size, bh // image_size
        big = tiles.reshape(gh, gw, side, side, hid).permute(0, 2, 1, 3, 4).reshape(gh * side, gw * side, hid)
        if unpad:
            rows, cols = _unpad_window(gh * side, gw * side, int(ow), int(oh))
            big = big[rows, cols]
            nl = image_newline.to(big.device, big.dtype).expand(big.shape[0], 1, hid)
            big = torch.cat([big, nl], 1)
        out.append(torch.cat([base, big.reshape(-1, hid)], 0))
    return out
