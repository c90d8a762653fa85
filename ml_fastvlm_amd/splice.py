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
    """-> start [B, L] int32, seqlen [B] int32, feat_row0 [B, L] int64, keep [B, L] bool, max_len (int; the one host sync).
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
    if int(slots.sum()) > n_feat:
        raise ValueError(f"the batch needs {int(slots.sum())} image-feature entries, {n_feat} given")
    img_idx = torch.clamp(base[:, None] + rank, max=n_feat - 1)
    feat_off = torch.cumsum(feat_lens, 0) - feat_lens
    tok_len = torch.where(is_img, feat_lens[img_idx], torch.ones_like(input_ids)) * keep
    start = torch.cumsum(tok_len, 1) - tok_len
    seqlen = tok_len.sum(1)
    if max_length is not None:
        seqlen = torch.clamp(seqlen, max=int(max_length))      # llava_arch.py:292-296
    feat_row0 = torch.where(is_img, feat_off[img_idx], torch.full_like(input_ids, -1))
    max_len = int(seqlen.max())
    return start.to(torch.int32), seqlen.to(torch.int32), feat_row0.to(torch.int64), keep, max_len


def multimodal_splice(input_ids: torch.Tensor, position_ids: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor],
                      labels: Optional[torch.Tensor], image_features, embed_weight: torch.Tensor,
                      padding_side: str = "right", max_length: Optional[int] = None):
    """Returns (None, position_ids, attention_mask, None, inputs_embeds, labels) exactly like the reference
    (`llava_arch.py:332`; the fourth entry is past_key_values, passed through by the caller)."""
    if embed_weight.device.type != "cuda":
        raise RuntimeError("multimodal_splice (MI355X): tensors must be on a HIP device - this path has no CPU implementation")
    dev = embed_weight.device
    feats, lens = flatten_features(image_features)
    feats = feats.to(device=dev, dtype=embed_weight.dtype).contiguous()
    ids = input_ids.to(dev).contiguous()
    am = None if attention_mask is None else attention_mask.to(dev)
    start, seqlen, row0, _, max_len = splice_plan(ids, am, lens.to(dev), max_length)
    B, L = ids.shape
    V, H = embed_weight.shape
    out = torch.empty((B, max_len, H), device=dev, dtype=embed_weight.dtype)
    mask_out = torch.empty((B, max_len), device=dev, dtype=torch.uint8)
    pos_out = torch.empty((B, max_len), device=dev, dtype=torch.int64)
    lab_in = None if labels is None else labels.to(device=dev, dtype=torch.int64).contiguous()
    lab_out = None if labels is None else torch.empty((B, max_len), device=dev, dtype=torch.int64)
    p = lambda t: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
    _lib.check(_lib.load().fvhd_op_splice(_lib.stream_ptr(dev), p(ids), p(start.contiguous()), p(seqlen.contiguous()), p(row0.contiguous()),
                                          p(lab_in), p(embed_weight.contiguous()), p(feats), p(out), p(mask_out), p(pos_out), p(lab_out),
                                          B, L, H, max_len, V, feats.shape[0], int(padding_side == "left"), _lib.dtype_code(embed_weight.dtype)),
               "fvhd_op_splice")
    new_mask = None if attention_mask is None else mask_out.to(attention_mask.dtype)      # llava_arch.py:325-328
    new_pos = None if position_ids is None else pos_out.to(position_ids.dtype)            # :330-331
    return None, new_pos, new_mask, None, out, lab_out


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
