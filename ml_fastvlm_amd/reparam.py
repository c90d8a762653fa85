"""Checkpoint ingest: training-mode (multi-branch) FastViT / FastViTHD state dict -> the inference-mode state dict the
MI355X tower (and the reference's `fastvithd()`, which is built with `inference_mode=True`, `mci.py:1472`) loads.

SURVEY.md 8(f) row 4.  The shipped FastVLM checkpoints are already re-parameterised; this is for checkpoints saved from the
training graph.  It is a restatement, on plain tensors, of what the reference does module by module:

* `MobileOneBlock.reparameterize / _get_kernel_bias / _fuse_bn_tensor`  (`mci.py:219-330`): conv+BN branches, the 1x1 scale
  branch zero-padded to k x k, and the BatchNorm-only skip branch (identity kernel) are each folded with their BatchNorm
  (w * gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var + eps)) and summed;
* `ReparamLargeKernelConv.get_kernel_bias`  (`mci.py:453-515`): large-kernel conv+BN plus the small-kernel conv+BN zero-padded
  to the large size;
* `RepMixer.reparameterize`  (`mci.py:819-859`): w = id + layer_scale * (mixer.w - norm.w), b = layer_scale * (mixer.b - norm.b),
  where mixer / norm are MobileOneBlocks folded as above;
* `RepCPE.reparameterize`  (`mci.py:1000-1039`): w = id + pe.weight, b = pe.bias.

Everything else (ConvFFN, attention, LayerNorm, layer scales of the blocks, SE, head) keeps its key and value.  Module types are
recognised from the key patterns of the training graph, hyper-parameters (kernel size, groups) from tensor shapes; the arithmetic
runs in the dtype of the checkpoint (use fp32 / fp64 tensors).  `tests/test_reparam.py` checks it against the reference's own
`reparameterize()` methods, tensor by tensor and through both forward passes.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

BN_EPS = 1e-5        # nn.BatchNorm2d default, used by every conv-bn branch of the reference (`mci.py:332-366, 517-550`)


def _fold(kernel: torch.Tensor, sd: Dict[str, torch.Tensor], bn: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv kernel [O, I/g, k, k] + BatchNorm statistics under `bn.` -> (kernel', bias')   (`_fuse_bn_tensor`, mci.py:282-330)"""
    gamma, beta = sd[bn + "weight"], sd[bn + "bias"]
    mean, var = sd[bn + "running_mean"], sd[bn + "running_var"]
    std = (var + BN_EPS).sqrt()
    return kernel * (gamma / std).reshape(-1, 1, 1, 1), beta - mean * gamma / std


def _identity_kernel(channels: int, input_dim: int, k: int, like: torch.Tensor) -> torch.Tensor:
    """`id_tensor` of the skip branch (mci.py:303-320): out channel i reads in channel i % input_dim at the centre tap"""
    idt = torch.zeros((channels, input_dim, k, k), dtype=like.dtype, device=like.device)
    idx = torch.arange(channels)
    idt[idx, idx % input_dim, k // 2, k // 2] = 1
    return idt


def _mobileone(sd: Dict[str, torch.Tensor], p: str, shape_hint: Optional[Tuple[int, int, int]] = None):
    """Fold the MobileOneBlock whose training keys live under `p` (with trailing dot).  Returns (kernel, bias, id_tensor or None).
    shape_hint = (channels, input_dim, k) for a block that has only the skip branch (RepMixer.norm)."""
    convs = sorted({m.group(1) for key in sd if (m := re.match(re.escape(p) + r"rbr_conv\.(\d+)\.conv\.weight$", key))}, key=int)
    kernel, bias = 0, 0
    ref = None
    for i in convs:
        w = sd[f"{p}rbr_conv.{i}.conv.weight"]
        ref = w
        kw, kb = _fold(w, sd, f"{p}rbr_conv.{i}.bn.")
        kernel, bias = kernel + kw, bias + kb
    if f"{p}rbr_scale.conv.weight" in sd:
        ws = sd[f"{p}rbr_scale.conv.weight"]                       # 1x1, padded to the block's kernel size (mci.py:253-259)
        kw, kb = _fold(ws, sd, f"{p}rbr_scale.bn.")
        k = ref.shape[-1] if ref is not None else shape_hint[2]
        pad = k // 2
        kernel, bias = kernel + torch.nn.functional.pad(kw, [pad, pad, pad, pad]), bias + kb
        ref = ref if ref is not None else ws
    idt = None
    if f"{p}rbr_skip.weight" in sd:
        if ref is not None:
            ch, inp, k = ref.shape[0], ref.shape[1], ref.shape[-1]
        else:
            ch, inp, k = shape_hint
        idt = _identity_kernel(ch, inp, k, sd[f"{p}rbr_skip.weight"])
        kw, kb = _fold(idt, sd, f"{p}rbr_skip.")
        kernel, bias = kernel + kw, bias + kb
    if isinstance(kernel, int):
        raise KeyError(f"no MobileOneBlock branches under '{p}'")
    return kernel, bias, idt


def is_training_state_dict(sd: Dict[str, torch.Tensor]) -> bool:
    return any(".rbr_" in k or ".lkb_origin." in k or k.endswith(".pe.weight") for k in sd)


def reparameterize_state_dict(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Training-mode FastViT state dict -> inference-mode state dict (keys relative to the FastViT module, as the input's)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    done = set()

    def emit(prefix: str, name: str, w: torch.Tensor, b: torch.Tensor):
        out[f"{prefix}{name}.weight"] = w.contiguous()
        out[f"{prefix}{name}.bias"] = b.contiguous()

    for key in sd:
        if key in done:
            continue
        # ---- RepMixer: <blk>.token_mixer.{norm, mixer, layer_scale}
        m = re.match(r"(.*\.token_mixer\.)(norm|mixer)\.rbr_", key)
        if m:
            p = m.group(1)
            if f"{p}reparam_conv.weight" not in out:
                mk, mb, idt = _mobileone(sd, p + "mixer.")
                ch, inp, k = mk.shape[0], mk.shape[1], mk.shape[-1]
                nk, nb, _ = _mobileone(sd, p + "norm.", (ch, inp, k))
                if idt is None:
                    idt = _identity_kernel(ch, inp, k, mk)
                if f"{p}layer_scale" in sd:                          # mci.py:830-837
                    ls = sd[f"{p}layer_scale"]
                    w = idt + ls.unsqueeze(-1) * (mk - nk)
                    b = torch.squeeze(ls) * (mb - nb)
                else:                                                 # mci.py:838-844
                    w, b = idt + mk - nk, mb - nb
                emit(p, "reparam_conv", w, b)
            done.update(k2 for k2 in sd if k2.startswith(p) and (".rbr_" in k2[len(p):] or k2 == f"{p}layer_scale"))
            continue
        if re.match(r".*\.token_mixer\.layer_scale$", key) and any(k2.startswith(key[: -len("layer_scale")] + "mixer.rbr_") for k2 in sd):
            continue                                                  # folded above (key order: layer_scale may come first)
        # ---- MobileOneBlock
        m = re.match(r"(.*?)rbr_(conv\.\d+\.|scale\.|skip\.)", key)
        if m:
            p = m.group(1)
            if f"{p}reparam_conv.weight" not in out:
                k_, b_, _ = _mobileone(sd, p)
                emit(p, "reparam_conv", k_, b_)
            done.update(k2 for k2 in sd if k2.startswith(p + "rbr_"))
            continue
        # ---- ReparamLargeKernelConv
        m = re.match(r"(.*?)(lkb_origin|small_conv)\.", key)
        if m:
            p = m.group(1)
            if f"{p}lkb_reparam.weight" not in out:
                k_, b_ = _fold(sd[f"{p}lkb_origin.conv.weight"], sd, f"{p}lkb_origin.bn.")
                if f"{p}small_conv.conv.weight" in sd:                # mci.py:459-466
                    sk, sb = _fold(sd[f"{p}small_conv.conv.weight"], sd, f"{p}small_conv.bn.")
                    pad = (k_.shape[-1] - sk.shape[-1]) // 2
                    k_, b_ = k_ + torch.nn.functional.pad(sk, [pad] * 4), b_ + sb
                emit(p, "lkb_reparam", k_, b_)
            done.update(k2 for k2 in sd if k2.startswith(p + "lkb_origin.") or k2.startswith(p + "small_conv."))
            continue
        # ---- RepCPE
        m = re.match(r"(.*?)pe\.(weight|bias)$", key)
        if m and f"{m.group(1)}pe.weight" in sd and sd[f"{m.group(1)}pe.weight"].dim() == 4:
            p = m.group(1)
            if f"{p}reparam_conv.weight" not in out:
                w = sd[f"{p}pe.weight"]
                emit(p, "reparam_conv", _identity_kernel(w.shape[0], w.shape[1], w.shape[-1], w) + w, sd[f"{p}pe.bias"])
            done.update({f"{p}pe.weight", f"{p}pe.bias"})
            continue
        out[key] = sd[key]
    return out


def load_training_checkpoint(tower, sd: Dict[str, torch.Tensor], strict: bool = True):
    """Re-parameterise `sd` if it comes from the training graph and load it into `tower.vision_tower.model`.

    A checkpoint saved from a bare `FastViT` carries the ImageNet classifier `head.{weight,bias}` (`mci.py:1412-1416`); `MCi` replaces
    that module by `GlobalPool2D` (`mobileclip/__init__.py:81-87`), whose `head.proj` is dead on the encode_images() path: the
    classifier keys are dropped and a missing `head.proj` keeps the tower's current value, so `strict` still vouches for every tensor
    the path uses."""
    sd = dict(sd)
    if is_training_state_dict(sd):
        sd = reparameterize_state_dict(sd)
    model = tower.vision_tower.model
    if "head.proj" not in sd:
        sd.pop("head.weight", None)
        sd.pop("head.bias", None)
        sd["head.proj"] = model.state_dict()["head.proj"]
    return model.load_state_dict(sd, strict=strict)
