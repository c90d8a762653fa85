"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's `encode_images()` path.

This is the parity oracle for the HIP path.  It is a plain functional restatement
(torch CPU, fp32 by default, fp64 on request) of the arithmetic the reference
performs in inference mode; each function cites the reference lines it follows
(paths relative to `/root/reference/llava/model/`).  It holds no modules and no
state: weights come in as a dict keyed by the reference's state-dict names.

Pinning: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md 8c), so the oracle is pinned against outputs of the *reference itself*
run in the build container: `oracle/make_golden.py` imports the reference's
`MobileCLIPVisionTower` unmodified (via `oracle/ref_import.py`) and writes
`tests/golden/*`; `tests/test_oracle_golden.py` checks this file against those
fixtures (and against the live reference when `/root/reference` is present).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product (`ml_fastvlm_amd`) never does.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

LAYERS = (2, 12, 24, 4, 2)                 # multimodal_encoder/mobileclip/mci.py:1455
EMBED_DIMS = (96, 192, 384, 768, 1536)     # mci.py:1456
HEAD_DIM = 32                              # mci.py:636
P = Dict[str, torch.Tensor]


def gelu(x: torch.Tensor) -> torch.Tensor:
    """`nn.GELU()` default = exact erf form (mci.py:108, 387, 870; projector builder.py:28)."""
    return F.gelu(x)


def mobileone(x, w, b, stride, padding, groups, act=True):
    """MobileOneBlock inference branch, SE = Identity: act(conv(x)+b)  (mci.py:194-198)."""
    y = F.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)
    return gelu(y) if act else y


def stem(x, p: P, prefix="patch_embed"):
    """convolutional_stem (mci.py:553-603): 3x3 s2 dense, 3x3 s2 depthwise, 1x1; GELU after each."""
    c = p[f"{prefix}.0.reparam_conv.weight"].shape[0]
    x = mobileone(x, p[f"{prefix}.0.reparam_conv.weight"], p[f"{prefix}.0.reparam_conv.bias"], 2, 1, 1)
    x = mobileone(x, p[f"{prefix}.1.reparam_conv.weight"], p[f"{prefix}.1.reparam_conv.bias"], 2, 1, c)
    x = mobileone(x, p[f"{prefix}.2.reparam_conv.weight"], p[f"{prefix}.2.reparam_conv.bias"], 1, 0, 1)
    return x


def repmixer(x, p: P, prefix):
    """RepMixer inference branch: depthwise 3x3 p1 + bias (mci.py:808-811)."""
    w = p[f"{prefix}.reparam_conv.weight"]
    return F.conv2d(x, w, p[f"{prefix}.reparam_conv.bias"], stride=1, padding=w.shape[-1] // 2, groups=w.shape[0])


def convffn(x, p: P, prefix):
    """ConvFFN.forward (mci.py:920-927): dw7x7 (no bias) -> BatchNorm2d(eval) -> 1x1 -> GELU -> 1x1."""
    w = p[f"{prefix}.conv.conv.weight"]
    y = F.conv2d(x, w, None, stride=1, padding=3, groups=w.shape[0])
    y = F.batch_norm(y, p[f"{prefix}.conv.bn.running_mean"], p[f"{prefix}.conv.bn.running_var"],
                     p[f"{prefix}.conv.bn.weight"], p[f"{prefix}.conv.bn.bias"], training=False, eps=1e-5)
    y = F.conv2d(y, p[f"{prefix}.fc1.weight"], p[f"{prefix}.fc1.bias"])
    y = gelu(y)
    y = F.conv2d(y, p[f"{prefix}.fc2.weight"], p[f"{prefix}.fc2.bias"])
    return y


def repmixer_block(x, p: P, prefix):
    """RepMixerBlock.forward (mci.py:1106-1109)."""
    x = repmixer(x, p, f"{prefix}.token_mixer")
    return x + p[f"{prefix}.layer_scale"] * convffn(x, p, f"{prefix}.convffn")


def layernorm_channel(x, w, b, eps=1e-5):
    """LayerNormChannel.forward (mci.py:617-623): per-pixel LN over C, biased variance."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[None, :, None, None] * x + b[None, :, None, None]


def mhsa(x, p: P, prefix):
    """MHSA.forward (mci.py:661-685): qkv (no bias) -> softmax((q*scale) k^T) v -> proj."""
    B, C, H, W = x.shape
    N = H * W
    nh = C // HEAD_DIM
    t = torch.flatten(x, start_dim=2).transpose(-2, -1)                       # (B, N, C)
    qkv = F.linear(t, p[f"{prefix}.qkv.weight"]).reshape(B, N, 3, nh, HEAD_DIM).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = (q * HEAD_DIM ** -0.5) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    t = (attn @ v).transpose(1, 2).reshape(B, N, C)
    t = F.linear(t, p[f"{prefix}.proj.weight"], p[f"{prefix}.proj.bias"])
    return t.transpose(-2, -1).reshape(B, C, H, W)


def attention_block(x, p: P, prefix):
    """AttentionBlock.forward (mci.py:1185-1188)."""
    n = layernorm_channel(x, p[f"{prefix}.norm.weight"], p[f"{prefix}.norm.bias"])
    x = x + p[f"{prefix}.layer_scale_1"] * mhsa(n, p, f"{prefix}.token_mixer")
    return x + p[f"{prefix}.layer_scale_2"] * convffn(x, p, f"{prefix}.convffn")


def patch_embed(x, p: P, prefix):
    """PatchEmbed.forward (mci.py:739-741): ReparamLargeKernelConv (dw7x7 s2 p3, groups=Cin,
    +b, GELU; mci.py:442-451) then MobileOneBlock 1x1 + b, GELU."""
    w = p[f"{prefix}.proj.0.lkb_reparam.weight"]
    cin = x.shape[1]
    y = gelu(F.conv2d(x, w, p[f"{prefix}.proj.0.lkb_reparam.bias"], stride=2, padding=3, groups=cin))
    return mobileone(y, p[f"{prefix}.proj.1.reparam_conv.weight"], p[f"{prefix}.proj.1.reparam_conv.bias"], 1, 0, 1)


def repcpe(x, p: P, prefix):
    """RepCPE inference branch: depthwise 7x7 p3 + bias (mci.py:992-995)."""
    w = p[f"{prefix}.reparam_conv.weight"]
    return F.conv2d(x, w, p[f"{prefix}.reparam_conv.bias"], stride=1, padding=w.shape[-1] // 2, groups=w.shape[0])


def conv_exp(x, p: P, prefix="conv_exp"):
    """FastViT.conv_exp (mci.py:1401-1411): MobileOneBlock dw3x3 (groups=Cin, Cout=2Cin) + b
    -> SEBlock (mci.py:72-81) -> GELU."""
    w = p[f"{prefix}.reparam_conv.weight"]
    y = F.conv2d(x, w, p[f"{prefix}.reparam_conv.bias"], stride=1, padding=1, groups=x.shape[1])
    b, c, h, wd = y.shape
    s = F.avg_pool2d(y, kernel_size=[h, wd])
    s = F.relu(F.conv2d(s, p[f"{prefix}.se.reduce.weight"], p[f"{prefix}.se.reduce.bias"]))
    s = torch.sigmoid(F.conv2d(s, p[f"{prefix}.se.expand.weight"], p[f"{prefix}.se.expand.bias"]))
    return gelu(y * s.view(-1, c, 1, 1))


def forward_features(x, p: P, taps: Optional[List[torch.Tensor]] = None):
    """FastViT.forward up to `image_embeddings` (mci.py:1427-1451); the GlobalPool2D logits
    (mci.py:1290-1302) are dead on this path (dropped by feature_select) and not computed.
    `taps`, when given, receives the NCHW output of the stem, each network entry and conv_exp."""
    x = stem(x, p)
    if taps is not None:
        taps.append(x)
    idx = 0
    n = len(LAYERS)
    for i in range(n):                                                        # constructor loop mci.py:1361-1399
        if i >= 3:                                                            # pos_embs: RepCPE before stages 3,4 (mci.py:1459)
            x = repcpe(x, p, f"network.{idx}")
            idx += 1
            if taps is not None:
                taps.append(x)
        for b in range(LAYERS[i]):
            if i < 3:
                x = repmixer_block(x, p, f"network.{idx}.{b}")
            else:
                x = attention_block(x, p, f"network.{idx}.{b}")
        idx += 1
        if taps is not None:
            taps.append(x)
        if i >= n - 1:
            break
        x = patch_embed(x, p, f"network.{idx}")
        idx += 1
        if taps is not None:
            taps.append(x)
    x = conv_exp(x, p)
    if taps is not None:
        taps.append(x)
    return x


def tower_forward(images: torch.Tensor, p: P, dtype=torch.float32, taps=None) -> torch.Tensor:
    """MobileCLIPVisionTower.forward_images + feature_select
    (multimodal_encoder/mobileclip_encoder.py:60-68, 77-88): [B,3,R,R] -> [B, (R/64)^2, 3072]."""
    pp = {k: v.to(dtype) for k, v in p.items() if v.is_floating_point()}
    with torch.no_grad():
        f = forward_features(images.to(dtype), pp, taps)
    B, C, H, W = f.shape
    return f.reshape(B, C, H * W).transpose(1, 2).to(images.dtype)


def projector(tokens: torch.Tensor, pj: P, dtype=torch.float32) -> torch.Tensor:
    """mlp2x_gelu (multimodal_projector/builder.py:23-30): Linear -> GELU -> Linear."""
    with torch.no_grad():
        h = F.linear(tokens.to(dtype), pj["0.weight"].to(dtype), pj["0.bias"].to(dtype))
        h = gelu(h)
        return F.linear(h, pj["2.weight"].to(dtype), pj["2.bias"].to(dtype)).to(tokens.dtype)


def encode_images(images, p: P, pj: P, dtype=torch.float32):
    """LlavaMetaForCausalLM.encode_images (llava_arch.py:141-144): tower -> projector."""
    return projector(tower_forward(images, p, dtype), pj, dtype)
