"""Static description of the FastViTHD encoder as the reference instantiates it.

Everything here is derived from the reference's hyper-parameters
(`llava/model/multimodal_encoder/mobileclip/mci.py:1454-1478`, `fastvithd()`:
layers [2,12,24,4,2], dims [96,192,384,768,1536], mlp_ratio 4, RepCPE(7x7) in
front of stages 3 and 4, RepMixer for stages 0-2, attention for 3-4,
inference_mode=True, cls_ratio 2.0) and from the module constructors it calls.
The key names are the reference's *inference-mode* state-dict keys
(`reparam_conv`, `lkb_reparam`, never `rbr_*`), so a FastVLM checkpoint loads
into our tower unchanged (SURVEY.md 8b "State-dict").

`network_plan()` is the flat list of "network entries" the reference iterates
over in `FastViT.forward_tokens` (`mci.py:1431-1434`).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

LAYERS = (2, 12, 24, 4, 2)                 # mci.py:1455
EMBED_DIMS = (96, 192, 384, 768, 1536)     # mci.py:1456
MLP_RATIO = 4                              # mci.py:1457
TOKEN_MIXERS = ("repmixer", "repmixer", "repmixer", "attention", "attention")  # mci.py:1460
HAS_CPE = (False, False, False, True, True)                                      # mci.py:1459
HEAD_DIM = 32                              # mci.py:636 (MHSA default head_dim)
CLS_RATIO = 2                              # mci.py:1328 (cls_ratio=2.0)
OUT_DIM = EMBED_DIMS[-1] * CLS_RATIO       # 3072 == mobileclip_l.json image_cfg.embed_dim
SE_RD = OUT_DIM // 16                      # mci.py:49 rd_ratio 0.0625 -> 192
PROJECTION_DIM = 768                       # mobileclip_l.json "embed_dim" (head.proj, dead on this path)
PATCH_SIZE = 64                            # mobileclip_l.json image_cfg.patch_size (total downsample)
BN_EPS = 1e-5                              # nn.BatchNorm2d default, mci.py:901
LN_EPS = 1e-5                              # mci.py:611


@dataclass(frozen=True)
class Entry:
    """One element of `FastViT.network` (or stem / head)."""
    kind: str          # 'stage' | 'patch_embed' | 'cpe'
    index: int         # position in self.network
    dim: int           # channels in
    dim_out: int       # channels out
    depth: int = 0     # number of blocks (stage only)
    mixer: str = ""    # 'repmixer' | 'attention' (stage only)


def network_plan() -> List[Entry]:
    """Mirror of the constructor loop `mci.py:1361-1399`."""
    plan: List[Entry] = []
    idx = 0
    n = len(LAYERS)
    for i in range(n):
        if HAS_CPE[i]:
            plan.append(Entry("cpe", idx, EMBED_DIMS[i], EMBED_DIMS[i]))
            idx += 1
        plan.append(Entry("stage", idx, EMBED_DIMS[i], EMBED_DIMS[i], LAYERS[i], TOKEN_MIXERS[i]))
        idx += 1
        if i >= n - 1:
            break
        plan.append(Entry("patch_embed", idx, EMBED_DIMS[i], EMBED_DIMS[i + 1]))
        idx += 1
    return plan


def _convffn(prefix: str, c: int, out: "OrderedDict[str, Tuple[Tuple[int, ...], str]]") -> None:
    h = c * MLP_RATIO
    out[f"{prefix}.convffn.conv.conv.weight"] = ((c, 1, 7, 7), "param")        # mci.py:885-894
    out[f"{prefix}.convffn.conv.bn.weight"] = ((c,), "param")                   # mci.py:901-907
    out[f"{prefix}.convffn.conv.bn.bias"] = ((c,), "param")
    out[f"{prefix}.convffn.conv.bn.running_mean"] = ((c,), "buffer")
    out[f"{prefix}.convffn.conv.bn.running_var"] = ((c,), "buffer")
    out[f"{prefix}.convffn.conv.bn.num_batches_tracked"] = ((), "buffer_i64")
    out[f"{prefix}.convffn.fc1.weight"] = ((h, c, 1, 1), "param")               # mci.py:908
    out[f"{prefix}.convffn.fc1.bias"] = ((h,), "param")
    out[f"{prefix}.convffn.fc2.weight"] = ((c, h, 1, 1), "param")               # mci.py:910
    out[f"{prefix}.convffn.fc2.bias"] = ((c,), "param")


def param_spec() -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """key -> (shape, kind) in the reference's registration order.

    kind is 'param', 'buffer' (float) or 'buffer_i64' (BatchNorm's counter).
    Keys are relative to the FastViT module (`...vision_tower.model.<key>`).
    """
    out: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    c0 = EMBED_DIMS[0]
    # convolutional_stem, mci.py:553-603 (three MobileOneBlocks, inference mode -> reparam_conv)
    out["patch_embed.0.reparam_conv.weight"] = ((c0, 3, 3, 3), "param")
    out["patch_embed.0.reparam_conv.bias"] = ((c0,), "param")
    out["patch_embed.1.reparam_conv.weight"] = ((c0, 1, 3, 3), "param")
    out["patch_embed.1.reparam_conv.bias"] = ((c0,), "param")
    out["patch_embed.2.reparam_conv.weight"] = ((c0, c0, 1, 1), "param")
    out["patch_embed.2.reparam_conv.bias"] = ((c0,), "param")
    for e in network_plan():
        p = f"network.{e.index}"
        if e.kind == "cpe":                                                     # RepCPE mci.py:975-984
            out[f"{p}.reparam_conv.weight"] = ((e.dim, 1, 7, 7), "param")
            out[f"{p}.reparam_conv.bias"] = ((e.dim,), "param")
        elif e.kind == "patch_embed":                                           # PatchEmbed mci.py:709-737
            out[f"{p}.proj.0.lkb_reparam.weight"] = ((e.dim_out, 1, 7, 7), "param")
            out[f"{p}.proj.0.lkb_reparam.bias"] = ((e.dim_out,), "param")
            out[f"{p}.proj.1.reparam_conv.weight"] = ((e.dim_out, e.dim_out, 1, 1), "param")
            out[f"{p}.proj.1.reparam_conv.bias"] = ((e.dim_out,), "param")
        else:
            c = e.dim
            for b in range(e.depth):
                q = f"{p}.{b}"
                if e.mixer == "repmixer":                                       # RepMixerBlock mci.py:1042-1109
                    out[f"{q}.layer_scale"] = ((c, 1, 1), "param")
                    out[f"{q}.token_mixer.reparam_conv.weight"] = ((c, 1, 3, 3), "param")
                    out[f"{q}.token_mixer.reparam_conv.bias"] = ((c,), "param")
                    _convffn(q, c, out)
                else:                                                           # AttentionBlock mci.py:1116-1188
                    out[f"{q}.layer_scale_1"] = ((c, 1, 1), "param")
                    out[f"{q}.layer_scale_2"] = ((c, 1, 1), "param")
                    out[f"{q}.norm.weight"] = ((c,), "param")
                    out[f"{q}.norm.bias"] = ((c,), "param")
                    out[f"{q}.token_mixer.qkv.weight"] = ((3 * c, c), "param")  # bias=False, mci.py:656
                    out[f"{q}.token_mixer.proj.weight"] = ((c, c), "param")
                    out[f"{q}.token_mixer.proj.bias"] = ((c,), "param")
                    _convffn(q, c, out)
    cl = EMBED_DIMS[-1]
    # conv_exp = MobileOneBlock(dw3x3, groups=1536, 1536->3072, use_se=True), mci.py:1401-1411
    # (MobileOneBlock assigns `self.se` before `self.reparam_conv`, mci.py:136-157, hence the order)
    out["conv_exp.se.reduce.weight"] = ((SE_RD, OUT_DIM, 1, 1), "param")        # SEBlock mci.py:55-68
    out["conv_exp.se.reduce.bias"] = ((SE_RD,), "param")
    out["conv_exp.se.expand.weight"] = ((OUT_DIM, SE_RD, 1, 1), "param")
    out["conv_exp.se.expand.bias"] = ((OUT_DIM,), "param")
    out["conv_exp.reparam_conv.weight"] = ((OUT_DIM, 1, 3, 3), "param")
    out["conv_exp.reparam_conv.bias"] = ((OUT_DIM,), "param")
    # GlobalPool2D head installed by MCi (mobileclip/__init__.py:48-53); dead on this path but
    # kept so checkpoints load with strict=True.
    out["head.proj"] = ((OUT_DIM, PROJECTION_DIM), "param")
    assert cl * CLS_RATIO == OUT_DIM
    return out


def tokens_per_side(image_size: int) -> int:
    return image_size // PATCH_SIZE
