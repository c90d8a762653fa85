"""TEST INFRASTRUCTURE - the oracle's arithmetic with bf16 rounding inserted exactly where the HIP path
stores a tensor (and nowhere else).

Why: the HIP path keeps activations in bf16 in HBM between kernels (fp32 accumulation and epilogues
inside a kernel, one rounding on store; GEMM weights rounded to bf16, depthwise taps / biases / layer
scales / norms in fp32).  Against the fp32 reference that storage noise accumulates over 44 blocks
to rel-L2 ~3e-2 with the stress-test synthetic weights (the reference's own bf16 run: 3.2-4.4e-2,
tests/golden/*.npz `ref_bf16_rel_l2`).  A bound that loose cannot see a small kernel bug, so the GPU
tests ALSO compare against this emulation, where only accumulation order, the A&S erf (1.5e-7) and
the approximate rcp/exp2 differ: that comparison is tight.

Rounding points mirror `fvhd_api.hip: encode_impl` one to one.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import fastvithd_oracle as O

LAYERS, DIMS = O.LAYERS, O.EMBED_DIMS


def rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _ffn(t, p, pre, ls):
    """dw7+BN (fp32 taps, folded) -> bf16 | fc1+GELU -> bf16 (registers or HBM) | fc2+ls+resid -> bf16."""
    w = p[f"{pre}.conv.conv.weight"]
    s = p[f"{pre}.conv.bn.weight"] / torch.sqrt(p[f"{pre}.conv.bn.running_var"] + 1e-5)
    b = p[f"{pre}.conv.bn.bias"] - p[f"{pre}.conv.bn.running_mean"] * s
    a = rb(F.conv2d(t, w * s[:, None, None, None], b, padding=3, groups=w.shape[0]))
    h = rb(O.gelu(F.conv2d(a, rb(p[f"{pre}.fc1.weight"]), p[f"{pre}.fc1.bias"])))
    y = F.conv2d(h, rb(p[f"{pre}.fc2.weight"]), p[f"{pre}.fc2.bias"])
    return rb(t + ls * y)


def _attention(n, p, pre):
    B, C, H, W = n.shape
    N, nh = H * W, C // 32
    t = n.flatten(2).transpose(1, 2)
    qkv = rb(F.linear(t, rb(p[f"{pre}.qkv.weight"]))).reshape(B, N, 3, nh, 32).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    s = (q @ k.transpose(-2, -1)) * 32 ** -0.5
    e = torch.exp(s - s.amax(-1, keepdim=True))
    o = (rb(e) @ v) / e.sum(-1, keepdim=True)            # P rounded to bf16 for the MFMA, row sum from fp32
    return rb(o.transpose(1, 2).reshape(B, N, C))         # attention output stored bf16


def step_fns(p):
    """[(name, fn)] in the library's step order (include/fvhd.h "step-level execution"): fn maps the NCHW fp32
    (bf16-representable) activation entering the step to the one leaving it, rounded where the HIP path stores.
    Step 0 takes the image batch; the last step returns [B, T, 3072] tokens (fp32 values, not yet rounded)."""
    p = {k: v.float() for k, v in p.items() if v.is_floating_point()}
    steps = []

    def stem(x):
        # stem[0] is an MFMA GEMM in the HIP path: image and taps rounded to bf16
        x = rb(O.gelu(F.conv2d(rb(x), rb(p["patch_embed.0.reparam_conv.weight"]), p["patch_embed.0.reparam_conv.bias"], stride=2, padding=1)))
        x = rb(O.gelu(F.conv2d(x, p["patch_embed.1.reparam_conv.weight"], p["patch_embed.1.reparam_conv.bias"], stride=2, padding=1, groups=96)))
        return rb(O.gelu(F.conv2d(x, rb(p["patch_embed.2.reparam_conv.weight"]), p["patch_embed.2.reparam_conv.bias"])))

    def cpe(pre, C):
        return lambda x: rb(F.conv2d(x, p[f"{pre}.reparam_conv.weight"], p[f"{pre}.reparam_conv.bias"], padding=3, groups=C))

    def rep(pre, C):
        def f(x):
            t = rb(F.conv2d(x, p[f"{pre}.token_mixer.reparam_conv.weight"], p[f"{pre}.token_mixer.reparam_conv.bias"],
                            padding=1, groups=C))
            return _ffn(t, p, f"{pre}.convffn", p[f"{pre}.layer_scale"])
        return f

    def att(pre, C):
        def f(x):
            n = rb(O.layernorm_channel(x, p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"]))
            o = _attention(n, p, f"{pre}.token_mixer")
            B_, _, H_, W_ = x.shape
            y = F.linear(o, rb(p[f"{pre}.token_mixer.proj.weight"]), p[f"{pre}.token_mixer.proj.bias"])
            y = y.transpose(1, 2).reshape(B_, C, H_, W_)
            x = rb(x + p[f"{pre}.layer_scale_1"] * y)
            return _ffn(x, p, f"{pre}.convffn", p[f"{pre}.layer_scale_2"])
        return f

    def down(pre, C):
        def f(x):
            y = rb(O.gelu(F.conv2d(x, p[f"{pre}.proj.0.lkb_reparam.weight"], p[f"{pre}.proj.0.lkb_reparam.bias"],
                                   stride=2, padding=3, groups=C)))
            return rb(O.gelu(F.conv2d(y, rb(p[f"{pre}.proj.1.reparam_conv.weight"]), p[f"{pre}.proj.1.reparam_conv.bias"])))
        return f

    def head(x):
        y = rb(F.conv2d(x, p["conv_exp.reparam_conv.weight"], p["conv_exp.reparam_conv.bias"], padding=1, groups=DIMS[4]))
        s = y.mean((2, 3), keepdim=True)
        s = F.relu(F.conv2d(s, p["conv_exp.se.reduce.weight"], p["conv_exp.se.reduce.bias"]))
        s = torch.sigmoid(F.conv2d(s, p["conv_exp.se.expand.weight"], p["conv_exp.se.expand.bias"]))
        f = O.gelu(y * s)
        B_, C_, H_, W_ = f.shape
        return f.reshape(B_, C_, H_ * W_).transpose(1, 2)

    steps.append(("stem", stem))
    idx = 0
    for i in range(5):
        C = DIMS[i]
        if i >= 3:
            steps.append((f"network.{idx} (RepCPE)", cpe(f"network.{idx}", C)))
            idx += 1
        for b in range(LAYERS[i]):
            pre = f"network.{idx}.{b}"
            steps.append((pre, rep(pre, C) if i < 3 else att(pre, C)))
        idx += 1
        if i == 4:
            break
        steps.append((f"network.{idx} (PatchEmbed)", down(f"network.{idx}", C)))
        idx += 1
    steps.append(("conv_exp", head))
    return steps


def emulate_tower(images, p, img_dtype=torch.bfloat16, out_dtype=torch.bfloat16):
    with torch.no_grad():
        x = images.to(img_dtype).float()
        for _, fn in step_fns(p):
            x = fn(x)
        return x.to(out_dtype)


def emulate_projector(tokens_bf16, pj, out_dtype=torch.bfloat16):
    with torch.no_grad():
        t = tokens_bf16.float()
        h = rb(O.gelu(F.linear(t, rb(pj["0.weight"].float()), pj["0.bias"].float())))
        return F.linear(h, rb(pj["2.weight"].float()), pj["2.bias"].float()).to(out_dtype)
