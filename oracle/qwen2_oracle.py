"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch, fp32 / fp64) of the Qwen2 prefill the reference delegates to the
third-party `transformers` package: `LlavaQwen2ForCausalLM.forward` -> `Qwen2ForCausalLM.forward(inputs_embeds=...)`
(`llava/model/language_model/llava_qwen.py:92-103`).  `transformers` is NOT under /root/reference; the reference pins 4.48.3
(`pyproject.toml:17`), this image has 5.15.0.  The published algorithm restated here (transformers/models/qwen2/modeling_qwen2.py,
same in both versions): `Qwen2RMSNorm.forward`, `Qwen2RotaryEmbedding.forward` + `rotate_half` / `apply_rotary_pos_emb`,
`repeat_kv` + `eager_attention_forward` with the causal + padding mask, `Qwen2MLP.forward`, `Qwen2DecoderLayer.forward`,
`Qwen2Model.forward`, `lm_head`.

PINNING: tests/test_qwen2_prefill.py checks every function here against the installed `transformers` modules themselves (random
tiny and 0.5B-shaped configs, left / right padding, explicit position ids) at <= 1e-5 - the library is importable both in the build
container and on the GPU box, so the pin runs in both places.  Only tests/ import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch


def rmsnorm(x, w, eps=1e-6):                       # Qwen2RMSNorm.forward
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_cos_sin(position_ids, head_dim, theta):   # Qwen2RotaryEmbedding.forward (default rope, attention_scaling 1)
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat((freqs, freqs), -1)
    return emb.cos(), emb.sin()                    # [B, T, head_dim]


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), -1)


def apply_rope(q, k, cos, sin):                    # apply_rotary_pos_emb, q / k [B, heads, T, head_dim]
    cos, sin = cos[:, None].to(q.dtype), sin[:, None].to(q.dtype)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def attention(q, k, v, key_valid=None):
    """q [B, nh, T, hd], k / v [B, nkv, T, hd] (rope applied) -> [B, T, nh * hd]; causal + key-padding mask (the 4-D mask
    `Qwen2Model.forward` builds from attention_mask for a prefill: key k visible to query t iff k <= t and attention_mask[b, k])."""
    B, nh, T, hd = q.shape
    rep = nh // k.shape[1]
    k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)             # repeat_kv
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    allow = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    if key_valid is not None:
        allow = allow & key_valid.bool()[:, None, None, :]
    s = s.masked_fill(~allow, torch.finfo(s.dtype).min)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B, T, nh * hd)


def prefill(inputs_embeds, sd: Dict[str, torch.Tensor], cfg, attention_mask=None, position_ids=None, dtype=torch.float32,
            n_layers: Optional[int] = None):
    """-> (logits of every position [B, T, V], hidden states after the last decoder layer [B, T, H], per-layer (k, v) after rope).
    sd: the model's state dict (keys "model.layers.<l>....", "model.norm.weight", "lm_head.weight" or the tied embedding);
    cfg: hidden_size, num_attention_heads, num_key_value_heads, num_hidden_layers, rms_norm_eps, rope theta (rope_theta)."""
    g = lambda k: sd[k].to(dtype)
    H, nh, nkv = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads
    hd = getattr(cfg, "head_dim", None) or H // nh
    theta = getattr(cfg, "rope_theta", None)
    if theta is None:
        theta = (getattr(cfg, "rope_parameters", None) or getattr(cfg, "rope_scaling", None) or {}).get("rope_theta", 10000.0)
    eps = cfg.rms_norm_eps
    x = inputs_embeds.to(dtype)
    B, T, _ = x.shape
    if position_ids is None:
        position_ids = torch.arange(T)[None].expand(B, T)
    cos, sin = rope_cos_sin(position_ids, hd, theta)
    kvs = []
    L = cfg.num_hidden_layers if n_layers is None else n_layers
    for l in range(L):
        p = f"model.layers.{l}."
        h = rmsnorm(x, g(p + "input_layernorm.weight"), eps)                      # Qwen2DecoderLayer.forward
        q = (h @ g(p + "self_attn.q_proj.weight").t() + g(p + "self_attn.q_proj.bias")).view(B, T, nh, hd).transpose(1, 2)
        k = (h @ g(p + "self_attn.k_proj.weight").t() + g(p + "self_attn.k_proj.bias")).view(B, T, nkv, hd).transpose(1, 2)
        v = (h @ g(p + "self_attn.v_proj.weight").t() + g(p + "self_attn.v_proj.bias")).view(B, T, nkv, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        kvs.append((k, v))
        a = attention(q, k, v, attention_mask)
        x = x + a @ g(p + "self_attn.o_proj.weight").t()
        h = rmsnorm(x, g(p + "post_attention_layernorm.weight"), eps)
        gate, up = h @ g(p + "mlp.gate_proj.weight").t(), h @ g(p + "mlp.up_proj.weight").t()
        x = x + (torch.nn.functional.silu(gate) * up) @ g(p + "mlp.down_proj.weight").t()   # Qwen2MLP.forward
    hidden = x
    head = sd["lm_head.weight"] if "lm_head.weight" in sd else sd["model.embed_tokens.weight"]
    logits = rmsnorm(x, g("model.norm.weight"), eps) @ head.to(dtype).t()
    return logits, hidden, kvs
