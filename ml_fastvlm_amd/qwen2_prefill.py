"""Qwen2 prefill on hand-written gfx950 kernels - the LLM half of FastVLM's time to first token.

The reference's prefill is `LlavaQwen2ForCausalLM.forward(inputs_embeds=...)` -> `transformers` `Qwen2ForCausalLM.forward`
(`llava/model/language_model/llava_qwen.py:92-103`; `generate`, `:138-143`, enters it for the first token).  `Qwen2Prefill` stands
where that module stands for the prefill step: it is built FROM the module (`from_hf`: same config, weights under the same
state-dict keys), takes the `inputs_embeds` / `attention_mask` / `position_ids` that `prepare_inputs_labels_for_multimodal` (or
`ml_fastvlm_amd.splice.multimodal_splice`) returns, and gives the logits of the last position - what `generate` samples the first
token from - plus, on request, the KV cache in `transformers`' per-layer layout so that the stock decode loop can continue.

    pre = Qwen2Prefill.from_hf(model)                     # model: Qwen2ForCausalLM / LlavaQwen2ForCausalLM on a HIP device
    logits = pre(inputs_embeds, attention_mask, position_ids)           # [B, vocab] fp32

Per decoder layer: RMSNorm -> ONE GEMM for q|k|v (+bias) -> rotary embedding in place -> causal grouped-query flash attention ->
o_proj GEMM with the residual in its epilogue -> RMSNorm -> ONE GEMM for gate|up with silu(gate)*up in its epilogue -> down_proj
GEMM with the residual in its epilogue: 8 launches, all through `libfvhd.so` (`fvhd_llm_*`, include/fvhd.h).  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


class Qwen2Prefill:
    def __init__(self, device_index: int, hidden: int, n_layers: int, n_heads: int, n_kv_heads: int, head_dim: int, intermediate: int,
                 vocab: int, rms_eps: float = 1e-6, rope_theta: float = 1e6):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.fvhd_llm_create(C.byref(h), device_index, hidden, n_layers, n_heads, n_kv_heads, head_dim, intermediate, vocab,
                                       float(rms_eps), float(rope_theta)), "fvhd_llm_create")
        self._h = h
        self.device = torch.device("cuda", device_index)
        self.hidden, self.n_layers, self.n_heads, self.n_kv_heads, self.head_dim = hidden, n_layers, n_heads, n_kv_heads, head_dim
        self.intermediate, self.vocab = intermediate, vocab

    # ---- construction from the reference's module -------------------------------------------------------------------------------
    @classmethod
    def from_hf(cls, model, device: Optional[torch.device] = None) -> "Qwen2Prefill":
        """model: a `transformers` Qwen2ForCausalLM (or the reference's LlavaQwen2ForCausalLM, whose decoder stack it is)."""
        cfg = model.config
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Qwen2Prefill (MI355X): the model must be on a HIP device - this path has no CPU implementation")
        scaling = getattr(cfg, "rope_scaling", None) or getattr(cfg, "rope_parameters", None)
        rope_type = (scaling or {}).get("rope_type", (scaling or {}).get("type", "default")) if isinstance(scaling, dict) else "default"
        if rope_type not in ("default", None):
            raise NotImplementedError(f"rope type {rope_type!r}: only the default rotary embedding of Qwen2 is implemented")
        if getattr(cfg, "use_sliding_window", False):
            raise NotImplementedError("sliding-window attention is off in every FastVLM checkpoint and is not implemented")
        theta = getattr(cfg, "rope_theta", None)
        if theta is None and isinstance(scaling, dict):
            theta = scaling.get("rope_theta")
        head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        self = cls(dev.index if dev.index is not None else torch.cuda.current_device(), cfg.hidden_size, cfg.num_hidden_layers,
                   cfg.num_attention_heads, cfg.num_key_value_heads, head_dim, cfg.intermediate_size, cfg.vocab_size,
                   getattr(cfg, "rms_norm_eps", 1e-6), float(theta if theta is not None else 1e6))
        mpe = getattr(cfg, "max_position_embeddings", None)
        if mpe:                                   # rows of the rotary table; positions beyond it are computed in the kernel, never clamped
            _lib.check(_lib.load().fvhd_llm_set_max_positions(self._h, int(mpe)), "fvhd_llm_set_max_positions")
        self.load_state_dict(model.state_dict())
        return self

    def load_state_dict(self, sd) -> None:
        """The decoder-stack tensors of a (Llava)Qwen2ForCausalLM state dict; `lm_head.weight` falls back to the embedding table
        (tie_word_embeddings).  Everything else in the dict (vision tower, projector) is ignored."""
        lib = _lib.load()
        seen_head = False
        for key, t in sd.items():
            k = key[6:] if key.startswith("model.") else key
            if not (k.startswith("layers.") or k == "norm.weight" or key == "lm_head.weight"):
                continue
            if k.startswith("layers.") and not any(k.endswith(s) for s in self._LAYER_SUFFIXES):
                continue
            seen_head |= key == "lm_head.weight"
            self._set(lib, key, t)
        if not seen_head:
            emb = sd.get("model.embed_tokens.weight", sd.get("embed_tokens.weight"))
            if emb is None:
                raise KeyError("neither lm_head.weight nor model.embed_tokens.weight in the state dict")
            self._set(lib, "lm_head.weight", emb)
        _lib.check(lib.fvhd_llm_finalize(self._h), "fvhd_llm_finalize")

    _LAYER_SUFFIXES = ("input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight", "self_attn.q_proj.bias",
                       "self_attn.k_proj.weight", "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                       "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")

    def _set(self, lib, key: str, t: torch.Tensor) -> None:
        """A tensor that already lives on this context's device is packed by ONE device-to-device copy (`fvhd_llm_set_tensor_device`:
        matrices as bf16, vectors as fp32 - converted on the device when the module holds another dtype); only host tensors take the
        host path.  (Round 3 moved every tensor device -> CPU -> temporary -> device: 15 GB through the host for the 7B model.)  Note the
        footprint: the packed copy (q|k|v concatenated, gate / up interleaved, bf16) lives NEXT to the module's own weights, which the
        stock decode loop keeps using - 2x the LLM's weight bytes on the device (0.99 GB / 15.2 GB more for Qwen2-0.5B / 7B)."""
        t = t.detach()
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        if t.device.type == "cuda" and t.device == self.device:
            want = torch.bfloat16 if t.dim() == 2 else torch.float32
            d = t.to(want).contiguous()
            with torch.cuda.device(self.device):
                _lib.check(lib.fvhd_llm_set_tensor_device(self._h, key.encode(), C.c_void_p(d.data_ptr()), _lib.dtype_code(want), shape, t.dim(),
                                                          _lib.stream_ptr(self.device)), f"fvhd_llm_set_tensor_device({key})")
            if d is not t and d.data_ptr() != t.data_ptr():
                d.record_stream(torch.cuda.current_stream(self.device))      # the temporary outlives the enqueued copy
            return
        if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            t = t.float()
        t = t.to("cpu").contiguous()
        _lib.check(lib.fvhd_llm_set_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), _lib.dtype_code(t.dtype), shape, t.dim()),
                   f"fvhd_llm_set_tensor({key})")

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().fvhd_llm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the prefill ---------------------------------------------------------------------------------------------------------------
    def reserve(self, batch: int, seq_len: int) -> None:
        """size the workspace now (growth synchronises the device and is refused during stream capture)"""
        _lib.check(_lib.load().fvhd_llm_reserve(self._h, int(batch), int(seq_len)), "fvhd_llm_reserve")

    def _check(self, inputs_embeds, attention_mask, position_ids):
        if not isinstance(inputs_embeds, torch.Tensor) or inputs_embeds.dim() != 3 or inputs_embeds.shape[2] != self.hidden \
                or inputs_embeds.shape[0] < 1 or inputs_embeds.shape[1] < 1:
            raise ValueError(f"expected inputs_embeds of shape [B, T, {self.hidden}], got "
                             f"{tuple(inputs_embeds.shape) if isinstance(inputs_embeds, torch.Tensor) else type(inputs_embeds)}")
        x = inputs_embeds.to(self.device).contiguous()
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x = x.float()
        B, T = x.shape[:2]
        am = None
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (B, T):
                raise ValueError(f"attention_mask must be [B, T] = {(B, T)}, got {tuple(attention_mask.shape)}")
            am = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
        pos = None
        if position_ids is not None:
            if tuple(position_ids.shape) != (B, T):
                raise ValueError(f"position_ids must be [B, T] = {(B, T)}, got {tuple(position_ids.shape)}")
            pos = position_ids.to(device=self.device, dtype=torch.int64).contiguous()
        return x, am, pos

    @torch.no_grad()
    def __call__(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                 return_kv: bool = False, out: Optional[torch.Tensor] = None):
        """-> logits [B, vocab] fp32 of the last position; with return_kv also (k, v): bf16 [n_layers, B, n_kv_heads, T, head_dim]."""
        x, am, pos = self._check(inputs_embeds, attention_mask, position_ids)
        B, T = x.shape[:2]
        if out is not None and (not isinstance(out, torch.Tensor) or tuple(out.shape) != (B, self.vocab) or out.dtype != torch.float32
                                or out.device != self.device or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous fp32 tensor [{B}, {self.vocab}] on {self.device} (its raw pointer goes to the kernel)")
        logits = out if out is not None else torch.empty((B, self.vocab), device=self.device, dtype=torch.float32)
        kc = vc = None
        if return_kv:
            kc = torch.empty((self.n_layers, B, self.n_kv_heads, T, self.head_dim), device=self.device, dtype=torch.bfloat16)
            vc = torch.empty_like(kc)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().fvhd_llm_prefill(self._h, _lib.ptr(x), _lib.dtype_code(x.dtype), _lib.ptr(am), _lib.ptr(pos), B, T,
                                                    _lib.ptr(logits), _lib.ptr(kc), _lib.ptr(vc), _lib.stream_ptr(self.device)), "fvhd_llm_prefill")
        return (logits, kc, vc) if return_kv else logits

    @property
    def workspace_generation(self) -> int:
        """increments whenever the library replaced its workspace (a graph the caller captured before stays valid - the old workspace is
        kept alive - but only a re-capture uses the new one)"""
        return int(_lib.load().fvhd_llm_workspace_generation(self._h))

    def hidden_states(self, rows: int) -> torch.Tensor:
        """tests: the residual stream after the last decoder layer of the previous prefill, [rows, hidden] bf16"""
        out = torch.empty((rows, self.hidden), device=self.device, dtype=torch.bfloat16)
        _lib.check(_lib.load().fvhd_llm_debug_hidden(self._h, _lib.ptr(out), rows, _lib.stream_ptr(self.device)), "fvhd_llm_debug_hidden")
        return out


def kv_to_dynamic_cache(k: torch.Tensor, v: torch.Tensor):
    """(k, v) of `Qwen2Prefill(..., return_kv=True)` -> a `transformers.DynamicCache` the stock decode loop continues from."""
    from transformers import DynamicCache
    cache = DynamicCache()
    for layer in range(k.shape[0]):
        cache.update(k[layer], v[layer], layer)
    return cache


def rope_table(positions: int, head_dim: int, theta: float = 1e6, device="cpu") -> torch.Tensor:
    """(cos, sin) table [positions, head_dim / 2, 2] fp32 exactly as the library builds it (and as Qwen2RotaryEmbedding.forward does:
    inv_freq = theta^(-2i / head_dim) in fp32, angle = position * inv_freq in fp32) - for the single-op tests."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = torch.arange(positions, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.stack([ang.cos(), ang.sin()], -1).contiguous().to(device)
