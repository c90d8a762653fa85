"""Factories mirroring the reference's registration points, plus the one-line hook that swaps
the MI355X tower into an unmodified checkout of apple/ml-fastvlm.

* `build_vision_tower`      <-> `llava/model/multimodal_encoder/builder.py:6-19`
* `build_vision_projector`  <-> `llava/model/multimodal_projector/builder.py:17-35`
* `encode_images`           <-> `LlavaMetaForCausalLM.encode_images`, `llava/model/llava_arch.py:141-144`
* `install_into_llava()`    patches the two names `llava_arch.py` imported (`llava_arch.py:22-23`)
  so `LlavaMetaModel.__init__` (`llava_arch.py:34-36`) builds our tower, and - with `splice=True` -
  replaces `prepare_inputs_labels_for_multimodal` (`llava_arch.py:146-332`) by the anyres feature merge
  + ONE splice kernel of `ml_fastvlm_amd/splice.py`; see INTEGRATION.md.
"""
from __future__ import annotations

import re

import torch
import torch.nn as nn

from .mobileclip_encoder import MobileCLIPVisionTower


def build_vision_tower(vision_tower_cfg, **kwargs):
    vision_tower = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if vision_tower is not None and "mobileclip" in vision_tower.lower():
        return MobileCLIPVisionTower(vision_tower, args=vision_tower_cfg, **kwargs)
    # CLIP / CLIP-S2 towers (multimodal_encoder/clip_encoder.py) are not FastViTHD and not on this path.
    raise ValueError(f"Unknown vision tower: {vision_tower}")


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


def build_vision_projector(config, delay_load=False, **kwargs):
    """Same module structure (and therefore the same state-dict keys `0.weight, 0.bias, 2.weight,
    2.bias`) as the reference; `encode_images` below recognises the `mlp2x_gelu` shape and routes
    it through the fused library call."""
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return nn.Linear(config.mm_hidden_size, config.hidden_size)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        depth = int(m.group(1))
        modules = [nn.Linear(config.mm_hidden_size, config.hidden_size)]
        for _ in range(1, depth):
            modules.append(nn.GELU())
            modules.append(nn.Linear(config.hidden_size, config.hidden_size))
        return nn.Sequential(*modules)
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")


def _is_mlp2x_gelu(p) -> bool:
    return (isinstance(p, nn.Sequential) and len(p) == 3 and isinstance(p[0], nn.Linear)
            and isinstance(p[1], nn.GELU) and isinstance(p[2], nn.Linear)
            and p[0].bias is not None and p[2].bias is not None
            and p[0].out_features % 32 == 0 and p[0].in_features % 32 == 0)


def library_projector(vision_tower, mm_projector) -> bool:
    """True when `mm_projector` runs on the library's GEMM kernels: our tower, an `mlp2x_gelu` projector on the tower's HIP device,
    and no gradient wanted through it (the HIP path is inference-only; a trainable projector under grad mode keeps autograd's
    nn.Sequential, as the reference's training does)."""
    return (isinstance(vision_tower, MobileCLIPVisionTower) and _is_mlp2x_gelu(mm_projector)
            and mm_projector[0].weight.device == vision_tower.device
            and not (torch.is_grad_enabled() and any(p.requires_grad for p in mm_projector.parameters())))


def project(vision_tower, mm_projector, image_features):
    """`mm_projector(image_features)` (llava_arch.py:143) for already-encoded tokens: `fvhd_project` whenever `library_projector`."""
    if library_projector(vision_tower, mm_projector):
        return vision_tower.project(image_features, mm_projector)
    return mm_projector(image_features)


def encode_images(vision_tower, mm_projector, images):
    """tower(images) -> mm_projector(features).  With our tower and an `mlp2x_gelu` projector on the
    same HIP device this is ONE library call (tokens stay in bf16 workspace between the two); a list of images is encoded as one
    batch and every piece projected by the library too (`MobileCLIPVisionTower.project`)."""
    if isinstance(images, torch.Tensor) and library_projector(vision_tower, mm_projector):
        return vision_tower.encode_images_with_projector(images, mm_projector)
    image_features = vision_tower(images)
    if isinstance(image_features, list):
        return [project(vision_tower, mm_projector, f) for f in image_features]
    return project(vision_tower, mm_projector, image_features)


def install_into_llava(splice: bool = False, prefill: bool = False, prefill_any_dtype: bool = False) -> None:
    """Make an unmodified `llava` package (the reference) build and call the MI355X tower; splice=True also routes
    `prepare_inputs_labels_for_multimodal` through the GPU splice (needs the embeddings on a HIP device); prefill=True also runs the
    PREFILL step of `LlavaQwen2ForCausalLM.forward` (`llava_qwen.py:92-103`: the first forward of `generate`, on `inputs_embeds`
    with an empty cache) on the hand-written Qwen2 kernels (`ml_fastvlm_amd.qwen2_prefill`), handing the KV cache to the stock
    decode loop.  The kernels compute in bf16: by default only a bf16 model takes them (an fp32 / fp16 model keeps the reference's forward
    and its precision); prefill_any_dtype=True opts such a model in knowingly (its prefill then runs in bf16, the cache is cast back)."""
    import llava.model.llava_arch as arch
    import llava.model.multimodal_encoder.builder as enc_builder

    ref_build = enc_builder.build_vision_tower

    def build(vision_tower_cfg, **kwargs):
        name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
        if name is not None and "mobileclip" in name.lower():
            return MobileCLIPVisionTower(name, args=vision_tower_cfg, **kwargs)
        return ref_build(vision_tower_cfg, **kwargs)

    enc_builder.build_vision_tower = build
    arch.build_vision_tower = build                      # llava_arch.py:22 imported the name

    def _encode_images(self, images):                    # replaces llava_arch.py:141-144
        return encode_images(self.get_model().get_vision_tower(), self.get_model().mm_projector, images)

    arch.LlavaMetaForCausalLM.encode_images = _encode_images
    if splice:
        arch.LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal = prepare_inputs_labels_for_multimodal
    if prefill:
        import llava.model.language_model.llava_qwen as lq
        cur = lq.LlavaQwen2ForCausalLM.forward
        lq.LlavaQwen2ForCausalLM.forward = _make_prefill_forward(getattr(cur, "_fvhd_orig", cur), any_dtype=prefill_any_dtype)


def _is_fresh_dynamic_cache(pkv) -> bool:
    """True only for what `generate` hands to its FIRST forward: an empty `transformers.DynamicCache` instance.  `None`, a StaticCache, a
    legacy tuple or a cache that already holds tokens all mean "not generate's prefill" and keep the reference's forward."""
    try:
        from transformers import DynamicCache
        return isinstance(pkv, DynamicCache) and pkv.get_seq_length() == 0
    except Exception:
        return False


def _make_prefill_forward(orig_forward, any_dtype: bool = False):
    """`LlavaQwen2ForCausalLM.forward` with its prefill step on `fvhd_llm_prefill`.  The kernel path returns the logits of the LAST
    position only ([B, 1, vocab] - what `generate` reads: `outputs.logits[:, -1, :]`), computes in bf16 and fills a DynamicCache, so it is
    taken only where the caller is demonstrably `generate`'s first step (advisor, round 3: a plain scoring forward under no_grad must keep
    its [B, T, vocab] logits): `use_cache` true AND `past_key_values` an EMPTY `DynamicCache` instance (generate creates it before the first
    forward; a bare `model(...)` call passes None), a bf16 model on a HIP device, a multi-token 2-D-masked `inputs_embeds`, no labels, no
    grad, no attention / hidden-state outputs.  Anything else - and any input the kernel path rejects (a 4-D mask, a rope type it does
    not implement, ...) - is the reference's own forward."""
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None,
                cache_position=None, **kwargs):
        if inputs_embeds is None and images is not None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, past_key_values, labels, images, image_sizes)

        def reference():
            return orig_forward(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                                inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache, output_attentions=output_attentions,
                                output_hidden_states=output_hidden_states, return_dict=return_dict, cache_position=cache_position, **kwargs)

        use = use_cache if use_cache is not None else getattr(self.config, "use_cache", False)
        eligible = (inputs_embeds is not None and inputs_embeds.dim() == 3 and inputs_embeds.shape[1] > 1 and labels is None
                    and not torch.is_grad_enabled() and inputs_embeds.device.type == "cuda"
                    and (any_dtype or (inputs_embeds.dtype == torch.bfloat16 and self.lm_head.weight.dtype == torch.bfloat16)) and bool(use) and _is_fresh_dynamic_cache(past_key_values)
                    and (attention_mask is None or attention_mask.dim() == 2)
                    and not output_attentions and not output_hidden_states and return_dict is not False)
        if not eligible:
            return reference()
        from transformers.modeling_outputs import CausalLMOutputWithPast
        from .qwen2_prefill import Qwen2Prefill
        try:
            pre = prefill_context(self)
            pos = position_ids
            if pos is None and attention_mask is not None:               # as prepare_inputs_for_generation derives them from the mask
                pos = torch.clamp(attention_mask.long().cumsum(-1) - 1, min=0)
            logits, k, v = pre(inputs_embeds, attention_mask, pos, return_kv=True)
        except (NotImplementedError, ValueError, KeyError) as e:         # an input / architecture the kernels do not cover
            if not getattr(self, "_fvhd_prefill_warned", False):
                import warnings
                warnings.warn(f"ml_fastvlm_amd: prefill stays on the reference forward ({type(e).__name__}: {e})")
                object.__setattr__(self, "_fvhd_prefill_warned", True)
            return reference()
        for layer in range(k.shape[0]):
            past_key_values.update(k[layer].to(inputs_embeds.dtype), v[layer].to(inputs_embeds.dtype), layer)
        return CausalLMOutputWithPast(loss=None, logits=logits[:, None, :], past_key_values=past_key_values)
    forward._fvhd_prefill = True
    forward._fvhd_orig = orig_forward
    return forward


def prefill_context(model):
    """The `Qwen2Prefill` context of a (Llava)Qwen2ForCausalLM, built on first use and rebuilt when its weights change (in place or by
    re-assignment).  `install_into_llava(prefill=True)` users call this once after loading the model so that the packing (device-to-device
    copies, ~0.1 s for 0.5B) is not part of the first request's TTFT."""
    from .qwen2_prefill import Qwen2Prefill
    key = tuple((p.data_ptr(), p._version) for p in (model.lm_head.weight, model.model.layers[0].self_attn.q_proj.weight,
                                                      model.model.layers[-1].mlp.down_proj.weight, model.model.norm.weight))
    pre = getattr(model, "_fvhd_prefill_ctx", None)
    if pre is None or pre[0] != key:
        pre = (key, Qwen2Prefill.from_hf(model))
        object.__setattr__(model, "_fvhd_prefill_ctx", pre)
    return pre[1]


def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, image_sizes=None):
    """Same contract as `LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` (`llava_arch.py:146-332`): the early outs, the
    list / 5-D image branch with its patch merge, then the embedding splice - as one gather kernel instead of the per-sample
    Python walk.  `self` is the reference's model object (it provides `get_vision_tower`, `encode_images`, `get_model`, `config`)."""
    from . import splice as S
    vision_tower = self.get_vision_tower()
    if vision_tower is None or images is None or input_ids.shape[1] == 1:           # llava_arch.py:150-152
        return input_ids, position_ids, attention_mask, past_key_values, None, labels
    cfg = self.config
    if isinstance(images, list) or images.ndim == 5:                               # tiles per image (anyres) or ragged lists
        tiles = [x.unsqueeze(0) if x.ndim == 3 else x for x in images] if isinstance(images, list) else list(images)
        feats = self.encode_images(torch.cat(tiles, 0))
        feats = list(torch.split(feats, [t.shape[0] for t in tiles], 0))
        merge = getattr(cfg, "mm_patch_merge_type", "flat")
        if merge.startswith("spatial") and getattr(cfg, "image_aspect_ratio", "square") != "anyres" and any(f.shape[0] > 1 for f in feats):
            raise NotImplementedError                                            # as the reference (llava_arch.py:184-185)
        tower_cfg = getattr(vision_tower, "config", None)
        size = getattr(vision_tower, "s2_image_size", None) or (tower_cfg["image_cfg"]["image_size"] if isinstance(tower_cfg, dict)
                                                                else getattr(tower_cfg, "image_size", None))
        newline = getattr(getattr(self, "model", None), "image_newline", None)
        image_features = S.merge_patch_features(feats, image_sizes if image_sizes is not None else [None] * len(feats), merge,
                                                getattr(cfg, "image_grid_pinpoints", None), size, newline)
    else:
        image_features = self.encode_images(images)
    if getattr(cfg, "tune_mm_mlp_adapter", False) and getattr(cfg, "mm_use_im_start_end", False):
        raise NotImplementedError                                                # llava_arch.py:214-215
    out = S.multimodal_splice(input_ids, position_ids, attention_mask, labels, image_features, self.get_model().embed_tokens.weight,
                              getattr(cfg, "tokenizer_padding_side", "right"), getattr(cfg, "tokenizer_model_max_length", None))
    return out[0], out[1], out[2], past_key_values, out[4], out[5]
