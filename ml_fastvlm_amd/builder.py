"""Factories mirroring the reference's registration points, plus the one-line hook that swaps
the MI355X tower into an unmodified checkout of apple/ml-fastvlm.

* `build_vision_tower`      <-> `llava/model/multimodal_encoder/builder.py:6-19`
* `build_vision_projector`  <-> `llava/model/multimodal_projector/builder.py:17-35`
* `encode_images`           <-> `LlavaMetaForCausalLM.encode_images`, `llava/model/llava_arch.py:141-144`
* `install_into_llava()`    patches the two names `llava_arch.py` imported (`llava_arch.py:22-23`)
  so `LlavaMetaModel.__init__` (`llava_arch.py:34-36`) builds our tower; see INTEGRATION.md.
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


def encode_images(vision_tower, mm_projector, images):
    """tower(images) -> mm_projector(features).  With our tower and an `mlp2x_gelu` projector on the
    same HIP device this is ONE library call (tokens stay in bf16 workspace between the two)."""
    if (isinstance(vision_tower, MobileCLIPVisionTower) and isinstance(images, torch.Tensor)
            and _is_mlp2x_gelu(mm_projector) and mm_projector[0].weight.device == vision_tower.device
            and not (torch.is_grad_enabled() and any(p.requires_grad for p in mm_projector.parameters()))):
        return vision_tower.encode_images_with_projector(images, mm_projector)
    image_features = vision_tower(images)
    if isinstance(image_features, list):
        return [mm_projector(f) for f in image_features]
    return mm_projector(image_features)


def install_into_llava() -> None:
    """Make an unmodified `llava` package (the reference) build and call the MI355X tower."""
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
