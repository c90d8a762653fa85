"""ml_fastvlm_amd - MI355X (gfx950) implementation of FastVLM's `encode_images()` hot path:
FastViTHD vision tower + `mlp2x_gelu` projector behind the reference's own tower API.

Public surface (mirrors `llava.model.multimodal_encoder` / `multimodal_projector`):
    MobileCLIPVisionTower, build_vision_tower, build_vision_projector, encode_images, project,
    install_into_llava, and `distributed` for the one-process-per-GPU data-parallel path.
"""
from .builder import build_vision_projector, build_vision_tower, encode_images, install_into_llava, library_projector, project  # noqa: F401
from .mobileclip_encoder import MobileCLIPVisionTower, load_model_config  # noqa: F401

__all__ = ["MobileCLIPVisionTower", "build_vision_tower", "build_vision_projector", "encode_images", "project",
           "library_projector", "install_into_llava", "load_model_config"]
