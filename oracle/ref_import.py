"""TEST INFRASTRUCTURE ONLY - imports the *reference's own* FastViTHD code, unmodified.

Used (a) by `oracle/make_golden.py` to generate the golden fixtures committed under
`tests/golden/`, (b) by the `-m "not gpu"` pinning tests when `/root/reference`
is present, and (c) on the GPU box - where `/root/reference` does not exist - by
`tests/test_gpu_reference.py` and bench.py's `cpu_baseline`, through the archive that
`oracle/stage_reference.py` stages under the git-ignored `oracle/_ref/` at build time.

The reference's hot path imports `timm` (registry + DropPath + two constants,
`mci.py:15-17`, `mobileclip/__init__.py:10`), which is not installed in this image.
`install_timm_stub()` registers an in-memory stand-in for exactly those names; every
line of arithmetic that then runs is the reference's own (`torch.nn` modules).
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

import torch.nn as nn

_MCI = "llava/model/multimodal_encoder/mobileclip/mci.py"


def _resolve_root() -> str:
    """FVHD_REFERENCE_ROOT, else the mounted checkout (build container), else the archive `oracle/stage_reference.py` staged under
    `oracle/_ref/` (GPU box), extracted into a scratch directory outside the repository."""
    env = os.environ.get("FVHD_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile(os.path.join("/root/reference", _MCI)):
        return "/root/reference"
    from . import stage_reference
    return stage_reference.unpack() or "/root/reference"


REFERENCE_ROOT = _resolve_root()
_REGISTRY = {}


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, _MCI))


def install_timm_stub() -> None:
    if "timm" in sys.modules and not getattr(sys.modules["timm"], "_fvhd_stub", False):
        return  # a real timm is present; use it

    def _mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)  # transformers calls find_spec("timm")
        m.__version__ = "1.0.15"                                 # the reference's pin (pyproject.toml:21)
        m.__path__ = []
        m._fvhd_stub = True
        sys.modules[name] = m
        return m

    timm, models, data, layers = map(_mod, ["timm", "timm.models", "timm.data", "timm.layers"])

    def register_model(fn):                       # used at mci.py:1454
        _REGISTRY[fn.__name__] = fn
        return fn

    def create_model(name, **kw):                 # used at mobileclip/__init__.py:46
        return _REGISTRY[name](**kw)

    class DropPath(nn.Module):                    # mci.py:17; never active (drop_path_rate=0)
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    models.register_model, models.create_model = register_model, create_model
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)            # mci.py:16 (unused on this path)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    layers.DropPath, layers.SqueezeExcite = DropPath, nn.Identity   # SqueezeExcite unused (use_se=False)
    timm.models, timm.data, timm.layers = models, data, layers


def import_reference():
    """Returns the reference modules needed by the golden generator."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    install_timm_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from llava.model.multimodal_encoder import mobileclip_encoder          # noqa: E402
    from llava.model.multimodal_encoder.mobileclip import mci             # noqa: E402
    from llava.model.multimodal_projector import builder as proj_builder   # noqa: E402
    return SimpleNamespace(mobileclip_encoder=mobileclip_encoder, mci=mci, proj_builder=proj_builder)


def build_reference_tower(res: int = 1024):
    """`MobileCLIPVisionTower("mobileclip_l_<res>", args)`, eager load (delay_load=False)."""
    ref = import_reference()
    tower = ref.mobileclip_encoder.MobileCLIPVisionTower(
        f"mobileclip_l_{res}", SimpleNamespace(unfreeze_mm_vision_tower=False))
    tower.eval()
    return tower


def build_reference_projector(hidden: int, mm_hidden: int = 3072):
    ref = import_reference()
    cfg = SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=mm_hidden, hidden_size=hidden)
    return ref.proj_builder.build_vision_projector(cfg).eval()
