"""MI355X drop-in for the reference's `MobileCLIPVisionTower`
(`llava/model/multimodal_encoder/mobileclip_encoder.py:13-116`).

Same constructor, methods, properties, state-dict key names, input/output shapes and error
behaviour; the forward pass runs the hand-written gfx950 kernels of `libfvhd.so` through the C ABI
of `include/fvhd.h` instead of `torch.nn` modules.  There is no CPU path: calling `forward` with
the parameters on a non-HIP device raises.

Differences that are deliberate and documented in DESIGN.md:
* arithmetic is bf16 activations / fp32 accumulation whatever `self.dtype` is (the tier's metric
  dtype); `self.dtype` still reports the parameters' dtype and the output is cast to
  `images.dtype` exactly as the reference does (`mobileclip_encoder.py:85-86`);
* inference only: `unfreeze_mm_vision_tower` (training through the tower,
  `mobileclip_encoder.py:70-75`) raises `NotImplementedError` when gradients are enabled.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import _lib
from . import fastvithd_spec as spec

# mobileclip/configs/mobileclip_l.json, image side only (the text tower is not on this path)
_MODEL_CONFIGS = {
    "mobileclip_l": {
        "embed_dim": spec.PROJECTION_DIM,
        "image_cfg": {"image_size": 1024, "model_name": "fastvithd", "embed_dim": spec.OUT_DIM,
                      "patch_size": spec.PATCH_SIZE},
    },
}


def load_model_config(model_name: str) -> Dict:
    """Mirror of `mobileclip.load_model_config` (`mobileclip/__init__.py:15-31`): the suffix after
    the second underscore-separated field is stripped; unknown names raise ValueError."""
    base = "_".join(model_name.split("_")[0:2])
    if base not in _MODEL_CONFIGS:
        raise ValueError(f"Unsupported model name: {base}")
    return copy.deepcopy(_MODEL_CONFIGS[base])


class _ParamTree(nn.Module):
    """A bare container: holds parameters/buffers/children under the reference's names."""


def _build_param_tree(root: nn.Module) -> None:
    for key, (shape, kind) in spec.param_spec().items():
        parts = key.split(".")
        mod = root
        for name in parts[:-1]:
            child = mod._modules.get(name)
            if child is None:
                child = _ParamTree()
                mod.add_module(name, child)
            mod = child
        leaf = parts[-1]
        if kind == "param":
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        elif kind == "buffer":
            mod.register_buffer(leaf, torch.ones(shape) if leaf == "running_var" else torch.zeros(shape))
        else:
            mod.register_buffer(leaf, torch.zeros(shape, dtype=torch.int64))


class FastViTHDWeights(nn.Module):
    """Stands where the reference has `MCi` (`mobileclip/__init__.py:34-58`): a child called
    `model` whose state-dict keys are FastViT's (`patch_embed.0.reparam_conv.weight`, ...)."""

    def __init__(self) -> None:
        super().__init__()
        self.model = _ParamTree()
        _build_param_tree(self.model)
        self._reset_parameters()

    def _reset_parameters(self) -> None:
        # Deterministic, non-degenerate values so that an un-loaded tower is still a valid
        # (if meaningless) encoder; real use goes through load_state_dict.
        from . import synth
        sd = synth.synthetic_state_dict(seed=0)
        self.model.load_state_dict(sd, strict=True)


class _DirtyHook:
    """load_state_dict post-hook (a picklable callable, unlike a lambda): any load into the subtree invalidates the packed copy."""

    def __init__(self, tower):
        self.tower = tower

    def __call__(self, module, incompatible_keys):
        self.tower._mark_dirty()


class MobileCLIPVisionTower(nn.Module):
    def __init__(self, vision_tower: str, args, delay_load: bool = False):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.tune_vision_tower = getattr(args, "unfreeze_mm_vision_tower", False)
        self.input_image_size = int(vision_tower.split("_")[-1])
        # MI355X-only options (no counterpart in the reference), read from the config object like the reference reads its own
        # switches.  Tri-state: None = leave the library's setting alone (fvhd_create reads FVHD_GRAPH from the environment,
        # INTEGRATION.md); True / False = explicit, pushed to the context before every call.
        # e4m3 MFMA operands in the MHSA core: the "fp8 MFMA attention path" BASELINE.json configs[4] names (include/fvhd.h:
        # fvhd_set_attention_fp8); opt-in, measured no faster than the bf16 parity path (DESIGN.md "fp8")
        fp8 = getattr(args, "mm_vision_attention_fp8", None)
        self.attention_fp8 = None if fp8 is None else bool(fp8)
        # hipGraph replay of the tower's interior launches (include/fvhd.h: fvhd_set_graph), for launch-bound batches
        graph = getattr(args, "mm_vision_hip_graph", None)
        self.hip_graph = None if graph is None else bool(graph)
        inv = getattr(args, "mm_vision_batch_invariant", None)
        self.batch_invariant = None if inv is None else bool(inv)
        # precision of the fused ConvFFN's hidden activation (include/fvhd.h "precision of the fused ConvFFN's hidden activation"):
        # "auto" (DEFAULT since round 5: the first `mm_vision_ffn_audit_batches` non-degenerate batches encoded after a weight load - or an
        # explicit `calibrate(images)` - run `audit_ranges()` first: one extra eager pass each, then only the blocks that need it run the
        # bf16 form), "half" (no audit: gelu(x)/4 in IEEE half - saturates beyond |fc1 output| = 262 016; the range guard below still
        # applies) or "bf16" (every block on the f32-GELU / bf16-operand form of the same kernel: no range limit, a few % slower).
        prec = getattr(args, "mm_vision_ffn_precision", None) or "auto"
        if prec not in ("half", "bf16", "auto"):
            raise ValueError(f"mm_vision_ffn_precision must be 'half', 'bf16' or 'auto', got {prec!r}")
        self.ffn_precision = prec
        self.ffn_audit_batches = max(1, int(getattr(args, "mm_vision_ffn_audit_batches", None) or 2))
        self._ffn_bf16_steps = set()            # steps an audit / the guard moved to the bf16 form (re-applied whenever the weights are re-packed)
        self._ffn_audits_left = self.ffn_audit_batches   # "auto": calibration batches the current weight set has still to see
        self._degenerate_seen = 0                        # constant (warm-up) batches looked at so far: each look is a host sync
        # the range guard (include/fvhd.h "range guard"): always on unless "off"; "strict" = poll synchronously after every call and
        # re-encode the batch when a block crossed its limit (the result of every call is then inside the proven range, at the price of
        # one host synchronisation per call); "on" (default) = asynchronous: the block is moved before the NEXT call, with a warning
        # (unset / None = tri-state like the other options: the library's own setting - on unless FVHD_RANGE_GUARD=0 - is left alone)
        guard = getattr(args, "mm_vision_range_guard", None)
        guard = "on" if guard is True else "off" if guard is False else guard
        if guard not in (None, "on", "off", "strict"):
            raise ValueError(f"mm_vision_range_guard must be 'on', 'off' or 'strict', got {guard!r}")
        self._range_guard = guard
        # expected batch size (sizes the library's workspace up front; it grows geometrically when a larger batch arrives)
        self._batch_hint = max(1, int(getattr(args, "mm_vision_max_batch", 1) or 1))
        self._ctx: Optional[_lib.Context] = None
        self._ctx_key = None
        self._dirty = True
        self._projector_src = None

        if not delay_load:
            self.load_model()
        elif getattr(args, "unfreeze_mm_vision_tower", False):
            self.load_model()
        else:
            self.cfg_only = load_model_config(self.vision_tower_name)

    # ---- construction ------------------------------------------------------------------------------
    def load_model(self, device_map=None):
        if self.is_loaded:
            print("{} is already loaded, `load_model` called again, skipping.".format(self.vision_tower_name))
            return
        model_cfg = load_model_config(self.vision_tower_name)
        model_cfg["image_cfg"]["image_size"] = self.input_image_size      # mobileclip_encoder.py:40
        if self.input_image_size % spec.PATCH_SIZE:
            raise ValueError(f"image size {self.input_image_size} is not a multiple of {spec.PATCH_SIZE}")
        self.cfg_only = model_cfg

        from transformers import CLIPImageProcessor                       # mobileclip_encoder.py:45-49
        r = model_cfg["image_cfg"]["image_size"]
        self.image_processor = CLIPImageProcessor(crop_size={"height": r, "width": r},
                                                  image_mean=[0.0, 0.0, 0.0], image_std=[1.0, 1.0, 1.0],
                                                  size={"shortest_edge": r})
        self.vision_tower = FastViTHDWeights()
        self.vision_tower.requires_grad_(bool(self.tune_vision_tower))
        # load_state_dict may be called on the tower, on an ancestor (the whole LLaVA model) or on any
        # descendant (tests load into `.vision_tower.model`): hook every module of the subtree.
        for mod in self.modules():
            mod.register_load_state_dict_post_hook(_DirtyHook(self))
        self.is_loaded = True
        self._dirty = True

    def _mark_dirty(self) -> None:
        self._dirty = True
        self._ffn_bf16_steps = set()            # new weights: an earlier range audit says nothing about them
        self._ffn_audits_left = self.ffn_audit_batches

    @property
    def range_guard(self) -> str:
        """"on" (default), "strict" or "off"; unset = "on" unless the library was created with FVHD_RANGE_GUARD=0 in the environment"""
        if self._range_guard is not None:
            return self._range_guard
        import os
        return "off" if os.environ.get("FVHD_RANGE_GUARD", "1") == "0" else "on"

    @property
    def _ffn_audited(self) -> bool:
        """"auto": has the current weight set seen all of its calibration batches?"""
        return self._ffn_audits_left <= 0

    def _apply(self, fn, *a, **kw):
        # .to() / .half() / .cuda(): parameter storage (and possibly values, through a dtype cast) changes
        self._dirty = True
        return super()._apply(fn, *a, **kw)

    # ---- weights -> library ------------------------------------------------------------------------
    def _context(self) -> "_lib.Context":
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                f"MobileCLIPVisionTower (MI355X): parameters are on {dev}; this tower has no CPU path - "
                "move the model to a HIP device (`.to('cuda')`).")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (idx, self.input_image_size)
        if self._ctx is None or self._ctx_key != key:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _lib.Context(idx, self.input_image_size, max_batch=self._batch_hint)
            self._ctx_key = key
            self._dirty = True
            self._projector_src = None
        if self._dirty:
            self._ctx.set_tensors(self.vision_tower.model.state_dict())     # one device-to-host transfer for the 629 tensors
            self._ctx.finalize()
            self._dirty = False
            self._apply_ffn_precision(self._ctx)
        if self._range_guard is not None:
            self._ctx.set_range_guard(self._range_guard != "off")
        if self.attention_fp8 is not None:
            self._ctx.set_attention_fp8(self.attention_fp8)
        if self.hip_graph is not None:
            self._ctx.set_graph(self.hip_graph)
        if self.batch_invariant is not None:
            self._ctx.set_batch_invariant(self.batch_invariant)
        return self._ctx

    def _apply_ffn_precision(self, ctx) -> None:
        for step in range(len(ctx.steps())):
            if ctx.ffn_precision(step) >= 0 and (self.ffn_precision == "bf16" or step in self._ffn_bf16_steps):
                ctx.set_ffn_precision(step, _lib.FFN_BF16)

    def audit_ranges(self, images: torch.Tensor, switch_above: float = 65504.0):
        """Range audit of the half-precision hidden activation of the fused ConvFFN kernels (no counterpart in the reference, whose bf16 /
        fp32 hidden tensor has no such limit).  Run it ONCE per checkpoint on a calibration batch of real, preprocessed images: every
        ConvFFN's fc1 output is materialised and reduced to max |.|; a fused block whose maximum exceeds `switch_above` (default 65 504: a
        factor 4 below the 262 016 where gelu(x)/4 saturates in f16) - or is not finite - is switched to the bf16-operand form of the
        kernel for all later calls of this tower (remembered across `.to()` / re-packing; a `load_state_dict` clears it).  Returns
        [{"step", "kind", "stage", "block", "max_abs_fc1", "precision", "switched"}] for every step that has a ConvFFN and warns when
        anything was switched.  Synchronises."""
        with torch.no_grad():
            images = self._check_images(images)
            ctx = self._context()
            self._grow(ctx, images.shape[0])
            before = [ctx.ffn_precision(i) for i in range(len(ctx.steps()))]
            maxes, switched = ctx.audit_ranges(images, switch_above)
        report = []
        for i, (kind, stage, block, *_rest) in enumerate(ctx.steps()):
            if kind not in ("repmixer_block", "attention_block"):
                continue
            now = ctx.ffn_precision(i)
            if now == _lib.FFN_BF16 and before[i] == _lib.FFN_HALF:
                self._ffn_bf16_steps.add(i)
            report.append({"step": i, "kind": kind, "stage": stage, "block": block, "max_abs_fc1": maxes[i],
                           "precision": {-1: "two GEMMs (bf16 hidden tensor in HBM)", 0: "half", 1: "bf16"}[now],
                           "switched": now == _lib.FFN_BF16 and before[i] == _lib.FFN_HALF})
        if switched:
            import warnings
            hot = [(r["step"], r["max_abs_fc1"]) for r in report if r["switched"]]
            warnings.warn(f"ml_fastvlm_amd: {switched} ConvFFN block(s) exceed |fc1 output| = {switch_above:g} on the calibration batch and now run "
                          f"the bf16-operand form of the fused kernel (step, max): {hot}")
        return report

    # The ctypes handle is process-local state, not model state: copies / pickles of a tower start without a context and
    # re-pack their weights on first use (copy.deepcopy of a model after its first forward used to fail on the handle).
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_ctx"], st["_ctx_key"], st["_dirty"], st["_projector_src"] = None, None, True, None
        return st

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():       # (the _DirtyHook objects are copied along and point at the copy: memo)
            object.__setattr__(new, k, copy.deepcopy(v, memo))
        return new

    def sync_weights(self) -> None:
        """Force a re-pack (e.g. after in-place edits of parameters that no hook can see)."""
        self._dirty = True
        self._context()

    # ---- forward (mobileclip_encoder.py:70-88) -----------------------------------------------------
    def forward(self, images: Union[torch.Tensor, List[torch.Tensor]]):
        if self.tune_vision_tower and torch.is_grad_enabled():
            raise NotImplementedError("the MI355X tower is inference-only (no backward through the HIP kernels)")
        with torch.no_grad():
            return self.forward_images(images)

    def _check_images(self, images: torch.Tensor) -> torch.Tensor:
        """Shape / dtype / device normalisation shared by every entry point: the library trusts B and the context's R, so a
        mis-shaped tensor must be rejected HERE (the reference raises a shape error inside its first conv)."""
        if not isinstance(images, torch.Tensor) or images.dim() != 4 or images.shape[1] != 3 \
                or images.shape[2] != images.shape[3] or images.shape[2] != self.input_image_size or images.shape[0] < 1:
            raise ValueError(f"expected images of shape [B,3,{self.input_image_size},{self.input_image_size}], "
                             f"got {tuple(images.shape) if isinstance(images, torch.Tensor) else type(images)}")
        images = images.to(device=self.device).contiguous()
        if images.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            images = images.float()
        return images

    def _grow(self, ctx, batch: int) -> None:
        """Workspace growth synchronises the device and re-allocates (include/fvhd.h): do it geometrically, not per batch size."""
        if batch > self._batch_hint:
            while self._batch_hint < batch:
                self._batch_hint *= 2
            ctx.reserve(self._batch_hint)

    # "auto" audits with a factor 16 (not 4) of headroom below the saturation point: the calibration batches are whatever arrives first
    AUTO_SWITCH_ABOVE = 16376.0

    def calibrate(self, images: torch.Tensor, switch_above: float = AUTO_SWITCH_ABOVE):
        """Explicit calibration (round 5): `audit_ranges(images)` on a batch of REAL, preprocessed images, after which "auto" does not audit
        incoming batches any more.  Call it once per checkpoint before serving (and before capturing the tower into a caller's hipGraph:
        the audit synchronises).  Returns the audit's report."""
        report = self.audit_ranges(images, switch_above)
        self._ffn_audits_left = 0
        return report

    @staticmethod
    def _degenerate(images: torch.Tensor) -> bool:
        """a batch that says nothing about activation ranges: every image constant (zeros / ones warm-up batches, dummy inputs)"""
        flat = images.reshape(images.shape[0], -1)
        return bool((flat.amax(dim=1) == flat.amin(dim=1)).all().item())

    def _auto_audit(self, images: torch.Tensor) -> None:
        if self.ffn_precision != "auto" or self._ffn_audits_left <= 0:
            return
        if torch.cuda.is_current_stream_capturing():
            return                              # the audit synchronises: a capturing caller calibrates up front
        if self._degenerate_seen < 8 and self._degenerate(images):
            self._degenerate_seen += 1          # (a host sync per look: at most 8 constant batches are examined, then the next batch counts)
            return                              # not counted: a zeros warm-up batch must not use up the calibration
        self.audit_ranges(images, self.AUTO_SWITCH_ABOVE)
        self._ffn_audits_left -= 1

    def _after_encode(self, ctx, rerun=None):
        """range guard: report (and remember across re-packs) the blocks the library moved to the bf16-operand form.  "strict": wait for this
        call's read-back and run the batch again if it crossed a limit."""
        if self.range_guard == "off" or torch.cuda.is_current_stream_capturing():
            return                              # (a capturing caller: the library's guard is inactive, and polling events is not capture-safe)
        import warnings
        strict = self.range_guard == "strict"
        # "strict": a block moved to the bf16-operand form changes its output, which can push a LATER block over its limit in the re-run -
        # so the re-run is polled too, until a pass crosses nothing (at most one pass per fused block: a block never moves back to half)
        for _ in range(64):
            hits = ctx.range_guard_poll(wait=strict)
            if not hits:
                return
            for step, _a in hits:
                self._ffn_bf16_steps.add(step)
            warnings.warn("ml_fastvlm_amd range guard: max |A| of the ConvFFN input exceeded the proven half-precision range in step(s) "
                          f"{[(s, float(a)) for s, a in hits]} (limits: {[ctx.range_guard_limit(s) for s, _ in hits]}); those blocks now run the "
                          "bf16-operand form of the fused kernel"
                          + ("" if strict else " - the batch that crossed the limit was computed on the half-precision form"))
            if not (strict and rerun is not None):
                return
            rerun()

    def _encode(self, images: torch.Tensor) -> torch.Tensor:
        images = self._check_images(images)
        ctx = self._context()
        self._auto_audit(images)
        self._grow(ctx, images.shape[0])
        out = torch.empty((images.shape[0], ctx.num_tokens, self.hidden_size), device=images.device, dtype=images.dtype)
        ctx.encode(images, out)
        self._after_encode(ctx, lambda: ctx.encode(images, out))
        return out

    def forward_images(self, images):
        if type(images) is list:
            if len(images) == 0:
                return []
            # the reference loops B=1 calls (mobileclip_encoder.py:78-83); images are independent, so
            # same-dtype lists are encoded as one batch and split back into [1, T, C] pieces.
            if all(im.dtype == images[0].dtype for im in images):
                feats = self._encode(torch.stack([im.to(self.device) for im in images], 0))
                return [feats[i:i + 1] for i in range(len(images))]
            return [self._encode(im.unsqueeze(0)) for im in images]
        return self._encode(images)

    # ---- the projector on the library's GEMM kernels (llava_arch.py:143) ----------------------------
    def _check_projector(self, projector: nn.Module):
        """shape check of the `mlp2x_gelu` weights (multimodal_projector/builder.py:23-30) - before the library is touched"""
        w0, b0, w2, b2 = projector[0].weight, projector[0].bias, projector[2].weight, projector[2].bias
        hid = w0.shape[0]
        if w0.dim() != 2 or w0.shape[1] != self.hidden_size or tuple(w2.shape) != (hid, hid) \
                or tuple(b0.shape) != (hid,) or tuple(b2.shape) != (hid,):
            raise ValueError(f"mlp2x_gelu projector must be Linear({self.hidden_size}, H) -> GELU -> Linear(H, H); got weights "
                             f"{tuple(w0.shape)}, {tuple(w2.shape)}")
        return w0, b0, w2, b2

    def _bind_projector(self, ctx, weights) -> int:
        """hand the projector weights to the context once per (storage, version); returns H"""
        src = tuple((t.data_ptr(), t._version, t.dtype) for t in weights)
        if self._projector_src != src:
            ctx.set_projector(*weights)
            self._projector_src = src
        return weights[0].shape[0]

    def project(self, tokens: torch.Tensor, projector: nn.Module) -> torch.Tensor:
        """`mm_projector(image_features)` (llava_arch.py:143) for tokens that are ALREADY encoded - the gather-before-projector leg of the
        multi-GPU path (FastVLM-7B, H = 3584 > 3072: distributed.gather_side) and list inputs: [..., 3072] -> [..., H] through
        `fvhd_project` (two hand-written GEMM launches with fused bias / GELU epilogues), never through torch.nn.Linear.  Inference only."""
        with torch.no_grad():
            if not isinstance(tokens, torch.Tensor) or tokens.dim() < 2 or tokens.shape[-1] != self.hidden_size or tokens.numel() == 0:
                raise ValueError(f"expected tokens of shape [..., {self.hidden_size}], got "
                                 f"{tuple(tokens.shape) if isinstance(tokens, torch.Tensor) else type(tokens)}")
            weights = self._check_projector(projector)
            tokens = tokens.to(device=self.device).contiguous()
            if tokens.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                tokens = tokens.float()
            ctx = self._context()
            hid = self._bind_projector(ctx, weights)
            rows = tokens.numel() // tokens.shape[-1]
            self._grow(ctx, -(-rows // ctx.num_tokens))
            out = torch.empty(tuple(tokens.shape[:-1]) + (hid,), device=tokens.device, dtype=tokens.dtype)
            ctx.project(tokens, out)
            return out

    # ---- encode_images = tower -> projector in one library call (llava_arch.py:141-144) -------------
    def encode_images_with_projector(self, images: torch.Tensor, projector: nn.Module) -> torch.Tensor:
        with torch.no_grad():
            images = self._check_images(images)
            weights = self._check_projector(projector)
            ctx = self._context()
            self._auto_audit(images)
            hid = self._bind_projector(ctx, weights)
            self._grow(ctx, images.shape[0])
            out = torch.empty((images.shape[0], ctx.num_tokens, hid), device=images.device, dtype=images.dtype)
            ctx.encode_images(images, out)
            self._after_encode(ctx, lambda: ctx.encode_images(images, out))
            return out

    # ---- properties (mobileclip_encoder.py:90-116) ---------------------------------------------------
    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @property
    def device(self):
        return next(self.vision_tower.parameters()).device

    @property
    def config(self):
        return self.cfg_only

    @property
    def hidden_size(self):
        return self.config["image_cfg"]["embed_dim"]

    @property
    def num_patches_per_side(self):
        return self.config["image_cfg"]["image_size"] // self.config["image_cfg"]["patch_size"]

    @property
    def num_patches(self):
        return (self.config["image_cfg"]["image_size"] // self.config["image_cfg"]["patch_size"]) ** 2
