#!/usr/bin/env python3
"""Parity on a REAL FastVLM checkpoint (the reference's `get_models.sh:8-13` downloads; none is available in the offline build
environment, where every tolerance is measured on seeded synthetic weights - VERDICT r3 "missing" #7).

    python tools/compare_checkpoint.py /path/to/llava-fastvithd_0.5b_stage3 --reference /path/to/ml-fastvlm [--images a.jpg b.png ...]

What it does, on one MI355X:
  1. loads the checkpoint's vision tower + mm_projector tensors (safetensors / .bin shards; keys `model.vision_tower.vision_tower.model.*`
     and `model.mm_projector.*`) into (a) the reference's own `MobileCLIPVisionTower` + `build_vision_projector` modules, executed by
     PyTorch-ROCm in fp32, and (b) `ml_fastvlm_amd.MobileCLIPVisionTower` (+ the same projector object) in bf16;
  2. preprocesses the given images (or 4 seeded random ones) with the reference's `process_images` ('pad' aspect ratio);
  3. runs the RANGE AUDIT of the half-precision ConvFFN hidden activation (`tower.audit_ranges`) and prints max |fc1 output| per block -
     a block above 65 504 is switched to the bf16-operand kernel and reported;
  4. prints rel-L2 / cosine / max-abs of tower tokens and projected tokens against the fp32 reference and, for scale, of the reference's
     own bf16 execution against its fp32 execution.  Expected: rel-L2 <= 1e-2, cosine >= 0.9999, at or below the
     reference's own bf16 error.  Exit code 1 if ours is more than 1.5x the reference's own bf16 error AND above 1e-2.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_tensors(ckpt_dir: str):
    sd = {}
    files = sorted(glob.glob(os.path.join(ckpt_dir, "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        for f in files:
            sd.update(load_file(f))
    else:
        for f in sorted(glob.glob(os.path.join(ckpt_dir, "pytorch_model*.bin"))):
            sd.update(torch.load(f, map_location="cpu"))
    if not sd:
        raise SystemExit(f"no *.safetensors / pytorch_model*.bin under {ckpt_dir}")
    tower = {k.split("vision_tower.vision_tower.model.", 1)[1]: v.float() for k, v in sd.items() if "vision_tower.vision_tower.model." in k}
    proj = {k.split("mm_projector.", 1)[1]: v.float() for k, v in sd.items() if "mm_projector." in k}
    return tower, proj


def metrics(got, want):
    a, b = got.double().flatten().cpu(), want.double().flatten().cpu()
    return {"rel_l2": ((a - b).norm() / b.norm()).item(), "cos": torch.nn.functional.cosine_similarity(a, b, dim=0).item(),
            "max_abs_over_absmax": ((a - b).abs().max() / b.abs().max()).item()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--reference", default=None, help="checkout of apple/ml-fastvlm (default: FVHD_REFERENCE_ROOT, /root/reference, or the "
                                                       "archive staged under oracle/_ref by __graft_entry__.build())")
    ap.add_argument("--images", nargs="*", default=[])
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    if a.reference:
        os.environ["FVHD_REFERENCE_ROOT"] = a.reference
    from oracle import ref_import                      # test infrastructure: the reference's own modules (timm stub if timm is absent)
    import ml_fastvlm_amd as fv

    tower_sd, proj_sd = load_tensors(a.checkpoint)
    cfg = json.load(open(os.path.join(a.checkpoint, "config.json")))
    name = cfg.get("mm_vision_tower", "mobileclip_l_1024")
    res, hidden = int(name.split("_")[-1]), int(cfg["hidden_size"])
    from ml_fastvlm_amd import reparam
    if reparam.is_training_state_dict(tower_sd):
        print("training-mode tower weights: re-parameterising (ml_fastvlm_amd.reparam)")
        tower_sd = dict(reparam.reparameterize_state_dict(tower_sd))

    ref_tower = ref_import.build_reference_tower(res)
    ref_tower.vision_tower.model.load_state_dict(tower_sd, strict=True)
    ref_proj = ref_import.build_reference_projector(hidden)
    ref_proj.load_state_dict(proj_sd, strict=True)
    ours = fv.MobileCLIPVisionTower(name, SimpleNamespace(unfreeze_mm_vision_tower=False))
    ours.vision_tower.model.load_state_dict(tower_sd, strict=True)
    proj = fv.build_vision_projector(SimpleNamespace(mm_projector_type=cfg.get("mm_projector_type", "mlp2x_gelu"), mm_hidden_size=3072, hidden_size=hidden))
    proj.load_state_dict(proj_sd, strict=True)

    if a.images:
        from PIL import Image
        ref_import.import_reference()                 # puts the reference checkout on sys.path
        from llava.mm_utils import process_images
        x = process_images([Image.open(p).convert("RGB") for p in a.images], ref_tower.image_processor, SimpleNamespace(image_aspect_ratio="pad"))
        x = x if isinstance(x, torch.Tensor) else torch.stack(x)
    else:
        x = torch.rand(4, 3, res, res, generator=torch.Generator().manual_seed(0))
    dev = torch.device(a.device)
    x = x.float().to(dev)
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        rt32, rp32 = ref_tower.to(dev, torch.float32), ref_proj.to(dev, torch.float32)
        want_t = rt32(x)
        want_p = rp32(want_t)
        rtb, rpb = ref_tower.to(dev, torch.bfloat16), ref_proj.to(dev, torch.bfloat16)
        refb_t = rtb(x.bfloat16()).float()
        refb_p = rpb(rtb(x.bfloat16())).float()
    ours, proj = ours.to(dev, torch.bfloat16), proj.to(dev, torch.bfloat16)
    report = ours.audit_ranges(x)
    hot = [r for r in report if r["switched"]]
    print("range audit: max |fc1 output| per ConvFFN block:", ", ".join(f"s{r['stage']}b{r['block']}={r['max_abs_fc1']:.3g}" for r in report))
    print(f"blocks switched to the bf16-operand kernel: {[(r['stage'], r['block']) for r in hot] or 'none'}")
    with torch.no_grad():
        got_t = ours(x)
        got_p = fv.encode_images(ours, proj, x)
    out = {"tower_ours_vs_ref_fp32": metrics(got_t, want_t), "tower_refbf16_vs_ref_fp32": metrics(refb_t, want_t),
           "projected_ours_vs_ref_fp32": metrics(got_p, want_p), "projected_refbf16_vs_ref_fp32": metrics(refb_p, want_p)}
    print(json.dumps(out, indent=1))
    bad = out["tower_ours_vs_ref_fp32"]["rel_l2"] > max(1e-2, 1.5 * out["tower_refbf16_vs_ref_fp32"]["rel_l2"])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
