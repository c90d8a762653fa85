"""Checkpoint ingest on the GPU (VERDICT r3 "missing" #3): a TRAINING-mode (multi-branch) FastViTHD state dict -
built from the reference's own classes out of the archive `oracle/stage_reference.py` ships to the GPU box - goes through
`ml_fastvlm_amd.reparam.load_training_checkpoint` into the HIP tower, and

 (a) the tower gives the SAME BITS as a tower loaded from the state dict the reference's own `reparameterize()` methods produce
     (`mci.py:219-330, 453-515, 819-859, 1000-1039`; ml-fastvit's `reparameterize_model` loop), and
 (b) agrees with the reference's TRAINING graph itself (every branch and BatchNorm still separate) executed by PyTorch-ROCm in fp32.
     STATED TOLERANCE for (b): rel-L2 <= 5.5e-2, cosine >= 0.998 - the whole-tower budget of tests/test_gpu_tower.py for weights that are
     not the well-conditioned "mild" profile (the reference's own bf16 execution sits at 4.4e-2 on such weights: DESIGN.md section 2).
"""
from types import SimpleNamespace

import pytest
import torch

import ml_fastvlm_amd as fv
from ml_fastvlm_amd import reparam, synth
from oracle import ref_import

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(),
                                 reason="reference not staged (run __graft_entry__.build() where /root/reference is mounted)")]
DEV = "cuda:0"
ARGS = SimpleNamespace(unfreeze_mm_vision_tower=False)


def _tower():
    return fv.MobileCLIPVisionTower("mobileclip_l_256", ARGS)


def test_training_checkpoint_through_load_training_checkpoint_matches_the_references_reparameterize_bit_for_bit():
    from test_reparam import _reference_reparameterize, _training_model
    train = _training_model()                                    # the reference's FastViT(inference_mode=False), fastvithd() hyper-parameters
    sd_train = {k: v.clone() for k, v in train.state_dict().items()}
    assert reparam.is_training_state_dict(sd_train) and any(".rbr_conv." in k for k in sd_train) and any(".lkb_origin." in k for k in sd_train)

    ours = _tower()
    missing, unexpected = reparam.load_training_checkpoint(ours, sd_train, strict=True)
    assert not missing and not unexpected
    ours = ours.to(DEV, torch.bfloat16)

    want_sd = {k: v for k, v in _reference_reparameterize(train).state_dict().items() if not k.startswith("head.")}
    theirs = _tower()
    want_sd["head.proj"] = ours.vision_tower.model.state_dict()["head.proj"].float().cpu()
    theirs.vision_tower.model.load_state_dict(want_sd, strict=True)
    theirs = theirs.to(DEV, torch.bfloat16)

    x = synth.synthetic_images(3, 256, seed=23).to(DEV)
    got, want = ours(x), theirs(x)
    assert got.shape == (3, 16, 3072) and torch.isfinite(got).all()
    assert torch.equal(got, want), "our re-parameterisation and the reference's reparameterize() must give the same tower, bit for bit"

    # (b) the reference's training graph itself, fp32 on PyTorch-ROCm: forward() up to conv_exp = the image embeddings MCi returns
    train = train.to(DEV).float().eval()
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        emb = train.conv_exp(train.forward_tokens(train.forward_embeddings(x)))          # mci.py:1436-1442
    ref = emb.flatten(2).transpose(1, 2)                                                  # feature_select, mobileclip_encoder.py:60-68
    a, b = got.double().cpu().flatten(), ref.double().cpu().flatten()
    rel = ((a - b).norm() / b.norm()).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    print(f"HIP tower from a training checkpoint vs the reference's training graph (fp32, PyTorch-ROCm): rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel <= 5.5e-2 and cos >= 0.998, (rel, cos)
