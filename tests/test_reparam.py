"""ml_fastvlm_amd.reparam against the reference's OWN re-parameterisation: the training-mode FastViTHD graph
is built from the reference's classes (mci.py:1305-1425 with the fastvithd() hyper-parameters of mci.py:1455-1474 and
inference_mode=False), every BatchNorm gets non-trivial statistics, the reference's `reparameterize()` methods produce its
inference-mode state dict, and ours must agree tensor by tensor - and load into our tower's key set (629 tensors)."""
import copy
from functools import partial

import pytest
import torch

from ml_fastvlm_amd import fastvithd_spec as spec
from ml_fastvlm_amd import reparam
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")


def _training_model():
    ref = ref_import.import_reference()
    mci = ref.mci
    torch.manual_seed(3)
    model = mci.FastViT(
        [2, 12, 24, 4, 2], token_mixers=("repmixer", "repmixer", "repmixer", "attention", "attention"),
        embed_dims=[96, 192, 384, 768, 1536], pos_embs=[None, None, None, partial(mci.RepCPE, spatial_shape=(7, 7)),
                                                        partial(mci.RepCPE, spatial_shape=(7, 7))],
        mlp_ratios=[4, 4, 4, 4, 4], downsamples=[True] * 5, norm_layer=mci.LayerNormChannel, stem_scale_branch=False,
        inference_mode=False)
    g = torch.Generator().manual_seed(5)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.4 + 0.8)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    for n, p in model.named_parameters():
        if n.endswith("layer_scale") or "layer_scale_" in n:
            p.data.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.1)
    return model.eval()


def _reference_reparameterize(model):
    model = copy.deepcopy(model)
    for module in model.modules():             # ml-fastvit's reparameterize_model loop: every module that knows how
        if hasattr(module, "reparameterize"):
            module.reparameterize()
    return model


def test_reparameterize_matches_the_reference_tensor_by_tensor():
    train = _training_model()
    sd_train = {k: v.clone() for k, v in train.state_dict().items()}
    assert reparam.is_training_state_dict(sd_train)
    want = _reference_reparameterize(train).state_dict()
    got = reparam.reparameterize_state_dict(sd_train)
    # the reference's classifier head (nn.Linear `head`) is replaced by MCi with GlobalPool2D (`head.proj`): not part of this check
    wk = [k for k in want if not k.startswith("head.")]
    gk = [k for k in got if not k.startswith("head.")]
    assert sorted(gk) == sorted(wk), (sorted(set(gk) ^ set(wk))[:10])
    for k in wk:
        assert got[k].shape == want[k].shape, k
        assert torch.allclose(got[k].float(), want[k].float(), rtol=1e-5, atol=1e-6), (k, (got[k].float() - want[k].float()).abs().max())
    assert not reparam.is_training_state_dict(got)
    # and it is exactly the key set our tower loads (plus head.proj, which MCi adds)
    ours = set(spec.param_spec()) - {"head.proj"}
    assert set(gk) == ours


def test_reparameterized_weights_reproduce_the_training_graph_forward():
    train = _training_model()
    x = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = train.forward_tokens(train.forward_embeddings(x))
    infer = _reference_reparameterize(train)
    sd = reparam.reparameterize_state_dict({k: v.clone() for k, v in train.state_dict().items()})
    missing, unexpected = infer.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    assert not [k for k in missing if not k.startswith("head.")] and not [k for k in unexpected if not k.startswith("head.")]
    with torch.no_grad():
        got = infer.forward_tokens(infer.forward_embeddings(x))
    assert torch.allclose(got, want, rtol=1e-3, atol=1e-4), (got - want).abs().max()


def test_inference_state_dict_passes_through_unchanged():
    from ml_fastvlm_amd import synth
    sd = synth.synthetic_state_dict(3)
    assert not reparam.is_training_state_dict(sd)
    out = reparam.reparameterize_state_dict(sd)
    assert list(out) == list(sd) and all(out[k] is sd[k] for k in sd)


def test_load_training_checkpoint_into_our_tower_strict():
    """`reparam.load_training_checkpoint` (the ingest entry point): a training-mode checkpoint of a bare FastViT - classifier head and
    all - loads STRICTLY into our tower's parameter set, every tensor equal to the reference's own reparameterize() output."""
    from types import SimpleNamespace
    import ml_fastvlm_amd as fv
    train = _training_model()
    sd_train = {k: v.clone() for k, v in train.state_dict().items()}
    tower = fv.MobileCLIPVisionTower("mobileclip_l_256", SimpleNamespace(unfreeze_mm_vision_tower=False))
    before = tower.vision_tower.model.state_dict()["head.proj"].clone()
    missing, unexpected = reparam.load_training_checkpoint(tower, sd_train, strict=True)
    assert not missing and not unexpected
    got = tower.vision_tower.model.state_dict()
    want = _reference_reparameterize(train).state_dict()
    for k, v in want.items():
        if not k.startswith("head."):
            assert torch.equal(got[k].float(), v.float()), k
    assert torch.equal(got["head.proj"], before)            # dead on this path: kept
    assert "head.weight" in sd_train                         # (the caller's dict is not modified)
