"""The REFERENCE ITSELF on the GPU box, beside our tower (VERDICT r2 "missing" #3 / next-round item 6).

`/root/reference` does not exist on the GPU box; `__graft_entry__.build()` stages the reference's own `llava` package, unmodified,
as `oracle/_ref/reference_llava.zip` (git-ignored, shipped with the tree like libfvhd.so; `oracle/stage_reference.py`), and
`oracle/ref_import.py` imports it from a scratch directory.  Two checks:

 (a) the reference's `MobileCLIPVisionTower` executed by PyTorch-ROCm in fp32 (1024x1024, B = 4) against our tower on the same
     weights and images - the first direct GPU-vs-reference comparison.  STATED TOLERANCE (mild weight set):
     rel-L2 <= 1e-2, cosine >= 0.9999, max-abs <= 3e-2 * absmax.
 (b) the reference's `LlavaQwen2ForCausalLM.generate(images=...)` (`predict.py:55-65` -> `llava_qwen.py:118-134` ->
     `llava_arch.py:146-332`), once un-patched (reference tower + reference splice walk, fp32 on the GPU), once after
     `install_into_llava(splice=True)` (our tower through the C ABI + the HIP splice kernel) with the SAME state dict loaded
     strictly: first-token logits agree to rel-L2 <= 3e-2 (bf16 tower arithmetic under an fp32 LLM) and the greedy token is
     the same wherever the reference's own top-2 margin exceeds that error; and once more with `prefill=True` (the language
     model's prefill on the hand-written Qwen2 kernels, the stock decode loop continuing from our KV cache): rel-L2 <= 4e-2.
MIOpen is switched off for the reference runs (`torch.backends.cudnn.flags(enabled=False)`: ATen's native HIP convolutions) so a
fresh box does not spend minutes compiling MIOpen kernels; the arithmetic is fp32 either way.
"""
import sys
from types import SimpleNamespace

import pytest
import torch

import ml_fastvlm_amd as fv
from ml_fastvlm_amd import synth
from oracle import ref_import

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_import.reference_available(),
                                 reason="reference not staged (run __graft_entry__.build() where /root/reference is mounted)")]
DEV = "cuda:0"
ARGS = SimpleNamespace(unfreeze_mm_vision_tower=False)


def _metrics(got, want):
    got, want = got.double().cpu().flatten(), want.double().cpu().flatten()
    rel = ((got - want).norm() / want.norm()).item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=0).item()
    mx = ((got - want).abs().max() / want.abs().max()).item()
    return rel, cos, mx


@pytest.mark.parametrize("res,batch", [(256, 2), (1024, 4)])
def test_our_tower_vs_the_reference_tower_on_pytorch_rocm(res, batch):
    sd = synth.synthetic_state_dict(1234, "mild")
    ref = ref_import.build_reference_tower(res)
    ref.vision_tower.model.load_state_dict(sd, strict=True)
    ref = ref.to(DEV, torch.float32)
    x = synth.synthetic_images(batch, res, seed=41).to(DEV)
    with torch.backends.cudnn.flags(enabled=False), torch.no_grad():
        want = ref(x)                                            # mobileclip_encoder.py:70-88 on PyTorch-ROCm, fp32
    ours = fv.MobileCLIPVisionTower(f"mobileclip_l_{res}", ARGS)
    ours.vision_tower.model.load_state_dict(sd, strict=True)
    ours = ours.to(DEV, torch.bfloat16)
    got = ours(x)                                                # fp32 images in -> fp32 tokens out, bf16 arithmetic inside
    assert got.shape == want.shape == (batch, (res // 64) ** 2, 3072) and got.dtype == want.dtype == torch.float32
    rel, cos, mx = _metrics(got, want)
    print(f"ours vs reference-on-GPU (fp32) r{res} B={batch}: rel-L2 {rel:.3e} cos {cos:.6f} max-abs/absmax {mx:.3e}")
    assert rel <= 1e-2 and cos >= 0.9999 and mx <= 3e-2, (rel, cos, mx)
    # the reference's own properties our tower mirrors
    assert ours.hidden_size == ref.hidden_size and ours.num_patches == ref.num_patches
    assert set(ours.state_dict()) == set(ref.state_dict())


def _llava_cfg(hidden=128):
    from transformers import Qwen2Config
    from llava.model.language_model.llava_qwen import LlavaConfig
    cfg = LlavaConfig(**Qwen2Config(vocab_size=1024, hidden_size=hidden, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                    num_key_value_heads=1, max_position_embeddings=2048).to_dict())        # head_dim 64, as Qwen2-0.5B
    cfg.mm_vision_tower, cfg.mm_projector_type, cfg.mm_hidden_size = "mobileclip_l_256", "mlp2x_gelu", 3072
    cfg.unfreeze_mm_vision_tower = True          # llava_arch.py:35 builds with delay_load=True; this forces the load
    cfg.tokenizer_padding_side, cfg.tokenizer_model_max_length = "right", 2048
    return cfg


def test_reference_generate_with_our_tower_underneath():
    ref_import.install_timm_stub()
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import llava.model.llava_arch as arch
    import llava.model.multimodal_encoder.builder as enc_builder
    from llava.model.language_model.llava_qwen import LlavaQwen2ForCausalLM
    import llava.model.language_model.llava_qwen as lq
    saved = (enc_builder.build_vision_tower, arch.build_vision_tower, arch.LlavaMetaForCausalLM.encode_images,
             arch.LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal, lq.LlavaQwen2ForCausalLM.forward)
    hidden = 128
    tower_sd = synth.synthetic_state_dict(1234, "mild")
    proj_sd = synth.synthetic_projector_state_dict(hidden, 1234)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 1024, (3, 14), generator=g)
    ids[:, 5] = -200                                             # IMAGE_TOKEN_INDEX (llava/constants.py:8)
    mask = torch.ones_like(ids)
    mask[1, 11:] = 0                                             # a right-padded sample
    images = synth.synthetic_images(3, 256, seed=17)

    def run(model):
        model = model.to(DEV).eval()
        with torch.inference_mode(), torch.backends.cudnn.flags(enabled=False):
            out = model.generate(ids.to(DEV), images=images.to(DEV), image_sizes=[(256, 256)] * 3, attention_mask=mask.to(DEV),
                                 do_sample=False, max_new_tokens=2, use_cache=True, output_scores=True, return_dict_in_generate=True,
                                 pad_token_id=0)
        return out.sequences.cpu(), out.scores[0].float().cpu(), out.scores[1].float().cpu()

    try:
        torch.manual_seed(0)
        ref_model = LlavaQwen2ForCausalLM(_llava_cfg(hidden))                       # UN-patched: the reference's own tower
        assert type(ref_model.get_vision_tower()).__module__.startswith("llava.")
        ref_model.get_vision_tower().vision_tower.model.load_state_dict(tower_sd, strict=True)
        ref_model.get_model().mm_projector.load_state_dict(proj_sd, strict=True)
        state = {k: v.clone() for k, v in ref_model.state_dict().items()}
        seq_ref, logits_ref, logits2_ref = run(ref_model)
        del ref_model

        fv.install_into_llava(splice=True)                                           # the whole reference-side patch
        ours_model = LlavaQwen2ForCausalLM(_llava_cfg(hidden))
        tower = ours_model.get_vision_tower()
        assert isinstance(tower, fv.MobileCLIPVisionTower)
        missing, unexpected = ours_model.load_state_dict(state, strict=True)         # the reference model's checkpoint, key for key
        assert not missing and not unexpected
        ours_model.get_model().mm_projector.requires_grad_(False)
        ctx_probe = {}
        real = tower.encode_images_with_projector

        def spy(images_, projector):
            ctx_probe["n"] = ctx_probe.get("n", 0) + 1
            return real(images_, projector)
        tower.encode_images_with_projector = spy
        seq, logits, logits2 = run(ours_model)
        assert ctx_probe.get("n", 0) == 1, "generate() must reach the fused library call exactly once (prefill)"

        # ... and with the PREFILL of the language model on the hand-written Qwen2 kernels too: first-token logits from
        # fvhd_llm_prefill, second-token logits from the stock decode step running on OUR KV cache
        # (this model is fp32: the bf16 prefill kernels are an explicit opt-in for it - by default only a bf16 model takes them)
        fv.install_into_llava(splice=True, prefill=True)
        assert getattr(lq.LlavaQwen2ForCausalLM.forward, "_fvhd_prefill", False)
        run(ours_model)
        assert getattr(ours_model, "_fvhd_prefill_ctx", None) is None, "an fp32 model must keep the reference's forward unless opted in"
        fv.install_into_llava(splice=True, prefill=True, prefill_any_dtype=True)
        seq_p, logits_p, logits2_p = run(ours_model)
        assert getattr(ours_model, "_fvhd_prefill_ctx", None) is not None, "generate() did not reach the Qwen2 prefill kernels"

        # ---- the DEFAULT patch on a bf16 model (what predict.py-style inference loads): generate() takes the kernels, every other
        # caller keeps the reference's forward (advisor, round 3) ----
        fv.install_into_llava(splice=True, prefill=True)
        bf = ours_model.to(torch.bfloat16)
        object.__setattr__(bf, "_fvhd_prefill_ctx", None)
        with torch.inference_mode():
            emb = torch.randn(2, 9, hidden, device=DEV, dtype=torch.bfloat16)
            scored = bf(inputs_embeds=emb)                                           # a scoring forward: no cache object, all positions wanted
            assert scored.logits.shape == (2, 9, 1024), "a plain forward must keep its [B, T, vocab] logits"
            assert getattr(bf, "_fvhd_prefill_ctx", None) is None, "a plain forward must not take the last-position-only kernel path"

        def gen(model, ids_, mask_, n=4):
            with torch.inference_mode(), torch.backends.cudnn.flags(enabled=False):
                out = model.generate(ids_.to(DEV), images=images.to(DEV, torch.bfloat16), image_sizes=[(256, 256)] * 3, attention_mask=mask_.to(DEV),
                                     do_sample=False, max_new_tokens=n, use_cache=True, output_scores=True, return_dict_in_generate=True, pad_token_id=0)
            return out.sequences.cpu(), [s.float().cpu() for s in out.scores]
        lmask = torch.ones_like(ids)
        lmask[1, :3] = 0                                                             # a LEFT-padded sample (padding_side of generation)
        for which, m_ in (("right-padded", mask), ("left-padded", lmask)):
            bf.config.tokenizer_padding_side = "left" if which == "left-padded" else "right"
            seq_k, sc_k = gen(bf, ids, m_)
            assert getattr(bf, "_fvhd_prefill_ctx", None) is not None, "generate() on a bf16 model must reach the Qwen2 prefill kernels"
            lq.LlavaQwen2ForCausalLM.forward = lq.LlavaQwen2ForCausalLM.forward._fvhd_orig
            seq_s, sc_s = gen(bf, ids, m_)                                           # same bf16 model, stock prefill
            fv.install_into_llava(splice=True, prefill=True)
            rel, cos, _ = _metrics(sc_k[0], sc_s[0])
            print(f"bf16 model, {which}: first-token logits kernels vs stock bf16 prefill rel-L2 {rel:.3e} cos {cos:.6f}; tokens {seq_k.tolist()} vs {seq_s.tolist()}")
            assert rel <= 4e-2 and cos >= 0.999, (which, rel, cos)
            for b in range(3):                                                       # greedy tokens agree while every margin exceeds the error
                for t in range(len(sc_s)):
                    top2 = sc_s[t][b].topk(2).values
                    if (top2[0] - top2[1]) <= 2 * (sc_k[t][b] - sc_s[t][b]).abs().max():
                        break
                    assert seq_k[b, t] == seq_s[b, t], (which, b, t)
        bf.config.tokenizer_padding_side = "right"
    finally:
        (enc_builder.build_vision_tower, arch.build_vision_tower, arch.LlavaMetaForCausalLM.encode_images,
         arch.LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal, lq.LlavaQwen2ForCausalLM.forward) = saved
    for name, a, b in (("first token, prefill kernels", logits_p, logits_ref), ("second token (stock decode on our KV cache)", logits2_p, logits2_ref)):
        rel, cos, mx = _metrics(a, b)
        print(f"{name}: logits vs the un-patched reference model rel-L2 {rel:.3e} cos {cos:.6f}")
        if name.startswith("first") or torch.equal(seq_p[:, 0], seq_ref[:, 0]):      # the second step is comparable only after the same first token
            assert rel <= 4e-2 and cos >= 0.999, (name, rel, cos)
    rel, cos, mx = _metrics(logits, logits_ref)
    print(f"first-token logits, patched vs un-patched reference model: rel-L2 {rel:.3e} cos {cos:.6f} max-abs/absmax {mx:.3e}")
    assert logits.shape == logits_ref.shape == (3, 1024)
    assert rel <= 3e-2 and cos >= 0.999, (rel, cos, mx)
    top2 = logits_ref.topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    err = (logits - logits_ref).abs().max(dim=-1).values
    for b in range(3):
        if margin[b] > 2 * err[b]:
            assert seq[b, 0] == seq_ref[b, 0], f"sample {b}: greedy first token differs although the reference's margin exceeds the error"


def test_drop_in_defaults_are_range_safe_on_a_saturating_checkpoint():
    """a checkpoint whose ConvFFN block leaves the half-precision range, the reference's model class, ONLY
    `install_into_llava()` + `model.encode_images(images)` - no precision option, no audit call - against the reference's own modules in
    fp32 on PyTorch-ROCm.  STATED TOLERANCE: the tower tolerance of SURVEY.md 8c on the projected tokens, rel-L2 <= 1.5e-2, cosine >= 0.9998
    (the scaled block is a harsher network than the mild profile: its first call must ALREADY be inside it)."""
    ref_import.install_timm_stub()
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import llava.model.llava_arch as arch
    import llava.model.multimodal_encoder.builder as enc_builder
    from llava.model.language_model.llava_qwen import LlavaQwen2ForCausalLM
    saved = (enc_builder.build_vision_tower, arch.build_vision_tower, arch.LlavaMetaForCausalLM.encode_images)
    hidden = 128
    # stage 1, block 3: fc1.weight x 2^27, layer scale x 2^-27 (tests/test_gpu_ffn_precision.py: hidden pre-activations up to ~5e7 on these
    # images, the typical unit an order of magnitude beyond the half form's 262 016)
    tower_sd = synth.synthetic_state_dict(1234, "mild")
    tower_sd["network.2.3.convffn.fc1.weight"] *= 2.0 ** 27
    tower_sd["network.2.3.layer_scale"] *= 2.0 ** -20          # (2^7 more than the inverse: this one block then carries about half of the stream)
    proj_sd = synth.synthetic_projector_state_dict(hidden, 1234)
    images = synth.synthetic_images(2, 256, seed=5).to(DEV)
    try:
        torch.manual_seed(0)
        ref_model = LlavaQwen2ForCausalLM(_llava_cfg(hidden))                       # un-patched: the reference's own tower and projector
        ref_model.get_vision_tower().vision_tower.model.load_state_dict(tower_sd, strict=True)
        ref_model.get_model().mm_projector.load_state_dict(proj_sd, strict=True)
        state = {k: v.clone() for k, v in ref_model.state_dict().items()}
        ref_model = ref_model.to(DEV).eval()
        with torch.inference_mode(), torch.backends.cudnn.flags(enabled=False):
            want = ref_model.encode_images(images).float().cpu()                     # llava_arch.py:141-144, fp32
        del ref_model

        fv.install_into_llava()                                                      # the whole patch; nothing else is configured
        cfg = _llava_cfg(hidden)
        cfg.mm_vision_batch_invariant = True     # (a 256-px, 2-image batch is below the fused kernels' fill rule: make them run - not a safety option)
        model = LlavaQwen2ForCausalLM(cfg)
        assert isinstance(model.get_vision_tower(), fv.MobileCLIPVisionTower)
        missing, unexpected = model.load_state_dict(state, strict=True)
        assert not missing and not unexpected
        model = model.to(DEV).eval()
        import warnings
        with torch.inference_mode(), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = model.encode_images(images).float().cpu()                          # FIRST call
        rel, cos, mx = _metrics(got, want)
        print(f"saturating checkpoint through install_into_llava() defaults: first encode_images rel-L2 {rel:.3e} cos {cos:.6f}")
        assert rel <= 1.5e-2 and cos >= 0.9998, (rel, cos, mx)
        tower = model.get_vision_tower()
        assert tower.ffn_precision == "auto" and tower.range_guard == "on"
        assert any("ConvFFN block" in str(w.message) for w in caught), "the switch is reported"
        with torch.inference_mode():
            assert torch.equal(model.encode_images(images).float().cpu(), got)
    finally:
        enc_builder.build_vision_tower, arch.build_vision_tower, arch.LlavaMetaForCausalLM.encode_images = saved
