"""Embedding splice: oracle restatement pinned against the reference METHOD (live, when mounted); the product's
plan (batched integer ops) checked against the oracle on CPU through a torch gather; the HIP kernel against the oracle on the GPU."""
import random
from types import SimpleNamespace

import pytest
import torch

from ml_fastvlm_amd import splice as S
from oracle import ref_import
from oracle import splice_oracle as O


def _case(seed, B=4, L=24, H=64, V=97, T=(16,), with_mask=True, with_labels=True, empty_sample=False):
    g = torch.Generator().manual_seed(seed)
    rnd = random.Random(seed)
    ids = torch.randint(0, V, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    n_feats = 0
    for b in range(B):
        n_img = rnd.choice([0, 1, 1, 2, 3])
        for p in rnd.sample(range(L), n_img):
            ids[b, p] = S.IMAGE_TOKEN_INDEX
        if with_mask:
            pad = rnd.randrange(0, L // 2)
            if rnd.random() < 0.5:
                mask[b, L - pad:] = 0           # right padding ...
            else:
                mask[b, :pad] = 0               # ... or left padding (drops whatever was there, image tokens included)
        if empty_sample and b == 1:
            mask[b] = 0
        kept = ids[b][mask[b].bool()]
        n_feats += max(1, int((kept == S.IMAGE_TOKEN_INDEX).sum()))
    feats = [torch.randn(rnd.choice(T), H, generator=g) for _ in range(n_feats + 1)]     # one spare entry
    labels = torch.randint(0, V, (B, L), generator=g) if with_labels else None
    W = torch.randn(V, H, generator=g)
    return ids, (mask if with_mask else None), labels, feats, W


def _run_plan_on_cpu(ids, mask, labels, feats, W, side, max_length):
    """the product's index plan + a torch gather standing in for the HIP kernel (same row semantics as csrc/splice.hip)"""
    flat, lens = S.flatten_features(feats)
    start, seqlen, row0, keep, max_len = S.splice_plan(ids, mask, lens, max_length)
    B, L = ids.shape
    out = torch.zeros(B, max_len, W.shape[1])
    am = torch.zeros(B, max_len, dtype=torch.bool)
    pos = torch.zeros(B, max_len, dtype=torch.long)
    lab = torch.full((B, max_len), S.IGNORE_INDEX, dtype=torch.long)
    for b in range(B):
        n = int(seqlen[b])
        shift = max_len - n if side == "left" else 0
        for u in range(n):
            j = int((start[b] <= u).nonzero().max())          # last position whose start <= u: what the kernel's binary search finds
            t = u + shift
            if row0[b, j] >= 0:
                out[b, t] = flat[int(row0[b, j]) + u - int(start[b, j])]
            else:
                out[b, t] = W[ids[b, j]]
                if labels is not None:
                    lab[b, t] = labels[b, j]
            am[b, t], pos[b, t] = True, u
    return out, am, pos, lab, max_len


@pytest.mark.parametrize("seed", range(12))
def test_plan_matches_oracle_on_cpu(seed):
    rnd = random.Random(100 + seed)
    side = rnd.choice(["right", "left"])
    max_length = rnd.choice([None, None, 20, 33])
    ids, mask, labels, feats, W = _case(seed, T=rnd.choice([(16,), (5, 9, 16)]), with_mask=seed % 3 != 0, with_labels=seed % 2 == 0,
                                        empty_sample=seed == 7)
    want = O.splice(ids, None, mask, labels, feats, W, side, max_length)
    out, am, pos, lab, max_len = _run_plan_on_cpu(ids, mask, labels, feats, W, side, max_length)
    assert out.shape == want[4].shape and torch.equal(out, want[4])
    full = O.splice(ids, torch.zeros(1, dtype=torch.long), torch.ones_like(ids) if mask is None else mask,
                    labels if labels is not None else torch.full_like(ids, 5), feats, W, side, max_length)
    assert torch.equal(am, full[2].bool()) and torch.equal(pos, full[1])
    if labels is not None:
        assert torch.equal(lab, want[5])


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")
@pytest.mark.parametrize("seed", range(10))
def test_oracle_matches_the_reference_method(seed):
    ref_import.import_reference()
    from llava.model.llava_arch import LlavaMetaForCausalLM
    rnd = random.Random(seed)
    side = rnd.choice(["right", "left"])
    max_length = rnd.choice([None, 18, 40])
    ids, mask, labels, feats, W = _case(seed, T=(16,) if seed % 2 else (5, 9, 16), with_mask=seed % 3 != 0, with_labels=seed % 2 == 0)
    emb = torch.nn.Embedding.from_pretrained(W)
    same_T = len({f.shape[0] for f in feats}) == 1

    class Fake:
        config = SimpleNamespace(tokenizer_padding_side=side, tokenizer_model_max_length=max_length, mm_patch_merge_type="flat")
        device = torch.device("cpu")

        def get_vision_tower(self):
            return object()

        def get_model(self):
            return SimpleNamespace(embed_tokens=emb)

        def encode_images(self, images):      # `images` is only a carrier here: the features are precomputed
            return torch.stack(feats, 0) if same_T else torch.cat(feats, 0)

    fake = Fake()
    if same_T:
        images = torch.zeros(len(feats), 3, 2, 2)                         # tensor path (llava_arch.py:209-210)
    else:
        images = [torch.zeros(f.shape[0], 3, 2, 2) for f in feats]        # list path: split sizes = rows per image, 'flat' merge (:154-164)
        Fake.encode_images = lambda self, cat: torch.cat(feats, 0).reshape(-1, 1, W.shape[1])
    pos_in = None if seed % 4 else torch.arange(ids.shape[1])
    got_ref = LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal(fake, ids, pos_in, mask, None, labels, images)
    want = O.splice(ids, pos_in, mask, labels, feats, W, side, max_length)
    for a, b, name in zip(got_ref, want, ("input_ids", "position_ids", "attention_mask", "past", "embeds", "labels")):
        assert (a is None) == (b is None), name
        if a is not None:
            assert a.dtype == b.dtype and torch.equal(a, b), name


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("seed", [0, 3, 7])
def test_hip_splice_matches_oracle(seed, dtype):
    side = "left" if seed % 2 else "right"
    max_length = None if seed != 3 else 40
    ids, mask, labels, feats, W = _case(seed, B=5, L=40, H=896, V=1000, T=(16,) if seed else (7, 16, 33), empty_sample=seed == 7)
    W, feats = W.to(dtype), [f.to(dtype) for f in feats]
    want = O.splice(ids, torch.arange(ids.shape[1]), mask, labels, feats, W, side, max_length)
    got = S.multimodal_splice(ids.cuda(), torch.arange(ids.shape[1]).cuda(), mask.cuda(), labels.cuda(), [f.cuda() for f in feats],
                              W.cuda(), side, max_length)
    torch.cuda.synchronize()
    assert got[0] is None and got[3] is None
    for a, b, name in ((got[4], want[4], "embeds"), (got[1], want[1], "position_ids"), (got[2], want[2], "attention_mask"), (got[5], want[5], "labels")):
        assert a.dtype == b.dtype and torch.equal(a.cpu(), b), name
    more = feats * 3        # without the mask every -200 is live: more feature entries are consumed
    none = S.multimodal_splice(ids.cuda(), None, None, None, torch.stack(more, 0).cuda() if seed else [f.cuda() for f in more], W.cuda(), side)
    assert none[1] is None and none[2] is None and none[5] is None


@pytest.mark.gpu
def test_hip_splice_prefill_shape_b8():
    """BASELINE.json configs[2] shape: 8 sequences of ~30 text tokens + one 256-token image each, H = 896, bf16."""
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 151936, (8, 30), generator=g)
    ids[:, 14] = S.IMAGE_TOKEN_INDEX
    feats = torch.randn(8, 256, 896, generator=g).to(torch.bfloat16)
    W = torch.randn(151936, 896, generator=g).to(torch.bfloat16)
    want = O.splice(ids, None, None, None, [f for f in feats], W)
    got = S.multimodal_splice(ids.cuda(), None, None, None, feats.cuda(), W.cuda())
    assert got[4].shape == (8, 285, 896) and torch.equal(got[4].cpu(), want[4])


@pytest.mark.parametrize("seed", range(8))
def test_sources_and_backward_match_autograd_of_the_oracle_walk(seed):
    """the splice must carry gradients to embed_tokens.weight and to the image features (training contract,
    llava_arch.py:257-283 builds inputs_embeds with differentiable ops).  `splice_sources` (the kernel's row -> source map, in torch)
    reproduces the forward bit for bit, and `splice_backward` equals autograd through the oracle's per-sample walk."""
    rnd = random.Random(200 + seed)
    side = rnd.choice(["right", "left"])
    max_length = rnd.choice([None, 20, 33])
    ids, mask, labels, feats, W = _case(seed, T=rnd.choice([(16,), (5, 9, 16)]), with_mask=seed % 3 != 0, empty_sample=seed == 5)
    Wg = W.clone().requires_grad_(True)
    fg = [f.clone().requires_grad_(True) for f in feats]
    want = O.splice(ids, None, mask, labels, fg, Wg, side, max_length)[4]
    g = torch.randn(want.shape, generator=torch.Generator().manual_seed(seed))
    want.backward(g)
    flat, lens = S.flatten_features(feats)
    start, seqlen, row0, _, max_len = S.splice_plan(ids, mask, lens, max_length)
    tr, fr = S.splice_sources(ids, start, seqlen, row0, max_len, side == "left")
    fwd = torch.zeros_like(want)
    fwd[tr >= 0] = W[tr[tr >= 0]]
    fwd[fr >= 0] = flat[fr[fr >= 0]]
    assert torch.equal(fwd, want.detach())
    gt, gf = S.splice_backward(g, tr, fr, W.shape[0], flat.shape[0])
    assert torch.allclose(gt, Wg.grad, rtol=1e-6, atol=1e-6)
    want_gf = torch.cat([f.grad if f.grad is not None else torch.zeros_like(f) for f in fg], 0)
    assert torch.allclose(gf, want_gf, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_fully_masked_batch_returns_empty_outputs_and_label_dtype():
    """llava_arch.py:297-322 with every position masked out: [B, 0, H] embeddings, [B, 0] mask / positions / labels, no kernel launch;
    labels keep the caller's dtype (the reference builds them with labels.dtype, :299)."""
    ids, _, labels, feats, W = _case(2, B=2, L=6, H=8)
    mask = torch.zeros_like(ids)
    lab32 = labels.to(torch.int32)
    out = S.multimodal_splice(ids.cuda(), torch.arange(6).cuda(), mask.cuda(), lab32.cuda(), [f.cuda() for f in feats], W.cuda())
    want = O.splice(ids, torch.arange(6), mask, lab32, feats, W)
    for a, b in ((out[4], want[4]), (out[2], want[2]), (out[1], want[1]), (out[5], want[5])):
        assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert out[4].shape == (2, 0, 8) and out[5].dtype == torch.int32
    mask[0, :3] = 1
    out = S.multimodal_splice(ids.cuda(), None, mask.cuda(), lab32.cuda(), [f.cuda() for f in feats], W.cuda())
    want = O.splice(ids, None, mask, lab32, feats, W)
    assert out[5].dtype == torch.int32 and torch.equal(out[5].cpu(), want[5]) and torch.equal(out[4].cpu(), want[4])


@pytest.mark.gpu
@pytest.mark.parametrize("side", ["right", "left"])
def test_hip_splice_is_differentiable(side):
    """with gradients enabled and a trainable embedding table / projector output, the HIP splice carries
    d(inputs_embeds) back to both (custom autograd.Function: kernel forward, scatter-add backward), equal to autograd through the
    oracle's walk on CPU; under no_grad / frozen inputs the plain kernel path is taken and the output has no grad_fn."""
    ids, mask, labels, feats, W = _case(11, B=4, L=30, H=64, V=300, T=(7, 16))
    Wc = W.clone().requires_grad_(True)
    fc = [f.clone().requires_grad_(True) for f in feats]
    want = O.splice(ids, None, mask, labels, fc, Wc, side, 50)[4]
    g = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(g)
    Wd = W.cuda().requires_grad_(True)
    fd = [f.cuda().requires_grad_(True) for f in feats]
    got = S.multimodal_splice(ids.cuda(), None, mask.cuda(), labels.cuda(), fd, Wd, side, 50)
    assert got[4].requires_grad and torch.equal(got[4].detach().cpu(), want.detach())
    assert not got[2].requires_grad and not got[5].requires_grad
    got[4].backward(g.cuda())
    assert torch.allclose(Wd.grad.cpu(), Wc.grad, rtol=1e-6, atol=1e-6)
    for a, b in zip(fd, fc):
        assert torch.allclose(a.grad.cpu() if a.grad is not None else torch.zeros_like(b), b.grad if b.grad is not None else torch.zeros_like(b),
                              rtol=1e-6, atol=1e-6)
    # projector-only training: frozen table, trainable features
    Wf = W.cuda()
    fd2 = [f.cuda().requires_grad_(True) for f in feats]
    out = S.multimodal_splice(ids.cuda(), None, mask.cuda(), labels.cuda(), fd2, Wf, side, 50)[4]
    out.sum().backward()
    assert fd2[0].grad is not None
    with torch.no_grad():
        assert S.multimodal_splice(ids.cuda(), None, mask.cuda(), None, fd2, Wf, side, 50)[4].grad_fn is None


def test_no_cpu_path():
    ids, mask, labels, feats, W = _case(0)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        S.multimodal_splice(ids, None, mask, labels, feats, W)


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")
@pytest.mark.parametrize("merge", ["flat", "spatial", "spatial_unpad"])
def test_anyres_feature_merge_matches_the_reference(merge):
    """llava_arch.py:165-206: list-of-tiles images through the reference method vs merge_patch_features + the splice oracle"""
    ref_import.import_reference()
    from llava.model.llava_arch import LlavaMetaForCausalLM
    g = torch.Generator().manual_seed(5)
    TS, side, H, V = 64, 4, 32, 50                        # tile size, token map side, hidden, vocabulary
    grids = [[64, 128], [128, 64], [128, 128], [192, 64]]
    sizes = [(150, 100), (80, 200), (64, 64), (300, 90)]  # (width, height) of the original pictures
    from ml_fastvlm_amd.preprocess import _best_resolution
    n_tiles = [1 + (lambda wh: (wh[0] // TS) * (wh[1] // TS))(_best_resolution(w, h, [tuple(p) for p in grids])) for (w, h) in sizes]
    n_tiles[2] = 1 if merge != "flat" else n_tiles[2]     # one picture comes as a single tile (the no-grid branch)
    feats = [torch.randn(n, side * side, H, generator=g) for n in n_tiles]
    newline = torch.randn(H, generator=g)
    ids = torch.randint(0, V, (len(sizes), 12), generator=g)
    ids[:, 5] = S.IMAGE_TOKEN_INDEX
    W = torch.randn(V, H, generator=g)
    emb = torch.nn.Embedding.from_pretrained(W)

    class Fake:
        config = SimpleNamespace(tokenizer_padding_side="right", tokenizer_model_max_length=None, mm_patch_merge_type=merge,
                                 image_aspect_ratio="anyres", image_grid_pinpoints=str(grids))
        device = torch.device("cpu")
        model = SimpleNamespace(image_newline=newline)

        def get_vision_tower(self):
            return SimpleNamespace(num_patches_per_side=side, config={"image_cfg": {"image_size": TS}})

        def get_model(self):
            return SimpleNamespace(embed_tokens=emb)

        def encode_images(self, cat):
            return torch.cat(feats, 0)

    images = [torch.zeros(n, 3, 2, 2) for n in n_tiles]
    ref = LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal(Fake(), ids, None, None, None, None, images, image_sizes=sizes)
    merged = S.merge_patch_features(feats, sizes, merge, grids, TS, newline)
    want = O.splice(ids, None, None, None, merged, W)
    assert ref[4].shape == want[4].shape and torch.equal(ref[4], want[4])


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")
@pytest.mark.parametrize("case", ["tensor", "anyres_unpad", "no_images", "decode_step"])
def test_prepare_inputs_replacement_routes_like_the_reference(case, monkeypatch):
    """builder.prepare_inputs_labels_for_multimodal (what install_into_llava(splice=True) installs) against the reference method,
    with the GPU kernel call replaced by the pinned CPU oracle: early outs, tensor and list-of-tiles branches, return tuple."""
    ref_import.import_reference()
    from llava.model.llava_arch import LlavaMetaForCausalLM
    from ml_fastvlm_amd import builder as B
    monkeypatch.setattr(S, "multimodal_splice", lambda ids, pos, am, lab, feats, w, side="right", max_length=None:
                        O.splice(ids, pos, am, lab, [f for f in feats] if isinstance(feats, torch.Tensor) else list(feats), w, side, max_length))
    g = torch.Generator().manual_seed(9)
    TS, side, H, V = 64, 4, 32, 50
    grids = [[64, 128], [128, 64], [128, 128]]
    ids = torch.randint(0, V, (2, 10), generator=g)
    ids[:, 4] = S.IMAGE_TOKEN_INDEX
    mask = torch.ones_like(ids)
    mask[1, 8:] = 0
    labels = torch.randint(0, V, (2, 10), generator=g)
    W = torch.randn(V, H, generator=g)
    emb = torch.nn.Embedding.from_pretrained(W)
    newline = torch.randn(H, generator=g)
    sizes = [(150, 100), (60, 64)]
    if case == "anyres_unpad":
        from ml_fastvlm_amd.preprocess import _best_resolution
        bw, bh = _best_resolution(*sizes[0], [tuple(p) for p in grids])
        n0 = 1 + (bw // TS) * (bh // TS)
        feats = [torch.randn(n0, side * side, H, generator=g), torch.randn(1, side * side, H, generator=g)]
        images = [torch.zeros(n0, 3, 2, 2), torch.zeros(3, 2, 2)]
    else:
        feats = [torch.randn(2, side * side, H, generator=g)]
        images = torch.zeros(2, 3, 2, 2)

    class Fake:
        config = SimpleNamespace(tokenizer_padding_side="right", tokenizer_model_max_length=None, image_aspect_ratio="anyres",
                                 mm_patch_merge_type="spatial_unpad" if case == "anyres_unpad" else "flat", image_grid_pinpoints=str(grids))
        device = torch.device("cpu")
        model = SimpleNamespace(image_newline=newline)

        def get_vision_tower(self):
            return SimpleNamespace(num_patches_per_side=side, config={"image_cfg": {"image_size": TS}})

        def get_model(self):
            return SimpleNamespace(embed_tokens=emb)

        def encode_images(self, x):
            return torch.cat(feats, 0)

    if case == "no_images":
        images = None
    if case == "decode_step":
        ids, mask, labels = ids[:, :1], mask[:, :1], labels[:, :1]
    args = (ids, None, mask, "PKV", labels, images)
    want = LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal(Fake(), *args, image_sizes=sizes)
    got = B.prepare_inputs_labels_for_multimodal(Fake(), *args, image_sizes=sizes)
    assert len(got) == len(want) == 6
    for a, b in zip(got, want):
        assert type(a) is type(b)
        if isinstance(a, torch.Tensor):
            assert a.dtype == b.dtype and torch.equal(a, b)
        else:
            assert a == b


@pytest.mark.gpu
def test_prepare_inputs_replacement_on_the_gpu():
    """the installed drop-in end to end on the device (list-of-tiles images, spatial_unpad merge, HIP splice) vs merge + oracle on CPU"""
    from ml_fastvlm_amd import builder as B
    from ml_fastvlm_amd.preprocess import _best_resolution
    g = torch.Generator().manual_seed(4)
    TS, side, H, V = 64, 4, 64, 80
    grids = [[64, 128], [128, 64], [128, 128]]
    sizes = [(150, 100), (60, 64)]
    bw, bh = _best_resolution(*sizes[0], [tuple(p) for p in grids])
    n0 = 1 + (bw // TS) * (bh // TS)
    feats = [torch.randn(n0, side * side, H, generator=g).to(torch.bfloat16), torch.randn(1, side * side, H, generator=g).to(torch.bfloat16)]
    newline = torch.randn(H, generator=g).to(torch.bfloat16)
    W = torch.randn(V, H, generator=g).to(torch.bfloat16)
    ids = torch.randint(0, V, (2, 12), generator=g)
    ids[:, 3] = S.IMAGE_TOKEN_INDEX
    mask = torch.ones_like(ids)
    mask[0, 10:] = 0
    labels = torch.randint(0, V, (2, 12), generator=g)

    class Fake:
        config = SimpleNamespace(tokenizer_padding_side="left", tokenizer_model_max_length=60, image_aspect_ratio="anyres",
                                 mm_patch_merge_type="spatial_unpad", image_grid_pinpoints=str(grids))
        model = SimpleNamespace(image_newline=newline.cuda())

        def get_vision_tower(self):
            return SimpleNamespace(num_patches_per_side=side, config={"image_cfg": {"image_size": TS}})

        def get_model(self):
            return SimpleNamespace(embed_tokens=SimpleNamespace(weight=W.cuda()))

        def encode_images(self, x):
            return torch.cat(feats, 0).cuda()

    images = [torch.zeros(n0, 3, 2, 2).cuda(), torch.zeros(3, 2, 2).cuda()]
    got = B.prepare_inputs_labels_for_multimodal(Fake(), ids.cuda(), None, mask.cuda(), "PKV", labels.cuda(), images, image_sizes=sizes)
    merged = S.merge_patch_features(feats, sizes, "spatial_unpad", grids, TS, newline)
    want = O.splice(ids, None, mask, labels, merged, W, "left", 60)
    assert got[0] is None and got[1] is None and got[3] == "PKV"
    assert torch.equal(got[4].cpu(), want[4]) and torch.equal(got[2].cpu(), want[2]) and torch.equal(got[5].cpu(), want[5])


@pytest.mark.gpu
def test_hip_splice_on_a_non_current_device():
    """fvhd_op_splice has no device guard; the host launches with the tensors' device current and leaves the caller's
    current device alone (one GPU visible: the check is the untouched current device and a correct result)."""
    dev = torch.device("cuda", torch.cuda.device_count() - 1)
    torch.cuda.set_device(0)
    ids, mask, labels, feats, W = _case(4, B=3, L=20, H=64, V=200)
    want = O.splice(ids, None, mask, labels, feats, W, "right", None)
    got = S.multimodal_splice(ids.to(dev), None, mask.to(dev), labels.to(dev), [f.to(dev) for f in feats], W.to(dev))
    assert torch.cuda.current_device() == 0 and got[4].device == dev
    assert torch.equal(got[4].cpu(), want[4]) and torch.equal(got[5].cpu(), want[5])
