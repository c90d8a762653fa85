"""Qwen2 prefill.  CPU: the oracle restatement pinned against the installed `transformers` Qwen2 modules (the
third-party code the reference calls at llava_qwen.py:92-103).  GPU: every new kernel against torch fp32 on identical bf16-rounded
operands, the whole prefill against the `transformers` module in fp32, and the KV cache hand-over to its decode loop.

STATED TOLERANCES (bf16 storage, fp32 accumulation against fp32 references):
  single ops: |err| <= 1e-2 |want| + 1e-2 rms(want);
  decoder stack: rel-L2 of the residual stream <= 5e-3 per layer (it is re-rounded to bf16 four times per layer), <= 1.5e-2 after the
  2-layer stacks tested here; logits rel-L2 <= 2e-2, cosine >= 0.9995; greedy token equal wherever the reference's top-2 margin
  exceeds twice the logit error."""
import ctypes as C

import pytest
import torch

from ml_fastvlm_amd import _lib
from oracle import qwen2_oracle as QO

DEV = "cuda:0"


def _cfg(hidden=128, layers=2, heads=2, kv=1, inter=256, vocab=512, theta=1e6, head_dim=None):
    from transformers import Qwen2Config
    cfg = Qwen2Config(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=kv, max_position_embeddings=4096, rms_norm_eps=1e-6, tie_word_embeddings=False)
    for holder in ("rope_parameters", "rope_scaling"):
        d = getattr(cfg, holder, None)
        if isinstance(d, dict):
            d["rope_theta"] = theta
    if hasattr(cfg, "rope_theta") and getattr(cfg, "rope_theta", None) is not None:
        cfg.rope_theta = theta
    cfg._attn_implementation = "eager"
    return cfg


def _model(cfg, seed=0):
    from transformers import Qwen2ForCausalLM
    torch.manual_seed(seed)
    m = Qwen2ForCausalLM(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                      # HF's init is N(0, 0.02) with zero biases and unit norms: give every tensor some life
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif "norm" in n:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p.shape[-1] ** 0.5))
    return m


def _inputs(B, T, H, seed=0, pad="none"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, H, generator=g)
    mask = torch.ones(B, T, dtype=torch.long)
    pos = torch.arange(T)[None].repeat(B, 1)
    if pad != "none":
        for b in range(B):
            n = (3 * b + 1) % max(2, T // 3)
            if n == 0:
                continue
            if pad == "left":
                mask[b, :n] = 0
                pos[b] = torch.clamp(torch.arange(T) - n, min=0)         # as prepare_inputs_labels_for_multimodal builds them (0 on padding)
            else:
                mask[b, T - n:] = 0
                pos[b, T - n:] = 0
    return x, mask, pos


def _metrics(got, want):
    got, want = got.double().cpu().flatten(), want.double().cpu().flatten()
    rel = ((got - want).norm() / want.norm()).item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=0).item()
    return rel, cos


# ------------------------------------------------------------------------------------------------- CPU: pin the oracle
@pytest.mark.parametrize("pad", ["none", "left", "right"])
def test_oracle_equals_transformers_qwen2(pad):
    cfg = _cfg(hidden=128, layers=2, heads=4, kv=2, inter=192, vocab=320)
    m = _model(cfg)
    x, mask, pos = _inputs(3, 19, 128, seed=5, pad=pad)
    with torch.no_grad():
        want = m(inputs_embeds=x, attention_mask=mask, position_ids=pos, use_cache=True)
    logits, hidden, kvs = QO.prefill(x, m.state_dict(), cfg, mask, pos)
    valid = mask.bool()
    assert torch.allclose(logits[valid], want.logits[valid], rtol=1e-4, atol=1e-5), (logits - want.logits)[valid].abs().max()
    pkv = want.past_key_values
    for l in range(cfg.num_hidden_layers):
        try:
            k_ref, v_ref = pkv[l][0], pkv[l][1]
        except Exception:
            k_ref, v_ref = pkv.layers[l].keys, pkv.layers[l].values
        assert torch.allclose(kvs[l][0], k_ref, rtol=1e-4, atol=1e-5) and torch.allclose(kvs[l][1], v_ref, rtol=1e-4, atol=1e-5)


def test_oracle_equals_transformers_qwen2_fp64_and_head_dim_128():
    cfg = _cfg(hidden=256, layers=1, heads=2, kv=1, inter=128, vocab=64)
    m = _model(cfg, seed=3).double()
    x, mask, pos = _inputs(2, 9, 256, seed=8)
    with torch.no_grad():
        want = m(inputs_embeds=x.double(), attention_mask=mask, position_ids=pos).logits
    got, _, _ = QO.prefill(x.double(), m.state_dict(), cfg, mask, pos, dtype=torch.float64)
    assert _metrics(got, want)[0] < 1e-6      # rope angles are fp32 on both sides (Qwen2RotaryEmbedding forces float32)


def test_host_refuses_cpu_and_bad_shapes():
    from ml_fastvlm_amd.qwen2_prefill import Qwen2Prefill, rope_table
    m = _model(_cfg())
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        Qwen2Prefill.from_hf(m)
    t = rope_table(7, 64, 1e6)
    assert t.shape == (7, 32, 2) and torch.allclose(t[:, :, 0] ** 2 + t[:, :, 1] ** 2, torch.ones(7, 32), atol=1e-6)
    cos, sin = QO.rope_cos_sin(torch.arange(7)[None], 64, 1e6)
    assert torch.equal(t[:, :, 0], cos[0, :, :32]) and torch.equal(t[:, :, 1], sin[0, :, :32])


# ------------------------------------------------------------------------------------------------- GPU: single ops
def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def _bf(t):
    return t.to(torch.bfloat16).float()


def _close(got, want, what, rtol=1e-2, atol_rms=1e-2):
    got, want = got.float().cpu(), want.float().cpu()
    assert torch.isfinite(got).all(), what
    tol = rtol * want.abs() + atol_rms * want.pow(2).mean().sqrt()
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} out of tolerance, max err {(got - want).abs().max():.4g}"


@pytest.mark.gpu
@pytest.mark.parametrize("M,H", [(5, 128), (300, 896), (64, 3584)])
def test_rmsnorm(M, H):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + H)
    x = _bf(torch.randn(M, H, generator=g) * 3)
    w = 1.0 + 0.3 * torch.randn(H, generator=g)
    xd, wd = x.to(DEV, torch.bfloat16), w.to(DEV)
    yd = torch.empty_like(xd)
    _lib.check(lib.fvhd_op_rmsnorm(_stream(), _p(xd), _p(yd), _p(wd), M, H, 1e-6), "rmsnorm")
    torch.cuda.synchronize()
    _close(yd, QO.rmsnorm(x, w, 1e-6), f"rmsnorm {M}x{H}")


@pytest.mark.gpu
@pytest.mark.parametrize("hd,nh,nkv", [(64, 14, 2), (128, 4, 2)])
def test_rope_in_place_and_kv_cache(hd, nh, nkv):
    from ml_fastvlm_amd.qwen2_prefill import rope_table
    lib = _lib.load()
    B, T = 3, 37
    g = torch.Generator().manual_seed(hd)
    width = (nh + 2 * nkv) * hd
    qkv = _bf(torch.randn(B * T, width, generator=g))
    pos = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])          # arbitrary positions: the table is indexed, not assumed
    table = rope_table(T, hd, 1e6, DEV)
    qd = qkv.to(DEV, torch.bfloat16)
    kc = torch.zeros(B, nkv, T, hd, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    _lib.check(lib.fvhd_op_rope(_stream(), _p(qd), _p(pos.to(DEV)), _p(table), _p(kc), _p(vc), B * T, T, nh, nkv, hd, T, 1e6), "rope")
    torch.cuda.synchronize()
    x = qkv.view(B, T, nh + 2 * nkv, hd)
    q, k, v = x[:, :, :nh].transpose(1, 2), x[:, :, nh:nh + nkv].transpose(1, 2), x[:, :, nh + nkv:].transpose(1, 2)
    cos, sin = QO.rope_cos_sin(pos, hd, 1e6)
    qr, kr = QO.apply_rope(q, k, cos, sin)
    got = qd.float().cpu().view(B, T, nh + 2 * nkv, hd)
    _close(got[:, :, :nh].transpose(1, 2), qr, "rope q")
    _close(got[:, :, nh:nh + nkv].transpose(1, 2), kr, "rope k")
    assert torch.equal(got[:, :, nh + nkv:], x[:, :, nh + nkv:]), "v heads must not be touched"
    assert torch.equal(kc.float().cpu(), got[:, :, nh:nh + nkv].transpose(1, 2)), "k cache = the rotated k rows, [B, nkv, T, hd]"
    assert torch.equal(vc.float().cpu(), v), "v cache = the v rows"
    # default positions (NULL) = 0..T-1 per sequence
    qd2 = qkv.to(DEV, torch.bfloat16)
    _lib.check(lib.fvhd_op_rope(_stream(), _p(qd2), _p(None), _p(table), _p(None), _p(None), B * T, T, nh, nkv, hd, T, 1e6), "rope")
    cos, sin = QO.rope_cos_sin(torch.arange(T)[None].expand(B, T), hd, 1e6)
    _close(qd2.float().cpu().view(B, T, -1, hd)[:, :, :nh].transpose(1, 2), QO.apply_rope(q, k, cos, sin)[0], "rope q default positions")
    # positions BEYOND the table (a caller continuing a long context): computed in the kernel from theta, not clamped to the table's edge
    # (advisor, round 3: the clamp gave plausible but wrong phases without an error)
    far = pos + 1000              # (fp32 phases: one ulp of inv_freq moves the angle by position * 6e-8 rad - kept small next to the tolerance)
    qd3 = qkv.to(DEV, torch.bfloat16)
    _lib.check(lib.fvhd_op_rope(_stream(), _p(qd3), _p(far.to(DEV)), _p(table), _p(None), _p(None), B * T, T, nh, nkv, hd, T, 1e6), "rope")
    cos, sin = QO.rope_cos_sin(far, hd, 1e6)
    _close(qd3.float().cpu().view(B, T, -1, hd)[:, :, :nh].transpose(1, 2), QO.apply_rope(q, k, cos, sin)[0], "rope q, positions beyond the table")


@pytest.mark.gpu
@pytest.mark.parametrize("hd,nh,nkv,K,cache", [(64, 14, 2, 896, True), (128, 4, 2, 256, False), (64, 2, 1, 128, True)])
def test_split_qkv_projection_with_rope_in_its_reduce(hd, nh, nkv, K, cache):
    """q|k|v projection as a split-K GEMM whose reduce adds the bias, rounds once, rotates q / k and fills the KV cache: bit-identical to the same
    partial sums + bias -> fvhd_op_rope (whose arithmetic is pinned to apply_rotary_pos_emb above), and close to the plain projection."""
    from ml_fastvlm_amd.qwen2_prefill import rope_table
    lib = _lib.load()
    B, T, splits = 3, 85, 2
    M, Mp = B * T, (B * T + 127) // 128 * 128
    width = (nh + 2 * nkv) * hd
    g = torch.Generator().manual_seed(K + hd)
    A = _bf(torch.randn(Mp, K, generator=g))
    W = _bf(torch.randn(width, K, generator=g) * K ** -0.5)
    bias = torch.randn(width, generator=g)
    pos = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])
    pos[0] += 500                                               # one sequence beyond the table
    table = rope_table(T, hd, 1e6, DEV)
    ad, wd, bd, pd = A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16), bias.to(DEV), pos.to(DEV)
    part = torch.empty(splits, Mp, width, device=DEV, dtype=torch.float32)
    qkv = torch.full((M, width), 7.0, device=DEV, dtype=torch.bfloat16)
    kc = torch.zeros(B, nkv, T, hd, device=DEV, dtype=torch.bfloat16) if cache else None
    vc = torch.zeros_like(kc) if cache else None
    _lib.check(lib.fvhd_op_qkv_splitk_rope(_stream(), _p(ad), _p(wd), _p(bd), _p(part), _p(qkv), _p(pd), _p(table), _p(kc), _p(vc), M, Mp, K, T, nh, nkv, hd,
                                           T, 1e6, splits), "split qkv + rope")
    torch.cuda.synchronize()
    # the same partial sums, reduced here, then the stand-alone rotary kernel
    ref = ((part[0] + part[1])[:M] + bd).to(torch.bfloat16).contiguous()
    plain = ref.clone()
    kc2 = torch.zeros_like(kc) if cache else None
    vc2 = torch.zeros_like(vc) if cache else None
    _lib.check(lib.fvhd_op_rope(_stream(), _p(ref), _p(pd), _p(table), _p(kc2), _p(vc2), M, T, nh, nkv, hd, T, 1e6), "rope")
    torch.cuda.synchronize()
    assert torch.equal(qkv, ref), "fused reduce + rope differs from reduce -> rope"
    if cache:
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2), "KV cache"
    _close(plain, A[:M] @ W.t() + bias, f"split q|k|v projection {M}x{width}x{K}")


@pytest.mark.gpu
@pytest.mark.parametrize("nh,nkv,K,B,T,cache", [(14, 2, 896, 8, 285, True), (2, 1, 128, 3, 85, True), (4, 2, 256, 1, 1, False), (14, 2, 896, 1, 281, True)])
def test_qkv_projection_with_rope_in_its_epilogue(nh, nkv, K, B, T, cache):
    """fvhd_op_gemm_qkv_rope (round 5): q|k|v projection + bias + rotary embedding + KV-cache copies in ONE launch (head_dim 64) -
    bit-identical to fvhd_op_gemm(EPI_BIAS) followed by fvhd_op_rope (whose arithmetic is pinned to apply_rotary_pos_emb above), padding rows
    included (projected, never rotated)."""
    from ml_fastvlm_amd.qwen2_prefill import rope_table
    lib = _lib.load()
    hd = 64
    M, Mp = B * T, (B * T + 255) // 256 * 256
    width = (nh + 2 * nkv) * hd
    if not lib.fvhd_gemm_qkv_rope_supported(Mp, width, K, hd, nh, nkv):
        pytest.skip("shape outside the streaming 128 x 128 kernel's rules")
    g = torch.Generator().manual_seed(K + T)
    A = _bf(torch.randn(Mp, K, generator=g))
    W = _bf(torch.randn(width, K, generator=g) * K ** -0.5)
    bias = torch.randn(width, generator=g)
    pos = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])
    pos[0] += 500                                               # one sequence beyond the table
    table = rope_table(T, hd, 1e6, DEV)
    ad, wd, bd, pd = A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16), bias.to(DEV), pos.to(DEV)
    mk = lambda: (torch.zeros(B, nkv, T, hd, device=DEV, dtype=torch.bfloat16) if cache else None)
    kc, vc, kc2, vc2 = mk(), mk(), mk(), mk()
    got = torch.full((Mp, width), 7.0, device=DEV, dtype=torch.bfloat16)
    _lib.check(lib.fvhd_op_gemm_qkv_rope(_stream(), _p(ad), _p(wd), _p(bd), _p(got), Mp, width, K, _p(pd), _p(table), _p(kc), _p(vc), M, T, nh, nkv, hd, T,
                                         1e6), "qkv + rope")
    ref = torch.empty_like(got)
    _lib.check(lib.fvhd_op_gemm(_stream(), _p(ad), _p(wd), _p(bd), _p(None), _p(None), _p(ref), Mp, width, K, _lib.EPI_BIAS, _lib.BF16), "qkv gemm")
    _lib.check(lib.fvhd_op_rope(_stream(), _p(ref), _p(pd), _p(table), _p(kc2), _p(vc2), M, T, nh, nkv, hd, T, 1e6), "rope")
    torch.cuda.synchronize()
    assert torch.equal(got, ref), "fused epilogue differs from projection -> rope"
    if cache:
        assert torch.equal(kc, kc2) and torch.equal(vc, vc2), "KV cache"
    # default positions (NULL)
    got2, ref2 = torch.empty_like(got), torch.empty_like(got)
    _lib.check(lib.fvhd_op_gemm_qkv_rope(_stream(), _p(ad), _p(wd), _p(bd), _p(got2), Mp, width, K, _p(None), _p(table), _p(None), _p(None), M, T, nh, nkv, hd, T,
                                         1e6), "qkv + rope, default positions")
    _lib.check(lib.fvhd_op_gemm(_stream(), _p(ad), _p(wd), _p(bd), _p(None), _p(None), _p(ref2), Mp, width, K, _lib.EPI_BIAS, _lib.BF16), "qkv gemm")
    _lib.check(lib.fvhd_op_rope(_stream(), _p(ref2), _p(None), _p(table), _p(None), _p(None), M, T, nh, nkv, hd, T, 1e6), "rope")
    torch.cuda.synchronize()
    assert torch.equal(got2, ref2)
    assert lib.fvhd_gemm_qkv_rope_supported(Mp, (nh + 2 * nkv) * 128, K, 128, nh, nkv) == 0, "head_dim 128 keeps the two launches"


@pytest.mark.gpu
@pytest.mark.parametrize("hd,nh,nkv,B,T,pad", [(64, 14, 2, 3, 285, "none"), (64, 4, 2, 4, 130, "left"), (64, 2, 1, 2, 64, "right"),
                                              (128, 4, 2, 2, 200, "left"), (128, 2, 2, 3, 17, "none"), (64, 2, 2, 1, 1, "none"),
                                              (64, 2, 1, 5, 300, "left")])        # left padding of up to 97 positions: whole key tiles masked
def test_causal_gqa_attention(hd, nh, nkv, B, T, pad):
    lib = _lib.load()
    g = torch.Generator().manual_seed(T + hd)
    width = (nh + 2 * nkv) * hd
    qkv = _bf(torch.randn(B * T, width, generator=g))
    _, mask, _ = _inputs(B, T, 8, pad=pad)
    qd = qkv.to(DEV, torch.bfloat16)
    od = torch.full((B * T, nh * hd), 7.0, device=DEV, dtype=torch.bfloat16)
    md = mask.to(DEV, torch.uint8)
    _lib.check(lib.fvhd_op_attention_causal(_stream(), _p(qd), _p(od), _p(md if pad != "none" else None), B, T, nh, nkv, hd), "attention")
    torch.cuda.synchronize()
    x = qkv.view(B, T, nh + 2 * nkv, hd)
    q, k, v = x[:, :, :nh].transpose(1, 2), x[:, :, nh:nh + nkv].transpose(1, 2), x[:, :, nh + nkv:].transpose(1, 2)
    want = QO.attention(q, k, v, mask)
    got = od.float().cpu().view(B, T, nh * hd)
    valid = mask.bool()
    assert torch.isfinite(got).all()
    _close(got[valid], want[valid], f"attention hd{hd} T{T} {pad}", rtol=2e-2, atol_rms=2e-2)      # P is rounded to bf16 for the PV MFMA (the tolerance of the tower's attention test)


@pytest.mark.gpu
def test_gemm_swiglu_and_residual_epilogues():
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    for M, K, I in ((300, 896, 4864), (2304, 128, 256), (256, 256, 64)):        # 2304 x 512 tiles: the streaming kernel at I = ... only when >= 512 tiles
        A = _bf(torch.randn(M, K, generator=g))
        Wg, Wu = _bf(torch.randn(I, K, generator=g) * K ** -0.5), _bf(torch.randn(I, K, generator=g) * K ** -0.5)
        Wi = torch.stack([Wg, Wu], 1).reshape(2 * I, K).contiguous()            # rows interleaved gate_j, up_j
        ad, wd = A.to(DEV, torch.bfloat16), Wi.to(DEV, torch.bfloat16)
        out = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
        _lib.check(lib.fvhd_op_gemm(_stream(), _p(ad), _p(wd), _p(None), _p(None), _p(None), _p(out), M, 2 * I, K, _lib.EPI_SWIGLU, _lib.BF16), "swiglu")
        torch.cuda.synchronize()
        _close(out, torch.nn.functional.silu(A @ Wg.t()) * (A @ Wu.t()), f"swiglu {M}x{K}x{I}")
        Wd = _bf(torch.randn(K, I, generator=g) * I ** -0.5)
        act = _bf(torch.randn(M, I, generator=g))
        res = _bf(torch.randn(M, K, generator=g))
        rd, actd, wdd = res.to(DEV, torch.bfloat16), act.to(DEV, torch.bfloat16), Wd.to(DEV, torch.bfloat16)     # (named: alive until the sync)
        _lib.check(lib.fvhd_op_gemm(_stream(), _p(actd), _p(wdd), _p(None), _p(None), _p(rd), _p(rd), M, K, I, _lib.EPI_RESID, _lib.BF16), "resid")
        torch.cuda.synchronize()
        _close(rd, res + act @ Wd.t(), f"resid {M}x{I}x{K}")
    # split-K (down_proj at the prefill shape): partial sums in slice order + residual, one rounding
    for M, N, K, splits in ((2304, 896, 4864, 4), (300, 128, 512, 2), (256, 256, 256, 1)):
        A, W, res = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5), _bf(torch.randn(M, N, generator=g))
        rd, ad, wd = res.to(DEV, torch.bfloat16), A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16)
        part = torch.empty(splits * M * N, device=DEV, dtype=torch.float32)
        _lib.check(lib.fvhd_op_gemm_splitk(_stream(), _p(ad), _p(wd), _p(rd), _p(rd), _p(part), M, N, K, splits), "splitk")
        torch.cuda.synchronize()
        _close(rd, res + A @ W.t(), f"split-K {M}x{N}x{K}/{splits}")
    # split-K whose reduce also applies the next RMSNorm: both outputs bit-identical to the two separate launches
    for M, N, K, splits in ((2304, 896, 4864, 4), (2304, 896, 896, 2), (300, 1536, 512, 2), (7, 3584, 256, 1)):
        A, W, res = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * K ** -0.5), _bf(torch.randn(M, N, generator=g))
        nw = (1.0 + 0.2 * torch.randn(N, generator=g)).to(DEV)
        ad, wd = A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16)
        part = torch.empty(splits * M * N, device=DEV, dtype=torch.float32)
        r0 = res.to(DEV, torch.bfloat16)
        n0 = torch.empty_like(r0)
        _lib.check(lib.fvhd_op_gemm_splitk(_stream(), _p(ad), _p(wd), _p(r0), _p(r0), _p(part), M, N, K, splits), "splitk")
        _lib.check(lib.fvhd_op_rmsnorm(_stream(), _p(r0), _p(n0), _p(nw), M, N, 1e-6), "rmsnorm")
        r1 = res.to(DEV, torch.bfloat16)
        n1 = torch.full_like(r1, 7.0)
        _lib.check(lib.fvhd_op_gemm_splitk_norm(_stream(), _p(ad), _p(wd), _p(r1), _p(r1), _p(part), M, N, K, splits, _p(nw), _p(n1), 1e-6), "splitk + norm")
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(n0, n1), f"split-K + norm {M}x{N}x{K}/{splits}"
    # fp32 logits without bias (lm_head), few rows
    A = _bf(torch.randn(8, 896, generator=g))
    W = _bf(torch.randn(1024, 896, generator=g) * 896 ** -0.5)
    out = torch.empty(8, 1024, device=DEV, dtype=torch.float32)
    ad, wd = A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16)
    _lib.check(lib.fvhd_op_gemm(_stream(), _p(ad), _p(wd), _p(None), _p(None), _p(None), _p(out), 8, 1024, 896, _lib.EPI_NONE, _lib.F32), "lm_head gemm")
    torch.cuda.synchronize()
    _close(out, A @ W.t(), "fp32 logits", rtol=2e-3, atol_rms=2e-3)


# ------------------------------------------------------------------------------------------------- GPU: the whole prefill
def _compare_prefill(cfg, B, T, pad, seed, layers_tol):
    from ml_fastvlm_amd.qwen2_prefill import Qwen2Prefill, kv_to_dynamic_cache
    m = _model(cfg, seed)
    x, mask, pos = _inputs(B, T, cfg.hidden_size, seed=seed + 10, pad=pad)
    x = _bf(x)
    sd = {k: (_bf(v) if v.dim() == 2 else v) for k, v in m.state_dict().items()}     # the matrices the library holds are bf16
    m.load_state_dict(sd)
    with torch.no_grad():
        want = m(inputs_embeds=x, attention_mask=mask, position_ids=pos).logits[:, -1]
    _, hidden, kvs = QO.prefill(x, sd, cfg, mask, pos)
    pre = Qwen2Prefill.from_hf(m.to(DEV))
    logits, kc, vc = pre(x.to(DEV, torch.bfloat16), mask.to(DEV), pos.to(DEV), return_kv=True)
    torch.cuda.synchronize()
    assert logits.shape == (B, cfg.vocab_size) and logits.dtype == torch.float32 and torch.isfinite(logits).all()
    valid = mask.bool()
    got_h = pre.hidden_states(B * T).float().cpu().view(B, T, -1)
    rel_h, cos_h = _metrics(got_h[valid], hidden[valid])
    rel, cos = _metrics(logits, want)
    print(f"prefill H={cfg.hidden_size} L={cfg.num_hidden_layers} B={B} T={T} pad={pad}: residual stream rel-L2 {rel_h:.3e} cos {cos_h:.6f}; "
          f"last-position logits rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel_h <= layers_tol and cos_h >= 0.9998, (rel_h, cos_h)
    if pad != "right":                               # with right padding position -1 is a padding row: meaningless in the reference too
        assert rel <= 2e-2 and cos >= 0.9995, (rel, cos)
        top2 = want.topk(2, -1).values
        err = (logits.cpu() - want).abs().max(-1).values
        for b in range(B):
            if top2[b, 0] - top2[b, 1] > 2 * err[b]:
                assert int(logits[b].argmax()) == int(want[b].argmax())
    # KV cache: rotated keys and values of the valid positions, in transformers' [B, nkv, T, hd] layer layout
    for l in range(cfg.num_hidden_layers):
        vm = valid[:, None, :, None].expand_as(kvs[l][0])
        rk, _ = _metrics(kc[l].float().cpu()[vm], kvs[l][0][vm])
        rv, _ = _metrics(vc[l].float().cpu()[vm], kvs[l][1][vm])
        assert rk <= layers_tol and rv <= layers_tol, (l, rk, rv)
    return m, pre, x, mask, pos, logits, kc, vc


@pytest.mark.gpu
@pytest.mark.parametrize("pad", ["none", "left", "right"])
def test_prefill_tiny_model_vs_transformers(pad):
    _compare_prefill(_cfg(hidden=128, layers=2, heads=2, kv=1, inter=256, vocab=512), 3, 70, pad, seed=1, layers_tol=1.5e-2)


@pytest.mark.gpu
def test_prefill_qwen2_05b_shapes_two_layers_b8():
    """BASELINE.json configs[2] shapes - hidden 896, 14 / 2 heads of 64, intermediate 4864, B = 8 x 285 tokens - two layers deep,
    vocabulary cut to 2048 rows (the lm_head GEMM is shape-generic in N)."""
    _compare_prefill(_cfg(hidden=896, layers=2, heads=14, kv=2, inter=4864, vocab=2048), 8, 285, "none", seed=2, layers_tol=1.5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"FVHD_LLM_QKVSPLIT": "2"}, {"FVHD_LLM_QKVSPLIT": "0", "FVHD_LLM_OSPLIT": "0", "FVHD_LLM_FUSENORM": "0"},
                                 {"FVHD_LLM_SPLITK": "0", "FVHD_LLM_OSPLIT": "0"}, {"FVHD_LLM_FUSEROPE": "1"}])
def test_prefill_launch_plans_agree_with_transformers(env, monkeypatch):
    """every launch plan of the decoder layer (fvhd_llm_create reads the switches): split q|k|v projection with bias + rotary embedding in its reduce;
    no fused norm / no split o_proj (the round-3 plan); no split-K at all; rotary embedding + KV-cache copies inside the q|k|v projection's epilogue (round 5, head_dim 64; opt-in: measured neutral)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _compare_prefill(_cfg(hidden=896, layers=2, heads=14, kv=2, inter=4864, vocab=2048), 4, 150, "left", seed=7, layers_tol=1.5e-2)
    _compare_prefill(_cfg(hidden=256, layers=3, heads=4, kv=2, inter=512, vocab=512), 3, 70, "none", seed=8, layers_tol=1.5e-2)


@pytest.mark.gpu
def test_prefill_qwen2_7b_shapes_one_layer():
    """BASELINE.json configs[3] widths (Qwen2-7B: hidden 3584, 28 / 4 heads of 128, intermediate 18944), one layer, 2 x 96 tokens."""
    _compare_prefill(_cfg(hidden=3584, layers=1, heads=28, kv=4, inter=18944, vocab=1024), 2, 96, "left", seed=3, layers_tol=1e-2)


@pytest.mark.gpu
def test_prefill_hands_its_kv_cache_to_the_transformers_decode_loop():
    """The drop-in contract of the prefill step: first token from our logits, then the stock `transformers` decode loop continues
    from OUR KV cache and produces the tokens it produces from its own prefill."""
    from ml_fastvlm_amd.qwen2_prefill import kv_to_dynamic_cache
    cfg = _cfg(hidden=128, layers=2, heads=2, kv=1, inter=256, vocab=512)
    m, pre, x, mask, pos, logits, kc, vc = _compare_prefill(cfg, 2, 40, "none", seed=5, layers_tol=1.5e-2)
    m = m.to(DEV).float()
    with torch.no_grad():
        ref = m(inputs_embeds=x.to(DEV), attention_mask=mask.to(DEV), position_ids=pos.to(DEV), use_cache=True)
        tok = ref.logits[:, -1].argmax(-1)
        step_ref = m(input_ids=tok[:, None], past_key_values=ref.past_key_values, use_cache=True,
                     attention_mask=torch.ones(2, 41, device=DEV, dtype=torch.long), position_ids=torch.full((2, 1), 40, device=DEV)).logits[:, -1]
        cache = kv_to_dynamic_cache(kc.float(), vc.float())
        step_ours = m(input_ids=tok[:, None], past_key_values=cache, use_cache=True,
                      attention_mask=torch.ones(2, 41, device=DEV, dtype=torch.long), position_ids=torch.full((2, 1), 40, device=DEV)).logits[:, -1]
    rel, cos = _metrics(step_ours, step_ref)
    print(f"decode step from our KV cache vs from transformers' own: logits rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel <= 2e-2 and cos >= 0.9995


# ------------------------------------------------------------------------------------------------- CPU: the reference-side patch
def test_install_into_llava_prefill_patch_keeps_the_references_forward_where_it_is_not_eligible():
    """install_into_llava(prefill=True) wraps LlavaQwen2ForCausalLM.forward (llava_qwen.py:66-116).  On CPU (or with labels, with a
    non-empty cache, for single-token steps) the wrapper must be the reference's forward, bit for bit."""
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip("reference tree not mounted / staged")
    import sys
    ref_import.install_timm_stub()
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import llava.model.language_model.llava_qwen as lq
    from transformers import Qwen2Config
    from ml_fastvlm_amd import builder
    cfg = lq.LlavaConfig(**Qwen2Config(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                       num_key_value_heads=1, max_position_embeddings=256).to_dict())
    torch.manual_seed(0)
    model = lq.LlavaQwen2ForCausalLM(cfg).eval()
    x = torch.randn(2, 7, 128)
    saved = lq.LlavaQwen2ForCausalLM.forward
    try:
        with torch.no_grad():
            want = model(inputs_embeds=x, use_cache=True)
        builder.install_into_llava.__globals__["_make_prefill_forward"]      # (the patch is importable without a GPU)
        lq.LlavaQwen2ForCausalLM.forward = builder._make_prefill_forward(saved)
        assert lq.LlavaQwen2ForCausalLM.forward._fvhd_prefill and lq.LlavaQwen2ForCausalLM.forward._fvhd_orig is saved
        with torch.no_grad():
            got = model(inputs_embeds=x, use_cache=True)                       # CPU tensors: not eligible -> the reference's own forward
            ids = torch.randint(0, 256, (2, 5))
            got_ids = model(input_ids=ids)
        assert torch.equal(got.logits, want.logits) and got.logits.shape == (2, 7, 256)
        assert got_ids.logits.shape == (2, 5, 256)
        lab = model(inputs_embeds=x, labels=torch.randint(0, 256, (2, 7)))     # training contract untouched
        assert lab.loss is not None and lab.loss.requires_grad
    finally:
        lq.LlavaQwen2ForCausalLM.forward = saved
