"""The XOR keys of the fused ConvFFN's transposing tile stage (csrc/ffn_fused.hip: ffn_slot_of), restated: the chunk -> LDS-slot map
must be a bijection of the tile's chunk ids (it is applied on the writing and on the reading side), and the fragment ds_read_b128
- served in the 16-lane groups of MI355X_MICROARCH.md "LDS" - must find its 16 rows in 16 different 16-B slots modulo 16 (256-B bank row).
Rounds 2-3 shipped keys that left 4-way (C = 96) and 2-way (C = 192) conflicts (PMC: 28 % / 8.4 % of the LDS cycles); this model
reproduces those numbers for the old keys.  CPU only; the source is parsed so that the two cannot drift apart."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml_fastvlm_amd", "csrc", "ffn_fused.hip")
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def _slot_fns():
    src = open(SRC).read()
    body = src[src.index("FVHD_DEV int ffn_slot_of(int id)"):]
    body = body[:body.index("}\n") + 1]
    exprs = re.findall(r"return (\(id \^ .*?\)) << 4;", body)
    assert len(exprs) == 3, body
    return {C: eval("lambda id: " + e) for C, e in zip((384, 192, 96), exprs)}     # C-style integer expressions are valid Python


def _old(C):
    sh = {384: 4, 192: 3, 96: 2}[C]
    return lambda id: id ^ ((id >> sh) & ((1 << sh) - 1))


def _worst_read_conflict(f, C):
    cpr, worst = C // 8, 0
    for ks in range(C // 16):
        for half in (0, 1):
            for g in GROUPS:
                slots = [f(li * cpr + 2 * ks + half) % 16 for li in g]
                worst = max(worst, max(slots.count(s) for s in set(slots)))
    return worst


@pytest.mark.parametrize("C", [96, 192, 384])
def test_stage_key_is_a_bijection_and_conflict_free(C):
    f = _slot_fns()[C]
    n = 32 * (C // 8)
    assert sorted(f(i) for i in range(n)) == list(range(n)), "not a permutation of the tile's chunk ids"
    assert _worst_read_conflict(f, C) == 1
    # the coalesced side: a ds_write_b128 is served in 8-lane groups of consecutive ids (32-bank modulus: 16-B slot mod 8)
    for i in range(n // 64):
        for g8 in range(8):
            slots = [f(i * 64 + g8 * 8 + l) % 8 for l in range(8)]
            assert len(set(slots)) == 8


def test_the_model_reproduces_the_measured_conflicts_of_the_old_keys():
    assert [_worst_read_conflict(_old(C), C) for C in (96, 192, 384)] == [4, 2, 1]
