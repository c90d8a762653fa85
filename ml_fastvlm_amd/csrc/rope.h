// Rotary position embedding of one (a, b) = (x[i4 ..], x[i4 + HD/2 ..]) pair of quads (apply_rotary_pos_emb, rotate_half form; transformers
// Qwen2) - shared by rope_kernel / splitk_bias_rope_kernel (llm.hip) and by the q|k|v projection's fused epilogue (gemm.hip, round 5).
#pragma once
#include "fvhd_common.h"

// (cos, sin) of i4 .. i4 + 3 at position p, then the rotation of one (a, b) = (x[i4..], x[i4 + HD/2 ..]) pair of quads - shared by the two kernels below
FVHD_DEV void rope_rotate(const f32x4 a, const f32x4 b, long p, int i4, const float* __restrict__ table, int HD, int P, float theta, bf16x4& ra, bf16x4& rb)
{
    f32x4 cs0, cs1;                                              // (cos, sin) of i4, i4+1 | i4+2, i4+3
    if (p >= 0 && p < P) {
        const float* tb = table + ((size_t)p * (HD / 2) + i4) * 2;
        cs0 = *(const f32x4*)tb;
        cs1 = *(const f32x4*)(tb + 4);
    } else {
        // a position outside the table (a caller continuing a context longer than the table, or a negative id): the phases are
        // computed here with the table's own formula - inv_freq_i = theta^(-2i / HD), angle = p * inv_freq_i in fp32 - instead of being
        // clamped to the table's edge (which gave plausible but wrong logits without an error: advisor, round 3)
        float cs[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float inv = 1.0f / powf(theta, (float)(2 * (i4 + k)) / (float)HD);      // the host's expression for the table rows
            const float ang = (float)p * inv;
            cs[2 * k] = cosf(ang);
            cs[2 * k + 1] = sinf(ang);
        }
        cs0 = f32x4{cs[0], cs[1], cs[2], cs[3]};
        cs1 = f32x4{cs[4], cs[5], cs[6], cs[7]};
    }
    const float c[4] = {cs0[0], cs0[2], cs1[0], cs1[2]}, s[4] = {cs0[1], cs0[3], cs1[1], cs1[3]};
    f32x4 oa, ob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        oa[k] = a[k] * c[k] - b[k] * s[k];
        ob[k] = b[k] * c[k] + a[k] * s[k];
    }
    ra = f32_to_bf4(oa);
    rb = f32_to_bf4(ob);
}

