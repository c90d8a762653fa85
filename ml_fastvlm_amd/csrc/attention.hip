// LayerNormChannel and multi-head self-attention of the FastViTHD AttentionBlocks (stages 3, 4).
//
//   LayerNormChannel.forward   mci.py:617-623   per-pixel LN over C, biased variance, eps 1e-5
//   MHSA.forward               mci.py:661-685   softmax((q * 32^-0.5) k^T) v, head_dim 32, no qkv bias
//
// NHWC makes the channel axis contiguous, so LN is one wave per token row (two-pass in registers,
// wave64 butterfly reductions) and the qkv GEMM output [B*N, 3C] already has q/k/v of head h at
// column offsets h*32, C + h*32, 2C + h*32 - attention reads it in place with a row stride of 3C.
//
// Attention kernel (flash-style, one (image, head, 128-query block) per 256-thread workgroup):
//   * each wave owns 2 x 16 queries (every K / V^T fragment read from LDS feeds two MFMAs: the 16x16x32 fragment is 1 KiB
//     per 16-cycle MFMA, i.e. the whole LDS bandwidth at one query block per wave); Q fragments live in registers.
//   * K/V stream through LDS in 64-key tiles, double buffered; K tile row-major [64][32] with the
//     same XOR slot swizzle as the GEMM (conflict-free ds_read_b128 fragment reads); V is stored
//     TRANSPOSED in LDS ([32 d][64 keys], 136-B row stride -> conflict-free ds_read_b64) because the
//     PV contraction runs over keys and an MFMA operand needs the contraction index contiguous
//     inside a lane.
//   * scores are computed transposed, S^T = K . Q^T (mfma(Kfrag, Qfrag)): the C/D layout then gives
//     every lane 4 consecutive keys of ONE query (q = lane & 15), so the softmax row statistics are
//     lane-local plus two xor-shuffles (lanes q, q+16, q+32, q+48), and two S^T fragments are
//     already a valid B operand of O^T = V^T . P^T for a permuted key order - the same permutation
//     the V^T fragment is read with - so P never leaves registers.
//   * online softmax in fp32 with exp2 (scale * log2 e folded), P rounded to bf16 for the MFMA,
//     row sums of those bf16 values accumulated in fp32 by the matrix cores (a ones fragment), O^T accumulators fp32; out-of-range keys are
//     masked to -1e30, out-of-range queries are not stored (any N works, e.g. 16 or 576).
#include "fvhd_common.h"

// ---------------------------------------------------------------------------------------------------
// LayerNorm over the channel axis: x [M, C] bf16 -> y [M, C] bf16.  One wave per row, C % 4 == 0.
template <int VPL>   // 8-byte vectors per lane (C = 256 * VPL when exact)
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int M, int C, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16* xr = x + (size_t)row * C;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            v[i] = bf4_to_f32(*(const bf16x4*)(xr + c));
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    bf16* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            const f32x4 wv = *(const f32x4*)(w + c), bv = *(const f32x4*)(b + c);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mean) * rstd * wv[k] + bv[k];
            *(bf16x4*)(yr + c) = f32_to_bf4(o);
        }
    }
}

extern "C" int fvhd_launch_layernorm(hipStream_t st, const void* x, void* y, const float* w, const float* b,
                                     int M, int C, float eps)
{
    if (C % 4 || C > 256 * 8) return (int)hipErrorInvalidValue;
    const int vpl = (C + 255) / 256;
    dim3 grid((M + 3) / 4), block(256);
    const bf16* xi = (const bf16*)x;
    bf16* yo = (bf16*)y;
#define LN_CASE(V) case V: hipLaunchKernelGGL((layernorm_kernel<V>), grid, block, 0, st, xi, yo, w, b, M, C, eps); break;
    switch (vpl) { LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8) }
#undef LN_CASE
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
#define ATT_D 32
#define ATT_KT 64          // keys per tile
#ifndef ATT_QW
#define ATT_QW 2           // 16-query blocks per wave: every K / V^T fragment read from LDS feeds ATT_QW MFMAs
#endif
#define ATT_QB (64 * ATT_QW)   // queries per workgroup
#ifndef ATT_VTR
#define ATT_VTR 1          // 1: bf16 V staged row-major and transposed by ds_read_b64_tr_b16 (round 6); 0: the round-1 ds_write_b16 transposition (A/B)
#endif
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) att_s16x4* att_lds_s16x4p;
#define ATT_VT_STRIDE 136  // bytes per V^T row in LDS (64 keys * 2 B + 8 B pad)

// fp8 (OCP e4m3, round to nearest even) pack of 8 fp32 values, k-slot order preserved: the operand of
// v_mfma_f32_16x16x32_fp8_fp8 holds k = 8g..8g+7 in one 64-bit register pair - the bf16 fragment layout at half the bytes.
// Saturating (round 5, advisor): the gfx950 conversion turns |x| > 448 into NaN - one such element of Q / K / V would poison a whole softmax
// row where the bf16 path has no such limit - so the values are clamped to the e4m3 range first (v_med3_f32, what HIP's satfinite
// conversions do).  v_med3_f32 alone would turn a NaN into -448 (with a NaN operand it returns the minimum of the three): a NaN input is
// passed through instead, so that it reaches the output as it does on the bf16 path (round 6, advisor).
FVHD_DEV long pack_fp8x8(f32x8 v)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] != v[i]) ? v[i] : __builtin_amdgcn_fmed3f(v[i], -448.0f, 448.0f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
#define ATT_VT8_STRIDE 68  // bytes per fp8 V^T row in LDS (64 keys + 4 B pad: row stride of 17 banks)

// FP8 = false: Q, K, V, P are bf16 MFMA operands (the parity path, the default).
// FP8 = true (BASELINE.json configs[4], "fp8 MFMA attention path"; opt-in: fvhd_set_attention_fp8): Q, K, V are rounded to OCP e4m3 when
// they are staged and P = exp(s - m) in [0, 1] when it is packed, both GEMMs run on v_mfma_f32_16x16x32_fp8_fp8 with fp32 accumulation;
// the running maximum stays fp32 and the denominator sums the SAME e4m3 P values that multiply V (a ones fragment, like the bf16 form).
// K / V tiles take half the LDS bytes.  Measured (rounds 1-2): correct, and no faster than bf16 - that MFMA issues at the bf16 rate,
// the kernel is exp-bound at head_dim 32 and attention is 9 % of the 1536^2 step (DESIGN.md "fp8"); kept so that configs[4] runs as named.
template <bool FP8>
__global__ __launch_bounds__(256) void attention_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                        int N, int C, float scale_log2e)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * (ATT_KT * 64 + ATT_D * ATT_VT_STRIDE)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    // 1-D grid in (image, head, query block) order, XCD-remapped: the query blocks of one (image, head) - which all stream the
    // same K / V rows - run on ONE XCD, so K / V come from HBM once, not once per XCD (round 1: a (qb, h, b) grid, consecutive
    // block ids = the 8 query blocks of a head = 8 different L2s: 654 MB fetched per launch against 151 MB of qkv).
    const int nqb = (N + ATT_QB - 1) / ATT_QB, nh = C / ATT_D;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = L % nqb, h = (L / nqb) % nh, b = L / (nqb * nh);
    const size_t row_stride = (size_t)3 * C;
    const bf16* base = qkv + (size_t)b * N * row_stride + h * ATT_D;

    // Q fragments (B operand: n = query = lr, k = d = 8g..8g+7), ATT_QW blocks of 16 queries per wave
    int q_idx[ATT_QW];
    bf16x8 qf[ATT_QW];
    long qf8[ATT_QW];
#pragma unroll
    for (int w = 0; w < ATT_QW; ++w) {
        q_idx[w] = qb * ATT_QB + (wave * ATT_QW + w) * 16 + lr;
        qf[w] = *(const bf16x8*)(base + (size_t)min(q_idx[w], N - 1) * row_stride + g * 8);
        if constexpr (FP8) qf8[w] = pack_fp8x8(bf8_to_f32(qf[w]));
    }

    // staging assignment: thread -> (key = tid>>2, 16-B chunk = tid&3) of the K and V tiles
    const int skey = tid >> 2, sch = tid & 3;
    const bf16* kptr = base + C + sch * 8;
    const bf16* vptr = base + 2 * C + sch * 8;
    const int k_dst = skey * 64 + ((sch ^ ((0 - (skey >> 2)) & 3)) << 4);
    // fp8 K tile: 32-B rows, 8-B slot s of row r at slot s ^ (2 * ((r >> 3) & 1)): the 32 lanes of one ds_read_b64 group
    // (16 rows x 2 slots) then cover all 64 banks once
    const int k8_dst = skey * 32 + ((sch ^ (((skey >> 3) & 1) << 1)) << 3);
    // bf16 V tile (round 6): row-major [key][32 d] like K, 64-B rows; the 32-B half (d 0..15 | 16..31) of a row is swapped for keys with
    // (key >> 2) & 1, so that the 8 keys x 32 B a half-wave of a transposing read touches cover all 64 banks once.
    // v_dst: this thread's 16-B chunk sch of key skey; v_src: this lane's source role for the transposing read (see the PV loop)
    const int v_dst = skey * 64 + ((sch ^ (((skey >> 2) & 1) << 1)) << 4);
    const unsigned v_src = lds_addr(lds) + (unsigned)(ATT_KT * 64 + (4 * g + (lr >> 2)) * 64 + (lr & 3) * 8);

    f32x4 o_acc[ATT_QW][2];                              // O^T[d = df*16 + 4g + r][q = lr]
    // softmax denominators on the matrix cores: a third "V^T fragment" of ones makes every row of l_acc the sum over the keys of the
    // SAME bf16 P values that multiply V (fp32 accumulate) - 2 more MFMAs per query block and tile instead of 16 v_add_f32 + 2
    // cross-lane shuffles + their waits (the kernel is VALU-bound: ~170 VALU instructions per 16 MFMAs)
    f32x4 l_acc[ATT_QW];
    float m_run[ATT_QW];
    bf16x8 ones;
    const long ones8 = 0x3838383838383838L;              // eight e4m3 1.0
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
#pragma unroll
    for (int w = 0; w < ATT_QW; ++w) {
        o_acc[w][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        o_acc[w][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        l_acc[w] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[w] = -1e30f;
    }

    const int ntiles = (N + ATT_KT - 1) / ATT_KT;
    u32x4 rk, rv;
    // running pointers of this thread's K / V rows: one 64-bit add per tile instead of a 64-bit multiply; only a tile that reaches
    // past N clamps its key index (a wave-uniform branch)
    const bf16* kcur = kptr + (size_t)min(skey, N - 1) * row_stride;
    const bf16* vcur = vptr + (size_t)min(skey, N - 1) * row_stride;
    const size_t tile_step = (size_t)ATT_KT * row_stride;
    rk = *(const u32x4*)kcur;
    rv = *(const u32x4*)vcur;
    for (int t = 0; t < ntiles; ++t) {
        char* kbuf = lds + (t & 1) * (ATT_KT * 64 + ATT_D * ATT_VT_STRIDE);
        char* vbuf = kbuf + ATT_KT * 64;
        if constexpr (FP8) {
            *(long*)(kbuf + k8_dst) = pack_fp8x8(bf8_to_f32(__builtin_bit_cast(bf16x8, rk)));
            const long v8 = pack_fp8x8(bf8_to_f32(__builtin_bit_cast(bf16x8, rv)));
#pragma unroll
            for (int i = 0; i < 8; ++i) *(unsigned char*)(vbuf + (sch * 8 + i) * ATT_VT8_STRIDE + skey) = (unsigned char)(v8 >> (8 * i));
        } else {
            *(u32x4*)(kbuf + k_dst) = rk;
#if ATT_VTR
            *(u32x4*)(vbuf + v_dst) = rv;               // V row-major like K: transposed by the READ (ds_read_b64_tr_b16), not by 8 ds_write_b16
#else
            const bf16x8 vv = __builtin_bit_cast(bf16x8, rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) *(bf16*)(vbuf + (sch * 8 + i) * ATT_VT_STRIDE + skey * 2) = vv[i];
#endif
        }
        __syncthreads();
        if (t + 1 < ntiles) {
            if ((t + 2) * ATT_KT <= N) {
                kcur += tile_step;
                vcur += tile_step;
            } else {
                const int key = min((t + 1) * ATT_KT + skey, N - 1);
                kcur = kptr + (size_t)key * row_stride;
                vcur = vptr + (size_t)key * row_stride;
            }
            rk = *(const u32x4*)kcur;
            rv = *(const u32x4*)vcur;
        }

        // S^T[key][q] for 4 key fragments of 16: one K fragment read per fragment, used by every query block
        f32x4 s[ATT_QW][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const int krow = kf * 16 + lr;
            if constexpr (FP8) {
                const long kfr = *(const long*)(kbuf + krow * 32 + ((g ^ (((krow >> 3) & 1) << 1)) << 3));
#pragma unroll
                for (int w = 0; w < ATT_QW; ++w)
                    s[w][kf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kfr, qf8[w], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else {
                const bf16x8 kfr = *(const bf16x8*)(kbuf + krow * 64 + ((g ^ ((0 - (krow >> 2)) & 3)) << 4));
#pragma unroll
                for (int w = 0; w < ATT_QW; ++w)
                    s[w][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[w], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
        }
        if ((t + 1) * ATT_KT > N) {
            // the one tile that reaches past N (none when N % 64 == 0): a REAL wave-uniform branch - as selects inside the loop below
            // the masking cost 3 VALU instructions per score on every tile (the asm statement keeps the block from being if-converted)
            asm volatile("; ragged key tile");
#pragma unroll
            for (int w = 0; w < ATT_QW; ++w)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * ATT_KT + kf * 16 + g * 4 + r >= N) s[w][kf][r] = -1e30f;
        }
        long pf8[ATT_QW][2];
        bf16x8 pf[ATT_QW][2];   // B operand of O^T = V^T . P^T: n = q = lr, k-slot j <-> key (j<4 ? 4g+j : 16+4g+j-4) of chunk c
#pragma unroll
        for (int w = 0; w < ATT_QW; ++w) {
            float mx = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[w][kf][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[w], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[w] - m_new) * scale_log2e);
            m_run[w] = m_new;
            const float mb = m_new * scale_log2e;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x8 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[w][2 * c][r], scale_log2e, -mb));
                    p[4 + r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[w][2 * c + 1][r], scale_log2e, -mb));
                }
                if constexpr (FP8) pf8[w][c] = pack_fp8x8(p);
                else pf[w][c] = f32_to_bf8(p);
            }
            // after the first tiles the running maximum rarely moves: alpha == 1 exactly (exp2(0)), and multiplying by it is the
            // identity - skipped when no lane of the wave saw a new maximum (same bits, 12 VALU instructions fewer per tile)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                l_acc[w] *= alpha;
                o_acc[w][0] *= alpha;
                o_acc[w][1] *= alpha;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (FP8) l_acc[w] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ones8, pf8[w][c], l_acc[w], 0, 0, 0);
                else l_acc[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[w][c], l_acc[w], 0, 0, 0);
            }
        }
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // A operand: row = d = df*16 + lr, k-slot j <-> same key permutation as pf; one read feeds every query block
                if constexpr (FP8) {
                    const char* vr = vbuf + (df * 16 + lr) * ATT_VT8_STRIDE + c * 32 + g * 4;
                    const unsigned lo = *(const unsigned*)(vr), hi = *(const unsigned*)(vr + 16);
                    const long vf = (long)(((unsigned long)hi << 32) | lo);
#pragma unroll
                    for (int w = 0; w < ATT_QW; ++w)
                        o_acc[w][df] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf, pf8[w][c], o_acc[w][df], 0, 0, 0);
                } else {
#if ATT_VTR
                    // lane (g, lr = 4 x + e) gets element e of the 8 bytes source lanes 4 j + x (j = 0..3) address: source lane (g, j, x) points at
                    // d = df*16 + 4 x .. + 3 of key c*32 + 4 g + j (+ 16 for the second half) -> 4 consecutive keys of d = df*16 + lr
                    const unsigned va = v_src + (unsigned)((t & 1) * (ATT_KT * 64 + ATT_D * ATT_VT_STRIDE) + c * 32 * 64) + (unsigned)((df ^ (g & 1)) << 5);
                    const att_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4p)(size_t)va);
                    const att_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4p)(size_t)(va + 16 * 64));
                    const bf16x4 lo = __builtin_bit_cast(bf16x4, lo4), hi = __builtin_bit_cast(bf16x4, hi4);
                    const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#else
                    const char* vr = vbuf + (df * 16 + lr) * ATT_VT_STRIDE + (c * 32 + g * 4) * 2;
                    const bf16x4 lo = *(const bf16x4*)(vr);
                    const bf16x4 hi = *(const bf16x4*)(vr + 32);
                    const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
#pragma unroll
                    for (int w = 0; w < ATT_QW; ++w)
                        o_acc[w][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[w][c], o_acc[w][df], 0, 0, 0);
                }
            }
        // no second barrier: the next iteration writes the OTHER buffer, and the buffer written two
        // iterations from now is only touched after every wave has passed the next __syncthreads().
    }

#pragma unroll
    for (int w = 0; w < ATT_QW; ++w)
        if (q_idx[w] < N) {
            const float inv = 1.0f / l_acc[w][0];
            bf16* orow = out + ((size_t)b * N + q_idx[w]) * C + h * ATT_D;
#pragma unroll
            for (int df = 0; df < 2; ++df)
                *(bf16x4*)(orow + df * 16 + g * 4) = f32_to_bf4(o_acc[w][df] * inv);
        }
}

// qkv [B*N, 3C] bf16 (q | k | v, head h at columns h*32) -> out [B*N, C] bf16.  C % 32 == 0.
// fp8 != 0 selects the e4m3 operand path.
extern "C" int fvhd_launch_attention(hipStream_t st, const void* qkv, void* out, int B, int N, int C, int fp8)
{
    if (C % ATT_D || B <= 0 || N <= 0) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(((N + ATT_QB - 1) / ATT_QB) * (C / ATT_D) * B));
    const float scale_log2e = 0.17677669529663687f * 1.4426950408889634f;   // 32^-0.5 * log2(e)
    if (fp8) hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, N, C, scale_log2e);
    else hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, N, C, scale_log2e);
    return (int)hipGetLastError();
}
