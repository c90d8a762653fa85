// Qwen2 prefill kernels: the LLM half of FastVLM's time-to-first-token, BASELINE.json configs[2] / [3].
//
// The reference runs the prefill through the third-party `transformers` Qwen2ForCausalLM (pinned 4.48.3 in the reference's
// pyproject.toml:17; call sites llava/model/language_model/llava_qwen.py:92-103 (forward) and :138-143 (generate)):
//     per layer   x = x + o_proj(attn(rope(q_proj(n1(x))), rope(k_proj(n1(x))), v_proj(n1(x))))        Qwen2DecoderLayer.forward
//                 x = x + down_proj(silu(gate_proj(n2(x))) * up_proj(n2(x)))                              Qwen2MLP.forward
//     n = Qwen2RMSNorm (fp32 statistics, eps 1e-6), rope = rotate_half form with theta = 1e6 (apply_rotary_pos_emb), grouped-query
//     attention with a causal + key-padding mask and scale head_dim^-0.5 (eager_attention_forward / repeat_kv), logits =
//     lm_head(norm(x)) of the last position.
// Here: the GEMMs are the tower's hand-written MFMA GEMM (gemm.hip; the q/k/v projections run as ONE GEMM over the concatenated
// weight, gate/up as ONE GEMM over row-interleaved weights whose epilogue forms silu(gate) * up, o_proj / down_proj add the
// residual in their epilogue), and this file adds what the tower did not have: RMSNorm, rotary embedding (in place on the packed
// q|k|v rows, optionally filling the KV cache), and a causal grouped-query flash attention for head_dim 64 / 128.
// Activations are bf16 rows [B*T (padded to 256), width], statistics and accumulation fp32 - the same arithmetic contract as the tower.
#include "fvhd_common.h"
#include "rope.h"

// ---------------------------------------------------------------------------------------------------
// Qwen2RMSNorm: y = w * x * rsqrt(mean(x^2) + eps).  One wave per row, two passes over the row (the second one hits L1 / L2).
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w,
                                                      int M, int H, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16* xr = x + (size_t)row * H;
    float s = 0.f;
    for (int c = lane * 8; c < H; c += 512) {
        const f32x8 v = bf8_to_f32(*(const bf16x8*)(xr + c));
#pragma unroll
        for (int k = 0; k < 8; ++k) s = __builtin_fmaf(v[k], v[k], s);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s) / (float)H + eps);
    bf16* yr = y + (size_t)row * H;
    for (int c = lane * 8; c < H; c += 512) {
        const f32x8 v = bf8_to_f32(*(const bf16x8*)(xr + c));
        const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
        f32x8 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = v[k] * rstd * w0[k]; o[4 + k] = v[4 + k] * rstd * w1[k]; }
        *(bf16x8*)(yr + c) = f32_to_bf8(o);
    }
}

// x, y [M, H] bf16 (may alias), w fp32 [H]; H % 8 == 0
extern "C" int fvhd_launch_rmsnorm(hipStream_t st, const void* x, void* y, const float* w, int M, int H, float eps)
{
    if (M <= 0 || H <= 0 || H % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, st, (const bf16*)x, (bf16*)y, w, M, H, eps);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Rotary position embedding (apply_rotary_pos_emb, rotate_half form) in place on the q and k heads of the packed projection
// rows qkv [M, (nh + 2 nkv) * HD]:   out[i] = x[i] cos_i - x[i + HD/2] sin_i,   out[i + HD/2] = x[i + HD/2] cos_i + x[i] sin_i
// with (cos_i, sin_i) = table[pos[row]][i]: fp32 [P][HD/2][2], computed on the host like Qwen2RotaryEmbedding.forward (fp32).
// One thread = 4 consecutive i of one head of one row.  With kcache / vcache != null the rotated k and the v heads are also written
// to the KV cache [B][nkv][T][HD] (the layout of transformers' DynamicCache layers) for a decode loop to continue from.
// (rope_rotate: rope.h)
__global__ __launch_bounds__(256) void rope_kernel(bf16* __restrict__ qkv, const long* __restrict__ pos, const float* __restrict__ table,
                                                   bf16* __restrict__ kcache, bf16* __restrict__ vcache, int M, int T, int nh, int nkv, int HD, int P,
                                                   float theta)
{
    const int per_head = HD / 8;                                 // threads per head: 4 i's each, HD / 2 i's
    const int nheads = nh + nkv + (vcache ? nkv : 0);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)M * nheads * per_head;
    if (idx >= total) return;
    const int i4 = (int)(idx % per_head) * 4;
    const int head = (int)((idx / per_head) % nheads);
    const int row = (int)(idx / ((long)per_head * nheads));
    const int width = (nh + 2 * nkv) * HD;
    if (head >= nh + nkv) {                                      // a v head: copy to the cache only
        const int j = head - nh - nkv;
        const bf16* src = qkv + (size_t)row * width + (nh + nkv + j) * HD;
        bf16* dst = vcache + (((size_t)(row / T) * nkv + j) * T + row % T) * HD;
        *(bf16x4*)(dst + i4) = *(const bf16x4*)(src + i4);
        *(bf16x4*)(dst + i4 + HD / 2) = *(const bf16x4*)(src + i4 + HD / 2);
        return;
    }
    bf16* xr = qkv + (size_t)row * width + head * HD;
    const long p = pos ? pos[row] : (long)(row % T);
    const f32x4 a = bf4_to_f32(*(const bf16x4*)(xr + i4)), b = bf4_to_f32(*(const bf16x4*)(xr + i4 + HD / 2));
    bf16x4 ra, rb;
    rope_rotate(a, b, p, i4, table, HD, P, theta, ra, rb);
    *(bf16x4*)(xr + i4) = ra;
    *(bf16x4*)(xr + i4 + HD / 2) = rb;
    if (kcache && head >= nh) {
        bf16* dst = kcache + (((size_t)(row / T) * nkv + (head - nh)) * T + row % T) * HD;
        *(bf16x4*)(dst + i4) = ra;
        *(bf16x4*)(dst + i4 + HD / 2) = rb;
    }
}

// The reduce of a split-K q|k|v projection (fvhd_launch_gemm_splitk_partials: fp32 partials [splits][Mp][width]) with everything that
// follows the projection in Qwen2Attention.forward: + bias, ONE rounding to bf16 (what the projection's own epilogue does), rotary
// embedding of the q and k heads, the KV-cache copies.  Same thread layout and the same arithmetic as rope_kernel - the result is
// bit-identical to reduce + bias -> rope_kernel; one launch and one pass over the packed rows less per decoder layer.
__global__ __launch_bounds__(256) void splitk_bias_rope_kernel(const float* __restrict__ part, int splits, size_t slice, const float* __restrict__ bias,
                                                               bf16* __restrict__ qkv, const long* __restrict__ pos, const float* __restrict__ table,
                                                               bf16* __restrict__ kcache, bf16* __restrict__ vcache, int M, int T, int nh, int nkv, int HD,
                                                               int P, float theta)
{
    const int per_head = HD / 8;
    const int nheads = nh + 2 * nkv;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)M * nheads * per_head;
    if (idx >= total) return;
    const int i4 = (int)(idx % per_head) * 4;
    const int head = (int)((idx / per_head) % nheads);
    const int row = (int)(idx / ((long)per_head * nheads));
    const int width = nheads * HD, col = head * HD + i4;
    const float* pr = part + (size_t)row * width + col;
    f32x4 a = *(const f32x4*)pr, b = *(const f32x4*)(pr + HD / 2);
    for (int s = 1; s < splits; ++s) { a += *(const f32x4*)(pr + s * slice); b += *(const f32x4*)(pr + s * slice + HD / 2); }
    if (bias) { a += *(const f32x4*)(bias + col); b += *(const f32x4*)(bias + col + HD / 2); }
    bf16x4 ra = f32_to_bf4(a), rb = f32_to_bf4(b);               // the projection's output as the reference holds it (bf16)
    bf16* xr = qkv + (size_t)row * width + head * HD;
    if (head < nh + nkv) {
        const long p = pos ? pos[row] : (long)(row % T);
        rope_rotate(bf4_to_f32(ra), bf4_to_f32(rb), p, i4, table, HD, P, theta, ra, rb);
    }
    *(bf16x4*)(xr + i4) = ra;
    *(bf16x4*)(xr + i4 + HD / 2) = rb;
    if (kcache && head >= nh) {
        bf16* dst = (head < nh + nkv ? kcache + (((size_t)(row / T) * nkv + (head - nh)) * T + row % T) * HD
                                     : vcache + (((size_t)(row / T) * nkv + (head - nh - nkv)) * T + row % T) * HD);
        *(bf16x4*)(dst + i4) = ra;
        *(bf16x4*)(dst + i4 + HD / 2) = rb;
    }
}

// qkv [M = B*T rows (only these are touched), (nh + 2 nkv) * HD] bf16 in place; pos int64 [M] or null (= t); table fp32 [P][HD/2][2]
extern "C" int fvhd_launch_rope(hipStream_t st, void* qkv, const long* pos, const float* table, void* kcache, void* vcache,
                                int M, int T, int nh, int nkv, int HD, int P, float theta)
{
    if (M <= 0 || T <= 0 || nh <= 0 || nkv <= 0 || HD % 8 || P <= 0 || !(theta > 0.f) || (kcache == nullptr) != (vcache == nullptr)) return (int)hipErrorInvalidValue;
    const long total = (long)M * (nh + nkv + (vcache ? nkv : 0)) * (HD / 8);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (bf16*)qkv, pos, table, (bf16*)kcache, (bf16*)vcache,
                       M, T, nh, nkv, HD, P, theta);
    return (int)hipGetLastError();
}

// partials fp32 [splits][Mp][(nh + 2 nkv) * HD] of the split-K projection (Mp >= M rows per slice) -> qkv rows [M, width] bf16 (+ bias, rotary, caches)
extern "C" int fvhd_launch_splitk_bias_rope(hipStream_t st, const float* partial, int splits, int Mp, const float* bias, void* qkv, const long* pos,
                                            const float* table, void* kcache, void* vcache, int M, int T, int nh, int nkv, int HD, int P, float theta)
{
    if (M <= 0 || Mp < M || splits < 1 || T <= 0 || nh <= 0 || nkv <= 0 || HD % 8 || P <= 0 || !(theta > 0.f) || (kcache == nullptr) != (vcache == nullptr))
        return (int)hipErrorInvalidValue;
    const int width = (nh + 2 * nkv) * HD;
    const long total = (long)M * (nh + 2 * nkv) * (HD / 8);
    hipLaunchKernelGGL(splitk_bias_rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, splits, (size_t)Mp * width, bias, (bf16*)qkv,
                       pos, table, (bf16*)kcache, (bf16*)vcache, M, T, nh, nkv, HD, P, theta);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Causal grouped-query attention (eager_attention_forward + repeat_kv + the causal / padding mask of the prefill):
//     out[b, t, h] = softmax_k( q[b,t,h] . k[b,k,h / (nh/nkv)] * HD^-0.5 + mask ) v[b,k,h / (nh/nkv)],   mask: k <= t and key_valid[b,k]
// Same flash structure as attention.hip (S^T = K . Q^T so the softmax statistics are lane-local + 2 shuffles, P stays in registers as
// the B operand of O^T = V^T . P^T under a key permutation, K / V^T tiles of 64 keys double-buffered in LDS), generalised to head_dim
// 64 / 128 (HD / 32 k-steps per score fragment, HD / 16 output fragments), with the query's own position as the causal limit: key
// tiles beyond the workgroup's last query are never loaded.
template <int HD>
struct LlmAttCfg {
    static constexpr int KT = 64, QW = 2, QB = 64 * QW;
    static constexpr int KBYTES = KT * HD * 2;                  // K tile: HD / 64 panels of [64 keys][64 d], 128-B rows, XOR-swizzled slots
    static constexpr int VSTRIDE = 136;                         // bytes per V^T row (64 keys * 2 B + 8 B pad)
    static constexpr int VBYTES = HD * VSTRIDE;
    static constexpr int LDS = 2 * (KBYTES + VBYTES);
};

template <int HD>
__global__ __launch_bounds__(256) void llm_attention_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, const unsigned char* __restrict__ key_valid,
                                                            int T, int nh, int nkv, float scale_log2e)
{
    using K = LlmAttCfg<HD>;
    constexpr int KS = HD / 32, DF = HD / 16, CH = HD / 8;      // score k-steps, output fragments, 16-B chunks per key row
    constexpr int NST = K::KT * CH / 256;                       // staging chunks per thread per matrix (2 / 4)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int nqb = (T + K::QB - 1) / K::QB;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = L % nqb, h = (L / nqb) % nh, b = L / (nqb * nh);
    const int hk = h / (nh / nkv);
    const int width = (nh + 2 * nkv) * HD;
    const bf16* rows = qkv + (size_t)b * T * width;
    const bf16* qbase = rows + h * HD;
    const bf16* kbase = rows + (nh + hk) * HD;
    const bf16* vbase = rows + (nh + nkv + hk) * HD;
    const unsigned char* kv = key_valid ? key_valid + (size_t)b * T : nullptr;

    int q_idx[K::QW];
    bf16x8 qf[K::QW][KS];
#pragma unroll
    for (int w = 0; w < K::QW; ++w) {
        q_idx[w] = qb * K::QB + (wave * K::QW + w) * 16 + lr;
        const bf16* qr = qbase + (size_t)min(q_idx[w], T - 1) * width;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[w][kk] = *(const bf16x8*)(qr + kk * 32 + g * 8);
    }
    // staging: chunk id c = i * 256 + tid -> key c / CH, 16-B chunk c % CH of the key's HD values
    int skey[NST], sch[NST], kdst[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int c = i * 256 + tid;
        skey[i] = c / CH;
        sch[i] = c % CH;
        kdst[i] = (sch[i] >> 3) * (K::KT * 128) + skey[i] * 128 + (((sch[i] & 7) ^ ((skey[i] >> 1) & 7)) << 4);
    }

    f32x4 o_acc[K::QW][DF];
    f32x4 l_acc[K::QW];                 // softmax denominators on the matrix cores (a ones fragment as one more V^T fragment: attention.hip)
    float m_run[K::QW];
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
#pragma unroll
    for (int w = 0; w < K::QW; ++w) {
#pragma unroll
        for (int df = 0; df < DF; ++df) o_acc[w][df] = f32x4{0.f, 0.f, 0.f, 0.f};
        l_acc[w] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[w] = -1e30f;
    }

    const int kmax = min(T, (qb + 1) * K::QB);                  // causal: no key beyond this workgroup's last query
    const int ntiles = (kmax + K::KT - 1) / K::KT;
    u32x4 rk[NST], rv[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const size_t ro = (size_t)min(skey[i], T - 1) * width + sch[i] * 8;
        rk[i] = *(const u32x4*)(kbase + ro);
        rv[i] = *(const u32x4*)(vbase + ro);
    }
    for (int t = 0; t < ntiles; ++t) {
        char* kbuf = lds + (t & 1) * (K::KBYTES + K::VBYTES);
        char* vbuf = kbuf + K::KBYTES;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            *(u32x4*)(kbuf + kdst[i]) = rk[i];
            const bf16x8 vv = __builtin_bit_cast(bf16x8, rv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) *(bf16*)(vbuf + (sch[i] * 8 + e) * K::VSTRIDE + skey[i] * 2) = vv[e];
        }
        __syncthreads();
        if (t + 1 < ntiles) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const size_t ro = (size_t)min((t + 1) * K::KT + skey[i], T - 1) * width + sch[i] * 8;
                rk[i] = *(const u32x4*)(kbase + ro);
                rv[i] = *(const u32x4*)(vbase + ro);
            }
        }
        // validity of the tile's 64 keys as one wave-uniform 64-bit mask (bit = key inside the tile)
        const int kl = t * K::KT + lane;
        const unsigned long long vmask = __ballot(kl < T && (!kv || kv[min(kl, T - 1)] != 0));

        f32x4 s[K::QW][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const int krow = kf * 16 + lr;
#pragma unroll
            for (int w = 0; w < K::QW; ++w) s[w][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int ks = kk * 4 + g;
                const bf16x8 kfr = *(const bf16x8*)(kbuf + (ks >> 3) * (K::KT * 128) + krow * 128 + (((ks & 7) ^ ((krow >> 1) & 7)) << 4));
#pragma unroll
                for (int w = 0; w < K::QW; ++w) s[w][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[w][kk], s[w][kf], 0, 0, 0);
            }
        }
        // masking only where a tile needs it - a padded key in it, or keys beyond the wave's FIRST query (the causal diagonal): a real
        // wave-uniform branch (as selects in the loop below the mask cost ~3 VALU instructions per score on every tile; the asm
        // statement keeps the block from being if-converted)
        const int q_first = qb * K::QB + wave * K::QW * 16;
        if (vmask != ~0ull || t * K::KT + K::KT - 1 > q_first) {
            asm volatile("; masked key tile");
#pragma unroll
            for (int w = 0; w < K::QW; ++w)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kin = kf * 16 + g * 4 + r;                    // key inside the tile
                        const bool ok = ((vmask >> kin) & 1ull) && (t * K::KT + kin <= q_idx[w]);
                        if (!ok) s[w][kf][r] = -1e30f;
                    }
        }
        bf16x8 pf[K::QW][2];
#pragma unroll
        for (int w = 0; w < K::QW; ++w) {
            float mx = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[w][kf][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[w], mx);
            // a row that has seen no visible key yet (a padding query in front of a left-padded sequence, or the leading all-padding
            // tiles of a real one) keeps m = -1e30; its exponent is taken against 0 instead, so every masked score gives exp2(-huge)
            // = 0 exactly - with the -1e30 reference the difference of two rounded 1.8e29 products is +-1e22, i.e. exp2 = inf -> NaN
            // rows whose k / v then poison every later query of the sequence through 0 x NaN
            const float m_ref = m_new <= -1e29f ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run[w] <= -1e29f ? -1e30f : m_run[w] - m_ref) * scale_log2e);
            m_run[w] = m_new;
            const float mb = m_ref * scale_log2e;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x8 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[w][2 * c][r], scale_log2e, -mb));
                    p[4 + r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[w][2 * c + 1][r], scale_log2e, -mb));
                }
                pf[w][c] = f32_to_bf8(p);
            }
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {      // alpha == 1 exactly once the running maximum stops moving
                l_acc[w] *= alpha;
#pragma unroll
                for (int df = 0; df < DF; ++df) o_acc[w][df] *= alpha;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) l_acc[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[w][c], l_acc[w], 0, 0, 0);
        }
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const char* vr = vbuf + (df * 16 + lr) * K::VSTRIDE + (c * 32 + g * 4) * 2;
                const bf16x4 lo = *(const bf16x4*)(vr);
                const bf16x4 hi = *(const bf16x4*)(vr + 32);
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int w = 0; w < K::QW; ++w) o_acc[w][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[w][c], o_acc[w][df], 0, 0, 0);
            }
    }

#pragma unroll
    for (int w = 0; w < K::QW; ++w)
        if (q_idx[w] < T) {
            // a query row whose keys are ALL masked (a padding position in front of a left-padded sequence) has l = 0: it is written
            // as zeros - finite, and never read (the reference produces equally meaningless rows there)
            const float inv = l_acc[w][0] > 0.f ? 1.0f / l_acc[w][0] : 0.f;
            bf16* orow = out + ((size_t)b * T + q_idx[w]) * ((size_t)nh * HD) + h * HD;
#pragma unroll
            for (int df = 0; df < DF; ++df) *(bf16x4*)(orow + df * 16 + g * 4) = f32_to_bf4(o_acc[w][df] * inv);
        }
}

template <int HD> static hipError_t launch_llm_attention(hipStream_t st, const bf16* qkv, bf16* out, const unsigned char* key_valid, int B, int T, int nh, int nkv)
{
    using K = LlmAttCfg<HD>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)llm_attention_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const float scale_log2e = (1.0f / sqrtf((float)HD)) * 1.4426950408889634f;
    const long grid = (long)((T + K::QB - 1) / K::QB) * nh * B;
    if (grid <= 0 || grid > 0x7fffffffl) return hipErrorInvalidValue;
    hipLaunchKernelGGL(llm_attention_kernel<HD>, dim3((unsigned)grid), dim3(256), K::LDS, st, qkv, out, key_valid, T, nh, nkv, scale_log2e);
    return hipGetLastError();
}

// qkv [B*T, (nh + 2 nkv) * HD] bf16 (q heads | k heads | v heads, rope already applied) -> out [B*T, nh * HD] bf16;
// key_valid uint8 [B, T] (the attention mask of the spliced batch) or null; HD in {64, 128}; nh % nkv == 0
extern "C" int fvhd_launch_llm_attention(hipStream_t st, const void* qkv, void* out, const unsigned char* key_valid, int B, int T, int nh, int nkv, int HD)
{
    if (B <= 0 || T <= 0 || nh <= 0 || nkv <= 0 || nh % nkv) return (int)hipErrorInvalidValue;
    if (HD == 64) return (int)launch_llm_attention<64>(st, (const bf16*)qkv, (bf16*)out, key_valid, B, T, nh, nkv);
    if (HD == 128) return (int)launch_llm_attention<128>(st, (const bf16*)qkv, (bf16*)out, key_valid, B, T, nh, nkv);
    return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// rows [B, T, H] of `dtype` -> dst [Mpad, H] bf16 (the first B*T rows; the padding rows stay as they are), and
// gather of one position per sequence: dst[b] = src[b*T + t_last]  (the rows the lm_head sees)
template <typename TIn>
__global__ __launch_bounds__(256) void cast_rows_kernel(const TIn* __restrict__ src, bf16* __restrict__ dst, long n)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v;
    if constexpr (sizeof(TIn) == 4) v = *(const f32x4*)(src + i);
    else if constexpr (__is_same(TIn, bf16)) v = bf4_to_f32(*(const bf16x4*)(src + i));
    else v = __builtin_convertvector(*(const f16x4*)(src + i), f32x4);
    *(bf16x4*)(dst + i) = f32_to_bf4(v);
}

extern "C" int fvhd_launch_cast_rows(hipStream_t st, const void* src, int dtype, void* dst, long n)
{
    if (n <= 0 || n % 4) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
    if (dtype == FVHD_F32) hipLaunchKernelGGL(cast_rows_kernel<float>, grid, block, 0, st, (const float*)src, (bf16*)dst, n);
    else if (dtype == FVHD_F16) hipLaunchKernelGGL(cast_rows_kernel<_Float16>, grid, block, 0, st, (const _Float16*)src, (bf16*)dst, n);
    else if (dtype == FVHD_BF16) hipLaunchKernelGGL(cast_rows_kernel<bf16>, grid, block, 0, st, (const bf16*)src, (bf16*)dst, n);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int T, int t_sel, int H)
{
    const int b = blockIdx.x;
    const bf16* s = src + ((size_t)b * T + t_sel) * H;
    bf16* d = dst + (size_t)b * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) *(u32x4*)(d + c) = *(const u32x4*)(s + c);
}

extern "C" int fvhd_launch_gather_rows(hipStream_t st, const void* src, void* dst, int B, int T, int t_sel, int H)
{
    if (B <= 0 || T <= 0 || t_sel < 0 || t_sel >= T || H % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, B, T, t_sel, H);
    return (int)hipGetLastError();
}
