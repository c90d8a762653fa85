// Fused ConvFFN MLP:  X <- X + ls * ( gelu(A . W1^T + b1) . W2^T + b2 )        (in place on X)
//
//   ConvFFN.fc1 -> GELU -> fc2 (mci.py:922-926) + layer scale + residual (mci.py:1106-1109 / 1187-1188);
//   A is the output of the (BatchNorm-folded) depthwise 7x7 (mci.py:921).
//
// 94 % of the encoder's FLOPs are these two 1x1 GEMMs.  Run as two kernels, the [M, 4C] hidden tensor
// makes stages 0-1 HBM-bound (arithmetic intensity 77 / 154 flop/B) and its bias+erf-GELU epilogue is
// as long as the GEMM main loop at K = C <= 384 (round-1 profile: fc1 260-396 TF/s).  Here the hidden
// activations never leave the register file ("flash-MLP", the same operand trick as the attention
// kernel):
//
//   * a wave owns 32 rows (pixels) of A for the whole kernel; the row block A^T lives in registers as
//     the B operands of v_mfma_f32_32x32x16_bf16 (C/16 fragments), the output block O^T[C x 32] as
//     C/32 fp32 accumulator tiles (AGPRs).
//   * per 32 hidden units: S^T[32h x 32m] = W1chunk . A^T  (C/16 MFMAs, W1 fragment = A operand read
//     from LDS) -> + b1, exact-erf GELU in fp32 -> round to bf16.  The C/D layout leaves lane
//     (m = lane&31, half = lane>>5) with the 16 hidden units h = (r&3) + 8(r>>2) + 4*half; regs 0-7 /
//     8-15 are, as they stand, valid B operands of the two K=16 steps of O^T += W2chunk . P^T for the
//     hidden order  k-slot (half, j) <-> h = 16kb + 8(j>>2) + 4half + (j&3).  W2 is stored by the host
//     with its hidden axis pre-permuted into exactly that order (fvhd_api.hip: pack_ffn), so the W2
//     fragment is a plain 16-B ds_read_b128 and P needs no cross-lane movement and no LDS round trip.
//   * W1 / W2 stream through LDS in slices of HS hidden units (24 KB each), shared by the 4 waves of
//     a workgroup, double buffered: the next slice's global loads are issued before the current
//     slice's MFMAs and written to the other buffer after them; one barrier per slice.
//     16-B slot XOR swizzles (per row stride) make the ds_write_b128 and the fragment ds_read_b128
//     conflict-free (lane groups of MI355X_MICROARCH "LDS").
//   * HBM traffic per row: A in (2C B), X in/out (4C B) - the algorithmic minimum; weights come from L2.
#include "fvhd_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C> FVHD_DEV int w1_off(int row, int slot)      // W1 slice [HS][C] bf16, slot = 16-B index in the row
{
    if constexpr (C == 384) return row * 768 + ((slot ^ (row & 15)) << 4);
    else if constexpr (C == 192) return row * 384 + ((slot ^ ((row >> 1) & 7)) << 4);
    else return row * 192 + ((slot ^ ((row >> 2) & 3)) << 4);            // C == 96
}

template <int HS> FVHD_DEV int w2_off(int row, int slot)     // W2 slice [C][HS] bf16
{
    if constexpr (HS == 32) return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
    else if constexpr (HS == 64) return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
    else return row * 256 + ((slot ^ (row & 15)) << 4);                   // HS == 128
}

template <int C, int HS, int OCC>
__global__ __launch_bounds__(256, OCC) void ffn_fused_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ W1, const float* __restrict__ b1,
    const bf16* __restrict__ W2s, const float* __restrict__ b2, const float* __restrict__ ls,
    bf16* X, int M, int nwg)
{
    constexpr int HID = 4 * C;
    constexpr int KS = C / 16;              // K=16 steps of GEMM1
    constexpr int NFR = C / 32;             // 32-wide output fragments of GEMM2
    constexpr int NSL = HID / HS;           // weight slices
    constexpr int CH = HS / 32;             // 32-hidden-unit chunks per slice
    constexpr int SPR1 = C / 8;             // 16-B slots per W1 row
    constexpr int SPR2 = HS / 8;            // 16-B slots per W2 row
    constexpr int SLICE_B = HS * C * 2;     // bytes of one W1 (or W2) slice
    constexpr int NCHUNK = HS * C / 8 / 256;   // 16-B chunks per thread per matrix per slice
    static_assert(HS * C / 8 % 256 == 0, "slice must split evenly over 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf0: W1 slice | W2 slice][buf1: ...][b1 fp32 HID]
    float* lb1 = (float*)(smem + 4 * SLICE_B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int blk = xcd_remap(blockIdx.x, nwg);
    const int m_row = blk * 128 + wave * 32 + li;
    const int m_ld = min(m_row, M - 1);

    // ---- A^T fragments (B operands), straight from global: lane reads 16 B of its own row per k-step
    bf16x8 afr[KS];
    {
        const bf16* arow = A + (size_t)m_ld * C + half * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[ks] = *(const bf16x8*)(arow + ks * 16);
    }

    // ---- staging assignment ----
    int s1_dst[NCHUNK], s2_dst[NCHUNK];
    const bf16* s1_src[NCHUNK];
    const bf16* s2_src[NCHUNK];
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
        const int idx = i * 256 + tid;
        const int r1 = idx / SPR1, c1 = idx % SPR1;
        s1_src[i] = W1 + (size_t)r1 * C + c1 * 8;                 // + slice * HS * C
        s1_dst[i] = w1_off<C>(r1, c1);
        const int r2 = idx / SPR2, c2 = idx % SPR2;
        s2_src[i] = W2s + (size_t)r2 * HS + c2 * 8;               // + slice * C * HS   (slice-major packing)
        s2_dst[i] = SLICE_B + w2_off<HS>(r2, c2);
    }
    u32x4 r1v[NCHUNK], r2v[NCHUNK];
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) { r1v[i] = *(const u32x4*)s1_src[i]; r2v[i] = *(const u32x4*)s2_src[i]; }
    for (int i = tid; i < HID / 4; i += 256) *(f32x4*)&lb1[i * 4] = *(const f32x4*)&b1[i * 4];
#pragma unroll
    for (int i = 0; i < NCHUNK; ++i) {
        *(u32x4*)(smem + s1_dst[i]) = r1v[i];
        *(u32x4*)(smem + s2_dst[i]) = r2v[i];
    }
    __syncthreads();

    f32x16 o[NFR];
#pragma unroll
    for (int i = 0; i < NFR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;

    for (int sl = 0; sl < NSL; ++sl) {
        const char* buf = smem + (sl & 1) * 2 * SLICE_B;
        if (sl + 1 < NSL) {
            const size_t so = (size_t)(sl + 1) * HS * C;
#pragma unroll
            for (int i = 0; i < NCHUNK; ++i) { r1v[i] = *(const u32x4*)(s1_src[i] + so); r2v[i] = *(const u32x4*)(s2_src[i] + so); }
        }
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            // GEMM1: S^T[32 hidden][32 rows]
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(buf + w1_off<C>(ch * 32 + li, ks * 2 + half));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, afr[ks], s, 0, 0, 0);
            }
            // bias + GELU; reg r <-> hidden h = (r&3) + 8(r>>2) + 4*half of this chunk
            const float* bp = lb1 + sl * HS + ch * 32 + 4 * half;
            f32x8 p0, p1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *(const f32x4*)(bp + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = gelu_erf(s[4 * q + j] + bv[j]);
                    if (q < 2) p0[4 * q + j] = g; else p1[4 * (q - 2) + j] = g;
                }
            }
            const bf16x8 pf0 = f32_to_bf8(p0), pf1 = f32_to_bf8(p1);
            // GEMM2: O^T[n][m] += W2chunk . P^T   (two K=16 steps)
#pragma unroll
            for (int nf = 0; nf < NFR; ++nf) {
                const bf16x8 w0 = *(const bf16x8*)(buf + SLICE_B + w2_off<HS>(nf * 32 + li, ch * 4 + half));
                const bf16x8 w1 = *(const bf16x8*)(buf + SLICE_B + w2_off<HS>(nf * 32 + li, ch * 4 + 2 + half));
                o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, pf0, o[nf], 0, 0, 0);
                o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pf1, o[nf], 0, 0, 0);
            }
        }
        if (sl + 1 < NSL) {
            char* nb = smem + ((sl + 1) & 1) * 2 * SLICE_B;
#pragma unroll
            for (int i = 0; i < NCHUNK; ++i) {
                *(u32x4*)(nb + s1_dst[i]) = r1v[i];
                *(u32x4*)(nb + s2_dst[i]) = r2v[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m][n0 .. n0+3], n0 = nf*32 + 8q + 4*half -------------------------
    if (m_row < M) {
        bf16* xr = X + (size_t)m_row * C;
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = nf * 32 + 8 * q + 4 * half;
                const f32x4 bv = *(const f32x4*)(b2 + n0), lv = *(const f32x4*)(ls + n0);
                const f32x4 rv = bf4_to_f32(*(const bf16x4*)(xr + n0));
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rv[j] + lv[j] * (o[nf][4 * q + j] + bv[j]);
                *(bf16x4*)(xr + n0) = f32_to_bf4(v);
            }
    }
}

template <int C, int HS, int OCC>
static hipError_t launch_ffn(hipStream_t st, const bf16* A, const bf16* W1, const float* b1, const bf16* W2s,
                             const float* b2, const float* ls, bf16* X, int M)
{
    const int nwg = (M + 127) / 128;
    const size_t shmem = (size_t)4 * HS * C * 2 + (size_t)4 * C * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<C, HS, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((ffn_fused_kernel<C, HS, OCC>), dim3(nwg), dim3(256), shmem, st, A, W1, b1, W2s, b2, ls, X, M, nwg);
    return hipGetLastError();
}

// Hidden-slice size used for channel count C (also needed by the host packer for the W2 layout).
extern "C" int fvhd_ffn_slice(int C) { return C == 384 ? 32 : C == 192 ? 32 : C == 96 ? 64 : 0; }

// A [M,C] bf16; W1 bf16 [4C][C]; W2s bf16 slice-major [4C/HS][C][HS] with the hidden axis permuted inside every
// 32-chunk (position 16kb+8half+j holds hidden 16kb+8(j>>2)+4half+(j&3)); b1 [4C], b2 [C], ls [C] fp32; X [M,C] in/out.
extern "C" int fvhd_launch_ffn_fused(hipStream_t st, const void* A, const void* W1, const float* b1, const void* W2s,
                                     const float* b2, const float* ls, void* X, int M, int C)
{
    const bf16* a = (const bf16*)A;
    const bf16* w1 = (const bf16*)W1;
    const bf16* w2 = (const bf16*)W2s;
    bf16* x = (bf16*)X;
    hipError_t e = hipErrorInvalidValue;
    if (M <= 0) return (int)e;
    if (C == 384) e = launch_ffn<384, 32, 1>(st, a, w1, b1, w2, b2, ls, x, M);
    else if (C == 192) e = launch_ffn<192, 32, 2>(st, a, w1, b1, w2, b2, ls, x, M);
    else if (C == 96) e = launch_ffn<96, 64, 2>(st, a, w1, b1, w2, b2, ls, x, M);
    return (int)e;
}
