// bf16 MFMA GEMM with fused epilogues:  out[M,N] = epi(A[M,K] . Wt[N,K]^T)
//
// Every 1x1 convolution / nn.Linear of the encode_images() path is this op once activations are NHWC
// (a 1x1 conv over [B,H,W,C] is a row-major GEMM over M = B*H*W rows); PyTorch stores both
// nn.Linear.weight and Conv2d(k=1).weight as [out, in] = [N, K] with K contiguous, which is exactly
// the "B^T" operand layout an MFMA wants, so weights are used as stored.
//   ConvFFN.fc1 + GELU, ConvFFN.fc2 (+ layer_scale * . + residual)      mci.py:908-910, 922-926, 1106-1109, 1185-1188
//   stem[2] / PatchEmbed.proj[1] 1x1 + GELU                             mci.py:587-598, 722-734
//   MHSA.qkv (no bias), MHSA.proj (+ layer_scale_1 * . + residual)      mci.py:656-658, 668, 681
//   mm_projector Linear/GELU/Linear                                     multimodal_projector/builder.py:23-30
//
// gfx950 design (v1: LDS-staged, register-prefetched, 4 waves, 16x16x32 bf16 MFMA):
//   * 128 x (96|128) output tile per 256-thread workgroup, waves 2(M) x 2(N); each wave owns 4 x NF
//     16x16 fragments, fp32 accumulators (64 or 48 VGPRs).
//   * operands are staged global -> VGPR (16 B/lane, coalesced: 8 or 4 lanes per 128/64-B row)
//     -> LDS with an XOR swizzle on the 16-B slot index that makes both the ds_write_b128 (8-lane
//     groups) and the fragment ds_read_b128 (the 4 non-contiguous 16-lane groups of MI355X_MICROARCH
//     "LDS") conflict-free; the next K-tile's global loads are issued before the MFMA block of the
//     current one.
//   * MFMA is issued "swapped": D = Wfrag x Afrag^T, so a lane ends up holding 4 *consecutive output
//     columns* of one output row (C/D layout: col = lane&15 -> m, row = 4*(lane>>4)+reg -> n).
//     bias / layer-scale are then float4 loads, the residual and the store are 8-B bf16x4 accesses,
//     and bias+GELU / bias+layer_scale+residual run in fp32 on the accumulators (one rounding, on store).
//   * block ids are remapped so each XCD's private L2 sees a contiguous range of the tile order, and the order walks
//     8 x 8 super-tiles (see the kernel): every A / W panel fetched from HBM feeds 8 resident tiles.
#include "fvhd_common.h"
#include "gemm_layout.h"
#include "rope.h"
#include <stdlib.h>

#define EPI_NONE 0
#define EPI_BIAS 1
#define EPI_BIAS_GELU 2
#define EPI_BIAS_LS_RESID 3
#define EPI_RESID 4        // out = resid + A.W^T                        (Qwen2 o_proj / down_proj + the decoder layer's skip)
static_assert(FVHD_EPI_SWIGLU_ID == 5 && FVHD_ODT_BF16_ID == FVHD_BF16, "gemm_layout.h ids");
#define EPI_SWIGLU 5       // (= FVHD_EPI_SWIGLU_ID of gemm_layout.h)  out[m][j] = silu(acc[m][2j]) * acc[m][2j+1], out is [M, N/2]: W rows interleaved gate_j, up_j (Qwen2MLP)

#define EPI_BIAS_ROPE 6    // out = rope(bf16(A.W^T + b)) on the q / k heads of the packed q|k|v row (+ the KV-cache copies): gemm_epilogue_rope, head_dim 64 only

// what the q|k|v projection's fused epilogue needs beyond the GEMM's own arguments (Qwen2Attention.forward after the projection):
// position ids, the rotary table, the KV cache, the number of REAL rows (the GEMM also writes the padding rows, unrotated)
struct RopeArgs {
    const long* pos; const float* table; bf16* kcache; bf16* vcache;
    int M, T, nh, nkv, P; float theta;
};

template <int BK>
FVHD_DEV int lds_off(int row, int ks)
{
    if constexpr (BK == 64) return row * 128 + ((ks ^ ((row >> 1) & 7)) << 4);
    else return row * 64 + ((ks ^ ((0 - (row >> 2)) & 3)) << 4);
}

// ---- which W row an LDS row of the W tile holds (round 5: whole 16-B stores in 64-B segments, for free) -------------------------
// The MFMA is issued "swapped" (D = Wfrag x Afrag^T): lane (lr, g) of fragment j ends up with D rows 4 g .. 4 g + 3 = the W rows that
// lanes 4 g .. 4 g + 3 fed.  With LDS row p holding W row p (rounds 1-4) those are the columns n0 + 16 j + 4 g .. + 3: an 8-B store per
// lane, 16 rows x 32 B per wave-instruction - and rocprofv3 counted 1.5-1.7x the algorithmic bytes written by every GEMM class
// (profiles/r04_pmc_summary.md; the fused ConvFFN, whose stores are whole lines, writes 1.00x).  Nothing forces LDS row p to hold W row p:
// the tile is filled row by row (LDS-DMA pieces of 8 rows / register staging), so the fill permutes the rows inside every block of 16 GRP
//     LDS row 16 GRP b + 16 jl + 4 q + t   <-   W row 16 GRP b + 4 GRP q + 4 jl + t          (jl < GRP, q < 4, t < 4)
// and lane (lr, g) then holds, over the GRP fragments j = GRP jb + jl, the 4 GRP CONSECUTIVE columns n0 + 16 GRP jb + 4 GRP g ..: one 16-B
// store (and one 16-B residual load) per lane, 16 rows x 64 B per wave-instruction.  Same products in the same K order per output
// element: identical bits.  GRP = 2 for bf16 outputs (8 columns = 16 B), 4 for SwiGLU (16 gate / up columns -> 8 outputs = 16 B; 4-B stores
// before), 1 (identity) for fp32 / f16 outputs (4 columns are 16 B already) and for the 96-wide tile (3 fragments per wave).
// (EpiGrp, wrow_of_lds_row, wpiece_row, wpiece_lane_row, epi_col: gemm_layout.h - shared with the CPU test of the mapping)
// ---- epilogue of one wave's (16 MF) x (16 NF) block.  GRP = 1: lane holds out[m][n .. n+3], m = mw + 16 i + lr, n = nw + 16 j + 4 g;
// GRP = 2 / 4: out[m][n .. n + 4 GRP - 1], n = nw + 16 GRP jb + 4 GRP g, as acc[i][GRP jb .. GRP jb + GRP - 1]
// ROWMAJOR (requested by the streaming kernels, whose occupancy the LDS ring fixes at one workgroup per CU; honoured for EPI_NONE only):
// rows outermost, see below.  Measured on one box (profiles/r05_gemm_epilogue_walk_ab.log, B = 32): the row-major walk brings WRITE_SIZE
// to 1.00-1.06x the algorithmic bytes for every class, costs the plain epilogue (qkv) nothing - and the epilogues that carry bias / GELU /
// layer-scale vectors 8-33 % of a launch (fc1 0.96 -> 1.09 ms per step, proj 0.356 -> 0.41, 1x1 0.35 -> 0.38, projector 0.081 -> 0.108;
// whole step 23.93 -> 24.20 ms).  The write amplification of the column-major walk (1.18-1.31x) is not a time cost: those keep it.  The
// register-prefetch kernel v1 keeps it too - the preloaded vectors of the row-major walk would take its GELU / residual variants from 168
// to 174-178 registers, i.e. from three resident workgroups to two.
// (-DFVHD_GEMM_EPI_COLMAJOR: the column-group-outermost walk everywhere, for same-box A/B runs: FVHD_VARIANT_TAG=cm)
template <int MF, int NF, int EPI, int ODT, bool ROWMAJOR_ = false>
FVHD_DEV void gemm_epilogue(f32x4 (&acc)[MF][NF], const float* __restrict__ bias, const float* __restrict__ ls, const bf16* resid, void* out,
                            int M, int N, int mw, int nw, int lr, int g)
{
    constexpr int GRP = EpiGrp<NF, EPI, ODT>::value;
#if defined(FVHD_GEMM_EPI_COLMAJOR)
    constexpr bool ROWMAJOR = false;
#elif defined(FVHD_GEMM_EPI_ROWMAJOR)           // the row-major walk for every epilogue of the streaming kernels (A/B: FVHD_VARIANT_TAG=rm)
    constexpr bool ROWMAJOR = ROWMAJOR_;
#else
    constexpr bool ROWMAJOR = ROWMAJOR_ && EPI == EPI_NONE;
#endif
    if constexpr (GRP == 1) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = nw + j * 16 + g * 4;
            if (n >= N) continue;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, lv = f32x4{1.f, 1.f, 1.f, 1.f};
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_LS_RESID) bv = *(const f32x4*)(bias + n);
            if constexpr (EPI == EPI_BIAS_LS_RESID) lv = *(const f32x4*)(ls + n);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = mw + i * 16 + lr;
                if (m >= M) continue;
                f32x4 v = acc[i][j] + bv;
                if constexpr (EPI == EPI_SWIGLU) {
                    // the lane's 4 consecutive columns are (gate, up, gate, up) of hidden units n/2, n/2 + 1: silu(gate) * up, written
                    // to the [M, N/2] activation (4-B store)
                    static_assert(ODT == FVHD_BF16, "SwiGLU writes the bf16 activation");
                    const f32x2 r = {v[0] * sigmoidf_fast(v[0]) * v[1], v[2] * sigmoidf_fast(v[2]) * v[3]};
                    *(bf16x2*)((bf16*)out + (size_t)m * (N / 2) + n / 2) = __builtin_convertvector(r, bf16x2);
                    continue;
                }
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = gelu_erf(v[c]);
                }
                const size_t o = (size_t)m * N + n;
                if constexpr (EPI == EPI_BIAS_LS_RESID || EPI == EPI_RESID) {
                    const f32x4 r = bf4_to_f32(*(const bf16x4*)(resid + o));
                    v = r + lv * v;
                }
                if constexpr (ODT == FVHD_BF16) *(bf16x4*)((bf16*)out + o) = f32_to_bf4(v);
                else if constexpr (ODT == FVHD_F16) *(f16x4*)((_Float16*)out + o) = __builtin_convertvector(v, f16x4);
                else *(f32x4*)((float*)out + o) = v;
            }
        }
    } else {
        static_assert(ODT == FVHD_BF16 && NF % GRP == 0, "grouped epilogue: bf16 outputs");
        // Row-major walk (i outer, column groups inner): a wave writes the 64-B segments of one 128-B line of its block back to back.  With
        // the groups outermost (first version of this epilogue) the second half of a line followed MF GELU blocks later and the L2 evicted
        // half-written lines in between: WRITE_SIZE stayed at 1.18-1.31x the algorithmic bytes (profiles/r05_gemm_layout.md).
        constexpr int NJB = NF / GRP;
        f32x4 bv[NJB][GRP], lv[NJB][GRP];
        int ncol[NJB];
        auto load_vectors = [&](int jb) {                          // bias / layer scale of column group jb
            ncol[jb] = nw + epi_col<GRP>(jb, g);
            const int nn = ncol[jb] < N ? ncol[jb] : 0;            // (columns past N are never stored: any valid address for the vectors)
#pragma unroll
            for (int c = 0; c < GRP; ++c) {
                bv[jb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                lv[jb][c] = f32x4{1.f, 1.f, 1.f, 1.f};
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_LS_RESID) bv[jb][c] = *(const f32x4*)(bias + nn + 4 * c);
                if constexpr (EPI == EPI_BIAS_LS_RESID) lv[jb][c] = *(const f32x4*)(ls + nn + 4 * c);
            }
        };
        auto emit = [&](int i, int jb) {                           // rows m = mw + 16 i + lr, columns ncol[jb] .. + 4 GRP - 1
            const int m = mw + i * 16 + lr, n = ncol[jb];
            if (m >= M || n >= N) return;
            f32x4 v[GRP];
#pragma unroll
            for (int c = 0; c < GRP; ++c) v[c] = acc[i][jb * GRP + c] + bv[jb][c];
            if constexpr (EPI == EPI_SWIGLU) {
                // 16 consecutive columns = (gate, up) of the 8 hidden units n/2 .. n/2 + 7: one 16-B store into the [M, N/2] activation
                static_assert(EPI != EPI_SWIGLU || GRP == 4, "SwiGLU: four fragments per group");
                f32x8 r;
#pragma unroll
                for (int c = 0; c < GRP; ++c) {
                    r[(2 * c) & 7] = v[c][0] * sigmoidf_fast(v[c][0]) * v[c][1];
                    r[(2 * c + 1) & 7] = v[c][2] * sigmoidf_fast(v[c][2]) * v[c][3];
                }
                *(bf16x8*)((bf16*)out + (size_t)m * (N / 2) + n / 2) = f32_to_bf8(r);
            } else {
                static_assert(EPI == EPI_SWIGLU || GRP == 2, "bf16 rows: two fragments per group");
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int c = 0; c < GRP; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[c][e] = gelu_erf(v[c][e]);
                }
                const size_t o = (size_t)m * N + n;
                if constexpr (EPI == EPI_BIAS_LS_RESID || EPI == EPI_RESID) {
                    const bf16x8 rr = *(const bf16x8*)(resid + o);
                    const f32x4 r0 = bf4_to_f32(__builtin_shufflevector(rr, rr, 0, 1, 2, 3)), r1 = bf4_to_f32(__builtin_shufflevector(rr, rr, 4, 5, 6, 7));
                    v[0] = r0 + lv[jb][0] * v[0];
                    v[1] = r1 + lv[jb][1] * v[1];
                }
                const bf16x4 b0 = f32_to_bf4(v[0]), b1 = f32_to_bf4(v[1]);
                *(bf16x8*)((bf16*)out + o) = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        if constexpr (ROWMAJOR) {
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) load_vectors(jb);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) emit(i, jb);
        } else {
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                load_vectors(jb);                                  // one group's vectors live at a time
#pragma unroll
                for (int i = 0; i < MF; ++i) emit(i, jb);
            }
        }
    }
}

// ---- q|k|v projection + rotary embedding + KV-cache copies in one epilogue (round 5; head_dim 64).  Since the tile-fill permutation a
// lane of a wave's 64-column block - ONE head of the packed row - holds columns 8 g .. 8 g + 7 (fragments 0, 1) and 32 + 8 g .. + 7
// (fragments 2, 3): exactly the (i, i + HD / 2) pairs rotate_half mixes.  Arithmetic = the projection's own epilogue (+ bias, one rounding
// to bf16) followed by rope_kernel's (fp32 rotation of the bf16 values, second rounding): bit-identical to the two launches.
template <int MF>
FVHD_DEV void gemm_epilogue_rope(f32x4 (&acc)[MF][4], const float* __restrict__ bias, bf16* out, const RopeArgs& r, int Mrows, int N,
                                 int mw, int nw, int lr, int g)
{
    constexpr int HD = 64;
    if (nw >= N) return;
    const int head = nw / HD, c0 = nw + 8 * g;
    const f32x4 b00 = *(const f32x4*)(bias + c0), b01 = *(const f32x4*)(bias + c0 + 4);
    const f32x4 b10 = *(const f32x4*)(bias + c0 + 32), b11 = *(const f32x4*)(bias + c0 + 36);
    const bool rotary = head < r.nh + r.nkv, cached = r.kcache != nullptr && head >= r.nh;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = mw + i * 16 + lr;
        if (m >= Mrows) continue;
        bf16x4 a0 = f32_to_bf4(acc[i][0] + b00), a1 = f32_to_bf4(acc[i][1] + b01);      // columns c0 .. c0 + 7
        bf16x4 p0 = f32_to_bf4(acc[i][2] + b10), p1 = f32_to_bf4(acc[i][3] + b11);      // their partners c0 + 32 ..
        const bool real = m < r.M;
        if (rotary && real) {
            const long p = r.pos ? r.pos[m] : (long)(m % r.T);
            rope_rotate(bf4_to_f32(a0), bf4_to_f32(p0), p, 8 * g, r.table, HD, r.P, r.theta, a0, p0);
            rope_rotate(bf4_to_f32(a1), bf4_to_f32(p1), p, 8 * g + 4, r.table, HD, r.P, r.theta, a1, p1);
        }
        const bf16x8 lo = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
        *(bf16x8*)(out + (size_t)m * N + c0) = lo;
        *(bf16x8*)(out + (size_t)m * N + c0 + 32) = hi;
        if (cached && real) {
            const int bt = m / r.T, t = m - bt * r.T;
            bf16* dst = rotary ? r.kcache + (((size_t)bt * r.nkv + (head - r.nh)) * r.T + t) * HD
                               : r.vcache + (((size_t)bt * r.nkv + (head - r.nh - r.nkv)) * r.T + t) * HD;
            *(bf16x8*)(dst + 8 * g) = lo;
            *(bf16x8*)(dst + 32 + 8 * g) = hi;
        }
    }
}

template <int NF, int BK, int EPI, int ODT>
__global__ __launch_bounds__(256) void gemm_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ ls, const bf16* resid, void* out_, int M, int N, int K, int tiles_n, int nwg, int ldk)
{
    // K = the K range one workgroup reduces; ldk = row stride of A and Wt.  Plain launches: ldk == K, gridDim.y == 1.  Split-K launches
    // (fvhd_launch_gemm_splitk): blockIdx.y = slice, A / Wt advance by slice * K columns, out = the slice's fp32 partial [M, N].
    constexpr int BM = 128, BN = 32 * NF, MF = 4;
    A += (size_t)blockIdx.y * K;
    Wt += (size_t)blockIdx.y * K;
    void* out = ODT == FVHD_F32 ? (void*)((float*)out_ + (size_t)blockIdx.y * M * N) : out_;
    constexpr int ROWB = BK * 2;
    constexpr int CPR = BK / 8;                       // 16-B chunks per tile row
    constexpr int A_CH = BM * CPR / 256;
    constexpr int W_CH = (BN * CPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * ROWB];
    char* ldsA = lds;
    char* ldsW = lds + BM * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    // Tile order.  Each XCD (private 4-MB L2) gets a contiguous range of the logical order (xcd_remap), and the logical order
    // walks SUPER-TILES of up to 8 (N) x 8 (M) tiles: the ~64 workgroups resident on an XCD then share 8 A panels and 8 W
    // panels, i.e. every panel byte fetched from HBM feeds 8 tiles.  Round 1 walked all tiles_n tiles of an M row first: at
    // N = 2304 / 3072 the whole weight matrix (3.5 / 4.7 MB) cycled through the 4-MB L2 once per M row and the stage-3/4 GEMMs
    // were fabric-bound (rocprofv3 FETCH_SIZE: 573 MB per qkv launch against 54 MB of operands, 5 TB/s - profiles/r02a).
    const int L = xcd_remap(blockIdx.x, nwg);
    constexpr int GN = 8;
    const int tiles_m = nwg / tiles_n;
    const int grp = L / (tiles_m * GN), rem = L - grp * tiles_m * GN;
    const int wg = min(GN, tiles_n - grp * GN);          // width of this N group (the last one may be narrower)
    const int tm = rem / wg, tn = grp * GN + (rem - tm * wg);
    const int m0 = tm * BM, n0 = tn * BN;

    u32x4 ra[A_CH], rw[W_CH];
    const bf16* a_src[A_CH];
    const bf16* w_src[W_CH];
    int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int idx = i * 256 + tid, row = idx / CPR, ks = idx % CPR;
        const int gr = min(m0 + row, M - 1);
        a_src[i] = A + (size_t)gr * ldk + ks * 8;
        a_dst[i] = lds_off<BK>(row, ks);
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int idx = i * 256 + tid, row = min(idx / CPR, BN - 1), ks = idx % CPR;
        const int gr = min(n0 + wrow_of_lds_row<EpiGrp<NF, EPI, ODT>::value>(row), N - 1);     // LDS row `row` holds this W row (see EpiGrp)
        w_src[i] = Wt + (size_t)gr * ldk + ks * 8;
        w_dst[i] = (idx < BN * CPR) ? lds_off<BK>(row, ks) : -1;
    }

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i]);
#pragma unroll
    for (int i = 0; i < W_CH; ++i) rw[i] = *(const u32x4*)(w_src[i]);

    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();                 // previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *(u32x4*)(ldsA + a_dst[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_CH; ++i)
            if (w_dst[i] >= 0) *(u32x4*)(ldsW + w_dst[i]) = rw[i];
        __syncthreads();
        if (kt + 1 < nk) {                           // prefetch next K tile while computing this one
            const int ko = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i] + ko);
#pragma unroll
            for (int i = 0; i < W_CH; ++i) rw[i] = *(const u32x4*)(w_src[i] + ko);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MF], wf[NF];
            const int ks = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(ldsA + lds_off<BK>(wm * 64 + i * 16 + lr, ks));
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const bf16x8*)(ldsW + lds_off<BK>(wn * 16 * NF + j * 16 + lr, ks));
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    gemm_epilogue<MF, NF, EPI, ODT>(acc, bias, ls, resid, out, M, N, m0 + wm * 64, n0 + wn * 16 * NF, lr, g);
}

// ---- v2: 256 x 128 tile, 8 waves, K tiles of 64 streamed by LDS-DMA through a ring of 3 stages ---------------------------
// v1 prefetches ONE K tile in registers: one MFMA block (~0.2 us) of cover against ~1-2 us of L2/HBM latency - rocprofv3 showed
// the stage-3/4 GEMMs at 22-31 % MFMA-busy with the waves waiting on vmcnt.  Here two K tiles (96 KB per CU) are in flight
// behind a counted vmcnt, no staging registers, ONE barrier per K tile (the barrier that publishes tile kt also retires tile
// kt - 1, whose stage the next DMA overwrites).  The LDS image is the same XOR-swizzled one as v1's (conflict-free ds_read_b128
// fragments): LDS-DMA writes lane-linearly, so the swizzle is applied on the GLOBAL side - lane l of a 1-KiB piece (8 rows x
// 128 B) fetches the 16-B chunk ks = (l & 7) ^ ((row >> 1) & 7) of row (l >> 3): still 8 whole 128-B lines per instruction.
// Each wave issues 4 of the 32 A pieces and 2 of the 16 W pieces of a K tile.  Taken when M % 256 == 0, N % 128 == 0, K % 64 == 0.
// v3 (round 3): the same streaming kernel with a 256 x 256 tile - 8 waves as 2 (M) x 4 (N), 128 x 64 per wave: 12 fragment reads per 32
// MFMAs instead of 8 per 16 (the LDS pipe was the co-bottleneck of v2), 128 accumulator registers per lane, two 64-KB stages (128 KB:
// one workgroup per CU, two waves per SIMD) - the "256^2 tile, glds, 2 LDS buffers, BK = 64, vmcnt(0) + barrier" structure of
// cdna_hip_programming.md 5.  Taken when N % 256 == 0 and there are at least two rounds of tiles.
// ring depth: K tiles of 64 -> 2 (256 x 256: 2 x 64 KB) or 3 (256 x 128: 3 x 48 KB) stages; K tiles of 32 at 8 waves (round-4 experiment,
// debug knobs 6 / 7) -> 4 x 32 KB (256 x 256) or 6 x 24 KB (256 x 128) stages: THREE / FIVE tiles of LDS-DMA in flight instead of one / two
// (the 64-KB stage of the 256 x 256 tile is issued in one burst right after the barrier and waited for at the next one)
template <int BN, int BK = 64, int NWV = 8> struct G2Cfg {
    static constexpr int RS = BK == 64 ? (BN == 256 ? 2 : 3) : (NWV == 4 ? 3 : BN == 256 ? 4 : 6), STAGE = (256 + BN) * BK * 2, LDS = RS * STAGE;
};
// the XOR key of lds_off<BK> as a function of the row (the LDS-DMA applies it on the global side)
template <int BK> FVHD_DEV int lds_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); }

FVHD_DEV void glds_piece(unsigned voff, const void* sbase, unsigned m0v)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(m0v) : "memory");
}

// BKT = 32 (debug-build experiment, knob 5): K tiles of 32 in a 3-stage ring = 72 KB at BN = 128, so that TWO 4-wave workgroups share a CU
// and one's prologue / epilogue runs behind the other's MFMAs; pieces are then 16 rows x 64 B.
// ABL (debug library only, knobs 8 / 9; wrong results, timing): 1 = no LDS-DMA (the K loop runs on stale LDS contents), 2 = no fragment
// reads and no MFMAs either side of the DMA (the DMA stream + barriers alone)
template <int EPI, int ODT, int NWV, int BN = 128, int BKT = 64, int ABL = 0>
__global__ __launch_bounds__(64 * NWV, (NWV == 4 && BKT == 32) ? 2 : 1) void gemm256_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ ls, const bf16* resid, void* out, int M, int N, int K, int tiles_n, int nwg)
{
    // BN = 128, NWV = 8: waves 4 (M) x 2 (N), 64 x 64 each (two per SIMD).  BN = 128, NWV = 4: 2 x 2, 128 x 64 each - 12 instead of 16 fragment
    // reads per 32 MFMAs, one wave per SIMD.  BN = 256, NWV = 8: 2 x 4, 128 x 64 each, two per SIMD.
    constexpr int BM = 256, BK = BKT, WN = BN / 64, WM = NWV / WN, MF = BM / WM / 16, NF = 4, RS = G2Cfg<BN, BK, NWV>::RS, STAGE = G2Cfg<BN, BK, NWV>::STAGE;
    constexpr int RPP = 1024 / (BK * 2), LPR = BK / 8;        // rows per 1-KiB piece (8 / 16), lanes (16-B chunks) per row (8 / 4)
    constexpr int PA = (BM / RPP) / NWV, PW = (BN / RPP) / NWV;   // 1-KiB pieces of the A / W tile per wave
    static_assert(WM * WN == NWV && MF * 16 * WM == BM, "wave grid");
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, g = lane >> 4;
    // Tile t of the launch (same super-tile order as v1: 8 x 8 tiles share their panels in one XCD's L2).  PERSISTENT form (round 5): with
    // gridDim.x < nwg a workgroup walks the tiles t = blockIdx.x, + gridDim.x, ... (gridDim.x a multiple of 8, so a tile stays on the XCD
    // the one-tile-per-workgroup launch would have given it) and issues the first K tiles of its NEXT tile before the epilogue of the
    // current one: the DMA round trip and the workgroup hand-over of a fresh launch slot hide behind the stores.  gridDim.x == nwg is the
    // one-tile form of rounds 2-4, unchanged.
    constexpr int GN = BN == 256 ? 4 : 8;
    const int tiles_m = nwg / tiles_n;
    auto tile_origin = [&](int t, int& m0_, int& n0_) {
        const int L = xcd_remap(t, nwg);
        const int grp = L / (tiles_m * GN), rem = L - grp * tiles_m * GN;
        const int wg = min(GN, tiles_n - grp * GN);
        const int tm = rem / wg, tn = grp * GN + (rem - tm * wg);
        m0_ = tm * BM; n0_ = tn * BN;
    };
    int tile = blockIdx.x, m0, n0;
    tile_origin(tile, m0, n0);

    // per-lane global byte offsets of this wave's pieces: rows 8 p + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7); the swizzle
    // term depends on the piece's parity only, the row base of a piece goes into the scalar base
    // W pieces: LDS rows 8 pi .. 8 pi + 7 hold the W rows wpiece_row(pi) + wpiece_lane_row(rip) (EpiGrp: the lane's output columns become
    // consecutive); A pieces are plain
    constexpr int GRP = EpiGrp<NF, EPI, ODT>::value;
    static_assert(GRP == 1 || BK == 64, "the W row permutation is written for 8-row pieces");
    const int rip = lane / LPR;
    unsigned va[2], vw[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int sw = lds_swz<BK>(par * RPP + rip);
        va[par] = (unsigned)((rip * K + (((lane % LPR) ^ sw) * 8)) * 2);
        vw[par] = (unsigned)((wpiece_lane_row<GRP>(rip) * K + (((lane % LPR) ^ sw) * 8)) * 2);
    }
    const char* abase = (const char*)(A + (size_t)(m0 + wave * RPP * PA) * K);   // A pieces PA wave .. PA wave + PA - 1 = rows RPP PA wave ..
    const char* wbase = (const char*)(Wt + (size_t)n0 * K);
    const unsigned lds0 = lds_addr(lds2);
    auto issue = [&](int kt) {                                 // K tile kt of the tile abase / wbase point at
        if constexpr (ABL == 1) return;
        const unsigned st = lds0 + (kt % RS) * STAGE;
        const size_t ko = (size_t)kt * BK * 2;
#pragma unroll
        for (int j = 0; j < PA; ++j) glds_piece(va[(wave * PA + j) & 1], abase + (size_t)j * RPP * K * 2 + ko, st + (wave * PA + j) * 1024);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int pi = wave * PW + j;                       // wave-uniform
            const int wr = GRP == 1 ? pi * RPP : wpiece_row<GRP>(pi);
            glds_piece(vw[pi & 1], wbase + (size_t)wr * K * 2 + ko, st + BM * BK * 2 + pi * 1024);
        }
    };

    const int nk = K / BK;
#pragma unroll
    for (int i = 0; i < RS - 1; ++i)
        if (i < nk) issue(i);
  for (;;) {
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
        // own pieces of tile kt have landed when at most the RS - 2 later tiles' PA + PW are outstanding.  (vmcnt <= n means at most n
        // operations are outstanding, and loads return in order: whatever else is in flight - the previous tile's epilogue stores in
        // the persistent form - all but the youngest n LOADS have landed.)
        // (near the end fewer later tiles exist: RS - 2 only while kt + RS - 2 < nk)
        {
            const int later = nk - 1 - kt < RS - 2 ? nk - 1 - kt : RS - 2;
            if (later == RS - 2 && RS > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RS - 2) * (PA + PW)) : "memory");
            else if (RS > 5 && later == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (PA + PW)) : "memory");
            else if (RS > 4 && later == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PA + PW)) : "memory");
            else if (RS > 3 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * (PA + PW)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                              // tile kt visible to every wave; tile kt - 1 fully consumed
        if (kt + RS - 1 < nk) issue(kt + RS - 1);     // into the stage tile kt - 1 occupied
        const char* ldsA = lds2 + (kt % RS) * STAGE;
        const char* ldsW = ldsA + BM * BK * 2;
        if constexpr (ABL == 2) continue;
        // (Round 4 measured the fragments of both k-steps read up front / double-buffered across the k-steps - 234 registers, all 24
        // ds_read_b128 ahead of the 64 MFMAs: no change, qkv 0.879 vs 0.892 ms per step - and the ablations of profiles/r04_gemm_ablations.log:
        // without the LDS-DMA the stage-3 qkv launch takes 142 us, with ONLY the DMA, the barriers and the epilogue stores 112 us, both
        // together 172 us, for 54 us of MFMA time: the DMA round trip per K tile and the un-overlapped stores bound it, not the reads.)
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MF], wf[NF];
            const int ks = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(ldsA + lds_off<BK>(wm * 16 * MF + i * 16 + lr, ks));
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const bf16x8*)(ldsW + lds_off<BK>(wn * 64 + j * 16 + lr, ks));
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    const int em0 = m0 + wm * 16 * MF, en0 = n0 + wn * 64;
    tile += gridDim.x;
    const bool more = tile < nwg;                              // workgroup-uniform
    if (more) {
        __syncthreads();                                       // every wave has read the last K tile: all stages are free
        tile_origin(tile, m0, n0);
        abase = (const char*)(A + (size_t)(m0 + wave * RPP * PA) * K);
        wbase = (const char*)(Wt + (size_t)n0 * K);
#pragma unroll
        for (int i = 0; i < RS - 1; ++i)
            if (i < nk) issue(i);                              // the next tile's first K tiles fly during this tile's epilogue
    }
    gemm_epilogue<MF, NF, EPI, ODT, true>(acc, bias, ls, resid, out, M, N, em0, en0, lr, g);
    if (!more) break;
  }
}

// v1s (round 4): v1's 128 x 128 tile / 4 waves (2 x 2, 64 x 64 each) with the operands streamed by LDS-DMA through a FOUR-stage ring of 32-KB
// K tiles (three tiles in flight per workgroup, one barrier per K tile) - for launches with at most ~one tile per CU, where v1's single
// register-prefetched tile leaves every K step exposed to one full L2 / HBM round trip (the prefill's q|k|v projection: 162 tiles of 14 K steps =
// 16.9 us, 1.2 us per step for 0.1 us of MFMA time; the tower's GEMMs at B = 1).  Same LDS image, fragment reads, MFMA order and epilogue as
// v1 - identical bits.  Split-K aware like v1 (blockIdx.y = K slice, ldk = row stride, fp32 partials).  Needs M % 128 == 0 (rows are not
// clamped: LDS-DMA), N % 128 == 0, K % 64 == 0.
constexpr int kG128Stages = 4, kG128Stage = (128 + 128) * 64 * 2, kG128Lds = kG128Stages * kG128Stage;
template <int EPI, int ODT>
__global__ __launch_bounds__(256, 1) void gemm128s_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ ls, const bf16* resid, void* out_, int M, int N, int K, int tiles_n, int nwg, int ldk, RopeArgs rope)
{
    constexpr int BM = 128, BK = 64, MF = 4, NF = 4, RS = kG128Stages, STAGE = kG128Stage, RPP = 8, PA = 4, PW = 4;
    A += (size_t)blockIdx.y * K;
    Wt += (size_t)blockIdx.y * K;
    void* out = ODT == FVHD_F32 ? (void*)((float*)out_ + (size_t)blockIdx.y * M * N) : out_;
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    const int L = xcd_remap(blockIdx.x, nwg);               // v1's super-tile order
    constexpr int GN = 8;
    const int tiles_m = nwg / tiles_n;
    const int grp = L / (tiles_m * GN), rem = L - grp * tiles_m * GN;
    const int wg = min(GN, tiles_n - grp * GN);
    const int tm = rem / wg, tn = grp * GN + (rem - tm * wg);
    const int m0 = tm * BM, n0 = tn * 128;

    // lane l of a 1-KiB piece (8 rows x 128 B) fetches chunk (l & 7) ^ swizzle(row) of row l >> 3; the key depends on the piece's parity only
    constexpr int GRP = EpiGrp<NF, EPI, ODT>::value;       // W rows permuted inside the tile: see EpiGrp
    const int rip = lane >> 3;
    unsigned va[2], vw[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int ch = ((lane & 7) ^ lds_swz<BK>(par * RPP + rip)) * 8;
        va[par] = (unsigned)((rip * ldk + ch) * 2);
        vw[par] = (unsigned)((wpiece_lane_row<GRP>(rip) * ldk + ch) * 2);
    }
    const char* abase = (const char*)(A + (size_t)(m0 + wave * RPP * PA) * ldk);
    const char* wbase = (const char*)(Wt + (size_t)n0 * ldk);
    const unsigned lds0 = lds_addr(lds2);
    auto issue = [&](int kt) {
        const unsigned st = lds0 + (kt % RS) * STAGE;
        const size_t ko = (size_t)kt * BK * 2;
#pragma unroll
        for (int j = 0; j < PA; ++j) glds_piece(va[j & 1], abase + (size_t)j * RPP * ldk * 2 + ko, st + (wave * PA + j) * 1024);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int pi = wave * PW + j;                       // wave-uniform (PW = 4: parity of pi = parity of j)
            glds_piece(vw[pi & 1], wbase + (size_t)wpiece_row<GRP>(pi) * ldk * 2 + ko, st + BM * BK * 2 + pi * 1024);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
#pragma unroll
    for (int i = 0; i < RS - 1; ++i)
        if (i < nk) issue(i);
    for (int kt = 0; kt < nk; ++kt) {
        // own pieces of tile kt have landed when at most the later tiles' pieces are outstanding (loads only in this loop; in order)
        const int later = nk - 1 - kt < RS - 2 ? nk - 1 - kt : RS - 2;
        if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PA + PW)) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // tile kt visible to every wave; tile kt - 1 fully consumed
        if (kt + RS - 1 < nk) issue(kt + RS - 1);     // into the stage tile kt - 1 occupied
        const char* ldsA = lds2 + (kt % RS) * STAGE;
        const char* ldsW = ldsA + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MF], wf[NF];
            const int ks = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(ldsA + lds_off<BK>(wm * 64 + i * 16 + lr, ks));
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[j] = *(const bf16x8*)(ldsW + lds_off<BK>(wn * 64 + j * 16 + lr, ks));
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    if constexpr (EPI == EPI_BIAS_ROPE) gemm_epilogue_rope<MF>(acc, bias, (bf16*)out, rope, M, N, m0 + wm * 64, n0 + wn * 64, lr, g);
    else gemm_epilogue<MF, NF, EPI, ODT, true>(acc, bias, ls, resid, out, M, N, m0 + wm * 64, n0 + wn * 64, lr, g);
}

// v4 (round 3, experiment): "ping-pong" 256 x 256 tile.  The 8 waves form two groups (rows 0-127 / 128-255 of the tile; one wave of each
// group per SIMD) that run ONE PHASE APART: while a group issues the fragment reads of a half K-tile (12 ds_read_b128 per wave) and its
// LDS-DMA pieces, the other group runs the 32 MFMAs of its own half K-tile out of registers, and a barrier ends every phase - each
// SIMD's matrix pipe always has a wave in its MFMA phase, the LDS / DMA work hides behind the partner's MFMAs (the schedule of
// cdna_hip_programming.md's 8-phase template, with half-K-tile phases).  Phases of group 0: L(t,0) M(t,0) L(t,1) M(t,1) = 4t .. 4t+3;
// group 1 is shifted by one.  K-tile t lives in stage t % 2 (64 KB: A 256 x 64, W 256 x 64, same XOR-swizzled image as v2 / v3);
// its last reader is group 1's L(t,1) in phase 4t+3, so the DMA of tile t+2 into the same stage is issued in phases 4t+4 / 4t+5 (4 + 4
// pieces per wave) and waited for (vmcnt(0)) before the barrier that ends phase 4t+7 - at least two phases of flight.
template <int EPI, int ODT>
__global__ __launch_bounds__(512) void gemm_pp_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ ls, const bf16* resid, void* out, int M, int N, int K, int tiles_n, int nwg)
{
    constexpr int BM = 256, BN = 256, BK = 64, MF = 8, NF = 4, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;               // group = M half; four waves side by side in N
    const int lr = lane & 15, g = lane >> 4;
    const int L = xcd_remap(blockIdx.x, nwg);
    constexpr int GN = 4;
    const int tiles_m = nwg / tiles_n;
    const int sg = L / (tiles_m * GN), rem = L - sg * tiles_m * GN;
    const int wgw = min(GN, tiles_n - sg * GN);
    const int tm = rem / wgw, tn = sg * GN + (rem - tm * wgw);
    const int m0 = tm * BM, n0 = tn * BN;

    // DMA: K-tile = 32 A pieces + 32 W pieces of 1 KiB (8 rows x 128 B); wave w owns A pieces 4w .. 4w+3 and W pieces 4w .. 4w+3
    constexpr int GRP = EpiGrp<NF, EPI, ODT>::value;       // W rows permuted inside the tile: see EpiGrp
    const int rip = lane >> 3;
    unsigned vo[2], vwo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int sw = ((par * 8 + rip) >> 1) & 7;
        vo[par] = (unsigned)((rip * K + (((lane & 7) ^ sw) * 8)) * 2);
        vwo[par] = (unsigned)((wpiece_lane_row<GRP>(rip) * K + (((lane & 7) ^ sw) * 8)) * 2);
    }
    const char* abase = (const char*)(A + (size_t)(m0 + wave * 32) * K);
    const char* wbase = (const char*)(Wt + (size_t)n0 * K);
    const unsigned lds0 = lds_addr(lds2);
    auto issue4 = [&](int kt, int which) {                  // which = 0: this wave's 4 A pieces, 1: its 4 W pieces
        const unsigned st = lds0 + (kt & 1) * STAGE + (which ? 256 * 128 : 0) + wave * 4096;
        if (which) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                glds_piece(vwo[j & 1], wbase + (size_t)wpiece_row<GRP>(wave * 4 + j) * K * 2 + (size_t)kt * BK * 2, st + j * 1024);
        } else {
            const char* base = abase + (size_t)kt * BK * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) glds_piece(vo[j & 1], base + (size_t)j * 8 * K * 2, st + j * 1024);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[MF], wf[NF];

    const int nk = K / BK;
    issue4(0, 0); issue4(0, 1);
    if (nk > 1) { issue4(1, 0); issue4(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one phase behind: this is its phase 0 (idle)

    auto load_phase = [&](int kt, int h) {                  // fragment reads of half K-tile (kt, h): k-slots 4h .. 4h+3
        const char* ldsA = lds2 + (kt & 1) * STAGE;
        const char* ldsW = ldsA + 256 * 128;
        const int ks = h * 4 + g;
#pragma unroll
        for (int j = 0; j < NF; ++j) wf[j] = *(const bf16x8*)(ldsW + lds_off<BK>(wn * 64 + j * 16 + lr, ks));
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *(const bf16x8*)(ldsA + lds_off<BK>(grp * 128 + i * 16 + lr, ks));
    };
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // End of a phase: own LDS reads complete (the stage may be overwritten two barriers from now), optionally own DMA complete.
#define PP_END(WAIT_VM)                                                                           \
    do {                                                                                          \
        if (WAIT_VM) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)

    for (int kt = 0; kt < nk; ++kt) {
        // DMA schedule (tiles 0 and 1 came with the prologue).  Tile kt + 1 goes into the stage of tile kt - 1, free from global phase
        // 4kt on; it must have landed by the barrier that ends phase 4kt + 3.  Group 0 (phases 4kt, 4kt+1 = its L / M(kt, 0)) issues its A
        // pieces in L(kt, 0) and its W pieces in M(kt, 0); group 1 (phases 4kt, 4kt+1 = its M(kt-1, 1) / L(kt, 0)) issues the A pieces of
        // tile kt + 2 in M(kt, 1) and the W pieces of tile kt + 1 in L(kt, 0): every piece has two phases of flight before its wait.
        const bool d1 = kt >= 1 && kt + 1 < nk, d2 = kt + 2 < nk;
        // ---- L(kt, 0)
        load_phase(kt, 0);
        if (d1) issue4(kt + 1, grp == 0 ? 0 : 1);
        PP_END(false);
        // ---- M(kt, 0)
        mfma_phase();
        if (d1 && grp == 0) issue4(kt + 1, 1);
        PP_END(false);
        // ---- L(kt, 1)
        load_phase(kt, 1);
        PP_END(grp == 1);                                   // group 1: A(kt+1) (issued in M(kt-1, 1)) and W(kt+1) have landed
        // ---- M(kt, 1)
        mfma_phase();
        if (d2 && grp == 1) issue4(kt + 2, 0);
        PP_END(grp == 0);                                   // group 0: A(kt+1), W(kt+1) have landed
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();             // group 0 matches group 1's extra phase
#undef PP_END
    gemm_epilogue<MF, NF, EPI, ODT, true>(acc, bias, ls, resid, out, M, N, m0 + grp * 128, n0 + wn * 64, lr, g);
}

#ifndef FVHD_GEMM_PERSIST_DEFAULT
#define FVHD_GEMM_PERSIST_DEFAULT 0     // the streaming 256 x (128 | 256) kernels as persistent workgroups (measured in profiles/r05_gemm_persist.log)
#endif
// Compute units of the current device (round 5, advisor: the dispatch rules below were written as multiples of MI355X's 256 CUs; a
// partitioned mode - CPX / DPX - or another part changes the count, and with it where "one round of tiles" ends).  Kernel choice only:
// every kernel gives identical bits.
static int cu_count()
{
    static int cached[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n = cached[dev & 63];
    if (n <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

#ifdef FVHD_DEBUG_KNOBS                 // A/B runs (tools/bench_ops.py gemm on libfvhd_ablate.so): 0 = always the v1 kernel
static int g_gemm_v2 = 1;
extern "C" void fvhd_debug_set_gemm_v2(int on) { g_gemm_v2 = on; }
#else
static constexpr int g_gemm_v2 = 1;
#endif
static const int g_gemm_nf6 = [] { const char* e = getenv("FVHD_GEMM_NF6"); return e ? atoi(e) : 1; }();    // A/B switch (round 6)

template <int EPI, int ODT, int NWV, int BN = 128, int BKT = 64, int ABL = 0>
static hipError_t launch_gemm256(hipStream_t st, const bf16* A, const bf16* Wt, const float* bias, const float* ls,
                                 const bf16* resid, void* out, int M, int N, int K)
{
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<EPI, ODT, NWV, BN, BKT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, G2Cfg<BN, BKT, NWV>::LDS);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const int tiles_m = M / 256, tiles_n = N / BN, nwg = tiles_m * tiles_n;
    constexpr int LDSB = G2Cfg<BN, BKT, NWV>::LDS;
    // persistent form (one workgroup per CU walks the tiles, see the kernel): FVHD_GEMM_PERSIST=1 / 0 in the environment overrides the default
    static const int persist = [] { const char* ev = getenv("FVHD_GEMM_PERSIST"); return ev ? atoi(ev) : FVHD_GEMM_PERSIST_DEFAULT; }();
    int grid = nwg;
    if (persist && ABL == 0) {
        const int g8 = cu_count() & ~7;                        // a multiple of the 8 XCDs: tile t stays on XCD t % 8
        if (g8 >= 8 && nwg > g8) grid = g8;
    }
    hipLaunchKernelGGL((gemm256_kernel<EPI, ODT, NWV, BN, BKT, ABL>), dim3(grid), dim3(64 * NWV), LDSB, st, A, Wt, bias, ls, resid, out, M, N, K, tiles_n, nwg);
    return hipGetLastError();
}

template <int EPI, int ODT>
static hipError_t launch_gemm_pp(hipStream_t st, const bf16* A, const bf16* Wt, const float* bias, const float* ls,
                                 const bf16* resid, void* out, int M, int N, int K)
{
    constexpr int LDS = 2 * 512 * 128;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, ODT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const int tiles_m = M / 256, tiles_n = N / 256, nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, ODT>), dim3(nwg), dim3(512), LDS, st, A, Wt, bias, ls, resid, out, M, N, K, tiles_n, nwg);
    return hipGetLastError();
}

static hipError_t dispatch_gemm_pp(hipStream_t st, const bf16* a, const bf16* w, const float* bias, const float* ls, const bf16* r, void* out,
                                   int M, int N, int K, int epi)
{
    switch (epi) {
    case EPI_NONE: return launch_gemm_pp<EPI_NONE, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS: return launch_gemm_pp<EPI_BIAS, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS_GELU: return launch_gemm_pp<EPI_BIAS_GELU, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS_LS_RESID: return launch_gemm_pp<EPI_BIAS_LS_RESID, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_RESID: return launch_gemm_pp<EPI_RESID, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_SWIGLU: return launch_gemm_pp<EPI_SWIGLU, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    }
    return hipErrorInvalidValue;
}

template <int BN>
static hipError_t dispatch_gemm256(hipStream_t st, const bf16* a, const bf16* w, const float* bias, const float* ls, const bf16* r, void* out,
                                   int M, int N, int K, int epi)
{
    switch (epi) {
    case EPI_NONE: return launch_gemm256<EPI_NONE, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS: return launch_gemm256<EPI_BIAS, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS_GELU: return launch_gemm256<EPI_BIAS_GELU, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_BIAS_LS_RESID: return launch_gemm256<EPI_BIAS_LS_RESID, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_RESID: return launch_gemm256<EPI_RESID, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    case EPI_SWIGLU: return launch_gemm256<EPI_SWIGLU, FVHD_BF16, 8, BN>(st, a, w, bias, ls, r, out, M, N, K);
    }
    return hipErrorInvalidValue;
}

// (knobs 5 / 6 / 7 of round 4 - K tiles of 32 with two workgroups per CU or 4- / 6-stage rings - measured slower everywhere
// (profiles/r03_gemm_2wg.log, r04_gemm_bk32_rings.log) and were removed in round 5 together with their dispatchers.)

// 128 x 128 streaming kernel (v1s).  K = the K range of ONE slice, ldk = the row stride, splits = gridDim.y (1 for a plain GEMM)
template <int EPI, int ODT>
static hipError_t launch_gemm128s(hipStream_t st, const bf16* A, const bf16* Wt, const float* bias, const float* ls, const bf16* resid, void* out,
                                  int M, int N, int K, int splits, int ldk, RopeArgs rope = RopeArgs{})
{
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm128s_kernel<EPI, ODT>, hipFuncAttributeMaxDynamicSharedMemorySize, kG128Lds);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    const int tiles_m = M / 128, tiles_n = N / 128, nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm128s_kernel<EPI, ODT>), dim3(nwg, splits), dim3(256), kG128Lds, st, A, Wt, bias, ls, resid, out, M, N, K, tiles_n, nwg, ldk, rope);
    return hipGetLastError();
}

static hipError_t dispatch_gemm128s(hipStream_t st, const bf16* a, const bf16* w, const float* bias, const float* ls, const bf16* r, void* out,
                                    int M, int N, int K, int epi)
{
    switch (epi) {
    case EPI_NONE: return launch_gemm128s<EPI_NONE, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    case EPI_BIAS: return launch_gemm128s<EPI_BIAS, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    case EPI_BIAS_GELU: return launch_gemm128s<EPI_BIAS_GELU, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    case EPI_BIAS_LS_RESID: return launch_gemm128s<EPI_BIAS_LS_RESID, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    case EPI_RESID: return launch_gemm128s<EPI_RESID, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    case EPI_SWIGLU: return launch_gemm128s<EPI_SWIGLU, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K, 1, K);
    }
    return hipErrorInvalidValue;
}

// the fp32 partials [splits][M][N] of a split-K GEMM: v1, or v1s when its shape rules hold and the launch is small enough for it
// measured (tools/bench_ops.py gemmsmall, profiles/r04_gemm_v1s.log): ahead of v1 up to ~200 workgroups - prefill q|k|v 16.9 -> 15.0 us, unsplit
// down_proj 63.4 -> 52.6, the tower at B = 1: stage-2 fc2 24.6 -> 21.3, stage-4 qkv / fc1 22.0 / 24.9 -> 19.0 / 21.5 - behind it from ~300 on
// (v1 then has 2-4 workgroups per CU covering each other's latency: stage-2 fc1 at B = 1, 384 tiles, 16.1 -> 21.0), and behind v1 for the
// slices of a split-K launch with 250-500 workgroups (llm down / 4: 39.7 -> 43.0)
#ifndef FVHD_GEMM128S_ROUNDS
#define FVHD_GEMM128S_ROUNDS 1             // v1s is taken up to this many workgroups per CU in one launch (0 = never): one round of one workgroup per CU
#endif
static bool take_gemm128s(int M, int N, int K, long workgroups)
{
    if (M % 128 || N % 128 || K % 64 || K < 128) return false;
    if (g_gemm_v2 == 10) return true;
    return g_gemm_v2 == 1 && workgroups <= (long)FVHD_GEMM128S_ROUNDS * cu_count();
}

static hipError_t launch_splitk_partials(hipStream_t st, const void* A, const void* Wt, float* partial, int M, int N, int K, int splits)
{
    const int tiles_m = (M + 127) / 128, tiles_n = N / 128, nwg = tiles_m * tiles_n, ks = K / splits;
    if (take_gemm128s(M, N, ks, (long)nwg * splits))
        return launch_gemm128s<EPI_NONE, FVHD_F32>(st, (const bf16*)A, (const bf16*)Wt, nullptr, nullptr, nullptr, (void*)partial, M, N, ks, splits, K);
    hipLaunchKernelGGL((gemm_kernel<4, 64, EPI_NONE, FVHD_F32>), dim3(nwg, splits), dim3(256), 0, st, (const bf16*)A, (const bf16*)Wt, nullptr, nullptr,
                       nullptr, (void*)partial, M, N, ks, tiles_n, nwg, K);
    return hipGetLastError();
}

template <int NF, int BK, int EPI, int ODT>
static hipError_t launch_gemm(hipStream_t st, const bf16* A, const bf16* Wt, const float* bias, const float* ls,
                              const bf16* resid, void* out, int M, int N, int K)
{
    constexpr int BM = 128, BN = 32 * NF;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_kernel<NF, BK, EPI, ODT>), dim3(nwg), dim3(256), 0, st,
                       A, Wt, bias, ls, resid, out, M, N, K, tiles_n, nwg, K);
    return hipGetLastError();
}

// ---- split-K: out = resid + A . Wt^T for SMALL M x N with a LONG K (Qwen2 down_proj at prefill: 2304 x 896 x 4864 = 126 tiles of
// 76 serial K steps on 256 CUs).  `splits` slices of K run as independent workgroups of the v1 kernel writing fp32 partials
// [splits][M][N]; a second kernel sums them in slice order (deterministic), adds the residual and rounds once to bf16.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const bf16* resid, bf16* out, long mn, int splits)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    f32x4 v = *(const f32x4*)(part + i);
    for (int s = 1; s < splits; ++s) v += *(const f32x4*)(part + (size_t)s * mn + i);
    if (resid) v += bf4_to_f32(*(const bf16x4*)(resid + i));
    *(bf16x4*)(out + i) = f32_to_bf4(v);
}

// The same reduce for rows that are whole hidden states (N = the model width), followed by the RMSNorm the NEXT operation starts with
// (Qwen2DecoderLayer: x = x + down_proj(...) / x + o_proj(...), then input_layernorm / post_attention_layernorm of x): one wave per row
// writes out = bf16(resid + sum of the slices) and nout = norm_w * out * rsqrt(mean(out^2) + eps).  Summation order, the rounding of `out`
// before the statistics and the order of the sum of squares are those of splitk_reduce_kernel followed by rmsnorm_kernel (llm.hip): the
// two outputs are bit-identical to the two separate launches - one launch and one pass over the row less per decoder layer.
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* __restrict__ part, const bf16* resid, bf16* out, const float* __restrict__ nw,
                                                                 bf16* __restrict__ nout, int M, int N, int splits, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t mn = (size_t)M * N, ro = (size_t)row * N;
    float ss = 0.f;
    for (int c = lane * 8; c < N; c += 512) {
        f32x4 v0 = *(const f32x4*)(part + ro + c), v1 = *(const f32x4*)(part + ro + c + 4);
        for (int s = 1; s < splits; ++s) { v0 += *(const f32x4*)(part + s * mn + ro + c); v1 += *(const f32x4*)(part + s * mn + ro + c + 4); }
        if (resid) { v0 += bf4_to_f32(*(const bf16x4*)(resid + ro + c)); v1 += bf4_to_f32(*(const bf16x4*)(resid + ro + c + 4)); }
        const bf16x4 b0 = f32_to_bf4(v0), b1 = f32_to_bf4(v1);
        *(bf16x4*)(out + ro + c) = b0;
        *(bf16x4*)(out + ro + c + 4) = b1;
        const f32x4 r0 = bf4_to_f32(b0), r1 = bf4_to_f32(b1);
#pragma unroll
        for (int k = 0; k < 4; ++k) ss = __builtin_fmaf(r0[k], r0[k], ss);
#pragma unroll
        for (int k = 0; k < 4; ++k) ss = __builtin_fmaf(r1[k], r1[k], ss);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)N + eps);
    for (int c = lane * 8; c < N; c += 512) {               // the lane re-reads its own stores
        const f32x4 r0 = bf4_to_f32(*(const bf16x4*)(out + ro + c)), r1 = bf4_to_f32(*(const bf16x4*)(out + ro + c + 4));
        const f32x4 w0 = *(const f32x4*)(nw + c), w1 = *(const f32x4*)(nw + c + 4);
        f32x4 o0, o1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o0[k] = r0[k] * rstd * w0[k]; o1[k] = r1[k] * rstd * w1[k]; }
        *(bf16x4*)(nout + ro + c) = f32_to_bf4(o0);
        *(bf16x4*)(nout + ro + c + 4) = f32_to_bf4(o1);
    }
}

// A [M, K] bf16, Wt [N, K] bf16, resid bf16 [M, N] or null (may alias out), out [M, N] bf16, partial: fp32 scratch [splits][M][N].
// K % (64 * splits) == 0, N % 128 == 0.  norm_w != null (fp32 [N], with norm_out [M, N] bf16, not aliasing out): the RMSNorm of the
// finished rows as well (splitk_reduce_norm_kernel).
extern "C" int fvhd_launch_gemm_splitk_norm(hipStream_t st, const void* A, const void* Wt, const void* resid, void* out, float* partial,
                                            int M, int N, int K, int splits, const float* norm_w, void* norm_out, float eps)
{
    if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || N % 128 || K % (64 * splits) || !partial) return (int)hipErrorInvalidValue;
    if ((norm_w == nullptr) != (norm_out == nullptr) || (norm_out && norm_out == out)) return (int)hipErrorInvalidValue;
    hipError_t e = launch_splitk_partials(st, A, Wt, partial, M, N, K, splits);
    if (e != hipSuccess) return (int)e;
    const long mn = (long)M * N;
    if (norm_w)
        hipLaunchKernelGGL(splitk_reduce_norm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, partial, (const bf16*)resid, (bf16*)out, norm_w,
                           (bf16*)norm_out, M, N, splits, eps);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, partial, (const bf16*)resid, (bf16*)out, mn, splits);
    return (int)hipGetLastError();
}

// The tower's residual GEMMs (ConvFFN.fc2, MHSA.proj: out = resid + ls * (A . Wt^T + bias), EPI_BIAS_LS_RESID) at SMALL batches: a handful of
// output tiles with a long K (B = 1, stage 2: 96 tiles of 24 K steps on 256 CUs; stage 4: 24 tiles of 96).  Same split as above, the reduce applies
// the epilogue in the epilogue's own order.
__global__ __launch_bounds__(256) void splitk_reduce_ls_kernel(const float* __restrict__ part, const float* __restrict__ bias, const float* __restrict__ ls,
                                                               const bf16* resid, bf16* out, long mn, int N, int splits)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    const int n = (int)(i % N);
    f32x4 v = *(const f32x4*)(part + i);
    for (int s = 1; s < splits; ++s) v += *(const f32x4*)(part + (size_t)s * mn + i);
    v = v + *(const f32x4*)(bias + n);
    const f32x4 r = bf4_to_f32(*(const bf16x4*)(resid + i));
    v = r + *(const f32x4*)(ls + n) * v;
    *(bf16x4*)(out + i) = f32_to_bf4(v);
}

// how many K slices fvhd_launch_gemm_splitk_ls should use for this shape (1 = the plain GEMM is the better launch): the largest power of two
// that keeps tiles x slices within two workgroups per CU and leaves every slice at least 6 K steps of 64, for K >= 2048 (stage-2 fc2 at B = 1,
// K = 1536: 23.5 us split in four against 21.3 us on the streaming 128 x 128 kernel)
extern "C" int fvhd_gemm_splitk_plan(int M, int N, int K)
{
    if (M <= 0 || N % 128 || K % 64) return 1;
    const long tiles = (long)((M + 127) / 128) * (N / 128);
    // at most two workgroups per CU - and never more (tile, slice) pairs than the caller's partial buffer holds (512: fvhd_api.hip kSplitKPartialBytes)
    const long cap = 2l * cu_count() < 512 ? 2l * cu_count() : 512;
    for (int sp = 16; sp > 1; sp >>= 1)
        if (tiles * sp <= cap && (K / 64) % sp == 0 && K / sp >= 384 && K >= 2048) return sp;     // (K < 2048: the reduce costs more than the K steps saved - v1s unsplit)
    return 1;
}

// A [M, K], Wt [N, K] bf16, bias / ls fp32 [N], resid [M, N] bf16 (may alias out), out [M, N] bf16; partial: fp32 scratch [splits][M][N]
extern "C" int fvhd_launch_gemm_splitk_ls(hipStream_t st, const void* A, const void* Wt, const float* bias, const float* ls, const void* resid, void* out,
                                          float* partial, int M, int N, int K, int splits)
{
    if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || N % 128 || K % (64 * splits) || !partial || !bias || !ls || !resid) return (int)hipErrorInvalidValue;
    hipError_t e = launch_splitk_partials(st, A, Wt, partial, M, N, K, splits);
    if (e != hipSuccess) return (int)e;
    const long mn = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_ls_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, partial, bias, ls, (const bf16*)resid, (bf16*)out,
                       mn, N, splits);
    return (int)hipGetLastError();
}

// only the partials [splits][M][N] (fp32): for a caller with its own reduce (llm.hip: bias + rotary embedding behind the q|k|v projection)
extern "C" int fvhd_launch_gemm_splitk_partials(hipStream_t st, const void* A, const void* Wt, float* partial, int M, int N, int K, int splits)
{
    if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || N % 128 || K % (64 * splits) || !partial) return (int)hipErrorInvalidValue;
    return (int)launch_splitk_partials(st, A, Wt, partial, M, N, K, splits);
}

extern "C" int fvhd_launch_gemm_splitk(hipStream_t st, const void* A, const void* Wt, const void* resid, void* out, float* partial,
                                       int M, int N, int K, int splits)
{
    return fvhd_launch_gemm_splitk_norm(st, A, Wt, resid, out, partial, M, N, K, splits, nullptr, nullptr, 0.f);
}

// q|k|v projection with the rotary embedding and the KV-cache copies in its epilogue (EPI_BIAS_ROPE): 1 if this shape takes it - head_dim 64
// (a wave's 64-column block = one head) on the streaming 128 x 128 kernel's shape rules; the caller launches projection + rope_kernel otherwise
extern "C" int fvhd_gemm_qkv_rope_supported(int Mp, int N, int K, int HD, int nh, int nkv)
{
    if (HD != 64 || Mp <= 0 || nh <= 0 || nkv <= 0 || N != (nh + 2 * nkv) * HD) return 0;
    return take_gemm128s(Mp, N, K, (long)(Mp / 128) * (N / 128)) ? 1 : 0;
}

// A [Mp, K] bf16, Wt [N, K] bf16 (q | k | v rows), bias fp32 [N], out [Mp, N] bf16; pos int64 [M] or null (= row % T), table fp32 [P][HD/2][2],
// kcache / vcache [B][nkv][T][HD] bf16 or both null; M <= Mp real rows
extern "C" int fvhd_launch_gemm_qkv_rope(hipStream_t st, const void* A, const void* Wt, const float* bias, void* out, int Mp, int N, int K,
                                         const long* pos, const float* table, void* kcache, void* vcache, int M, int T, int nh, int nkv, int HD, int P, float theta)
{
    if (!A || !Wt || !bias || !out || !table || M <= 0 || M > Mp || T <= 0 || P <= 0 || !(theta > 0.f) || (kcache == nullptr) != (vcache == nullptr))
        return (int)hipErrorInvalidValue;
    if (!fvhd_gemm_qkv_rope_supported(Mp, N, K, HD, nh, nkv)) return (int)hipErrorInvalidValue;
    const RopeArgs r{pos, table, (bf16*)kcache, (bf16*)vcache, M, T, nh, nkv, P, theta};
    return (int)launch_gemm128s<EPI_BIAS_ROPE, FVHD_BF16>(st, (const bf16*)A, (const bf16*)Wt, bias, nullptr, nullptr, out, Mp, N, K, 1, K, r);
}

template <int NF, int BK>
static hipError_t dispatch_epi(hipStream_t st, const bf16* A, const bf16* Wt, const float* bias, const float* ls,
                               const bf16* resid, void* out, int M, int N, int K, int epi, int odt)
{
    if (odt == FVHD_BF16) {
        switch (epi) {
        case EPI_NONE: return launch_gemm<NF, BK, EPI_NONE, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        case EPI_BIAS: return launch_gemm<NF, BK, EPI_BIAS, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        case EPI_BIAS_GELU: return launch_gemm<NF, BK, EPI_BIAS_GELU, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        case EPI_BIAS_LS_RESID: return launch_gemm<NF, BK, EPI_BIAS_LS_RESID, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        case EPI_RESID: return launch_gemm<NF, BK, EPI_RESID, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        case EPI_SWIGLU: return launch_gemm<NF, BK, EPI_SWIGLU, FVHD_BF16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        }
    } else if (epi == EPI_BIAS) {
        if (odt == FVHD_F16) return launch_gemm<NF, BK, EPI_BIAS, FVHD_F16>(st, A, Wt, bias, ls, resid, out, M, N, K);
        if (odt == FVHD_F32) return launch_gemm<NF, BK, EPI_BIAS, FVHD_F32>(st, A, Wt, bias, ls, resid, out, M, N, K);
    } else if (epi == EPI_NONE && odt == FVHD_F32) {                 // fp32 logits of the lm_head
        return launch_gemm<NF, BK, EPI_NONE, FVHD_F32>(st, A, Wt, bias, ls, resid, out, M, N, K);
    }
    return hipErrorInvalidValue;
}

// A [M,K] bf16, Wt [N,K] bf16, bias/ls fp32 [N], resid bf16 [M,N] (may alias out), out [M,N] of out_dtype.
// Requirements: K % 32 == 0, N % 16 == 0.  Non-bf16 outputs only with epi == EPI_BIAS (f16 / f32) and EPI_NONE (f32).
// EPI_SWIGLU: out is [M, N/2].
extern "C" int fvhd_launch_gemm(hipStream_t st, const void* A, const void* Wt, const float* bias, const float* ls,
                                const void* resid, void* out, int M, int N, int K, int epi, int out_dtype)
{
    if (M <= 0 || N <= 0 || K <= 0 || (K % 32) || (N % 16)) return (int)hipErrorInvalidValue;
    const bf16* a = (const bf16*)A;
    const bf16* w = (const bf16*)Wt;
    const bf16* r = (const bf16*)resid;
    // streaming kernel where it has at least two full rounds of tiles (one 8-wave workgroup per CU): measured against v1 at B = 32
    // (tools/bench_ops.py gemm, profiles/r02_gemm_v1_v2.log) fc1 272 -> 264 us, fc2 (K = 3072) 215 -> 196, stage-5 qkv 177 -> 155,
    // projector fc 68 -> 56; with 384 tiles (stage-5 proj / fc2, 1.5 rounds) v1 stays ahead.  A 4-wave variant with 128 x 64 per
    // wave (fewer fragment reads, one wave per SIMD) was 10-25 % slower: nothing covers the LDS read latency.
    // g_gemm_v2 (debug build; 10 = the 128 x 128 streaming kernel v1s wherever legal, split-K partials included): 0 = v1 only, 1 = the rule below, 2 = the 256 x 128 tile wherever it is legal, 3 = the 256 x 256 tile wherever legal,
    // 4 = the ping-pong 256 x 256 kernel wherever legal, 5 = 256 x 128 / 4 waves / BK 32 / two workgroups per CU wherever legal
    // v1s: launches of at most ~one 128 x 128 tile per CU (the prefill's q|k|v projection, the tower's GEMMs at B = 1): three K tiles in flight
    if (out_dtype == FVHD_BF16 && epi >= EPI_NONE && epi <= EPI_SWIGLU && take_gemm128s(M, N, K, (long)(M / 128) * (N / 128)))
        return (int)dispatch_gemm128s(st, a, w, bias, ls, r, out, M, N, K, epi);
    if (g_gemm_v2 && out_dtype == FVHD_BF16 && M % 256 == 0 && N % 128 == 0 && K % 64 == 0 && K >= 128) {
        const long long t128 = (long long)(M / 256) * (N / 128), t256 = N % 256 == 0 ? (long long)(M / 256) * (N / 256) : 0;
#ifdef FVHD_DEBUG_KNOBS
        if ((g_gemm_v2 == 8 || g_gemm_v2 == 9) && t256 > 0 && (epi == EPI_NONE || epi == EPI_BIAS_GELU)) {     // ablations of the 256 x 256 kernel
            if (g_gemm_v2 == 8) return (int)(epi == EPI_NONE ? launch_gemm256<EPI_NONE, FVHD_BF16, 8, 256, 64, 1>(st, a, w, bias, ls, r, out, M, N, K)
                                                              : launch_gemm256<EPI_BIAS_GELU, FVHD_BF16, 8, 256, 64, 1>(st, a, w, bias, ls, r, out, M, N, K));
            return (int)(epi == EPI_NONE ? launch_gemm256<EPI_NONE, FVHD_BF16, 8, 256, 64, 2>(st, a, w, bias, ls, r, out, M, N, K)
                                         : launch_gemm256<EPI_BIAS_GELU, FVHD_BF16, 8, 256, 64, 2>(st, a, w, bias, ls, r, out, M, N, K));
        }
#endif
        // measured (tools/bench_ops.py gemm, profiles/r03_gemm_tiles.log, B = 32): the 256 x 256 tile wins from N = 2304 on when it has
        // ~2 rounds of tiles - stage-3 qkv 186 -> 165 us, fc1 268 -> 242, stage-4 fc1 224 -> 204, 7B projector 239 / 267 -> 211 / 239 -
        // ties or loses below (N = 768: 64 -> 74 us) and with 1.3 rounds (prefill gate|up, 342 tiles: 59 -> 65).  All three kernels
        // produce identical bits (same K order per output element).
        const long long ncu = cu_count();              // 256 on MI355X: the measured thresholds below are 1.75 / 2 / 4 / 0.5 rounds of it
        const bool use256 = g_gemm_v2 == 3 ? t256 > 0 : (g_gemm_v2 == 1 && N >= 2304 && t256 * 4 >= ncu * 7);
        // 256 x 128 (one workgroup per CU): from two rounds of tiles on, unless the last round is mostly empty (B = 8 stage-3 qkv, 576 tiles = 2.25
        // rounds: v1 46.9 vs 52.1 us; the prefill's gate|up, 684 tiles = 2.67 rounds, stays: 58.7 vs 62.3) - and already from half a round when K is
        // long (K >= 3072: B = 8 stage-3 fc2 64.7 -> 55.5 us, the 0.5B projector's fc 66.5 -> 54.7; profiles/r04_gemm_small_batch.log, r03_gemm_tiles.log)
        const long long rounds128 = (t128 + ncu - 1) / ncu;
        const bool full_enough = t128 >= 4 * ncu || t128 * 5 >= rounds128 * ncu * 4;            // >= 80 % of the slots of its rounds
        const bool use128 = g_gemm_v2 == 2 || (g_gemm_v2 == 1 && ((t128 >= 2 * ncu && full_enough) || (t128 * 2 >= ncu && K >= 3072)));
        // ping-pong kernel: same tile, +0-6 % over the plain 256 x 256 kernel, the more the longer K (profiles/r03_gemm_tiles.log: stage-4 fc2
        // K = 6144 185 -> 177 us, 7B projector K = 3584 243 -> 228 us = 0.92 PF/s; K = 768 shapes tie) - taken for K >= 3072
        if ((g_gemm_v2 == 4 && t256 > 0) || (g_gemm_v2 == 1 && K >= 3072 && t256 * 2 >= ncu))
            return (int)dispatch_gemm_pp(st, a, w, bias, ls, r, out, M, N, K, epi);
        if (use256) return (int)dispatch_gemm256<256>(st, a, w, bias, ls, r, out, M, N, K, epi);
        if (use128) return (int)dispatch_gemm256<128>(st, a, w, bias, ls, r, out, M, N, K, epi);
    }
    const bool nf3 = (N % 128 != 0) && (N % 96 == 0);
    const bool bk64 = (K % 64 == 0);
    hipError_t e;
    // N = 192 with many row tiles (PatchEmbed's 1x1 + GELU after stage 0: M = B * 16384): ONE 128 x 192 tile per row block instead of two
    // 128 x 96 tiles - A is read once and the tile count halves; same K order per output element (identical bits).  Round 6, FVHD_GEMM_NF6=0: off.
    if (g_gemm_nf6 && nf3 && bk64 && N % 192 == 0 && epi == EPI_BIAS_GELU && out_dtype == FVHD_BF16 && (long long)((M + 127) / 128) * (N / 192) >= 4ll * cu_count())
        return (int)launch_gemm<6, 64, EPI_BIAS_GELU, FVHD_BF16>(st, a, w, bias, ls, r, out, M, N, K);
    if (nf3) e = bk64 ? dispatch_epi<3, 64>(st, a, w, bias, ls, r, out, M, N, K, epi, out_dtype)
                      : dispatch_epi<3, 32>(st, a, w, bias, ls, r, out, M, N, K, epi, out_dtype);
    else     e = bk64 ? dispatch_epi<4, 64>(st, a, w, bias, ls, r, out, M, N, K, epi, out_dtype)
                      : dispatch_epi<4, 32>(st, a, w, bias, ls, r, out, M, N, K, epi, out_dtype);
    return (int)e;
}
