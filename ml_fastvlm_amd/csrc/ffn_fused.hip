// Fused ConvFFN MLP:  X <- X + ls * ( gelu(A . W1^T + b1) . W2^T + b2 )        (in place on X)
//
//   ConvFFN.fc1 -> GELU -> fc2 (mci.py:922-926) + layer scale + residual (mci.py:1106-1109 / 1187-1188);
//   A is the output of the (BatchNorm-folded) depthwise 7x7 (mci.py:921).
//
// 94 % of the encoder's FLOPs are these two 1x1 GEMMs.  Run as two kernels, the [M, 4C] hidden tensor
// makes stages 0-1 HBM-bound and its bias + erf-GELU epilogue is as long as the GEMM main loop at
// K = C <= 384.  Here the hidden activations never leave the register file ("flash-MLP"):
//
//   * a wave owns a block of 32 rows (pixels); the row block A^T lives in registers as the B operands of
//     v_mfma_f32_32x32x16_bf16 (C/16 fragments), the output block O^T[C x 32] as C/32 fp32 accumulator tiles.
//   * per chunk of 32 hidden units: S^T[32h x 32m] = W1chunk . A^T (C/16 MFMAs, accumulator pre-loaded
//     with b1) -> erf-GELU in fp32 -> bf16.  The C/D layout leaves lane (m = lane&31, half =
//     lane>>5) with hidden units h = (r&3) + 8(r>>2) + 4*half; regs 0-7 / 8-15 are, as they stand, the
//     B operands of the two K=16 steps of O^T += W2chunk . P^T for the hidden order
//     k-slot (kb, half, j) <-> h = 16kb + 8(j>>2) + 4half + (j&3).  The host stores W2 with its hidden
//     axis pre-permuted into exactly that order, so P needs no cross-lane movement and no LDS trip.
//   * software pipeline, skewed by two chunks, so that the VALU work of the GELU runs in the issue
//     shadow of MFMAs that do not depend on it:   iteration t issues
//         GEMM1(chunk t)  ||  GELU(chunk t-1)  ||  GEMM2(chunk t-2).
//   * W1 / W2 chunk images (64*C bytes each) are pre-swizzled on the host into the exact byte order the
//     LDS wants (16-B slot XOR swizzles that make the fragment ds_read_b128 conflict-free for the lane
//     groups of MI355X_MICROARCH "LDS"), so staging is a linear LDS-DMA copy
//     (global_load_lds_dwordx4: no VGPRs, no ds_write pass), double buffered per matrix, issued one
//     iteration ahead; one barrier per chunk.  The chunk stream is CYCLIC: chunk NCH-1 is followed by chunk 0.
//   * HBM traffic per row: A in (2C B), X in/out (4C B) - the algorithmic minimum; weights come from L2.
//
// Round 2 (what the round-1 profile asked for: 26-43 % of the kernel was A / X-in / X-out streaming with the MFMA pipe idle):
// the big launches run PERSISTENT workgroups, one per CU, that keep the chunk pipeline going across row tiles:
//
//   ffn_persist8_kernel (C = 96, 192; 8 waves = two per SIMD, 256-row tiles, <= 256 registers per lane)
//     - the pipeline never drains: GEMM1 of tile k+1's first chunks runs beside the GELU / GEMM2 of tile k's last chunks
//       (the hidden sum is order-free and the weight stream cyclic, so a tile may start at any time);
//     - tile k+1's A rows arrive by LDS-DMA into a per-wave LDS region while tile k computes (fully coalesced 1-KiB pieces,
//       XOR-swizzled on the SOURCE side so that the fragment ds_read_b128 is conflict-free); X of tile k follows into
//       the same region once the A^T fragments are in registers, the finished tile is rounded once, written back into
//       that region in MFMA layout and leaves as coalesced 16-B stores; nothing of it waits for HBM inside the loop;
//     - eight waves share one weight stream (round 1: two 4-wave workgroups per CU each streamed their own copy: the
//       L2 -> LDS path, 28-41 B/clk/CU, was as busy as the MFMA pipe).
//   ffn_persist4_kernel (C = 384; 4 waves = one per SIMD: the 496-register accumulator file leaves no second wave and
//       96 KB of weight ring leaves no room for a 96-KB row tile in LDS)
//     - the weight stream keeps running across tiles (no W1[0] refetch bubble), X rows are loaded in MFMA layout two
//       iterations before the epilogue, the next tile's A^T fragments are loaded as the epilogue retires registers, and
//       both are L2-prefetched half a tile ahead by 4-B-per-line LDS-DMA "touch" loads.
//   Small launches (M < 65536 rows: not enough tiles to fill 256 persistent workgroups) keep the one-tile-per-workgroup kernel.
#include "fvhd_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C> __host__ __device__ __forceinline__ int w1_off(int row, int slot)      // W1 chunk [32][C] bf16, slot = 16-B index in the row
{
    if constexpr (C == 384) return row * 768 + ((slot ^ (row & 15)) << 4);
    else if constexpr (C == 192) return row * 384 + ((slot ^ ((row >> 1) & 7)) << 4);
    else return row * 192 + ((slot ^ ((row >> 2) & 3)) << 4);            // C == 96
}

__host__ __device__ __forceinline__ int w2_off(int row, int slot)                        // W2 chunk [C][32] bf16 (64-B rows)
{
    return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
}

// erf GELU (fvhd_common.h: Phi(x) = 0.5 + xc Q(xc^2), degree-7 Q) on PAIRS of hidden values: every step is one v_pk_*_f32,
// so a pair costs 10 packed + 2 v_med3 issue slots (6 per value; + one packed add where the bias is not already in the
// accumulator).  Cut into six half-stages of 2-3 instructions; the software pipeline below advances one pair by one
// half-stage per unit.
struct GeluSt { f32x2 x, xc, u, q; };
#define FFN_PK(c) (f32x2{c, c})
template <int H> FVHD_DEV void gelu_half(GeluSt& g, f32x2 x, f32x2& out)
{
    if constexpr (H == 0) {
        g.x = x;
        g.xc = f32x2{__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    } else if constexpr (H == 1) {
        g.u = g.xc * g.xc;
        g.q = __builtin_elementwise_fma(FFN_PK(FVHD_GELU_C7), g.u, FFN_PK(FVHD_GELU_C6));
    } else if constexpr (H == 2) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C5));
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C4));
    } else if constexpr (H == 3) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C3));
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C2));
    } else if constexpr (H == 4) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C1));
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU_C0));
    } else {
        out = g.x * __builtin_elementwise_fma(g.xc, g.q, FFN_PK(0.5f));
    }
}

// 16 B/lane LDS-DMA (global -> LDS, no VGPR staging): LDS destination = wave-uniform byte address `lds_dst` + lane*16.
// Issued from inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in the loop hipcc's waitcnt pass
// degrades every counted lgkmcnt(N) of the fragment ds_reads to lgkmcnt(0) (measured: 96 counted waits without the
// builtin, 0 with it), which exposes the full LDS latency before every MFMA.  hipcc does not count an asm load, so the
// consumer side waits explicitly: s_waitcnt vmcnt(0) before the barrier that publishes the images (ffn_wait_dma).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
FVHD_DEV void ffn_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One pipeline iteration t, for the NB 32-row blocks a wave owns:
//     s_out <- b1(t) + GEMM1(chunk t);   p_out <- GELU(s_in = S(t-1));   O += GEMM2(chunk t-2, p_in = P(t-2)).
// The iteration is cut into NM = 2*KS*NB slots of one MFMA each.  Fragment f (f even: W1 k-step f/2, f odd: W2 fragment
// f/2) is read from LDS once and feeds NB consecutive slots (one per row block), so consecutive MFMAs never share an
// accumulator, and the instruction order is pinned slot by slot with sched_barrier:
//     slot m = { MFMA m | LDS fragment read (PF fragments ahead) | one 1-KiB LDS-DMA piece of next iteration's weights
//                (some slots) | 96*NB/NM GELU half-stages (2-3 VALU each, from independent dependency chains) }
// i.e. every 32-cycle MFMA carries a handful of independent single-issue fillers - what one wave can issue in its shadow
// (MI355X_MICROARCH "one wave per SIMD") - instead of 200+ VALU in a lump between two MFMA bursts.
template <int C, int NB, int WAVES, bool DO_A, bool DO_B, bool DO_C, bool DO_DMA, int VAR, int PF, bool BPRE = (C == 96), int DW = WAVES>
FVHD_DEV void ffn_iter(const bf16x8 (&afr)[NB][C / 16], f32x16 (&o)[NB][C / 32], f32x16 (&s_out)[NB], const f32x16 (&s_in)[NB],
                       bf16x8 (&p_out)[NB][2], const bf16x8 (&p_in)[NB][2], const char* const (&w1p)[C / 48],
                       const char* const (&w2p)[2], const int ring, const float* b1_cur, const float* b1_prev, int half,
                       const char* dma_src1, unsigned dma_dst1, bool dma1, const char* dma_src2, unsigned dma_dst2, bool dma2, int uwave,
                       unsigned dma_mask = ~0u)
{
    constexpr int KS = C / 16, NF = 2 * KS, NM = NF * NB;   // fragments, MFMA slots
    constexpr int NA = C / 48, CHB = 64 * C, NG = CHB / 1024;
    constexpr int UPS = 96 * NB / NM;       // GELU half-stage units per slot (16 values x 6 half-stages per block) = 48 / KS
    constexpr int TP = 2 * NG, NPW = (TP + DW - 1) / DW;   // 1-KiB DMA pieces per iteration (W1 then W2), per issuing wave (waves 0 .. DW-1)
    // VAR (ablation / tuning bits; 0 in production): 1 no weight DMA, 2 no GELU math, 4 no GEMM2 MFMAs, 8 no GEMM1 MFMAs,
    // 16 instruction order not pinned (hipcc schedules), 32 all DMA pieces issued in the first slots, 64 scalar (unpacked) GELU
    constexpr bool PIN = !(VAR & 16);
    constexpr int DSTRIDE = (VAR & 32) ? 1 : ((NM / 2) / NPW > 0 ? (NM / 2) / NPW : 1);   // DMA issue spread over the first half
    // fragment addresses = per-lane pointer (swizzle resolved once per kernel) + compile-time immediate:
    //   W1 k-step ks: w1p[ks % NA] + (ks / NA) * NA * 32;   W2 n-fragment nf, k-step kk: w2p[kk] + nf * 2048
#define FFN_LD_W1(ks) (*(const bf16x8*)(w1p[(ks) % NA] + ring * CHB + ((ks) / NA) * NA * 32))
#define FFN_LD_W2(f) (*(const bf16x8*)(w2p[(f) & 1] + ring * CHB + ((f) >> 1) * 2048))
#define FFN_LD(f) (((f) & 1) ? FFN_LD_W2((f) >> 1) : FFN_LD_W1((f) >> 1))
#define FFN_LD_ON(f) (((f) & 1) ? DO_C : DO_A)
    bf16x8 wf[NF];
    f32x4 bv[4];
    GeluSt gs[NB][8];
    f32x2 gout[NB][8];
    // b1: either the GEMM1 accumulator starts from it (BPRE: 16 more live registers at the top of the iteration, no VALU), or
    // it is added in the first GELU half-stage (value r <-> hidden (r&3) + 8(r>>2) + 4*half)
    if constexpr (DO_A && BPRE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)(b1_cur + 8 * q + 4 * half);
    }
    if constexpr (DO_B && !BPRE) bv[0] = *(const f32x4*)(b1_prev + 4 * half);
#pragma unroll
    for (int f = 0; f < PF; ++f)
        if (FFN_LD_ON(f)) wf[f] = FFN_LD(f);
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int f = m / NB, nb = m % NB;
        if ((f & 1) == 0) {
            if constexpr (DO_A) {
                const int ks = f >> 1;
                if constexpr (VAR & 8) asm volatile("" ::"v"(wf[f]));      // ablation: no GEMM1 MFMA
                else if (ks == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = BPRE ? bv[r >> 2][r & 3] : 0.0f;
                    s_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], afr[nb][ks], z, 0, 0, 0);
                } else s_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], afr[nb][ks], s_out[nb], 0, 0, 0);
            }
        } else {
            if constexpr (DO_C) {
                const int g = f >> 1;
                if constexpr (VAR & 4) asm volatile("" ::"v"(wf[f]), "v"(p_in[nb][g & 1]));   // ablation: no GEMM2 MFMA
                else o[nb][g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], p_in[nb][g & 1], o[nb][g >> 1], 0, 0, 0);
            }
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);  // the MFMA opens the slot; everything below issues in its shadow
        if (nb == 0 && f + PF < NF && FFN_LD_ON(f + PF)) wf[f + PF] = FFN_LD(f + PF);
        if constexpr (DO_DMA && !(DW == WAVES && TP % DW == 0)) {
            // Persistent 8-wave kernels: waves 0 .. DW-1 each own NPW CONSECUTIVE pieces of the flat list (W1's NG pieces, then
            // W2's), issued as one run right behind the first MFMA: the whole iteration to land, 8 + NPW instructions in all.
            // NO BRANCH (the iteration must stay one basic block: its order is pinned with sched_barrier; with control flow in it
            // LLVM sank a whole iteration's GELU into the loop latch): a wave without pieces issues the run with EXEC = 0.
            static_assert(TP % DW == 0 && NG % NPW == 0 && (NPW == 2 || NPW == 4), "whole runs inside one matrix");
            if (m == 0) {
                const int flat0 = uwave * NPW;                           // wave-uniform (uwave lives in an SGPR)
                const bool second = flat0 >= NG;
                const int pc0 = second ? flat0 - NG : flat0;
                glds16_run_masked<NPW>((second ? dma_src2 : dma_src1) + pc0 * 1024, (threadIdx.x & 63) * 16,
                                       (second ? dma_dst2 : dma_dst1) + (unsigned)pc0 * 1024u, dma_mask);
            }
        } else if constexpr (DO_DMA) {      // next iteration's weight images, one 1-KiB piece per DSTRIDE slots
            if (m % DSTRIDE == 0 && m / DSTRIDE < NPW) {
                const int flat = (m / DSTRIDE) * DW + uwave;
                const int lane16 = (threadIdx.x & 63) * 16;
                const bool second = flat >= NG;
                const int pc = second ? flat - NG : flat;
                const char* src = (second ? dma_src2 : dma_src1) + pc * 1024 + lane16;
                const unsigned dst = (second ? dma_dst2 : dma_dst1) + (unsigned)pc * 1024u;
                if (second ? dma2 : dma1) glds16(src, dst);              // compile-time or loop-invariant flags only
            }
        }
        if constexpr (DO_B) {               // GELU; value r of block gb <-> hidden h = (r&3) + 8(r>>2) + 4*half
            if constexpr (!BPRE) {
#pragma unroll
                for (int q = 1; q < 4; ++q)     // bias of values 4q..4q+3 (first used by unit 24*q*NB): read ~2 slots ahead
                    if (m == ((24 * q * NB) / UPS >= 2 ? (24 * q * NB) / UPS - 2 : 0)) bv[q] = *(const f32x4*)(b1_prev + 8 * q + 4 * half);
            }
#pragma unroll
            for (int u = m * UPS / 2; u < (m + 1) * UPS / 2; ++u) {     // unit u = ((pair j, half-stage h), block gb)
                const int gb = u % NB, jh = u / NB, h = jh % 6, r = 2 * (jh / 6);
                f32x2 sv = {s_in[gb][r], s_in[gb][r + 1]};
                if (!BPRE && (h == 0 || ((VAR & (2 | 64)) && h == 5))) sv += f32x2{bv[r >> 2][r & 3], bv[r >> 2][(r & 3) + 1]};
                if (VAR & 2) {               // ablation bit 1: no GELU math
                    if (h == 5) gout[gb][r >> 1] = sv;
                } else {
                    if constexpr (VAR & 64) {
                        if (h == 5) gout[gb][r >> 1] = f32x2{gelu_erf(sv[0]), gelu_erf(sv[1])};
                    } else
                    if (h == 0) gelu_half<0>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                    else if (h == 1) gelu_half<1>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                    else if (h == 2) gelu_half<2>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                    else if (h == 3) gelu_half<3>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                    else if (h == 4) gelu_half<4>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                    else gelu_half<5>(gs[gb][r >> 1], sv, gout[gb][r >> 1]);
                }
                if (h == 5 && (r & 3) == 2) {
                    const bf16x4 pk = f32_to_bf4(f32x4{gout[gb][(r >> 1) - 1][0], gout[gb][(r >> 1) - 1][1], gout[gb][r >> 1][0], gout[gb][r >> 1][1]});
#pragma unroll
                    for (int j = 0; j < 4; ++j) p_out[gb][r >> 3][(r & 4) + j] = pk[j];
                }
            }
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
    // anchor: the values this iteration produces are "used" here, inside its block, so that no pass can sink their
    // producers (the GELU chain above all) below the block - e.g. into the loop latch, out of the MFMA shadow
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if constexpr (DO_B) asm volatile("" : "+v"(p_out[nb][0]), "+v"(p_out[nb][1]));
        if constexpr (DO_A) asm volatile("" : "+v"(s_out[nb]));
    }
#undef FFN_LD_W1
#undef FFN_LD_W2
#undef FFN_LD
#undef FFN_LD_ON
}

// chunk id of a 32-row tile (= row * CPR + 16-B chunk in the row; the tile is contiguous in HBM, so id*16 is also its byte
// offset) -> LDS byte offset of the transposing stage.  The XOR key (id >> SH) & SWZ equals (3*row + const) mod 2^SH for
// the rows of one fragment read (CPR = 3 * 2^SH), a bijection on row mod 2^SH: the 16 lanes of a ds_read_b128 group land
// in 16 different 16-B slots; it only touches the low SH bits, so aligned groups of 64 chunks (the 1-KiB pieces of an
// LDS-DMA) map onto themselves and the map is an involution (used for the DMA's source-side swizzle).
template <int C> FVHD_DEV int ffn_slot_of(int id)
{
    constexpr int SH = C == 384 ? 4 : C == 192 ? 3 : 2, SWZ = (1 << SH) - 1;
    return (id ^ ((id >> SH) & SWZ)) << 4;
}

// ---------------------------------------------------------------------------------------------------------------------
// One tile (32 * WAVES rows) per workgroup: the small-launch path (and the round-1 structure).
template <int C, int NB, int WAVES, int VAR = 0, int PF = 3, int OCC = WAVES / 4>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void ffn_fused_kernel(
    const bf16* __restrict__ A, const char* __restrict__ w1img, const char* __restrict__ w2img,
    const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ ls,
    bf16* X, int M, int nwg)
{
    constexpr int HID = 4 * C, KS = C / 16, NFR = C / 32, NCH = HID / 32;
    constexpr int CHB = 64 * C;             // bytes of one W1 (or W2) chunk image
    constexpr int NG = CHB / 1024;          // 1-KiB DMA pieces per chunk image
    static_assert(NCH % 2 == 0, "pipeline is unrolled by two");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [W1 ring: 2 x CHB][W2 ring: 2 x CHB][b1 fp32 HID]   - one array: see cdna_hip_programming.md 5 item 4(a)
    char* w1ring = smem;
    char* w2ring = smem + 2 * CHB;
    float* lb1 = (float*)(smem + 4 * CHB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int blk = xcd_remap(blockIdx.x, nwg);
    const int row0 = blk * (32 * NB * WAVES) + wave * (32 * NB);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);          // provably uniform: DMA bases stay in SGPRs / M0
    const unsigned lds_w1 = __builtin_amdgcn_readfirstlane(lds_addr(w1ring)), lds_w2 = __builtin_amdgcn_readfirstlane(lds_addr(w2ring));

    for (int i = tid; i < HID / 4; i += WAVES * 64) *(f32x4*)&lb1[i * 4] = *(const f32x4*)&b1[i * 4];

    // per-lane fragment pointers (ring slot 0), see ffn_iter
    const char* w1p[C / 48];
    const char* w2p[2];
#pragma unroll
    for (int k = 0; k < C / 48; ++k) w1p[k] = w1ring + w1_off<C>(li, 2 * k + half);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2p[k] = w2ring + w2_off(li, 2 * k + half);

    bf16x8 afr[NB][KS];
    f32x16 o[NB][NFR];
    f32x16 s0[NB], s1[NB];          // s[k&1] holds S(k)
    bf16x8 p0[NB][2], p1[NB][2];    // p[k&1] holds P(k)
    // ---- A^T fragments (B operands).  A wave's 32 rows are 32*C*2 contiguous bytes of HBM, but the MFMA wants lane
    // (row, k-half) to hold 16 B of ITS row: loading that directly is 32 rows x 32 B per instruction (a quarter of every
    // 128-B line per request).  For C <= 192 the tile is read fully coalesced (lane L of load i takes 16-B chunk i*64+L)
    // and transposed through LDS - the (still empty) weight rings - with a 16-B-slot XOR swizzle that keeps both sides
    // conflict-free.
    static_assert(NB == 1 && WAVES == 4, "the LDS transpose uses one quarter of the 4*CHB ring area per wave");
    constexpr int CPR = C / 8;               // 16-B chunks per row
    constexpr int NL = 32 * CPR / 64;        // coalesced 16-B loads per lane for one 32-row tile (== KS)
    char* stage = smem + wave * (32 * C * 2);
    const size_t tile_b = (size_t)row0 * C * 2, last_b = (size_t)M * C * 2 - 16;     // rows >= M: any valid address (never stored)
    constexpr bool CIO = C <= 192;           // C = 384: measured slower (register spills in the edges, serialised W1[0] DMA)
    if constexpr (CIO) {
        u32x4 t[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t gb = tile_b + (size_t)(i * 64 + lane) * 16;
            t[i] = *(const u32x4*)((const char*)A + (gb < last_b ? gb : last_b));
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) *(u32x4*)(stage + ffn_slot_of<C>(i * 64 + lane)) = t[i];
        // same wave wrote and reads: LDS operations of one wave are processed in order, no barrier needed
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[0][ks] = *(const bf16x8*)(stage + ffn_slot_of<C>(li * CPR + ks * 2 + half));
    } else {
        // straight from global: lane reads 16 B of its own row per k-step
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int m_ld = min(row0 + nb * 32 + li, M - 1);
        const bf16* arow = A + (size_t)m_ld * C + half * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[nb][ks] = *(const bf16x8*)(arow + ks * 16);
    }
    }
    if constexpr (CIO) __syncthreads();      // every wave has its fragments: the rings may now receive weights
#pragma unroll
    for (int g = 0; g < (NG + WAVES - 1) / WAVES; ++g) {             // W1[0]
        const int piece = g * WAVES + uwave;
        if (NG % WAVES == 0 || piece < NG)
            glds16(w1img + piece * 1024 + lane * 16, lds_w1 + piece * 1024);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int i = 0; i < NFR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nb][i][r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[nb][r] = 0.f; s1[nb][r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { p0[nb][0][r] = 0; p0[nb][1][r] = 0; p1[nb][0][r] = 0; p1[nb][1][r] = 0; }
    }

    // Iteration t reads W1[t] (ring slot t&1) and W2[t-2] (ring slot t&1); it issues the DMA of W1[t+1] and W2[t-1] into the
    // other slots, which every wave finished reading before this iteration's barrier.  FFN_SYNC = vmcnt(0) (the DMA is
    // invisible to hipcc's own counting) + barrier: the images issued during the previous iteration have landed and are
    // visible to every wave.  (w1img carries one zero chunk past the end, so W1[t+1] is always a valid source.)
#define FFN_DMA_ARGS(t) w1img + (size_t)((t) < NCH ? (t) + 1 : 0) * CHB, lds_w1 + (((t) + 1) & 1) * CHB, (t) < NCH, \
                        w2img + (size_t)((t) >= 1 ? (t) - 1 : 0) * CHB, lds_w2 + (((t) + 1) & 1) * CHB, (t) >= 1, uwave
#define FFN_SYNC() ffn_wait_dma(); __syncthreads()
#define FFN_B1PREV(cur) ((cur) == lb1 ? lb1 + (NCH - 1) * 32 : (cur) - 32)     // b1 of the chunk before `cur` (cyclic)
    FFN_SYNC();
    ffn_iter<C, NB, WAVES, true, false, false, true, VAR, PF>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(0));
    FFN_SYNC();
    ffn_iter<C, NB, WAVES, true, true, false, true, VAR, PF>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + 32, FFN_B1PREV(lb1 + 32), half, FFN_DMA_ARGS(1));
#pragma unroll 1
    for (int t = 2; t < NCH; t += 2) {
        FFN_SYNC();                  // even t: S(t) -> s0, GELU(s1) -> p1, GEMM2 reads p0
        ffn_iter<C, NB, WAVES, true, true, true, true, VAR, PF>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1 + t * 32, FFN_B1PREV(lb1 + t * 32), half, FFN_DMA_ARGS(t));
        FFN_SYNC();                  // odd t:  S(t) -> s1, GELU(s0) -> p0, GEMM2 reads p1
        ffn_iter<C, NB, WAVES, true, true, true, true, VAR, PF>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + (t + 1) * 32, FFN_B1PREV(lb1 + (t + 1) * 32), half, FFN_DMA_ARGS(t + 1));
    }
    // the residual tile of X is fetched now, coalesced like A^T above (the A^T registers are dead from here on), so that
    // its HBM latency hides behind the last two pipeline iterations instead of stalling the epilogue
    u32x4 xq[CIO ? NL : 1];
    bf16x4 xres[NB][NFR][4];
    if constexpr (CIO) {
        int xln = lane;
        asm volatile("" : "+v"(xln));
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t gb = tile_b + (size_t)(i * 64 + xln) * 16;
            xq[i] = *(const u32x4*)((const char*)X + (gb < last_b ? gb : last_b));
        }
    } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const bf16* xrd = X + (size_t)min(row0 + nb * 32 + li, M - 1) * C + 4 * half;
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int q = 0; q < 4; ++q) xres[nb][nf][q] = *(const bf16x4*)(xrd + nf * 32 + 8 * q);
    }
    }
    FFN_SYNC();                      // t = NCH (even): GELU(S(NCH-1) in s1) -> p1, GEMM2(chunk NCH-2) reads p0; DMA W2[NCH-1]
    ffn_iter<C, NB, WAVES, false, true, true, true, VAR, PF>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(NCH));
    FFN_SYNC();                      // t = NCH + 1: GEMM2(chunk NCH-1) reads p1
    ffn_iter<C, NB, WAVES, false, false, true, false, VAR, PF>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(NCH));
#undef FFN_DMA_ARGS

    // ---- epilogue.  The accumulators are in MFMA layout (lane = row li, 4 consecutive channels n0 = nf*32 + 8q + 4*half per
    // register quad); X and the output want the coalesced layout.  Two trips through the (now idle) ring area of this wave:
    // X: coalesced registers -> LDS -> MFMA layout;  x + ls * (o + b2) in fp32, ONE rounding to bf16 -> LDS -> coalesced store.
    if constexpr (CIO) {
    __syncthreads();                         // all waves are done reading the weight rings
    // per-lane offsets of the epilogue from an OPAQUE copy of the lane id: otherwise LLVM hoists ~40 address registers above the
    // chunk loop and spills accumulators around them (116 B of scratch per lane at C = 192: +117 MB of HBM writes per launch in
    // the PMC pass of profiles/r02a)
    int eln = lane;
    asm volatile("" : "+v"(eln));
    const int eli = eln & 31, ehalf = eln >> 5;
#pragma unroll
    for (int i = 0; i < NL; ++i) *(u32x4*)(stage + ffn_slot_of<C>(i * 64 + eln)) = xq[i];
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n0 = nf * 32 + 8 * q + 4 * ehalf;
            char* slot = stage + ffn_slot_of<C>(eli * CPR + nf * 4 + q) + ehalf * 8;
            const f32x4 bv = *(const f32x4*)(b2 + n0), lv = *(const f32x4*)(ls + n0);
            const f32x4 rv = bf4_to_f32(*(const bf16x4*)slot);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = rv[j] + lv[j] * (o[0][nf][4 * q + j] + bv[j]);
            *(bf16x4*)slot = f32_to_bf4(v);
        }
    const int rows_ok = M - row0;            // rows of this tile that exist
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int id = i * 64 + eln;
        const u32x4 v = *(const u32x4*)(stage + ffn_slot_of<C>(id));
        if (id < rows_ok * CPR) *(u32x4*)((char*)X + tile_b + (size_t)id * 16) = v;
    }
    } else {
    // ---- epilogue: lane holds out[m][n0 .. n0+3], n0 = nf*32 + 8q + 4*half -------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int m_row = row0 + nb * 32 + li;
        if (m_row < M) {
            bf16* xr = X + (size_t)m_row * C;
#pragma unroll
            for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = nf * 32 + 8 * q + 4 * half;
                    const f32x4 bv = *(const f32x4*)(b2 + n0), lv = *(const f32x4*)(ls + n0);
                    const f32x4 rv = bf4_to_f32(xres[nb][nf][q]);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rv[j] + lv[j] * (o[nb][nf][4 * q + j] + bv[j]);
                    *(bf16x4*)(xr + n0) = f32_to_bf4(v);
                }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent, 8 waves (two per SIMD), 256-row tiles, continuous chunk pipeline: C = 96 / 192 (header).
// Workgroup g owns tiles g, g + G, g + 2G, ...; wave w owns rows [tile*256 + 32w, +32) of each.
//
// Who waits for what.  The first version of this kernel issued the tile traffic (A rows in, X rows in, rows out) from every
// wave and kept the per-iteration "s_waitcnt vmcnt(0); s_barrier" of the weight ring: correct, and no faster than the
// one-tile kernel - an ablation showed the run time to be the SUM of the HBM time (4.2 TB/s for the algorithmic bytes at
// C = 96), the MFMA time, the GELU time and the weight-DMA time.  vmcnt(0) waits for everything a wave has in flight, so each
// boundary iteration stalled the whole (lock-stepped) chip for an HBM round trip and HBM idled in between.  Now:
//   * waves 0-5 issue the weight pieces (L2-resident, ~0.5-0.9 us) and only those: their vmcnt(0) never sees HBM latency;
//   * wave 7 ("aux") issues no weight piece; it trickles the tile traffic of ALL waves as LDS-DMA pieces, 12 per iteration in
//     fixed windows of the tile period, with up to 60 in flight, and waits with COUNTED vmcnt (its queue holds loads only,
//     which retire in order) at exactly two points per tile and phase group - before the barrier that precedes the first
//     reader.  Windows whose tile does not exist issue 4-B dummies instead so that the counts stay exact;
//   * the finished rows leave as coalesced 16-B stores right after a barrier (a whole iteration until the next vmcnt(0));
//     wave 7's own rows are stored by wave 6 one barrier later (stores in wave 7's queue would break the counted waits).
//
// LDS: [W1 ring 2 x CHB][W2 ring 2 x CHB][b1, b2, ls fp32][256 B scratch][stage A: RPG x 32 rows x 2C B][stage X: same]
//   C = 96:  one phase group, 8 regions each (96 KB);   C = 192: 96 + 96 KB do not fit beside the 48-KB ring, so the waves form
//   TWO PHASE GROUPS (waves 0-3 / 4-7: the two waves of every SIMD are in different groups) half a tile apart that time-share
//   4 + 4 regions - which also means that one wave of every SIMD is in steady state while the other runs its epilogue.
//   (Cost: half a tile of idle slots at the start and the end of a launch.)
// Group-local schedule of tile k (t = iteration of the tile; every step below sits right AFTER the barrier of iteration t):
//     t = 0   A^T(k) fragments <- stage A region (aux wave: A window complete before that barrier, vmcnt(48))
//     t = 2   epilogue of tile k-1 (its last GEMM2 ran in iteration 1): X(k-1) in stage X (aux: vmcnt(12)), x + ls (o + b2) ->
//             bf16 -> LDS -> coalesced stores; O <- 0            t = 3   wave 6 stores wave 7's rows
//     aux windows (at the END of iterations, 12 pieces each):  A(k+1) -> stage A in [a0, a0+4),  X(k) -> stage X in [a0+4, a0+8),
//             a0 = 1 (C = 96) / NCH/2 + 1 (C = 192): after the last reader of the region, >= 5 iterations before the first.
#ifdef FVHD_FFN_ABLATE
// VAR & 128 (ablation build): workgroup 0 stamps s_memtime at four points of its first 40 iteration pairs, per wave, into LDS
// and dumps them here at the end:  [wave][pair][0 before the wait, 1 after the barrier, 2 after the even body, 3 after the hooks]
__device__ unsigned g_ffn_stamps[8 * 40 * 4];
extern "C" int fvhd_debug_ffn_stamps(unsigned* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ffn_stamps), sizeof(g_ffn_stamps));
}
#define FFN_STAMP(k) do { if constexpr ((VAR & 128) != 0) { if (blockIdx.x == 0 && it < 80 && (threadIdx.x & 63) == 0) \
    stamps[(uwave * 40 + (it >> 1)) * 4 + (k)] = (unsigned)__builtin_readcyclecounter(); } } while (0)
#else
#define FFN_STAMP(k) do { } while (0)
#endif
template <int C, int VAR = 0, int PF = 3, bool TOUCH = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ffn_persist8_kernel(
    const bf16* __restrict__ A, const char* __restrict__ w1img, const char* __restrict__ w2img,
    const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ ls,
    bf16* X, int M, int ntw)
{
    constexpr int WAVES = 8, DW = 6, STW = 6, AUXW = 7, NB = 1;   // waves 0-5: weight pieces (2 NG is a multiple of 6); 6: row stores; 7: row loads
    constexpr int HID = 4 * C, KS = C / 16, NFR = C / 32, NCH = HID / 32;
    constexpr int CHB = 64 * C;
    constexpr int CPR = C / 8, NL = 32 * CPR / 64;        // 16-B chunks per row; 1-KiB pieces (= coalesced 16-B accesses per lane) per wave tile
    constexpr int WTB = 32 * C * 2;                       // bytes of one wave tile
    constexpr int NGRP = C == 192 ? 2 : 1, RPG = WAVES / NGRP, OFF1 = NCH / 2;
    constexpr int WLEN = 4, PPI = RPG * NL / WLEN;        // aux windows: 4 iterations x 12 pieces
    constexpr int AW0 = (NGRP == 2 ? NCH / 2 : 0) + 1;    // group-local start of the A(k+1) window
    constexpr int XGAP = 2;                               // the X(k) window starts XGAP iterations after the A window ends: the store
                                                          // wave reads the X regions (rows of tile k-1) during iterations 3 .. 6
    constexpr int SH = C == 192 ? 3 : 2, SWZ = (1 << SH) - 1;
    static_assert(NCH % 4 == 0 && NCH >= 12 && RPG * NL == WLEN * PPI && PPI == 12 && AW0 + 2 * WLEN + XGAP <= NCH + 1, "aux schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1ring = smem;
    char* w2ring = smem + 2 * CHB;
    float* lb1 = (float*)(smem + 4 * CHB);
    float* lb2 = lb1 + HID;                               // b2, layer scale: read by the epilogue from LDS (its loads must not queue behind HBM traffic)
    float* lls = lb2 + C;
    char* scratch = smem + 4 * CHB + (HID + 2 * C) * 4;   // 256 B: destination of the dummy loads (never read)
    char* stA = scratch + 256;
    char* stX = stA + RPG * WTB;
#ifdef FVHD_FFN_ABLATE
    unsigned* stamps = (unsigned*)(stX + RPG * WTB);      // 5 KB behind the stage regions (VAR & 128 only: the launcher adds the bytes)
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int grp = NGRP == 2 ? uwave >> 2 : 0, slot = uwave % RPG;
    const int off = grp * OFF1;                           // global iteration at which this wave's tile 0 starts
    const unsigned dma_mask = (unsigned)((uwave - DW) >> 31);   // all ones on waves 0 .. DW-1, else 0: scalar integer ALU, so it stays in an
                                                                // SGPR (a ?: here became v_cndmask and hipcc fed the VGPR to the asm's "s" operand)
    const unsigned lds_w1 = __builtin_amdgcn_readfirstlane(lds_addr(w1ring)), lds_w2 = __builtin_amdgcn_readfirstlane(lds_addr(w2ring));
    const unsigned lds_stA = __builtin_amdgcn_readfirstlane(lds_addr(stA)), lds_stX = __builtin_amdgcn_readfirstlane(lds_addr(stX));
    const unsigned lds_scr = __builtin_amdgcn_readfirstlane(lds_addr(scratch));
    char* stageA = stA + slot * WTB;
    char* stageX = stX + slot * WTB;
    const int G = gridDim.x;
    const int ntl = (ntw - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup (>= 1: the launcher keeps G <= ntw)
    const size_t last_b = (size_t)M * C * 2 - 16;         // rows >= M: any valid address (never stored)

    for (int i = tid; i < HID / 4; i += WAVES * 64) *(f32x4*)&lb1[i * 4] = *(const f32x4*)&b1[i * 4];
    for (int i = tid; i < C / 4; i += WAVES * 64) { *(f32x4*)&lb2[i * 4] = *(const f32x4*)&b2[i * 4]; *(f32x4*)&lls[i * 4] = *(const f32x4*)&ls[i * 4]; }

    const char* w1p[C / 48];
    const char* w2p[2];
#pragma unroll
    for (int k = 0; k < C / 48; ++k) w1p[k] = w1ring + w1_off<C>(li, 2 * k + half);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2p[k] = w2ring + w2_off(li, 2 * k + half);

    // byte offset of the rows wave w owns in workgroup-tile j (j = 0 .. ntl-1)
    auto wtile_b = [&](int j, int w) -> size_t { return ((size_t)((int)blockIdx.x + j * G) * 256 + (size_t)w * 32) * C * 2; };
    // The tile-boundary code below runs once per 2 * NCH iterations but sits inside the hot loop: every per-lane offset it uses
    // is derived from an OPAQUE copy of the lane id taken inside the block, so that LLVM cannot hoist ~60 loop-invariant
    // address registers out of the loop and then spill the accumulators around them (cdna_hip_programming.md, persistent
    // attention pitfalls: "recompute per block").
    auto opaque_lane = [&]() -> int { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };
    // One LDS-DMA piece of a wave tile: LDS slot s = i*64 + lane (linear) <- global chunk s ^ key(s): the source-side swizzle
    // (ffn_slot_of); i may be a run-time (wave-uniform) value.
    auto dma_piece = [&](const bf16* src, size_t tb, int i, int ln, unsigned lds_region) {
        const int sidx = i * 64 + ln;
        const size_t gb = tb + (size_t)((sidx ^ ((sidx >> SH) & SWZ)) << 4);
        glds16((const char*)src + (gb < last_b ? gb : last_b), lds_region + (unsigned)i * 1024u);
    };

    bf16x8 afr[NB][KS];
    f32x16 o[NB][NFR];
    f32x16 s0[NB], s1[NB];
    bf16x8 p0[NB][2], p1[NB][2];

    auto load_afr = [&]() {
        const int ln = opaque_lane();
        const char* base = stageA;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[0][ks] = *(const bf16x8*)(base + ffn_slot_of<C>((ln & 31) * CPR + ks * 2 + (ln >> 5)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the region may be overwritten by an LDS-DMA after the next barrier
    };
    // store wave: NP12 = 12 coalesced 16-B pieces (whole wave tiles) of the finished rows of group g's tile j, pieces q0 .. q0+11
    // of the group's RPG * NL.  LDS slot s = i*64 + lane holds global chunk s ^ key(s) = tb/16 + i*64 + (lane ^ key(lane)).
    auto store_pieces = [&](int g, int j, int q0) {
        const int ln = opaque_lane();
        const unsigned swz16 = (unsigned)((ln ^ ((ln >> SH) & SWZ)) << 4);
        constexpr int RPI = PPI / NL;
#pragma unroll
        for (int rr = 0; rr < RPI; ++rr) {
            const int r = q0 / NL + rr;
            const size_t tb = wtile_b(j, g * RPG + r);
            const char* reg = stX + r * WTB + ln * 16;
            char* dst = (char*)X + tb + swz16;
            if (tb + WTB <= (size_t)M * C * 2) {                      // all 32 rows exist (wave-uniform; false only in the last tile)
#pragma unroll
                for (int i = 0; i < NL; ++i) *(u32x4*)(dst + i * 1024) = *(const u32x4*)(reg + i * 1024);
            } else {
                const long rows_ok = (long)M - (long)(tb / ((size_t)C * 2));      // may be <= 0
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int id = (i * 64 + ln) ^ ((ln >> SH) & SWZ);            // global chunk of LDS slot i*64 + lane
                    const u32x4 v = *(const u32x4*)(reg + i * 1024);
                    if ((long)id < rows_ok * CPR) *(u32x4*)(dst + i * 1024) = v;
                }
            }
        }
    };
    // store wave, after the barrier of global iteration `it`: iterations 3 .. 6 of a group's tile carry the rows of its previous tile
    auto store_issue = [&](int it) {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int phg = it - g * OFF1;
            if (phg < NCH + 3) continue;
            const int tg = phg % NCH, j = phg / NCH - 1;
            if (tg >= 3 && tg < 3 + WLEN && j < ntl) store_pieces(g, j, (tg - 3) * PPI);
        }
    };
    auto zero_o = [&]() {
#pragma unroll
        for (int i = 0; i < NFR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][i][r] = 0.0f;
    };
    // epilogue of workgroup-tile j: X(j) is in stage X.  One output fragment (32 channels) at a time, so that the live set
    // stays small: the whole accumulator file is still in use (the pipeline did not drain).
    auto epilogue = [&](int j) {
        const int ln = opaque_lane();
        const int eli = ln & 31, ehalf = ln >> 5;
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = nf * 32 + 8 * q + 4 * ehalf;
                char* sl = stageX + ffn_slot_of<C>(eli * CPR + nf * 4 + q) + ehalf * 8;
                const f32x4 bv = *(const f32x4*)(lb2 + n0), lv = *(const f32x4*)(lls + n0);
                const f32x4 rv = bf4_to_f32(*(const bf16x4*)sl);
                f32x4 v;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = rv[jj] + lv[jj] * (o[0][nf][4 * q + jj] + bv[jj]);
                *(bf16x4*)sl = f32_to_bf4(v);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][nf][r] = 0.0f;    // O <- 0 for the next tile (its first GEMM2 runs this iteration)
            __builtin_amdgcn_sched_barrier(0);                  // keep the fragments sequential: registers, not latency, are short here
        }
    };
    // aux wave, end of global iteration `it`: this iteration's 12 pieces (or nothing outside the windows)
    auto aux_issue = [&](int it) {
        const int ln = opaque_lane();
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int u = it - g * OFF1 - AW0 + 2 * NCH;          // >= 0 (it >= 0, OFF1 + AW0 <= 2 NCH); shifts the tile index by two
            const int ku = u / NCH - 2, tu = u % NCH;             // window position tu of the period that started in tile ku of group g
            if (tu < WLEN || (tu >= WLEN + XGAP && tu < 2 * WLEN + XGAP)) {
                const bool isA = tu < WLEN;
                const int j = isA ? ku + 1 : ku;                  // A(ku+1) / X(ku)
                const int q0 = (isA ? tu : tu - WLEN - XGAP) * PPI;
                if (j >= 0 && j < ntl) {
                    const bf16* src = isA ? A : X;
                    const unsigned reg0 = isA ? lds_stA : lds_stX;
                    // the swizzle key of chunk i*64 + lane does not depend on i (it reads lane bits only): piece i of a wave
                    // tile = global [tb + 1024 i + 16 (lane ^ key(lane))): one per-lane offset serves every piece
                    const unsigned swz16 = (unsigned)((ln ^ ((ln >> SH) & SWZ)) << 4);
                    constexpr int RPI = PPI / NL > 0 ? PPI / NL : 1;          // whole regions per iteration (C = 96: 2, C = 192: 1)
                    static_assert(PPI % NL == 0 && NL % 2 == 0, "an iteration's pieces are whole wave tiles");
#pragma unroll
                    for (int rr = 0; rr < RPI; ++rr) {
                        const int r = q0 / NL + rr;                           // region (= wave of the group) this iteration fills
                        const size_t tb = wtile_b(j, g * RPG + r);
                        const unsigned ldst = reg0 + (unsigned)r * WTB;
                        if (tb + WTB <= (size_t)M * C * 2) {                  // all 32 rows exist (wave-uniform; false only in the last tile)
                            const char* sb = (const char*)src + tb;
#pragma unroll
                            for (int i0 = 0; i0 < NL; i0 += 4) {
                                if (NL - i0 >= 4) glds16_run<4>(sb + i0 * 1024, swz16, ldst + i0 * 1024);
                                else glds16_run<2>(sb + i0 * 1024, swz16, ldst + i0 * 1024);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < NL; ++i) dma_piece(src, tb, i, ln, ldst);
                        }
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < PPI; ++jj) glds4(w1img + ln * 4, lds_scr);      // keeps the counted waits exact
                }
            }
        }
    };

    // ---- prologue: A(0) of group 0 -> its stage A regions (every wave its own), W1[0] -> ring
    if (grp == 0) {
        const int ln = opaque_lane();
#pragma unroll
        for (int i = 0; i < NL; ++i) dma_piece(A, wtile_b(0, uwave), i, ln, lds_stA + (unsigned)slot * WTB);
    }
    if (uwave < DW) {
        constexpr int NG2 = CHB / 1024;
#pragma unroll
        for (int g = 0; g < (NG2 + DW - 1) / DW; ++g) {
            const int piece = g * DW + uwave;
            if (piece < NG2) glds16(w1img + piece * 1024 + lane * 16, lds_w1 + piece * 1024);
        }
    }
    ffn_wait_dma();
    zero_o();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int r = 0; r < 8; ++r) afr[0][ks][r] = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[0][r] = 0.f; s1[0][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 8; ++r) { p0[0][0][r] = 0; p0[0][1][r] = 0; p1[0][0][r] = 0; p1[0][1][r] = 0; }

    // cyclic weight stream: global iteration `it` works on chunk c = it % NCH: it reads W1[c] and W2[c-2] from ring slot it&1
    // (NCH is even) and issues W1[c+1], W2[c-1] into the other slot
#define FFN_DMA_ARGS(c) w1img + (size_t)(((c) + 1) % NCH) * CHB, lds_w1 + (((c) + 1) & 1) * CHB, true, \
                        w2img + (size_t)(((c) + NCH - 1) % NCH) * CHB, lds_w2 + (((c) + 1) & 1) * CHB, true, uwave, dma_mask
    // ONE iteration body for every wave and every iteration (anything else costs registers at C = 192): a wave whose tile
    // sequence has not started yet, or is over, runs the same MFMAs on stale registers - its partner on the SIMD would have
    // the pipe to itself otherwise, so the launch edges cost about what idle waves would.  Whatever such iterations leave in
    // S / P / O is never stored: O is cleared after the barrier of iteration 2 of every tile (epilogue), P and S are
    // overwritten before they are read for a real tile.
#define FFN_ITER(SO, SI, PO, PI, RING, CC) \
    ffn_iter<C, NB, WAVES, true, true, true, !(VAR & 1), VAR, PF, (C == 96), DW>( \
        afr, o, SO, SI, PO, PI, w1p, w2p, RING, lb1 + (CC) * 32, lb1 + (((CC) + NCH - 1) % NCH) * 32, half, FFN_DMA_ARGS(CC))
    const int ph_end = ntl * NCH;                          // phase (= it - off) at which this wave's last tile ends (+ 2 drain iterations)
    const int it_end = ph_end + 4 + (NGRP - 1) * OFF1;    // ... + the iteration pair that holds the last epilogue
#pragma unroll 1
    for (int it = 0; it < it_end; it += 2) {
        const int ph = it - off, c = it % NCH;             // c is even; this wave's tile iteration t = ph % NCH (= c or c + NCH/2)
        const int t = (ph + NCH) % NCH;
        // ---------------- even iteration
        FFN_STAMP(0);
        if (uwave == AUXW) {                               // counted waits of the aux wave (loads only, in order): see the header
            bool w48 = false, w12 = false;
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
                const int tg = (it - g * OFF1 + NCH) % NCH;
                w48 |= tg == 0;
                w12 |= tg == 2;
            }
            if (w48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            if (w12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (uwave < DW) ffn_wait_dma();             // (the store wave never waits for its stores)
        __syncthreads();
        FFN_STAMP(1);
        if (ph >= 0 && ph < ph_end && t == 0) load_afr();                    // tile ph / NCH starts
        if (ph >= 0 && ph <= ph_end + 2 && t == 2) {
            if (ph >= NCH) epilogue(ph / NCH - 1);                            // the previous tile is complete (also the last one)
            else zero_o();                                                    // first tile: drop what the start-up iterations accumulated
        }
        if (uwave == STW) store_issue(it);
        FFN_ITER(s0, s1, p1, p0, 0, c);
        FFN_STAMP(2);
        if (uwave == AUXW) aux_issue(it);
        FFN_STAMP(3);
        // ---------------- odd iteration
        if (uwave < DW) ffn_wait_dma();
        __syncthreads();
        if (uwave == STW) store_issue(it + 1);
        FFN_ITER(s1, s0, p0, p1, 1, c + 1);
        if (uwave == AUXW) aux_issue(it + 1);
    }
    // the loop ends with iteration 3 of the last group's pseudo-tile ntl: its last tile's rows of iterations 4 .. 6 remain
    if (uwave == STW) {
#pragma unroll 1
        for (int q0 = PPI; q0 < RPG * NL; q0 += PPI) store_pieces(NGRP - 1, ntl - 1, q0);
    }
    ffn_wait_dma();                                        // nothing may be in flight towards this workgroup's LDS when it ends
#ifdef FVHD_FFN_ABLATE
    if constexpr ((VAR & 128) != 0) {
        __syncthreads();
        if (blockIdx.x == 0) for (int i = tid; i < 8 * 40 * 4; i += 512) g_ffn_stamps[i] = stamps[i];
    }
#endif
#undef FFN_ITER
#undef FFN_DMA_ARGS
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent, 4 waves (one per SIMD), 128-row tiles: C = 384 (header).  The pipeline drains at the end of every tile (the
// A^T registers become the residual's, there is no second set), but the weight stream does not stop, the loads are issued
// where their latency hides, and both A(k+1) and X(k) are in L2 when they are asked for: wave 3 touches one 8-KB slice of
// them at the END of every iteration 2 .. 25 of a tile (a 4-B dummy otherwise), so that HBM sees a trickle, not a burst of
// 256 x 192 KB at every tile boundary, and waits with vmcnt(1): loads retire in order, so "all but the newest" = its weight
// pieces of this iteration plus the previous touch, which had a whole iteration (~1.8 us) to come back.
// The residual rows are requested before the two drain iterations and the barrier in between waits with vmcnt(48): the 48
// row loads stay in flight, the weight pieces issued before them are complete (round 1 waited vmcnt(0) there, i.e. for the rows).
template <int C, int VAR = 0, int PF = 3, bool TOUCH = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_persist4_kernel(
    const bf16* __restrict__ A, const char* __restrict__ w1img, const char* __restrict__ w2img,
    const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ ls,
    bf16* X, int M, int ntw)
{
    constexpr int WAVES = 4, NB = 1, AUXW = TOUCH ? 3 : 4, DW = 4;
    constexpr int HID = 4 * C, KS = C / 16, NFR = C / 32, NCH = HID / 32;
    constexpr int CHB = 64 * C, NG = CHB / 1024;
    static_assert(NCH % 2 == 0 && NFR * 4 == 48, "pipeline is unrolled by two; vmcnt(48) = the residual row loads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1ring = smem;
    char* w2ring = smem + 2 * CHB;
    float* lb1 = (float*)(smem + 4 * CHB);
    float* lb2 = lb1 + HID;
    float* lls = lb2 + C;
    char* scratch = smem + 4 * CHB + (HID + 2 * C) * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_w1 = __builtin_amdgcn_readfirstlane(lds_addr(w1ring)), lds_w2 = __builtin_amdgcn_readfirstlane(lds_addr(w2ring));
    const unsigned lds_scr = __builtin_amdgcn_readfirstlane(lds_addr(scratch));
    const int G = gridDim.x;
    const int ntl = (ntw - (int)blockIdx.x + G - 1) / G;
    const size_t last_b = (size_t)M * C * 2 - 16;

    for (int i = tid; i < HID / 4; i += WAVES * 64) *(f32x4*)&lb1[i * 4] = *(const f32x4*)&b1[i * 4];
    for (int i = tid; i < C / 4; i += WAVES * 64) { *(f32x4*)&lb2[i * 4] = *(const f32x4*)&b2[i * 4]; *(f32x4*)&lls[i * 4] = *(const f32x4*)&ls[i * 4]; }
    auto opaque_lane = [&]() -> int { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };   // see ffn_persist8_kernel

    const char* w1p[C / 48];
    const char* w2p[2];
#pragma unroll
    for (int k = 0; k < C / 48; ++k) w1p[k] = w1ring + w1_off<C>(li, 2 * k + half);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2p[k] = w2ring + w2_off(li, 2 * k + half);

    auto row0_of = [&](int j) -> long { return ((long)((int)blockIdx.x + j * G) * 128 + (long)uwave * 32); };
    // aux wave, end of iteration t of tile k: slice (t - 2) of the 24 that cover X(k) (12 x 8 KB, needed first) and A(k+1)
    auto touch_slice = [&](int k, int t) {
        const int j = t - 2, ln = opaque_lane();
        const bool isA = j >= 12;
        const int tile = isA ? k + 1 : k;
        const char* src = w1img + ln * 4;                     // dummy: exactly one load per call keeps vmcnt(1) exact
        if (j >= 0 && j < 24 && tile < ntl) {
            const int w = (j % 12) / 3, part = j % 3;         // wave tile w (24 KB = 192 lines), 64 lines per touch
            const size_t tb = ((size_t)((int)blockIdx.x + tile * G) * 128 + (size_t)w * 32) * C * 2;
            const size_t gb = tb + (size_t)(part * 64 + ln) * 128;
            src = (const char*)(isA ? A : X) + (gb < last_b ? gb : last_b);
        }
        glds4(src, lds_scr);
    };

    bf16x8 afr[NB][KS];
    f32x16 o[NB][NFR];
    f32x16 s0[NB], s1[NB];
    bf16x8 p0[NB][2], p1[NB][2];
    auto load_afr = [&](int j) {             // lane reads 16 B of its own row per k-step (4 consecutive k-steps share a 128-B line)
        const int ln = opaque_lane();
        const long m_ld = min(row0_of(j) + (ln & 31), (long)M - 1);
        const bf16* arow = A + (size_t)m_ld * C + (ln >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) afr[0][ks] = *(const bf16x8*)(arow + ks * 16);
    };

    if (uwave < DW) {
#pragma unroll
        for (int g = 0; g < (NG + DW - 1) / DW; ++g) {               // W1[0]
            const int piece = g * DW + uwave;
            if (piece < NG) glds16(w1img + piece * 1024 + lane * 16, lds_w1 + piece * 1024);
        }
    }
    ffn_wait_dma();                                                  // (also the aux wave: its queue starts empty)
#define FFN_DMA_ARGS(c) w1img + (size_t)(((c) + 1) % NCH) * CHB, lds_w1 + (((c) + 1) & 1) * CHB, true, \
                        w2img + (size_t)(((c) + NCH - 1) % NCH) * CHB, lds_w2 + (((c) + 1) & 1) * CHB, true, uwave
#define FFN_SYNC4() ffn_wait_dma(); __syncthreads()
#define FFN_SYNC4T() if (uwave == AUXW) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else ffn_wait_dma(); __syncthreads()   /* a touch was issued last */
#pragma unroll 1
    for (int k = 0; k < ntl; ++k) {
        load_afr(k);                 // L2 hits from the second tile on (touched half a tile ago); waited for by the barrier below
#pragma unroll
        for (int i = 0; i < NFR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][i][r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[0][r] = 0.f; s1[0][r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { p0[0][0][r] = 0; p0[0][1][r] = 0; p1[0][0][r] = 0; p1[0][1][r] = 0; }
        FFN_SYNC4();                  // W1[0] is in ring slot 0: from the prologue, or issued by iteration NCH-1 of the previous tile
        ffn_iter<C, NB, WAVES, true, false, false, true, VAR, PF, false, DW>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(0));
        FFN_SYNC4();
        ffn_iter<C, NB, WAVES, true, true, false, true, VAR, PF, false, DW>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + 32, FFN_B1PREV(lb1 + 32), half, FFN_DMA_ARGS(1));
#pragma unroll 1
        for (int t = 2; t < NCH; t += 2) {
            if (t == 2) { FFN_SYNC4(); } else { FFN_SYNC4T(); }
            ffn_iter<C, NB, WAVES, true, true, true, !(VAR & 1), VAR, PF, false, DW>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1 + t * 32, FFN_B1PREV(lb1 + t * 32), half, FFN_DMA_ARGS(t));
            if (uwave == AUXW) touch_slice(k, t);
            FFN_SYNC4T();
            ffn_iter<C, NB, WAVES, true, true, true, !(VAR & 1), VAR, PF, false, DW>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + (t + 1) * 32, FFN_B1PREV(lb1 + (t + 1) * 32), half, FFN_DMA_ARGS(t + 1));
            if (uwave == AUXW) touch_slice(k, t + 1);
        }
        // residual rows in MFMA layout (the A^T registers are dead from here on); latency hides behind the two drain iterations
        bf16x4 xres[NFR][4];
        const int eln = opaque_lane(), eli = eln & 31, ehalf = eln >> 5;
        const long m_row = row0_of(k) + eli;
        {
            const bf16* xrd = X + (size_t)min(m_row, (long)M - 1) * C + 4 * ehalf;
#pragma unroll
            for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
                for (int q = 0; q < 4; ++q) xres[nf][q] = *(const bf16x4*)(xrd + nf * 32 + 8 * q);
        }
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");   // NFR * 4 row loads may stay in flight; everything older (weight pieces, touch) is complete
        __syncthreads();              // GELU(S(NCH-1)) -> p1, GEMM2(chunk NCH-2) reads p0; the stream wraps: W1[0] was issued by iteration NCH-1
        ffn_iter<C, NB, WAVES, false, true, true, true, VAR, PF, false, DW>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half,
                                                                  w1img, lds_w1, false, w2img + (size_t)(NCH - 1) * CHB, lds_w2 + CHB, true, uwave);
        FFN_SYNC4();                  // GEMM2(chunk NCH-1) reads p1
        ffn_iter<C, NB, WAVES, false, false, true, false, VAR, PF, false, DW>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(1));
        // ---- epilogue: lane holds out[m][n0 .. n0+3], n0 = nf*32 + 8q + 4*half; then the next tile's A^T fragments
        bf16* xr = X + (size_t)min(m_row, (long)M - 1) * C;
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = nf * 32 + 8 * q + 4 * ehalf;
                const f32x4 bv = *(const f32x4*)(lb2 + n0), lv = *(const f32x4*)(lls + n0);
                const f32x4 rv = bf4_to_f32(xres[nf][q]);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rv[j] + lv[j] * (o[0][nf][4 * q + j] + bv[j]);
                if (m_row < M) *(bf16x4*)(xr + n0) = f32_to_bf4(v);
            }
            __builtin_amdgcn_sched_barrier(0);          // fragment by fragment: the residual / accumulator registers retire as we go
        }
    }
    ffn_wait_dma();
#undef FFN_DMA_ARGS
#undef FFN_SYNC4
#undef FFN_SYNC4T
#undef FFN_SYNC
}

// ---------------------------------------------------------------------------------------------------------------------
// FVHD_FFN_PERSIST (read once): 1 = persistent kernels for launches of >= 65536 rows, 0 (default) = one tile per workgroup.
// Measured (MI355X, B = 32 stage shapes, same box, us): persistent 630 / 474 / 389 vs one-tile 602 / 464 / 387 at C = 96 / 192 /
// 384: correct (tests/test_gpu_ops.py runs both) but not faster yet, and a persistent workgroup owns its CU (150 KB of LDS), so
// the two half-batch streams of fvhd_encode cannot overlap it with anything.  What the s_memtime stamps of the ablation build
// showed (profiles/r02_ffn_persist_notes.md): the iteration body runs at ~50 % MFMA utilisation whatever feeds it (the same as an
// idealised microbenchmark of the chunk loop with the GELU in it, tools/ubench/ffn_mix.hip); every 1-KiB VMEM instruction costs
// its issuing wave ~50 cycles, so the aux wave's 12 pieces are 600 cycles that seven waves wait out at the barrier; the younger
// wave of a SIMD runs ~20 % behind the older one.
static int ffn_persist_mode()
{
    static const int mode = [] { const char* e = getenv("FVHD_FFN_PERSIST"); return e ? atoi(e) : 0; }();
    return mode;
}

template <typename K> static hipError_t ffn_set_lds(K kernel, size_t shmem, bool* done)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done[dev & 63]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e == hipSuccess) done[dev & 63] = true;
    return e;
}

template <int C, int NB, int WAVES, int VAR = 0, int PF = 3, int OCC = WAVES / 4>
static hipError_t launch_ffn(hipStream_t st, const bf16* A, const char* w1img, const char* w2img, const float* b1,
                             const float* b2, const float* ls, bf16* X, int M)
{
    constexpr int ROWS = 32 * NB * WAVES;
    const int nwg = (M + ROWS - 1) / ROWS;
    const size_t shmem = (size_t)4 * 64 * C + (size_t)4 * C * 4 + 256;
    static bool attr_set[64];                // the attribute is per device
    hipError_t e = ffn_set_lds(ffn_fused_kernel<C, NB, WAVES, VAR, PF, OCC>, shmem, attr_set);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ffn_fused_kernel<C, NB, WAVES, VAR, PF, OCC>), dim3(nwg), dim3(WAVES * 64), shmem, st, A, w1img, w2img, b1, b2, ls, X, M, nwg);
    return hipGetLastError();
}

static int ffn_num_cus()
{
    static int n[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!n[dev & 63]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev & 63] = v;
    }
    return n[dev & 63];
}

template <int C, int VAR = 0, int PF = 3, bool TOUCH = true>
static hipError_t launch_ffn_persist8(hipStream_t st, const bf16* A, const char* w1img, const char* w2img, const float* b1,
                                      const float* b2, const float* ls, bf16* X, int M)
{
    const int ntw = (M + 255) / 256;
    const size_t shmem = (size_t)4 * 64 * C + (size_t)(4 * C + 2 * C) * 4 + 256 + (size_t)2 * (C == 96 ? 8 : 4) * 32 * C * 2 + ((VAR & 128) ? 8 * 40 * 16 : 0);
    static bool attr_set[64];
    hipError_t e = ffn_set_lds(ffn_persist8_kernel<C, VAR, PF, TOUCH>, shmem, attr_set);
    if (e != hipSuccess) return e;
    const int G = ntw < ffn_num_cus() ? ntw : ffn_num_cus();
    hipLaunchKernelGGL((ffn_persist8_kernel<C, VAR, PF, TOUCH>), dim3(G), dim3(512), shmem, st, A, w1img, w2img, b1, b2, ls, X, M, ntw);
    return hipGetLastError();
}

template <int C, int VAR = 0, int PF = 3, bool TOUCH = true>
static hipError_t launch_ffn_persist4(hipStream_t st, const bf16* A, const char* w1img, const char* w2img, const float* b1,
                                      const float* b2, const float* ls, bf16* X, int M)
{
    const int ntw = (M + 127) / 128;
    const size_t shmem = (size_t)4 * 64 * C + (size_t)(4 * C + 2 * C) * 4 + 256;
    static bool attr_set[64];
    hipError_t e = ffn_set_lds(ffn_persist4_kernel<C, VAR, PF, TOUCH>, shmem, attr_set);
    if (e != hipSuccess) return e;
    const int G = ntw < ffn_num_cus() ? ntw : ffn_num_cus();
    hipLaunchKernelGGL((ffn_persist4_kernel<C, VAR, PF, TOUCH>), dim3(G), dim3(256), shmem, st, A, w1img, w2img, b1, b2, ls, X, M, ntw);
    return hipGetLastError();
}

#ifdef FVHD_FFN_ABLATE
// Ablation build only (libfvhd_ablate.so, `FVHD_FFN_ABLATE=1 python -m ml_fastvlm_amd.build`; never the shipped library):
// FVHD_FFN_VARIANT picks one of the instantiated variants (results are wrong by construction for VAR & 15).
static int ffn_variant()
{
    static const int v = [] { const char* e = getenv("FVHD_FFN_VARIANT"); return e ? atoi(e) : 0; }();
    return v;
}
template <int C> static hipError_t launch_ffn_variant(hipStream_t st, const bf16* a, const char* w1, const char* w2, const float* b1,
                                                      const float* b2, const float* ls, bf16* x, int M)
{
#define FFN_P(VV, PP, TT) (C == 384 ? launch_ffn_persist4<384, VV, PP, TT>(st, a, w1, w2, b1, b2, ls, x, M) \
                                    : C == 192 ? launch_ffn_persist8<192, VV, PP, TT>(st, a, w1, w2, b1, b2, ls, x, M) \
                                               : launch_ffn_persist8<96, VV, PP, TT>(st, a, w1, w2, b1, b2, ls, x, M))
    switch (ffn_variant()) {
    case 1: return FFN_P(1, 3, true);
    case 2: return FFN_P(2, 3, true);
    case 3: return FFN_P(3, 3, true);
    case 15: return FFN_P(15, 3, true);
    case 16: return FFN_P(16, 3, true);
    case 32: return FFN_P(32, 3, true);
    case 48: return FFN_P(48, 3, true);
    case 64: return FFN_P(64, 3, true);
    case 80: return FFN_P(80, 3, true);
    case 128: return C == 384 ? FFN_P(0, 3, true) : (C == 192 ? launch_ffn_persist8<192, 128, 3, true>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn_persist8<96, 128, 3, true>(st, a, w1, w2, b1, b2, ls, x, M));
    case 131: return C == 384 ? FFN_P(0, 3, true) : (C == 192 ? launch_ffn_persist8<192, 131, 3, true>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn_persist8<96, 131, 3, true>(st, a, w1, w2, b1, b2, ls, x, M));
    case 143: return C == 384 ? FFN_P(0, 3, true) : (C == 192 ? launch_ffn_persist8<192, 143, 3, true>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn_persist8<96, 143, 3, true>(st, a, w1, w2, b1, b2, ls, x, M));
    case 100: return FFN_P(0, 3, false);
    case 101: return FFN_P(0, 2, true);
    case 102: return FFN_P(0, 5, true);
    default: return FFN_P(0, 3, true);
    }
#undef FFN_P
}
#endif

// 1 if the fused kernel exists for this channel count
extern "C" int fvhd_ffn_fused_supported(int C) { return C == 384 || C == 192 || C == 96; }

// Host-side packer: fc1 [4C][C] and fc2 [C][4C] (fp32, the reference's layouts) -> bf16 chunk images in LDS byte order.
//   w1img: (4C/32 + 1) chunks of 64*C bytes (last chunk zero);  w2img: 4C/32 chunks of 64*C bytes.
static uint16_t to_bf16(float f)      // round-to-nearest-even, as torch's .to(bfloat16)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

extern "C" int fvhd_ffn_pack_host(int C, const float* fc1, const float* fc2, uint16_t* w1img, uint16_t* w2img)
{
    if (!fvhd_ffn_fused_supported(C)) return 1;
    const int HID = 4 * C, NCH = HID / 32, CHE = 32 * C;   // bf16 elements per chunk image
    for (int i = 0; i < (NCH + 1) * CHE; ++i) w1img[i] = 0;
    for (int ch = 0; ch < NCH; ++ch) {
        uint16_t* i1 = w1img + (size_t)ch * CHE;
        uint16_t* i2 = w2img + (size_t)ch * CHE;
        for (int row = 0; row < 32; ++row)
            for (int slot = 0; slot < C / 8; ++slot) {
                const int off = (C == 384 ? w1_off<384>(row, slot) : C == 192 ? w1_off<192>(row, slot) : w1_off<96>(row, slot)) / 2;
                for (int e = 0; e < 8; ++e) i1[off + e] = to_bf16(fc1[(size_t)(ch * 32 + row) * C + slot * 8 + e]);
            }
        for (int n = 0; n < C; ++n)
            for (int slot = 0; slot < 4; ++slot) {
                const int off = w2_off(n, slot) / 2;
                for (int e = 0; e < 8; ++e) {
                    const int pos = slot * 8 + e, kb = pos >> 4, hf = (pos >> 3) & 1, j = pos & 7;
                    const int h = 16 * kb + 8 * (j >> 2) + 4 * hf + (j & 3);
                    i2[off + e] = to_bf16(fc2[(size_t)n * HID + ch * 32 + h]);
                }
            }
    }
    return 0;
}

// A [M,C] bf16; w1img / w2img from fvhd_ffn_pack_host (device copies); b1 [4C], b2 [C], ls [C] fp32; X [M,C] in/out.
extern "C" int fvhd_launch_ffn_fused(hipStream_t st, const void* A, const void* w1img, const float* b1, const void* w2img,
                                     const float* b2, const float* ls, void* X, int M, int C)
{
    const bf16* a = (const bf16*)A;
    const char* w1 = (const char*)w1img;
    const char* w2 = (const char*)w2img;
    bf16* x = (bf16*)X;
    hipError_t e = hipErrorInvalidValue;
    if (M <= 0) return (int)e;
    const bool persist = ffn_persist_mode() != 0 && M >= 65536;
#ifdef FVHD_FFN_ABLATE
    if (persist && ffn_variant() != 0)
        return (int)(C == 384 ? launch_ffn_variant<384>(st, a, w1, w2, b1, b2, ls, x, M) : C == 192 ? launch_ffn_variant<192>(st, a, w1, w2, b1, b2, ls, x, M)
                                                                                                      : C == 96 ? launch_ffn_variant<96>(st, a, w1, w2, b1, b2, ls, x, M) : e);
#endif
    if (C == 384) e = persist ? launch_ffn_persist4<384>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn<384, 1, 4>(st, a, w1, w2, b1, b2, ls, x, M);
    else if (C == 192) e = persist ? launch_ffn_persist8<192>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn<192, 1, 4, 0, 3, 2>(st, a, w1, w2, b1, b2, ls, x, M);
    // C = 96: 152 registers since the epilogue offsets stopped being hoisted -> three workgroups (waves) per SIMD: the kernel is
    // VALU-issue-bound (16 GELUs per 12 MFMAs), a third instruction stream per SIMD is what it needs
    else if (C == 96) e = persist ? launch_ffn_persist8<96>(st, a, w1, w2, b1, b2, ls, x, M) : launch_ffn<96, 1, 4, 0, 3, 3>(st, a, w1, w2, b1, b2, ls, x, M);
    return (int)e;
}
