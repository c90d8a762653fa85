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
// Round 2 built PERSISTENT variants of this kernel (one workgroup per CU keeping the chunk pipeline going across row tiles, tile
// traffic by LDS-DMA from a dedicated wave with counted vmcnt); they ended at parity with the one-tile kernels below (630 / 474 /
// 389 vs 602 / 464 / 387 us at C = 96 / 192 / 384: profiles/r02_ffn_persist_notes.md) and were removed in round 3 (git: 815bb98).
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

// erf GELU (fvhd_common.h: Phi(x) = 0.5 + xc Q(xc^2); degree-5 Q on clamp(x, +-3.5) in this kernel, see FVHD_GELU5_*) on PAIRS of
// hidden values: every step is one v_pk_*_f32, so a pair costs 8 packed + 2 v_med3 issue slots (5 per value; + one packed add where
// the bias is not already in the accumulator).  Cut into six half-stages of 2-3 instructions; the software pipeline below advances one pair by one
// half-stage per unit.
struct GeluSt { f32x2 x, xc, u, q; };
#define FFN_PK(c) (f32x2{c, c})
#ifndef FVHD_FFN_GELU_DEG
#define FVHD_FFN_GELU_DEG 5          // 7 = the degree-7 fit every other kernel uses (A/B builds: -DFVHD_FFN_GELU_DEG=7)
#endif
template <int H> FVHD_DEV void gelu_half(GeluSt& g, f32x2 x, f32x2& out)
{
#if FVHD_FFN_GELU_DEG == 7
    if constexpr (H == 0) {
        g.x = x;
        g.xc = f32x2{__builtin_amdgcn_fmed3f(x[0], -FVHD_GELU_CLAMP, FVHD_GELU_CLAMP), __builtin_amdgcn_fmed3f(x[1], -FVHD_GELU_CLAMP, FVHD_GELU_CLAMP)};
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
#else
    if constexpr (H == 0) {
        g.x = x;
        g.xc = f32x2{__builtin_amdgcn_fmed3f(x[0], -FVHD_GELU5_CLAMP, FVHD_GELU5_CLAMP), __builtin_amdgcn_fmed3f(x[1], -FVHD_GELU5_CLAMP, FVHD_GELU5_CLAMP)};
    } else if constexpr (H == 1) {
        g.u = g.xc * g.xc;
        g.q = __builtin_elementwise_fma(FFN_PK(FVHD_GELU5_C5), g.u, FFN_PK(FVHD_GELU5_C4));
    } else if constexpr (H == 2) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU5_C3));
    } else if constexpr (H == 3) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU5_C2));
    } else if constexpr (H == 4) {
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU5_C1));
        g.q = __builtin_elementwise_fma(g.q, g.u, FFN_PK(FVHD_GELU5_C0));
    } else {
        out = g.x * __builtin_elementwise_fma(g.xc, g.q, FFN_PK(0.5f));
    }
#endif
}

// Half-precision form (round 3): the kernels are POWER-limited (1.39 kW of the 1.4-kW package cap at C = 96 / 192, 1.31 kW at C = 384:
// profiles/r03_power_probe.log), so VALU instructions cost wall time even where they hide behind the MFMAs in cycles, and packed-f16
// VALU does two values per instruction at the price of one scalar f32 one (tools/ubench/f16_rate.hip: the chunk loop without its LDS
// side 508 -> 467 us per 1000 chunks at C = 192, 348 -> 294 at C = 96).  Measured on the kernel, sustained: 481 -> 450 us at C = 96,
// 357 -> 336 at C = 192, 339 -> 336 at C = 384 (profiles/r03_ffn_f16.log).  GEMM1 delivers x' = x / 4 (W1 and b1 pre-scaled by 1/4 - exact; b1 rides in the accumulator), so that every coefficient
// of   Phi(x) = clamp01(0.5 + x' Q'(min(x'^2, (3.5/4)^2)))   is O(1..10) in f16;  y' = x' Phi = gelu(x) / 4 is the f16 B operand of GEMM2
// (v_mfma_f32_32x32x16_f16) as it stands, W2 packed as f16(4 W2).  11 packed instructions per PAIR: cvt (round toward zero: saturates at
// |x| = 262016 instead of overflowing), mul, min, 5 fma, fma + clamp modifier, mul.  Accuracy of the hidden activation against the exact
// erf GELU (tools/ubench/g16.py): relative 0.5-1.0e-3, against 1.7e-3 for f32 math + rounding to bf16 - P now carries 11 mantissa bits
// instead of 8; |Phi error| <= 1.4e-3; Phi = 1 exactly from x = 3.5 on and 0 exactly below -3.51 (c0 is nudged one ulp up for that; at -3.5
// itself the half-precision sum leaves 2^-13 against the true 2.3e-4) - tests/test_gelu_f16.py restates the sequence in numpy.
#ifndef FVHD_FFN_CIO384
#define FVHD_FFN_CIO384 1            // C = 384: A / X tiles read and written as whole lines through an LDS transpose, like C <= 192 (0: 16-B / 8-B row accesses)
#endif
#ifndef FVHD_FFN_PAIRWAIT
#define FVHD_FFN_PAIRWAIT 1          // fragment reads in pairs behind one explicit s_waitcnt (0: one read and one compiler-placed wait per MFMA)
#endif
#ifndef FVHD_FFN_DMARUN
#define FVHD_FFN_DMARUN 1            // weight DMA as runs of consecutive pieces (0: one statement per piece, the round 1-3 form)
#endif
#ifndef FVHD_FFN_F16
#define FVHD_FFN_F16 2               // 2: every C;  1: C <= 192 only;  0: the f32 GELU + bf16 GEMM2 of rounds 1-2   (A/B builds: -DFVHD_FFN_F16=0)
#endif
template <int C> __host__ __device__ constexpr bool ffn_f16() { return FVHD_FFN_F16 == 2 || (FVHD_FFN_F16 == 1 && C <= 192); }
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct GeluSt16 { f16x2 x, u, q; };
#define FFN_H2(bits) (__builtin_bit_cast(f16x2, (unsigned)(bits) * 0x10001u))
// The ten packed instructions of one pair, ONE per call (round 4): the iteration issues stage k of several INDEPENDENT pairs in a slot
// and stage k + 1 in the next, so that no packed instruction directly follows the one it depends on - in the half-stage form of round 3
// hipcc separated every such pair with an s_nop (66 of the 833 instructions of two iterations at C = 384).
// c_k' = 4 * 16^k * FVHD_GELU5_Ck rounded to f16 (c0' + 1 ulp):  1.5927734375, -4.12109375, 8.703125, -11.7890625, 8.9375, -2.84375
template <int K> FVHD_DEV void gelu16_stage(GeluSt16& g, f32x2 x, f16x2& out)
{
    if constexpr (K == 0) g.x = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
    else if constexpr (K == 1) g.u = g.x * g.x;
    else if constexpr (K == 2) g.u = __builtin_elementwise_min(g.u, FFN_H2(0x3a20));          // (3.5 / 4)^2 = 0.765625
    else if constexpr (K == 3) g.q = __builtin_elementwise_fma(FFN_H2(0xc1b0), g.u, FFN_H2(0x4878));
    else if constexpr (K == 4) g.q = __builtin_elementwise_fma(g.q, g.u, FFN_H2(0xc9e5));
    else if constexpr (K == 5) g.q = __builtin_elementwise_fma(g.q, g.u, FFN_H2(0x485a));
    else if constexpr (K == 6) g.q = __builtin_elementwise_fma(g.q, g.u, FFN_H2(0xc41f));
    else if constexpr (K == 7) g.q = __builtin_elementwise_fma(g.q, g.u, FFN_H2(0x3e5f));
    else if constexpr (K == 8) asm("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(g.q) : "v"(g.x), "v"(g.q), "v"(FFN_H2(0x3800)));
    else out = g.x * g.q;
}
template <int K = 0> FVHD_DEV void gelu16_dispatch(int k, GeluSt16& g, f32x2 x, f16x2& out)
{
    if constexpr (K < 10) {
        if (k == K) gelu16_stage<K>(g, x, out);
        else gelu16_dispatch<K + 1>(k, g, x, out);
    }
}

// 16 B/lane LDS-DMA (global -> LDS, no VGPR staging): LDS destination = wave-uniform byte address `lds_dst` + lane*16.
// Issued from inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in the loop hipcc's waitcnt pass
// degrades every counted lgkmcnt(N) of the fragment ds_reads to lgkmcnt(0) (measured: 96 counted waits without the
// builtin, 0 with it), which exposes the full LDS latency before every MFMA.  hipcc does not count an asm load, so the
// consumer side waits explicitly: s_waitcnt vmcnt(0) before the barrier that publishes the images (ffn_wait_dma).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
#ifdef FVHD_FFN_ABL_NOWAIT      // ablation (wrong results, timing only): the chunk loop never waits for its weight DMA
FVHD_DEV void ffn_wait_dma() {}
#else
FVHD_DEV void ffn_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// One pipeline iteration t, for the NB 32-row blocks a wave owns:
//     s_out <- b1(t) + GEMM1(chunk t);   p_out <- GELU(s_in = S(t-1));   O += GEMM2(chunk t-2, p_in = P(t-2)).
// The iteration is cut into NM = 2*KS*NB slots of one MFMA each.  Fragment f (f even: W1 k-step f/2, f odd: W2 fragment
// f/2) is read from LDS once and feeds NB consecutive slots (one per row block), so consecutive MFMAs never share an
// accumulator, and the instruction order is pinned slot by slot with sched_barrier:
//     slot m = { MFMA m | LDS fragment read (PF fragments ahead) | one 1-KiB LDS-DMA piece of next iteration's weights
//                (some slots) | 96*NB/NM GELU half-stages (2-3 VALU each, from independent dependency chains) }
// i.e. every 32-cycle MFMA carries a handful of independent single-issue fillers - what one wave can issue in its shadow
// (MI355X_MICROARCH "one wave per SIMD") - instead of 200+ VALU in a lump between two MFMA bursts.
template <int C, int NB, int WAVES, bool DO_A, bool DO_B, bool DO_C, bool DO_DMA, int VAR, int PF, bool F16 = ffn_f16<C>(), bool BPRE = (C == 96 || F16)>
FVHD_DEV void ffn_iter(const bf16x8 (&afr)[NB][C / 16], f32x16 (&o)[NB][C / 32], f32x16 (&s_out)[NB], const f32x16 (&s_in)[NB],
                       bf16x8 (&p_out)[NB][2], const bf16x8 (&p_in)[NB][2], const char* const (&w1p)[C / 48],
                       const char* const (&w2p)[2], const int ring, const float* b1_cur, const float* b1_prev, int half,
                       const char* dma_src1, unsigned dma_dst1, bool dma1, const char* dma_src2, unsigned dma_dst2, bool dma2, int uwave)
{
    constexpr int KS = C / 16, NF = 2 * KS, NM = NF * NB;   // fragments, MFMA slots
    constexpr int NA = C / 48, CHB = 64 * C, NG = CHB / 1024;
    constexpr int UPS = 96 * NB / NM;       // GELU half-stage units per slot (16 values x 6 half-stages per block) = 48 / KS
    constexpr int DW = WAVES, TP = 2 * NG, NPW = (TP + DW - 1) / DW;   // 1-KiB DMA pieces per iteration (W1 then W2), per wave
    static_assert(TP % DW == 0, "every wave issues the same number of pieces");
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
    GeluSt16 gs16[NB][8];
    f16x2 gout16[NB][8];
    // b1: either the GEMM1 accumulator starts from it (BPRE: 16 more live registers at the top of the iteration, no VALU), or
    // it is added in the first GELU half-stage (value r <-> hidden (r&3) + 8(r>>2) + 4*half)
    if constexpr (DO_A && BPRE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)(b1_cur + 8 * q + 4 * half);
    }
    if constexpr (DO_B && !BPRE) bv[0] = *(const f32x4*)(b1_prev + 4 * half);
    // PAIRW (round 4, steady-state iterations of the half-precision form with one block per wave): fragments are fetched in PAIRS -
    // slot 2j issues the reads of fragments 2j + 4 and 2j + 5, slot 2j + 1 none - behind ONE explicit s_waitcnt per pair
    // (lgkmcnt(2): everything but the pair read last has landed, i.e. fragments 2j and 2j + 1); hipcc's own insertion then finds the
    // second MFMA's operand already covered and adds nothing: 24 instead of 49 s_waitcnt per iteration at C = 384, each of which
    // took an issue slot of the issue-bound loop.  (A count that is too permissive would only make hipcc add its own wait back.)
    constexpr bool PAIRW = FVHD_FFN_PAIRWAIT && PIN && F16 && BPRE && NB == 1 && DO_A && DO_B && DO_C && !(VAR & (4 | 8));
    constexpr int PFE = PAIRW ? 4 : PF;         // fragments in flight ahead of the MFMA stream
#pragma unroll
    for (int f = 0; f < PFE; ++f)
        if (FFN_LD_ON(f)) wf[f] = FFN_LD(f);
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int f = m / NB, nb = m % NB;
        if constexpr (PAIRW) {
            if ((f & 1) == 0) {
                // fragment reads issued after f + 1: {f + 2, f + 3} while they exist
                if (f + 2 < NF) __builtin_amdgcn_s_waitcnt(0xc27f);     // lgkmcnt(2)
                else __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
            }
        }
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
                else if constexpr (F16)
                    o[nb][g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[f]), __builtin_bit_cast(f16x8, p_in[nb][g & 1]), o[nb][g >> 1], 0, 0, 0);
                else o[nb][g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[f], p_in[nb][g & 1], o[nb][g >> 1], 0, 0, 0);
            }
        }
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);  // the MFMA opens the slot; everything below issues in its shadow
        if constexpr (PAIRW) {
            if ((f & 1) == 0) {
                if (f + 4 < NF) wf[f + 4] = FFN_LD(f + 4);
                if (f + 5 < NF) wf[f + 5] = FFN_LD(f + 5);
            }
        } else if (nb == 0 && f + PF < NF && FFN_LD_ON(f + PF)) wf[f + PF] = FFN_LD(f + PF);
        if constexpr (DO_DMA) {             // next iteration's weight images
#if FVHD_FFN_DMARUN
            // RUNS of 3-4 consecutive 1-KiB pieces per statement (glds16_run: one M0 setting, scalar base + the constant lane offset,
            // the instruction offset advances the global and the LDS side together): 2-3 issue slots per piece instead of 7 (M0 save /
            // set / nop / restore + a 64-bit VALU address per piece) - the iteration is issue-bound (~6 instructions per MFMA gap
            // against the ~5 one wave per SIMD hides), and the per-piece form spent 84 of its ~300 instructions at C = 384 on this.
            // Waves 0-1 stream W1, waves 2-3 W2; a wave's PPW pieces are consecutive in the (linear) chunk image.
            // (DW waves: the first DW / 2 stream W1, the others W2)
            constexpr int PPW = TP / DW, RUN = PPW % 4 == 0 ? 4 : PPW % 3 == 0 ? 3 : PPW, NRUN = PPW / RUN, RSTRIDE = (NM / 2) / NRUN > 0 ? (NM / 2) / NRUN : 1;
            static_assert(TP % DW == 0 && PPW % RUN == 0 && RUN <= 4 && DW % 2 == 0 && NG % PPW == 0, "a wave's pieces lie in one matrix");
            if (m % RSTRIDE == 0 && m / RSTRIDE < NRUN) {
                const int second = uwave >= DW / 2;                                       // wave-uniform (SGPR)
                const unsigned pc0 = (unsigned)((uwave % (DW / 2)) * PPW + (m / RSTRIDE) * RUN) * 1024u;
                const char* sb = (second ? dma_src2 : dma_src1) + pc0;
                const unsigned dst = (second ? dma_dst2 : dma_dst1) + pc0;
                if (second ? dma2 : dma1) glds16_run<RUN>(sb, (threadIdx.x & 63) * 16u, dst);
            }
#else
            if (m % DSTRIDE == 0 && m / DSTRIDE < NPW) {    // one 1-KiB piece per DSTRIDE slots
                const int flat = (m / DSTRIDE) * DW + uwave;
                const int lane16 = (threadIdx.x & 63) * 16;
                const bool second = flat >= NG;
                const int pc = second ? flat - NG : flat;
                const char* src = (second ? dma_src2 : dma_src1) + pc * 1024 + lane16;
                const unsigned dst = (second ? dma_dst2 : dma_dst1) + (unsigned)pc * 1024u;
                if (second ? dma2 : dma1) glds16(src, dst);              // compile-time or loop-invariant flags only
            }
#endif
        }
        if constexpr (DO_B) {               // GELU; value r of block gb <-> hidden h = (r&3) + 8(r>>2) + 4*half
            if constexpr (!BPRE) {
#pragma unroll
                for (int q = 1; q < 4; ++q)     // bias of values 4q..4q+3 (first used by unit 24*q*NB): read ~2 slots ahead
                    if (m == ((24 * q * NB) / UPS >= 2 ? (24 * q * NB) / UPS - 2 : 0)) bv[q] = *(const f32x4*)(b1_prev + 8 * q + 4 * half);
            }
            if constexpr (F16) {
                // half-precision form: slot m issues stage k = m % 10 of the UPS pairs of group m / 10 (NM / 12 groups of UPS = 48 / KS
                // pairs: 8 NB pairs per iteration) - one instruction per pair and slot, all of them independent of each other
                constexpr int NGRP = NM / 12;
                static_assert(NGRP * UPS == 8 * NB && NGRP * 10 <= NM, "ten stages of every pair group fit into the iteration's slots");
                if (m / 10 < NGRP) {
                    const int k = m % 10;
#pragma unroll
                    for (int i = 0; i < UPS; ++i) {
                        const int pr = (m / 10) * UPS + i, gb = pr % NB, r = 2 * (pr / NB);      // pair (values r, r + 1) of block gb
                        const f32x2 sv = {s_in[gb][r], s_in[gb][r + 1]};
                        if constexpr (VAR & 2) {     // ablation: no GELU math (conversion only)
                            if (k == 9) gout16[gb][r >> 1] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(sv[0], sv[1]));
                        } else gelu16_dispatch(k, gs16[gb][r >> 1], sv, gout16[gb][r >> 1]);
                        if (k == 9) {
                            f16x8 pt = __builtin_bit_cast(f16x8, p_out[gb][r >> 3]);
                            pt[r & 7] = gout16[gb][r >> 1][0];
                            pt[(r & 7) + 1] = gout16[gb][r >> 1][1];
                            p_out[gb][r >> 3] = __builtin_bit_cast(bf16x8, pt);
                        }
                    }
                }
            } else
#pragma unroll
            for (int u = m * UPS / 2; u < (m + 1) * UPS / 2; ++u) {     // unit u = ((pair j, half-stage h), block gb)
                // f32 form (FVHD_FFN_BF16 precision): one pair at a time (register budget at C = 384)
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
// offset) -> LDS byte offset of the transposing stage: a bijection of the tile's chunk ids (XOR of low id bits with higher ones), the
// same on the writing (lane-linear ids) and the reading side (row li, chunk 2 ks + half).  A fragment ds_read_b128 is served in 16-lane
// groups (rows {0-3, 12-15, 20-27} / the rest), conflict-free iff their 16 slots differ modulo 16:
//   C = 384 (CPR 48 = 3 * 16): id mod 16 is the same for every row; key = bits 4-7 of id = (3 row + const) mod 16 - 16 different slots.
//   C = 192 (CPR 24): round 2-3 used key = bits 3-5 -> low 3 bits only: rows r and r + 8 (24 r = 8 * 3 r) collided, 2-way (PMC: 8.4 %
//       of the LDS cycles); bit 6 of id (= floor(0.375 row): differs inside every colliding pair of a group) now goes into bit 3.
//   C = 96 (CPR 12): the old key (bits 2-3 into bits 0-1) only moved entropy that bits 2-3 already carried: 4 slots per group, 4-way
//       (PMC: 28 %); bits 4-5 of id (= floor(0.75 row)) into bits 0-1 give 16 different slots (checked exhaustively on the host).
template <int C> FVHD_DEV int ffn_slot_of(int id)
{
    if constexpr (C == 384) return (id ^ ((id >> 4) & 15)) << 4;
    else if constexpr (C == 192) return (id ^ (((id >> 3) & 7) ^ (((id >> 6) & 1) << 3))) << 4;
    else return (id ^ ((id >> 4) & 3)) << 4;
}

// ---------------------------------------------------------------------------------------------------------------------
// One tile (32 * WAVES rows) per workgroup: the small-launch path (and the round-1 structure).
template <int C, int NB, int WAVES, int VAR = 0, int PF = 3, int OCC = WAVES / 4, bool F16 = ffn_f16<C>()>
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
    // C = 384 with the coalesced tile I/O (FVHD_FFN_CIO384): the transposing stage of the tile prologue must not sit in the W1 ring
    // (W1[0] is streamed in while the A tile is staged), so it starts at the W2 ring and runs 4 * 24 KB = 96 KB from there, past the
    // 4 * CHB of the rings; b1 moves behind it.  C <= 192: the stage is the (still empty) ring area itself.
    constexpr bool CIO = C <= 192 || FVHD_FFN_CIO384;
    constexpr int STAGE0 = (C == 384 && CIO) ? 2 * CHB : 0;
    constexpr int STAGE_END = CIO ? STAGE0 + WAVES * (32 * C * 2) : 0;     // (more than 4 waves at C <= 192: the stage outgrows the rings)
    float* lb1 = (float*)(smem + (STAGE_END > 4 * CHB ? STAGE_END : 4 * CHB));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int blk = xcd_remap(blockIdx.x, nwg);
    const int row0 = blk * (32 * NB * WAVES) + wave * (32 * NB);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);          // provably uniform: DMA bases stay in SGPRs / M0
    const unsigned lds_w1 = __builtin_amdgcn_readfirstlane(lds_addr(w1ring)), lds_w2 = __builtin_amdgcn_readfirstlane(lds_addr(w2ring));

    for (int i = tid; i < HID / 4; i += WAVES * 64)
        *(f32x4*)&lb1[i * 4] = *(const f32x4*)&b1[i * 4] * (F16 ? 0.25f : 1.0f);          // half-precision form: GEMM1 delivers x / 4

    // per-lane fragment pointers (ring slot 0), see ffn_iter
    const char* w1p[C / 48];
    const char* w2p[2];
#pragma unroll
    for (int k = 0; k < C / 48; ++k) w1p[k] = w1ring + w1_off<C>(li, 2 * k + half);
#pragma unroll
    for (int k = 0; k < 2; ++k) w2p[k] = w2ring + w2_off(li, 2 * k + half);

#define FFN_W1_FIRST()                                                                                                  \
    _Pragma("unroll") for (int g = 0; g < (NG + WAVES - 1) / WAVES; ++g) {             /* W1[0] */                      \
        const int piece = g * WAVES + uwave;                                                                           \
        if (NG % WAVES == 0 || piece < NG) glds16(w1img + piece * 1024 + lane * 16, lds_w1 + piece * 1024);            \
    }
    // (STAGE0 != 0: the stage does not overlap the W1 ring - W1[0] streams in while the A tile is staged; the W2 ring, which the stage
    // does overlap, is first written during iteration 1, two barriers after the last fragment read)
    if constexpr (CIO && STAGE0 != 0) { FFN_W1_FIRST() }
    bf16x8 afr[NB][KS];
    f32x16 o[NB][NFR];
    f32x16 s0[NB], s1[NB];          // s[k&1] holds S(k)
    bf16x8 p0[NB][2], p1[NB][2];    // p[k&1] holds P(k)
    // ---- A^T fragments (B operands).  A wave's 32 rows are 32*C*2 contiguous bytes of HBM, but the MFMA wants lane
    // (row, k-half) to hold 16 B of ITS row: loading that directly is 32 rows x 32 B per instruction (a quarter of every
    // 128-B line per request).  For C <= 192 the tile is read fully coalesced (lane L of load i takes 16-B chunk i*64+L)
    // and transposed through LDS - the (still empty) weight rings - with a 16-B-slot XOR swizzle that keeps both sides
    // conflict-free.
    static_assert((NB == 1 || NB == 2) && WAVES % 4 == 0, "every wave transposes its tiles through its own 32-row stage, one block at a time");
    constexpr int CPR = C / 8;               // 16-B chunks per row
    constexpr int NL = 32 * CPR / 64;        // coalesced 16-B loads per lane for one 32-row tile (== KS)
    char* stage = smem + STAGE0 + wave * (32 * C * 2);
    const size_t tile_b = (size_t)row0 * C * 2, last_b = (size_t)M * C * 2 - 16;     // rows >= M: any valid address (never stored)
    constexpr size_t BLK_B = (size_t)32 * C * 2;      // bytes of one 32-row block (contiguous in HBM)
    if constexpr (CIO) {
        u32x4 t[NB][NL];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const size_t gb = tile_b + nb * BLK_B + (size_t)(i * 64 + lane) * 16;
                t[nb][i] = *(const u32x4*)((const char*)A + (gb < last_b ? gb : last_b));
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {       // the blocks of a wave go through its staging area one after the other
#pragma unroll
            for (int i = 0; i < NL; ++i) *(u32x4*)(stage + ffn_slot_of<C>(i * 64 + lane)) = t[nb][i];
            // same wave wrote and reads: LDS operations of one wave are processed in order, no barrier needed
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) afr[nb][ks] = *(const bf16x8*)(stage + ffn_slot_of<C>(li * CPR + ks * 2 + half));
        }
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
    if constexpr (CIO && STAGE0 == 0) __syncthreads();      // every wave has its fragments: the rings may now receive weights
    if constexpr (!(CIO && STAGE0 != 0)) { FFN_W1_FIRST() }
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
    ffn_iter<C, NB, WAVES, true, false, false, !(VAR & 1), VAR, PF, F16>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(0));
    FFN_SYNC();
    ffn_iter<C, NB, WAVES, true, true, false, !(VAR & 1), VAR, PF, F16>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + 32, FFN_B1PREV(lb1 + 32), half, FFN_DMA_ARGS(1));
#pragma unroll 1
    for (int t = 2; t < NCH; t += 2) {
        FFN_SYNC();                  // even t: S(t) -> s0, GELU(s1) -> p1, GEMM2 reads p0
        ffn_iter<C, NB, WAVES, true, true, true, !(VAR & 1), VAR, PF, F16>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1 + t * 32, FFN_B1PREV(lb1 + t * 32), half, FFN_DMA_ARGS(t));
        FFN_SYNC();                  // odd t:  S(t) -> s1, GELU(s0) -> p0, GEMM2 reads p1
        ffn_iter<C, NB, WAVES, true, true, true, !(VAR & 1), VAR, PF, F16>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1 + (t + 1) * 32, FFN_B1PREV(lb1 + (t + 1) * 32), half, FFN_DMA_ARGS(t + 1));
    }
    // the residual tile of X is fetched now, coalesced like A^T above (the A^T registers are dead from here on), so that
    // its HBM latency hides behind the last two pipeline iterations instead of stalling the epilogue
    u32x4 xq[CIO ? NB : 1][CIO ? NL : 1];
    bf16x4 xres[NB][NFR][4];
    // (C = 96 with two blocks per wave: the 48 registers of the prefetched X tile would not fit beside the pipeline's live state under
    // the 256-register limit of two waves per SIMD - there the tile is fetched after the last iteration instead)
    constexpr bool XPRE = !(C == 96 && NB == 2);
#define FFN_LOAD_X()                                                                                                   \
    if constexpr (CIO) {                                                                                               \
        int xln = lane;                                                                                                \
        asm volatile("" : "+v"(xln));                                                                                  \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                              \
            _Pragma("unroll") for (int i = 0; i < NL; ++i) {                                                           \
                const size_t gb = tile_b + nb * BLK_B + (size_t)(i * 64 + xln) * 16;                                   \
                xq[nb][i] = *(const u32x4*)((const char*)X + (gb < last_b ? gb : last_b));                             \
            }                                                                                                          \
    } else {                                                                                                           \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                            \
            const bf16* xrd = X + (size_t)min(row0 + nb * 32 + li, M - 1) * C + 4 * half;                              \
            _Pragma("unroll") for (int nf = 0; nf < NFR; ++nf)                                                         \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) xres[nb][nf][q] = *(const bf16x4*)(xrd + nf * 32 + 8 * q); \
        }                                                                                                              \
    }
    if constexpr (XPRE) { FFN_LOAD_X() }
    FFN_SYNC();                      // t = NCH (even): GELU(S(NCH-1) in s1) -> p1, GEMM2(chunk NCH-2) reads p0; DMA W2[NCH-1]
    ffn_iter<C, NB, WAVES, false, true, true, !(VAR & 1), VAR, PF, F16>(afr, o, s0, s1, p1, p0, w1p, w2p, 0, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(NCH));
    FFN_SYNC();                      // t = NCH + 1: GEMM2(chunk NCH-1) reads p1
    ffn_iter<C, NB, WAVES, false, false, true, false, VAR, PF, F16>(afr, o, s1, s0, p0, p1, w1p, w2p, 1, lb1, FFN_B1PREV(lb1), half, FFN_DMA_ARGS(NCH));
    if constexpr (!XPRE) { FFN_LOAD_X() }
#undef FFN_LOAD_X
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
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *(u32x4*)(stage + ffn_slot_of<C>(i * 64 + eln)) = xq[nb][i];
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
            for (int j = 0; j < 4; ++j) v[j] = rv[j] + lv[j] * (o[nb][nf][4 * q + j] + bv[j]);
            *(bf16x4*)slot = f32_to_bf4(v);
        }
    const int rows_ok = M - row0 - nb * 32;  // rows of this block that exist
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int id = i * 64 + eln;
        const u32x4 v = *(const u32x4*)(stage + ffn_slot_of<C>(id));
        if (id < rows_ok * CPR) *(u32x4*)((char*)X + tile_b + nb * BLK_B + (size_t)id * 16) = v;
    }
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

#undef FFN_SYNC

// =====================================================================================================================
// C = 384, half-precision form: WAVE PAIRS (round 4).  With 32 rows per wave every weight fragment read from LDS
// (1 KiB per wave) feeds ONE MFMA, and 64 rows per wave do not fit the register file at C = 384 (A^T 192 + O^T 384).  Here two waves
// share 64 rows (two 32-row blocks) and split the two GEMMs the other way:
//   GEMM1 by K: wave r of the pair holds A^T[k-steps 12 r .. 12 r + 11] of BOTH blocks (96 registers) and computes the partial
//               S_r^T[32 h x 64 rows] of its K half - 12 W1 fragments, each feeding two MFMAs;
//   exchange:   a wave keeps the partial of its OWN block (block r) and passes the other one to its partner through LDS (4 KiB);
//               the next iteration adds the partner's partial of its own block and runs the GELU on that block only;
//   GEMM2 by N: wave r holds O^T[channels 192 r .. 192 r + 191] of BOTH blocks (192 registers); it needs P of both blocks: its own
//               from registers, the partner's through LDS (2 KiB) - 12 W2 fragments, each feeding two MFMAs.
// Per wave and chunk: 48 MFMAs as before, but 24 fragment reads + 6 KiB of exchange reads instead of 48 fragment reads
// (LDS read traffic 30 KiB instead of 48 KiB), and half the GELU work per wave is unchanged (16 values per lane).  The exchange is
// double-buffered by iteration parity and published by the one barrier per chunk the weight ring already needs; the pipeline skew
// (GEMM1(t) | reduce + GELU(t-1) | GEMM2(t-2)) is the one of ffn_iter.  "Own block" is always index 0 of a wave's register arrays.
// MEASURED (profiles/r04_ffn_pair384_power.log, same box, sustained): correct (all fused-FFN op tests), and SLOWER - 324 us against 304 us
// per launch for the one-block-per-wave kernel with the same round-4 changes, 422 against 393 mJ; the chip answers with a higher clock
// (2.05 against 1.79 GHz) but the pair needs 22 % more cycles: the LDS traffic it saves was not what the loop waited for, and the two
// exchanges put an LDS round trip and the partner's progress into every chunk's dependency chain.  Compiled only with
// -DFVHD_FFN_PAIR384=1 (kept as the record of the experiment; the shipped library does not contain it).
#ifndef FVHD_FFN_PAIR384
#define FVHD_FFN_PAIR384 0
#endif
#if FVHD_FFN_PAIR384
template <bool DO_A, bool DO_B, bool DO_C, bool DO_DMA>
FVHD_DEV void ffn_pair_iter(const bf16x8 (&afr)[2][12], f32x16 (&o)[2][6], f32x16 (&s_out)[2], const f32x16& s_prev, bf16x8 (&p_out)[2],
                            const bf16x8 (&p_own)[2], const char* const (&w1p)[8], const char* const (&w2p)[2], const int ring,
                            const float* b1_cur, int half, const char* xs_rd, char* xs_wr, const char* px_rd, char* px_wr,
                            const char* dma_src1, unsigned dma_dst1, bool dma1, const char* dma_src2, unsigned dma_dst2, bool dma2, int uwave)
{
    constexpr int C = 384, CHB = 64 * C, NG = CHB / 1024, NFRG = 24, NM = 48, PF = 2;
    // fragment j: j < 12 -> W1, this wave's k-step j;  j >= 12 -> W2 fragment g = j - 12 (n-fragment g >> 1 of this wave's six, k-step g & 1)
#define FFP_LD(j) ((j) < 12 ? *(const bf16x8*)(w1p[(j) % 8] + ring * CHB + ((j) / 8) * 256)                                  \
                            : *(const bf16x8*)(w2p[((j) - 12) & 1] + ring * CHB + (((j) - 12) >> 1) * 2048))
#define FFP_ON(j) ((j) < 12 ? DO_A : DO_C)
    bf16x8 wf[NFRG];
    f32x4 bv[4];
    f32x4 xs[4];                     // the partner's partial S(t-1) of this wave's own block
    bf16x8 pp[2];                    // the partner's P(t-2) (its own block = block 1 here)
    f32x16 sf;                       // S(t-1) of the own block, complete
    GeluSt16 gs16[8];
    f16x2 gout16[8];
    if constexpr (DO_A) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)(b1_cur + 8 * q + 4 * half);       // (the second wave of a pair reads zeros)
    }
    if constexpr (DO_B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[q] = *(const f32x4*)(xs_rd + q * 1024);
    }
    if constexpr (DO_C) {
#pragma unroll
        for (int k = 0; k < 2; ++k) pp[k] = *(const bf16x8*)(px_rd + k * 1024);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (FFP_ON(j)) wf[j] = FFP_LD(j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int j = m >> 1, nb = m & 1;
        if (j < 12) {
            if constexpr (DO_A) {
                if (j == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = bv[r >> 2][r & 3];
                    s_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], afr[nb][j], z, 0, 0, 0);
                } else s_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], afr[nb][j], s_out[nb], 0, 0, 0);
            }
        } else {
            if constexpr (DO_C) {
                const int g = j - 12;
                const bf16x8 pb = nb == 0 ? p_own[g & 1] : pp[g & 1];
                o[nb][g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[j]), __builtin_bit_cast(f16x8, pb), o[nb][g >> 1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // the MFMA opens the slot; everything below issues in its shadow
        if (nb == 0 && j + PF < NFRG && FFP_ON(j + PF)) wf[j + PF] = FFP_LD(j + PF);
        if constexpr (DO_DMA) {                 // next iteration's weight images: three runs of four 1-KiB pieces per wave (see ffn_iter)
            if (m % 8 == 0 && m / 8 < 3) {
                const int second = uwave >> 1;
                const unsigned pc0 = (unsigned)((uwave & 1) * 12 + (m / 8) * 4) * 1024u;
                const char* sb = (second ? dma_src2 : dma_src1) + pc0;
                const unsigned dst = (second ? dma_dst2 : dma_dst1) + pc0;
                if (second ? dma2 : dma1) glds16_run<4>(sb, (threadIdx.x & 63) * 16u, dst);
            }
        }
        if constexpr (DO_B) {
            // reduce: S(t-1) = own partial + the partner's (4 scalar adds per slot in slots 3..6: the exchange reads were issued at the top);
            // then the ten GELU stages of the pair groups {0,1} {2,3} {4,5} {6,7}: stage k of group g in slot 7 + 10 g + k
            if (m >= 3 && m < 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sf[4 * (m - 3) + i] = s_prev[4 * (m - 3) + i] + xs[m - 3][i];
            }
            if (m >= 7 && m < 47) {
                const int g = (m - 7) / 10, k = (m - 7) % 10;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int pr = 2 * g + i, r = 2 * pr;
                    const f32x2 sv = {sf[r], sf[r + 1]};
                    gelu16_dispatch(k, gs16[pr], sv, gout16[pr]);
                    if (k == 9) {
                        f16x8 pt = __builtin_bit_cast(f16x8, p_out[r >> 3]);
                        pt[r & 7] = gout16[pr][0];
                        pt[(r & 7) + 1] = gout16[pr][1];
                        p_out[r >> 3] = __builtin_bit_cast(bf16x8, pt);
                    }
                }
            }
        }
        if constexpr (DO_A) {                   // the partner's block: its partial is complete after slot 23; handed over in slots 26..29
            if (m >= 26 && m < 30) *(f32x4*)(xs_wr + (m - 26) * 1024) = f32x4{s_out[1][4 * (m - 26)], s_out[1][4 * (m - 26) + 1], s_out[1][4 * (m - 26) + 2], s_out[1][4 * (m - 26) + 3]};
        }
        if constexpr (DO_B) {                   // P(t-1) of the own block for the partner's GEMM2 two iterations on
            if (m == 47) {
                *(bf16x8*)(px_wr) = p_out[0];
                *(bf16x8*)(px_wr + 1024) = p_out[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DO_B) asm volatile("" : "+v"(p_out[0]), "+v"(p_out[1]));
    if constexpr (DO_A) asm volatile("" : "+v"(s_out[0]));
#undef FFP_LD
#undef FFP_ON
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_pair384_kernel(
    const bf16* __restrict__ A, const char* __restrict__ w1img, const char* __restrict__ w2img,
    const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ ls, bf16* X, int M, int nwg)
{
    constexpr int C = 384, HID = 4 * C, NCH = HID / 32, CHB = 64 * C, NG = CHB / 1024, CPR = C / 8, NL = 32 * CPR / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [W1 ring 2 CHB][W2 ring 2 CHB][48 KiB: second half of the tile stage / the exchange buffers][b1 / 4][32 zeros]
    char* w1ring = smem;
    char* w2ring = smem + 2 * CHB;
    char* xbase = smem + 4 * CHB;                                  // X[wave][parity] 4 KiB each, then PX[wave][parity] 2 KiB each
    float* lb1 = (float*)(smem + 4 * CHB + 49152);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int role = wave & 1;
    const int blk = xcd_remap(blockIdx.x, nwg);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_w1 = __builtin_amdgcn_readfirstlane(lds_addr(w1ring)), lds_w2 = __builtin_amdgcn_readfirstlane(lds_addr(w2ring));
    for (int i = tid; i < HID / 4 + 8; i += 256)
        *(f32x4*)&lb1[i * 4] = i < HID / 4 ? *(const f32x4*)&b1[i * 4] * 0.25f : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG / 4; ++g) glds16(w1img + (g * 4 + uwave) * 1024 + lane * 16, lds_w1 + (g * 4 + uwave) * 1024);     // W1[0]

    // per-lane fragment pointers: W1 k-step ks (local) = global k-step 12 role + ks, 16-B slot 2 (12 role + ks) + half of row li;
    // w1p[j] serves local k-steps j and j + 8 (immediate + 256), see the derivation at the call; W2 n-fragment 6 role + nf'
    const char* w1p[8];
    const char* w2p[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ksg = 12 * role + j;                              // global k-step of local k-step j
        w1p[j] = w1ring + w1_off<C>(li, 2 * ksg + half);            // (local k-step j + 8 = global ksg + 8: slot + 16 = + 256 B, same XOR key)
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) w2p[k] = w2ring + w2_off(li, 2 * k + half) + role * (6 * 2048);

    // ---- A^T: every wave reads the tile of its own block whole lines at a time into its stage; after a barrier it takes its K half
    // of both blocks (own block from its own stage, the partner's block from the partner's)
    const int row_own = blk * 128 + (wave >> 1) * 64 + role * 32;
    char* stage_own = smem + 2 * CHB + wave * (32 * C * 2);
    char* stage_par = smem + 2 * CHB + (wave ^ 1) * (32 * C * 2);
    const size_t tile_b = (size_t)row_own * C * 2, last_b = (size_t)M * C * 2 - 16;
    {
        u32x4 t[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t gb = tile_b + (size_t)(i * 64 + lane) * 16;
            t[i] = *(const u32x4*)((const char*)A + (gb < last_b ? gb : last_b));
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) *(u32x4*)(stage_own + ffn_slot_of<C>(i * 64 + lane)) = t[i];
    }
    __syncthreads();
    bf16x8 afr[2][12];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        afr[0][ks] = *(const bf16x8*)(stage_own + ffn_slot_of<C>(li * CPR + (12 * role + ks) * 2 + half));
        afr[1][ks] = *(const bf16x8*)(stage_par + ffn_slot_of<C>(li * CPR + (12 * role + ks) * 2 + half));
    }
    f32x16 o[2][6];
    f32x16 sA[2], sB[2];
    bf16x8 pA[2], pB[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nb][i][r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[nb][r] = 0.f; sB[nb][r] = 0.f; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) { pA[0][r] = 0; pA[1][r] = 0; pB[0][r] = 0; pB[1][r] = 0; }

    // exchange slots (per lane): iteration t writes X / PX [own wave][t & 1] and reads X / PX [partner][(t - 1) & 1]
    char* x_mine = xbase + wave * 8192 + lane * 16;
    const char* x_part = xbase + (wave ^ 1) * 8192 + lane * 16;
    char* px_mine = xbase + 32768 + wave * 4096 + lane * 16;
    const char* px_part = xbase + 32768 + (wave ^ 1) * 4096 + lane * 16;
    const float* b1base = role ? lb1 + HID : lb1;                  // second wave of the pair: zeros (the bias enters the sum once)
    const int b1step = role ? 0 : 32;
#define FFP_DMA(t) w1img + (size_t)((t) < NCH ? (t) + 1 : 0) * CHB, lds_w1 + (((t) + 1) & 1) * CHB, (t) < NCH, \
                   w2img + (size_t)((t) >= 1 ? (t) - 1 : 0) * CHB, lds_w2 + (((t) + 1) & 1) * CHB, (t) >= 1, uwave
#define FFP_X(t) x_part + (((t) + 1) & 1) * 4096, x_mine + ((t) & 1) * 4096, px_part + (((t) + 1) & 1) * 2048, px_mine + ((t) & 1) * 2048
#define FFP_SYNC() ffn_wait_dma(); __syncthreads()
    FFP_SYNC();
    ffn_pair_iter<true, false, false, true>(afr, o, sA, sB[0], pB, pA, w1p, w2p, 0, b1base, half, FFP_X(0), FFP_DMA(0));
    FFP_SYNC();
    ffn_pair_iter<true, true, false, true>(afr, o, sB, sA[0], pA, pB, w1p, w2p, 1, b1base + b1step, half, FFP_X(1), FFP_DMA(1));
#pragma unroll 1
    for (int t = 2; t < NCH; t += 2) {
        FFP_SYNC();                  // even t: S(t) -> sA, reduce + GELU(sB) -> pB, GEMM2 reads pA (own) and the partner's P(t-2)
        ffn_pair_iter<true, true, true, true>(afr, o, sA, sB[0], pB, pA, w1p, w2p, 0, b1base + t * b1step, half, FFP_X(t), FFP_DMA(t));
        FFP_SYNC();
        ffn_pair_iter<true, true, true, true>(afr, o, sB, sA[0], pA, pB, w1p, w2p, 1, b1base + (t + 1) * b1step, half, FFP_X(t + 1), FFP_DMA(t + 1));
    }
    // residual tile of the own block, whole lines (the A^T registers are dead from here on)
    u32x4 xq[NL];
    {
        int xln = lane;
        asm volatile("" : "+v"(xln));
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t gb = tile_b + (size_t)(i * 64 + xln) * 16;
            xq[i] = *(const u32x4*)((const char*)X + (gb < last_b ? gb : last_b));
        }
    }
    FFP_SYNC();                      // t = NCH (even): reduce + GELU(S(NCH-1) in sB) -> pB, GEMM2(chunk NCH-2) reads pA; DMA W2[NCH-1]
    ffn_pair_iter<false, true, true, true>(afr, o, sA, sB[0], pB, pA, w1p, w2p, 0, b1base, half, FFP_X(NCH), FFP_DMA(NCH));
    FFP_SYNC();                      // t = NCH + 1: GEMM2(chunk NCH-1) reads pB and the partner's
    ffn_pair_iter<false, false, true, false>(afr, o, sB, sA[0], pA, pB, w1p, w2p, 1, b1base, half, FFP_X(NCH + 1), FFP_DMA(NCH));
#undef FFP_DMA
#undef FFP_X
#undef FFP_SYNC

    // ---- epilogue: both waves of a pair hold channels 192 role .. + 191 of BOTH blocks.  Every wave puts the residual tile of its own
    // block into its stage; after a barrier each wave updates its channel half in its own and in the partner's stage
    // (x + ls * (o + b2), one rounding to bf16); after another barrier every wave stores its own block as whole lines.
    __syncthreads();
    int eln = lane;
    asm volatile("" : "+v"(eln));
    const int eli = eln & 31, ehalf = eln >> 5;
#pragma unroll
    for (int i = 0; i < NL; ++i) *(u32x4*)(stage_own + ffn_slot_of<C>(i * 64 + eln)) = xq[i];
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        char* stg = nb == 0 ? stage_own : stage_par;
#pragma unroll
        for (int nf = 0; nf < 6; ++nf)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = (6 * role + nf) * 32 + 8 * q + 4 * ehalf;
                char* slot = stg + ffn_slot_of<C>(eli * CPR + (6 * role + nf) * 4 + q) + ehalf * 8;
                const f32x4 bv = *(const f32x4*)(b2 + n0), lv = *(const f32x4*)(ls + n0);
                const f32x4 rv = bf4_to_f32(*(const bf16x4*)slot);
                f32x4 v;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = rv[jj] + lv[jj] * (o[nb][nf][4 * q + jj] + bv[jj]);
                *(bf16x4*)slot = f32_to_bf4(v);
            }
    }
    __syncthreads();
    const int rows_ok = M - row_own;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int id = i * 64 + eln;
        const u32x4 v = *(const u32x4*)(stage_own + ffn_slot_of<C>(id));
        if (id < rows_ok * CPR) *(u32x4*)((char*)X + tile_b + (size_t)id * 16) = v;
    }
}

static hipError_t launch_ffn_pair384(hipStream_t st, const bf16* A, const char* w1img, const char* w2img, const float* b1,
                                     const float* b2, const float* ls, bf16* X, int M)
{
    const int nwg = (M + 127) / 128;
    const size_t shmem = (size_t)4 * 64 * 384 + 49152 + (size_t)4 * 384 * 4 + 128 + 256;
    static bool attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)ffn_pair384_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(ffn_pair384_kernel, dim3(nwg), dim3(256), shmem, st, A, w1img, w2img, b1, b2, ls, X, M, nwg);
    return hipGetLastError();
}
#endif   // FVHD_FFN_PAIR384

template <typename K> static hipError_t ffn_set_lds(K kernel, size_t shmem, bool* done)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done[dev & 63]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e == hipSuccess) done[dev & 63] = true;
    return e;
}

template <int C, int NB, int WAVES, int VAR = 0, int PF = 3, int OCC = WAVES / 4, bool F16 = ffn_f16<C>()>
static hipError_t launch_ffn(hipStream_t st, const bf16* A, const char* w1img, const char* w2img, const float* b1,
                             const float* b2, const float* ls, bf16* X, int M)
{
    constexpr int ROWS = 32 * NB * WAVES;
    const int nwg = (M + ROWS - 1) / ROWS;
    const size_t rings = (size_t)4 * 64 * C, stage_end = (C == 384 ? (FVHD_FFN_CIO384 ? (size_t)2 * 64 * C : 0) : 0) + ((C <= 192 || FVHD_FFN_CIO384) ? (size_t)WAVES * 32 * C * 2 : 0);
    const size_t shmem = (stage_end > rings ? stage_end : rings) + (size_t)4 * C * 4 + 256;
    static bool attr_set[64];                // the attribute is per device
    hipError_t e = ffn_set_lds(ffn_fused_kernel<C, NB, WAVES, VAR, PF, OCC, F16>, shmem, attr_set);
    if (e != hipSuccess) return e;
    // (Round 3 tried starting half of the first generation of workgroups 6-45 us late so that the memory phases of one half of the chip fall
    // into the chunk loops of the other: no gain in sustained operation - 354 / 369 / 493 us per launch at C = 384 / 192 / 96 with or
    // without, the whole step 26.83 -> 26.84 .. 27.4 ms - back-to-back launches already overlap at their tails; profiles/r03_ffn_stagger.log.)
    hipLaunchKernelGGL((ffn_fused_kernel<C, NB, WAVES, VAR, PF, OCC, F16>), dim3(nwg), dim3(WAVES * 64), shmem, st, A, w1img, w2img, b1, b2, ls, X, M, nwg);
    return hipGetLastError();
}

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

static uint16_t to_f16(float f)       // round-to-nearest-even, subnormals kept, saturating at +-65504 (a weight never gets there)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);                // >= 65520 rounds past the largest finite: saturate
    if (a < 0x33000001u) return sign;                                       // < 2^-25 (or the tie at 2^-25) -> 0
    const int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                               // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                             // normal: keep 11 bits; subnormal: fewer
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
    m >>= shift;
    if (rest > half || (rest == half && (m & 1u))) ++m;
    const uint32_t bits = e >= -14 ? (uint32_t)((e + 15 - 1) << 10) + m : m;   // the hidden bit of m carries into the exponent field
    return (uint16_t)(sign | bits);
}

// precision: FVHD_FFN_HALF (0) = the half-precision hidden activation (gelu16_stage; the default), FVHD_FFN_BF16 (1) = the f32 GELU with a
// bf16 hidden operand (no range limit: for blocks whose fc1 output can exceed the f16 form's 262 016, fvhd_audit_ranges)
extern "C" int fvhd_ffn_pack_host(int C, const float* fc1, const float* fc2, uint16_t* w1img, uint16_t* w2img, int precision)
{
    if (!fvhd_ffn_fused_supported(C) || precision < 0 || precision > 1) return 1;
    // half-precision form (see gelu16_stage): W1 carries the factor 1/4 (exact in bf16), W2 is f16(4 W2)
    const bool f16 = precision == 0;
    const float s1 = f16 ? 0.25f : 1.0f;
    const int HID = 4 * C, NCH = HID / 32, CHE = 32 * C;   // bf16 elements per chunk image
    for (int i = 0; i < (NCH + 1) * CHE; ++i) w1img[i] = 0;
    for (int ch = 0; ch < NCH; ++ch) {
        uint16_t* i1 = w1img + (size_t)ch * CHE;
        uint16_t* i2 = w2img + (size_t)ch * CHE;
        for (int row = 0; row < 32; ++row)
            for (int slot = 0; slot < C / 8; ++slot) {
                const int off = (C == 384 ? w1_off<384>(row, slot) : C == 192 ? w1_off<192>(row, slot) : w1_off<96>(row, slot)) / 2;
                for (int e = 0; e < 8; ++e) i1[off + e] = to_bf16(s1 * fc1[(size_t)(ch * 32 + row) * C + slot * 8 + e]);
            }
        for (int n = 0; n < C; ++n)
            for (int slot = 0; slot < 4; ++slot) {
                const int off = w2_off(n, slot) / 2;
                for (int e = 0; e < 8; ++e) {
                    const int pos = slot * 8 + e, kb = pos >> 4, hf = (pos >> 3) & 1, j = pos & 7;
                    const int h = 16 * kb + 8 * (j >> 2) + 4 * hf + (j & 3);
                    const float w = fc2[(size_t)n * HID + ch * 32 + h];
                    i2[off + e] = f16 ? to_f16(4.0f * w) : to_bf16(w);
                }
            }
    }
    return 0;
}

// A [M,C] bf16; w1img / w2img from fvhd_ffn_pack_host (device copies); b1 [4C], b2 [C], ls [C] fp32; X [M,C] in/out.
// largest |4 * W2| the half-precision form can hold: a block whose fc2 weights exceed it must be packed with FVHD_FFN_BF16
extern "C" float fvhd_ffn_half_w2_limit(void) { return 65504.0f / 4.0f; }

extern "C" int fvhd_launch_ffn_fused(hipStream_t st, const void* A, const void* w1img, const float* b1, const void* w2img,
                                     const float* b2, const float* ls, void* X, int M, int C, int precision)
{
    const bf16* a = (const bf16*)A;
    const char* w1 = (const char*)w1img;
    const char* w2 = (const char*)w2img;
    bf16* x = (bf16*)X;
    hipError_t e = hipErrorInvalidValue;
    if (M <= 0) return (int)e;
#ifdef FVHD_FFN_ABLATE
    // ablation build only (libfvhd_ablate.so): FVHD_FFN_VARIANT = 1 PF 4, 2 no GELU math (wrong results), 3 compiler-scheduled iteration,
    // 4 scalar GELU, 5 PF 2, 6 no weight DMA (wrong results), 7 no GELU + no DMA
    static const int variant = [] { const char* ev = getenv("FVHD_FFN_VARIANT"); return ev ? atoi(ev) : 0; }();
#define FFN_V(CC, OCC)                                                                                                  \
    switch (variant) {                                                                                                 \
    case 1: return (int)launch_ffn<CC, 1, 4, 0, 4, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                              \
    case 2: return (int)launch_ffn<CC, 1, 4, 2, 3, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                              \
    case 3: return (int)launch_ffn<CC, 1, 4, 16, 3, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                             \
    case 4: return (int)launch_ffn<CC, 1, 4, 64, 3, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                             \
    case 5: return (int)launch_ffn<CC, 1, 4, 0, 2, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                              \
    case 6: return (int)launch_ffn<CC, 1, 4, 1, 3, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                              \
    case 7: return (int)launch_ffn<CC, 1, 4, 3, 3, OCC>(st, a, w1, w2, b1, b2, ls, x, M);                              \
    default: break;                                                                                                    \
    }
    if (C == 384) { FFN_V(384, 1) } else if (C == 192) { FFN_V(192, 2) } else if (C == 96) { FFN_V(96, 3) }
#undef FFN_V
#endif
    if (precision == 1) {      // f32 GELU, bf16 hidden operand (the round-1..3a kernel): no range limit on the fc1 output
        if (C == 384) e = launch_ffn<384, 1, 4, 0, 3, 1, false>(st, a, w1, w2, b1, b2, ls, x, M);
        else if (C == 192) e = launch_ffn<192, 1, 4, 0, 3, 2, false>(st, a, w1, w2, b1, b2, ls, x, M);
        else if (C == 96) e = launch_ffn<96, 1, 4, 0, 3, 3, false>(st, a, w1, w2, b1, b2, ls, x, M);
        return (int)e;
    }
    if (precision != 0) return (int)e;
#ifndef FVHD_FFN_OCC96
#define FVHD_FFN_OCC96 2
#endif
#ifndef FVHD_FFN_NB192
#define FVHD_FFN_NB192 1             // 32-row blocks per wave at C = 192 (2: 64 rows per wave, 256 per workgroup, one wave per SIMD)
#endif
#ifndef FVHD_FFN_NB96
#define FVHD_FFN_NB96 1
#endif
#ifndef FVHD_FFN_W192
#define FVHD_FFN_W192 4              // waves per workgroup at C = 192 (8: 256 rows share one weight stream, one workgroup per CU, still two waves per SIMD)
#endif
#ifndef FVHD_FFN_W96
#define FVHD_FFN_W96 4               // ... at C = 96 (12: 384 rows per workgroup, three waves per SIMD)
#endif
#if FVHD_FFN_PAIR384
    if (C == 384) e = launch_ffn_pair384(st, a, w1, w2, b1, b2, ls, x, M);
    else
#endif
    if (C == 384) e = launch_ffn<384, 1, 4>(st, a, w1, w2, b1, b2, ls, x, M);
    // measured alternatives, compiled only on request (profiles/r04_ffn_nb2_power.log, r04_ffn_pairwait_cio384_w8_power.log): 64 rows per
    // wave (NB 2) 341 -> 356 us at C = 192, 455 -> 524 at C = 96; 8- / 12-wave workgroups sharing one weight stream 342 -> 352 / 463 -> 487
#if FVHD_FFN_W192 == 8
    else if (C == 192) e = launch_ffn<192, 1, 8, 0, 3, 2>(st, a, w1, w2, b1, b2, ls, x, M);
#endif
#if FVHD_FFN_W96 == 12
    else if (C == 96) e = launch_ffn<96, 1, 12, 0, 3, 3>(st, a, w1, w2, b1, b2, ls, x, M);
#endif
#if FVHD_FFN_NB192 == 2
    else if (C == 192) e = launch_ffn<192, 2, 4, 0, 3, 1>(st, a, w1, w2, b1, b2, ls, x, M);
#endif
#if FVHD_FFN_NB96 == 2
    else if (C == 96) e = launch_ffn<96, 2, 4, 0, 3, FVHD_FFN_OCC96>(st, a, w1, w2, b1, b2, ls, x, M);
#endif
    else if (C == 192) e = launch_ffn<192, 1, 4, 0, 3, 2>(st, a, w1, w2, b1, b2, ls, x, M);
    // C = 96: 152 registers since the epilogue offsets stopped being hoisted -> three workgroups (waves) per SIMD: the kernel is
    // VALU-issue-bound (16 GELUs per 12 MFMAs), a third instruction stream per SIMD is what it needs
    else if (C == 96) e = launch_ffn<96, 1, 4, 0, 3, 3>(st, a, w1, w2, b1, b2, ls, x, M);
    return (int)e;
}
