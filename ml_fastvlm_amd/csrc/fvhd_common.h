// Shared device helpers for the FastViTHD gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FVHD_F32 0
#define FVHD_F16 1
#define FVHD_BF16 2

#define FVHD_DEV __device__ __forceinline__

// ---- bf16 <-> f32 ---------------------------------------------------------------------------------
FVHD_DEV f32x8 bf8_to_f32(bf16x8 v) { return __builtin_convertvector(v, f32x8); }
FVHD_DEV f32x4 bf4_to_f32(bf16x4 v) { return __builtin_convertvector(v, f32x4); }
FVHD_DEV bf16x8 f32_to_bf8(f32x8 v) { return __builtin_convertvector(v, bf16x8); }   // v_cvt_pk_bf16_f32 (RNE)
FVHD_DEV bf16x4 f32_to_bf4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }

// ---- erf GELU -------------------------------------------------------------------------------------
// gelu(x) = x * Phi(x) - the *erf* GELU the reference uses (nn.GELU() default, mci.py:108/387/870), not the tanh form.
//   Phi(x) = 0.5 + xc * Q(xc^2),  xc = clamp(x, -4, 4),  Q = degree-7 polynomial: minimax fit (LP on 6000 Chebyshev nodes
//   of [0, 4]) under the constraint Phi(4) = 1, with C0 nudged by 3e-7 so that the fp32 Horner chain gives Q(16) = 0.125
//   EXACTLY: Phi(+-4) = 1 / 0 bit-exactly, i.e. gelu(x) = x for x >= 4 and 0 for x <= -4 whatever the magnitude of the
//   pre-activation (the round-1 degree-9 fit saturated at Phi(-4) = 3.2e-5: gelu(-60) = -1.9e-3).
//   |Phi error| <= 3.3e-5 on [-4, 4] (= 1 - Phi(4): the price of the exact tails), |gelu error| <= 1.3e-4 absolute,
//   <= 5e-5 relative for x > 0: 1/40 of the bf16 half-ulp every activation is rounded with right after.
// 11 full-rate VALU per value (v_med3, v_mul, 7 v_fma, v_fma, v_mul); no v_rcp / v_exp (quarter rate).  The ConvFFN
// kernels spend 14-47 % of their chunk loop on it (tools/ubench/ffn_mix.hip), hence the low degree.
#define FVHD_GELU_CLAMP 4.0f
#define FVHD_GELU_C0 3.988050222e-01f
#define FVHD_GELU_C1 -6.606452912e-02f
#define FVHD_GELU_C2 9.582614526e-03f
#define FVHD_GELU_C3 -1.021786709e-03f
#define FVHD_GELU_C4 7.637563249e-05f
#define FVHD_GELU_C5 -3.731796596e-06f
#define FVHD_GELU_C6 1.056969481e-07f
#define FVHD_GELU_C7 -1.304839459e-09f
FVHD_DEV float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -FVHD_GELU_CLAMP, FVHD_GELU_CLAMP);
    const float u = xc * xc;
    float q = __builtin_fmaf(FVHD_GELU_C7, u, FVHD_GELU_C6);
    q = __builtin_fmaf(q, u, FVHD_GELU_C5);
    q = __builtin_fmaf(q, u, FVHD_GELU_C4);
    q = __builtin_fmaf(q, u, FVHD_GELU_C3);
    q = __builtin_fmaf(q, u, FVHD_GELU_C2);
    q = __builtin_fmaf(q, u, FVHD_GELU_C1);
    q = __builtin_fmaf(q, u, FVHD_GELU_C0);
    return x * __builtin_fmaf(xc, q, 0.5f);
}
// The fused ConvFFN kernel (ffn_fused.hip) alone evaluates a degree-5 Q on clamp(x, +-3.5): its GELU output is rounded to bf16 as the
// MFMA operand P on the spot, and the polynomial's error - |Phi error| <= 2.33e-4 (= 1 - Phi(3.5): exact upper tail, Phi(-3.5) = 3e-8),
// |gelu error| <= 8.2e-4 absolute, rel-L2 2.5e-4 on N(0,1) pre-activations - is a quarter of that rounding (rel-L2 1.1e-3): the
// teacher-forced RepMixerBlock steps measure 5.9e-4 with either polynomial (round 3, gpurun c1), while steps whose OUTPUT is a GELU
// (stem, PatchEmbed) would go from 1.2e-3 to 2.3e-3 - so every other kernel keeps the degree-7 fit above.  9 instead of 11 VALU per
// value; the GELU is issue-bound work that does not overlap the MFMAs (tools/ubench/ffn_mix.hip): C = 96 / 192 launches -4 %.
#define FVHD_GELU5_CLAMP 3.5f
#define FVHD_GELU5_C0 3.980601132e-01f
#define FVHD_GELU5_C1 -6.438287348e-02f
#define FVHD_GELU5_C2 8.499878459e-03f
#define FVHD_GELU5_C3 -7.195603685e-04f
#define FVHD_GELU5_C4 3.409395140e-05f
#define FVHD_GELU5_C5 -6.780236390e-07f

FVHD_DEV float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// ---- XCD-aware block id remap (guide T1, bijective form) -----------------------------------------
// Hardware places block b on XCD b % 8; give each XCD a contiguous run of logical tiles so that
// neighbouring tiles (which share operand panels) hit the same private L2.
FVHD_DEV int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- range-guard maxima (round 5) ------------------------------------------------------------------------------------
// A kernel that reduces max |.| for the range guard publishes it with ONE atomicMax per workgroup into slot blockIdx.x % FVHD_AMAX_SLOTS of a
// row of FVHD_AMAX_SLOTS words (fp32 bit patterns of non-negative numbers); the reader takes the maximum of the row.  One word for the whole
// launch (first version) serialised 1536-6144 atomics on one address as the workgroups of a launch finish together: +10 us on a 52-us launch.
#define FVHD_AMAX_SLOTS 64

// ---- wave64 reductions ----------------------------------------------------------------------------
FVHD_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- LDS-DMA ---------------------------------------------------------------------------------------
// global_load_lds_dwordx4: every active lane copies 16 B from its own global address straight into LDS at
// M0 + lane*16 - no VGPR round trip.  M0 is compiler-reserved: saved and restored inside the statement
// (cdna_hip_programming.md 5.7).  Tracked by vmcnt; the consumer waits with s_waitcnt vmcnt(0) before its barrier.
FVHD_DEV void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// same, address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset
FVHD_DEV void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// 4 B/lane variant: used as an L2 prefetch ("touch": every lane names one 128-B line, the bytes land in a scratch LDS
// area nobody reads) - no VGPR destination, so nothing to keep alive or to wait for beyond the usual vmcnt
FVHD_DEV void glds4(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// NP (<= 4) consecutive 1-KiB pieces in ONE statement: LDS [lds_dst + 1024 i, +1 KiB) <- global sbase + voff + 1024 i.  The
// instruction offset of global_load_lds advances the global AND the LDS address (tools/ubench/dma_offset.hip), so one M0
// setting serves four pieces: 2 + NP instructions instead of 5 NP, and no VALU address arithmetic at all (saddr form).
template <int NP> FVHD_DEV void glds16_run(const void* sbase, unsigned voff, unsigned lds_dst)
{
    static_assert(NP >= 1 && NP <= 4, "13-bit instruction offset");
    unsigned keep;
    if constexpr (NP == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (NP == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (NP == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else glds16_s(sbase, voff, lds_dst);
}
FVHD_DEV unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
