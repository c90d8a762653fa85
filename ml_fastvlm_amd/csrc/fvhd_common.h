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
//   Phi(x) = 0.5 + xc * Q(xc^2),  xc = clamp(x, -4, 4),  Q = degree-9 polynomial (least-squares fit on Chebyshev nodes of
//   u = x^2 in [0, 16]); |Phi error| <= 6.2e-6 in fp32 Horner form on [-4, 4] (fit 1.7e-6 + cancellation), i.e.
//   |gelu error| <= 6.2e-6 |x|: <= 1/8 of a bf16 half-ulp wherever gelu(x) is not itself below 1e-3.  Outside the clamp
//   Phi saturates at Phi(+-4) = 1 - 3.2e-5 / 3.2e-5 (gelu(-8) = -2.5e-4 instead of 0).
// Why not A&S 7.1.26 (round 1, 1.5e-7): that form needs v_rcp + v_exp (quarter rate, 8 cycles each) and half-rate
// abs/max fix-ups - ~50 VALU cycles per wave64 value; this one is 12 full-rate FMA/MUL + one v_med3 (~30 cycles) and
// packs two values per v_pk_fma_f32, halving the issue slots it takes from the MFMA stream in the fused ConvFFN kernel.
// Every activation it feeds is rounded to bf16 (relative 2e-3 half-ulp) right after.
#define FVHD_GELU_C0 3.989380888e-01f
#define FVHD_GELU_C1 -6.647037283e-02f
#define FVHD_GELU_C2 9.945140159e-03f
#define FVHD_GELU_C3 -1.168552637e-03f
#define FVHD_GELU_C4 1.084709610e-04f
#define FVHD_GELU_C5 -7.841504780e-06f
#define FVHD_GELU_C6 4.224180292e-07f
#define FVHD_GELU_C7 -1.572596130e-08f
#define FVHD_GELU_C8 3.561182860e-10f
#define FVHD_GELU_C9 -3.658831230e-12f
FVHD_DEV float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = xc * xc;
    float q = __builtin_fmaf(FVHD_GELU_C9, u, FVHD_GELU_C8);
    q = __builtin_fmaf(q, u, FVHD_GELU_C7);
    q = __builtin_fmaf(q, u, FVHD_GELU_C6);
    q = __builtin_fmaf(q, u, FVHD_GELU_C5);
    q = __builtin_fmaf(q, u, FVHD_GELU_C4);
    q = __builtin_fmaf(q, u, FVHD_GELU_C3);
    q = __builtin_fmaf(q, u, FVHD_GELU_C2);
    q = __builtin_fmaf(q, u, FVHD_GELU_C1);
    q = __builtin_fmaf(q, u, FVHD_GELU_C0);
    return x * __builtin_fmaf(xc, q, 0.5f);
}

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
FVHD_DEV unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
