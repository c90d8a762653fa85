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

// ---- exact-erf GELU -------------------------------------------------------------------------------
// gelu(x) = x * Phi(x),  Phi(x) = 0.5 * erfc(-x / sqrt(2)).
// erfc(z), z >= 0, by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16 resolution):
//   erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z).
// This is the *erf* GELU the reference uses (nn.GELU() default, mci.py:108/387/870), not the tanh form.
// Written for minimum VALU issue slots (11 full-rate VALU + v_rcp + v_exp):  gelu(x) = max(x, 0) - |x| * Phi(-|x|).
FVHD_DEV float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    float q = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    q = __builtin_fmaf(q, t, 0.5f * 1.421413741f);
    q = __builtin_fmaf(q, t, 0.5f * -0.284496736f);
    q = __builtin_fmaf(q, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((x * -0.72134752044448170f) * x);      // exp(-x^2 / 2)
    const float h = (q * t) * e;                                                   // Phi(-|x|) = 0.5 erfc(|x| / sqrt 2)
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));          // relu without the canonicalising self-max that fmaxf() costs
    return __builtin_fmaf(-ax, h, r);
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
