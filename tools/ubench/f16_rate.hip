// Is a half-precision GELU worth it inside the fused ConvFFN (gfx950)?   hipcc --offload-arch=gfx950 -O3 f16_rate.hip -o f16_rate
//
//  (1) rate   : shader cycles per wave64 instruction for the packed-f16 VALU family next to the f32 instructions the kernels use
//               today (16 independent dependency chains, 1 / 2 waves per SIMD).
//  (2) shadow : one wave per SIMD, a stream of independent v_mfma_f32_32x32x16_bf16 with N filler instructions of one kind behind
//               every MFMA: cycles per MFMA slot as N grows = how many fillers of that kind ride in the 32-cycle shadow for free.
//  (3) chunk  : the chunk loop of the fused ConvFFN without its LDS side (fragments from registers): NM MFMAs per chunk (48 / 24 /
//               12 = C 384 / 192 / 96), the erf-GELU of the previous chunk's 16 values per lane dealt out over the MFMA slots and
//               its result fed to the GEMM2 MFMAs as their B operand, in five forms:
//                 0 none | 1 packed f32 degree 5 (the production form) | 2 scalar f32 degree 5 (asm v_fma_f32) |
//                 3 packed f16 degree 5 -> f16 P (GEMM2 = v_mfma_f32_32x32x16_f16) | 4 packed f16 degree 4
//               reports shader cycles per chunk against the MFMA floor (32 x NM) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DEV __device__ __forceinline__

#define G5_CLAMP 3.5f
#define G5_C0 3.980601132e-01f
#define G5_C1 -6.438287348e-02f
#define G5_C2 8.499878459e-03f
#define G5_C3 -7.195603685e-04f
#define G5_C4 3.409395140e-05f
#define G5_C5 -6.780236390e-07f

// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rate_k(float* out, long long* cyc, int iters, float seed)
{
    unsigned u[16];
    float a[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x * 0.001f; u[i] = 0x3c003c00u + i + threadIdx.x; }
    const unsigned w = 0x38003800u + (unsigned)seed;      // (0.5, 0.5) as f16x2
    const float wf = seed * 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(wf));
                if (MODE == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w));
                if (MODE == 2) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(w));
                if (MODE == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(w));
                if (MODE == 4) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(w));
                if (MODE == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(a[i]));
                if (MODE == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(a[i]));
                if (MODE == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(a[i]));
                if (MODE == 8) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(wf));
                if (MODE == 9) asm volatile("v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(w));
                if (MODE == 10) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (MODE == 11 && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(f32x2*)&a[2 * i]) : "v"(f32x2{wf, wf}));
                if (MODE == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(wf));
                if (MODE == 13) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(u[i]) : : "a0");
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + __uint_as_float(u[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
template <int KIND, int N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void shadow_k(float* out, long long* cyc, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j) * seed); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned u[16];
    float f[16];
    for (int i = 0; i < 16; ++i) { f[i] = seed + i + lane * 0.001f; u[i] = 0x3c003c00u + i + lane; }
    const unsigned w = 0x38003800u + (unsigned)seed;
    const float wf = seed * 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int i = (m * N + n) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(wf));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(f32x2*)&f[i & 14]) : "v"(f32x2{wf, wf}));
                if (KIND == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w));
                if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(f[i]));
                if (KIND == 4) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(wf));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += f[i] + __uint_as_float(u[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// (2b) lds shadow: the same MFMA stream with N ds_read_b128 (1 KiB per wave, lane-linear = conflict-free) behind every MFMA, the data
// not consumed (USE = 0) or used as the A operand of the MFMA PF slots later (USE = 1, the production pattern: counted lgkmcnt).
// WPS waves per SIMD (1: 4 waves per CU; 2: 8 waves per CU).
template <int N, int USE, int WPS>
__global__ __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void lds_shadow_k(float* out, long long* cyc, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256 * WPS) ((float*)lds)[i] = 0.001f * (i & 1023) * seed;
    __syncthreads();
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j) * seed); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (wave & 3) * 8192;
    bf16x8 fr[8];      // reads are inline asm (the compiler neither reorders nor counts them): explicit lgkmcnt below
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[i] = a;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (USE) {                                       // fr[m] was requested three slots ago: at most the 2 younger reads may be in flight
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                asm volatile("" : "+v"(fr[m]));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m], b, acc[m], 0, 0, 0);
            } else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (USE) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(m + 3) & 7]) : "v"(base), "n"(((m + n) & 7) * 1024));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(m * N + n) & 7]) : "v"(base), "n"(((m * N + n) & 7) * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!USE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(fr[i])); s += (float)fr[i][0]; }
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// GELU forms, each cut into 6 stages per PAIR of values (the production kernel's half-stages) so that the dealing-out is identical
struct StF32 { f32x2 x, xc, u, q; };
template <int H> DEV void gelu_pk32(StF32& g, f32x2 x, f32x2& out)
{
    if constexpr (H == 0) { g.x = x; g.xc = f32x2{__builtin_amdgcn_fmed3f(x[0], -G5_CLAMP, G5_CLAMP), __builtin_amdgcn_fmed3f(x[1], -G5_CLAMP, G5_CLAMP)}; }
    else if constexpr (H == 1) { g.u = g.xc * g.xc; g.q = __builtin_elementwise_fma(f32x2{G5_C5, G5_C5}, g.u, f32x2{G5_C4, G5_C4}); }
    else if constexpr (H == 2) g.q = __builtin_elementwise_fma(g.q, g.u, f32x2{G5_C3, G5_C3});
    else if constexpr (H == 3) g.q = __builtin_elementwise_fma(g.q, g.u, f32x2{G5_C2, G5_C2});
    else if constexpr (H == 4) { g.q = __builtin_elementwise_fma(g.q, g.u, f32x2{G5_C1, G5_C1}); g.q = __builtin_elementwise_fma(g.q, g.u, f32x2{G5_C0, G5_C0}); }
    else out = g.x * __builtin_elementwise_fma(g.xc, g.q, f32x2{0.5f, 0.5f});
}
#define SFMA(d, a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c))
template <int H> DEV void gelu_sc32(StF32& g, f32x2 x, f32x2& out)
{
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if constexpr (H == 0) { g.x[e] = x[e]; g.xc[e] = __builtin_amdgcn_fmed3f(x[e], -G5_CLAMP, G5_CLAMP); }
        else if constexpr (H == 1) { asm volatile("v_mul_f32 %0, %1, %1" : "=v"(g.u[e]) : "v"(g.xc[e])); float c5 = G5_C5; asm volatile("v_mov_b32 %0, %1" : "=v"(g.q[e]) : "s"(c5)); SFMA(g.q[e], g.q[e], g.u[e], G5_C4); }
        else if constexpr (H == 2) SFMA(g.q[e], g.q[e], g.u[e], G5_C3);
        else if constexpr (H == 3) SFMA(g.q[e], g.q[e], g.u[e], G5_C2);
        else if constexpr (H == 4) { SFMA(g.q[e], g.q[e], g.u[e], G5_C1); SFMA(g.q[e], g.q[e], g.u[e], G5_C0); }
        else { float t; SFMA(t, g.xc[e], g.q[e], 0.5f); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(out[e]) : "v"(g.x[e]), "v"(t)); }
    }
}
// half precision.  The kernel would receive x' = x / 4 straight from GEMM1 (W1, b1 pre-scaled by 1/4 on the host, W2 by 4: exact), so
// that every coefficient of Phi = clamp01(0.5 + x' Q'(min(x'^2, (3.5/4)^2))) is O(1..10) in f16:  10 packed instructions per PAIR
// (cvt, mul, min, 5 fma, fma+clamp, mul), the result IS the f16 B operand of GEMM2 (v_mfma_f32_32x32x16_f16).
// tools/ubench/g16.py: relative error of the hidden activation 0.5-1.0e-3 against 1.7e-3 for f32 math + bf16 rounding.
struct StF16 { f16x2 x, xc, u, q; };
#define H2(c) (f16x2{(_Float16)(c), (_Float16)(c)})
template <int DEG, int H> DEV void gelu_pk16(StF16& g, f32x2 x, f16x2& out)
{
    constexpr float s = 4.0f, s2 = s * s;
    constexpr float c0 = G5_C0 * s, c1 = G5_C1 * s * s2, c2 = G5_C2 * s * s2 * s2, c3 = G5_C3 * s * s2 * s2 * s2, c4 = G5_C4 * s * s2 * s2 * s2 * s2,
                    c5 = G5_C5 * s * s2 * s2 * s2 * s2 * s2;
    if constexpr (H == 0) g.x = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
    else if constexpr (H == 1) {
        g.u = __builtin_elementwise_min(g.x * g.x, H2(G5_CLAMP * G5_CLAMP / s2));
        if constexpr (DEG == 5) g.q = __builtin_elementwise_fma(H2(c5), g.u, H2(c4)); else g.q = H2(c4);
    } else if constexpr (H == 2) g.q = __builtin_elementwise_fma(g.q, g.u, H2(c3));
    else if constexpr (H == 3) g.q = __builtin_elementwise_fma(g.q, g.u, H2(c2));
    else if constexpr (H == 4) { g.q = __builtin_elementwise_fma(g.q, g.u, H2(c1)); g.q = __builtin_elementwise_fma(g.q, g.u, H2(c0)); }
    else {
        f16x2 phi;
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(phi) : "v"(g.x), "v"(g.q), "v"(H2(0.5f)));
        out = g.x * phi;
    }
}

template <int NM, int G>
DEV void chunk_iter(const bf16x8 (&wfr)[4], const bf16x8 (&afr)[4], f32x16 (&o)[NM / 4], f32x16& s_out, const f32x16& s_in, bf16x8 (&p_out)[2], const bf16x8 (&p_in)[2])
{
    // s_out <- GEMM1 (this chunk); GELU(s_in = previous chunk) -> p_out; GEMM2 with p_in (the chunk before)
    constexpr int UPS = 96 / NM;              // GELU half-stage units (one pair, one stage) per MFMA slot: 2 / 4 / 8
    StF32 g32[8];
    StF16 g16[8];
    f32x2 o32[8];
    f16x2 o16[8];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if ((m & 1) == 0) {
            if (m == 0) s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[0], afr[0], o[NM / 4 - 1], 0, 0, 0);    // (starts from a varying accumulator: two identical chains would be merged)
            else s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[(m >> 1) & 3], afr[(m >> 1) & 3], s_out, 0, 0, 0);
        } else {
            const int gq = m >> 1;
            if (G >= 3) o[gq >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wfr[gq & 3]), __builtin_bit_cast(f16x8, p_in[gq & 1]), o[gq >> 1], 0, 0, 0);
            else o[gq >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[gq & 3], p_in[gq & 1], o[gq >> 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (G > 0) {
#pragma unroll
            for (int uu = m * UPS / 2; uu < (m + 1) * UPS / 2; ++uu) {
                // two pairs in flight, their stages alternating: consecutive VALU instructions are independent
                const int h = (uu % 12) / 2, r = 2 * (2 * (uu / 12) + (uu & 1));
                const f32x2 sv = {s_in[r], s_in[r + 1]};
#define ST(FN, ST_, O_) do { if (h == 0) FN<0>(ST_, sv, O_); else if (h == 1) FN<1>(ST_, sv, O_); else if (h == 2) FN<2>(ST_, sv, O_); \
                             else if (h == 3) FN<3>(ST_, sv, O_); else if (h == 4) FN<4>(ST_, sv, O_); else FN<5>(ST_, sv, O_); } while (0)
#define ST16(D, ST_, O_) do { if (h == 0) gelu_pk16<D, 0>(ST_, sv, O_); else if (h == 1) gelu_pk16<D, 1>(ST_, sv, O_); else if (h == 2) gelu_pk16<D, 2>(ST_, sv, O_); \
                              else if (h == 3) gelu_pk16<D, 3>(ST_, sv, O_); else if (h == 4) gelu_pk16<D, 4>(ST_, sv, O_); else gelu_pk16<D, 5>(ST_, sv, O_); } while (0)
                if (G == 1) ST(gelu_pk32, g32[r >> 1], o32[r >> 1]);
                if (G == 2) ST(gelu_sc32, g32[r >> 1], o32[r >> 1]);
                if (G == 3) ST16(5, g16[r >> 1], o16[r >> 1]);
                if (G == 4) ST16(4, g16[r >> 1], o16[r >> 1]);
                if (h == 5) {
                    if (G <= 2) {
                        if ((r & 3) == 2) {
                            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                            const b2 lo = __builtin_convertvector(o32[(r >> 1) - 1], b2), hi = __builtin_convertvector(o32[r >> 1], b2);
                            p_out[r >> 3][(r & 4) + 0] = lo[0]; p_out[r >> 3][(r & 4) + 1] = lo[1]; p_out[r >> 3][(r & 4) + 2] = hi[0]; p_out[r >> 3][(r & 4) + 3] = hi[1];
                        }
                    } else {
                        f16x8 t = __builtin_bit_cast(f16x8, p_out[r >> 3]);
                        t[(r & 6) + 0] = o16[r >> 1][0]; t[(r & 6) + 1] = o16[r >> 1][1];
                        p_out[r >> 3] = __builtin_bit_cast(bf16x8, t);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (G > 0) asm volatile("" : "+v"(p_out[0]), "+v"(p_out[1]));
    asm volatile("" : "+v"(s_out));
}

template <int NM, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void chunk_k(float* out, long long* cyc, int chunks, float seed)
{
    const int lane = threadIdx.x & 63;
    bf16x8 wfr[4], afr[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 8; ++j) { wfr[k][j] = (__bf16)(0.01f * ((lane * 7 + j * 3 + k) % 23 - 11) * seed); afr[k][j] = (__bf16)(0.02f * ((lane * 5 + j + k * 9) % 19 - 9)); }
    f32x16 o[NM / 4], s0, s1;
    for (int i = 0; i < NM / 4; ++i) for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.01f * r; s1[r] = 0.02f * r; }
    bf16x8 p0[2], p1[2];
    for (int j = 0; j < 8; ++j) { p0[0][j] = (__bf16)0.f; p0[1][j] = (__bf16)0.f; p1[0][j] = (__bf16)0.f; p1[1][j] = (__bf16)0.f; }
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int c = 0; c < chunks; c += 2) {
        chunk_iter<NM, G>(wfr, afr, o, s0, s1, p1, p0);
        chunk_iter<NM, G>(wfr, afr, o, s1, s0, p0, p1);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NM / 4; ++i) for (int r = 0; r < 16; ++r) s += o[i][r];
    for (int r = 0; r < 16; ++r) s += s0[r] + s1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
struct Res { double cyc, ms; };
template <typename K> Res launch(K kern, int grid, int iters, int block = 256)
{
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)grid * block * 4); (void)hipMalloc(&cyc, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters / 10 + 1, 1.0f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, cyc, iters, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return {(double)c, (double)ms};
}

template <int MODE> void rate(const char* name)
{
    const int iters = 4000;
    const double n = (double)iters * 4 * (MODE == 11 ? 8 : 16);
    const Res r1 = launch(rate_k<MODE>, 256, iters), r2 = launch(rate_k<MODE>, 512, iters);
    printf("rate   %-22s %6.2f cycles per instruction at 1 wave/SIMD, %6.2f per SIMD at 2 waves/SIMD\n", name, r1.cyc / n, r2.cyc / n / 2);
}
template <int KIND, int N> double shadow1() { const int iters = 2000; return launch(shadow_k<KIND, N>, 256, iters).cyc / (iters * 8.0); }
template <int KIND> void shadow(const char* name)
{
    printf("shadow %-18s cycles per MFMA slot with N fillers:  N=0 %5.1f  2 %5.1f  4 %5.1f  6 %5.1f  8 %5.1f  12 %5.1f  16 %5.1f\n", name,
           shadow1<KIND, 0>(), shadow1<KIND, 2>(), shadow1<KIND, 4>(), shadow1<KIND, 6>(), shadow1<KIND, 8>(), shadow1<KIND, 12>(), shadow1<KIND, 16>());
}
template <int N, int USE, int WPS> double ldsshadow1() { const int iters = 2000; return launch(lds_shadow_k<N, USE, WPS>, 256, iters, 256 * WPS).cyc / (iters * 8.0 * WPS); }
template <int USE, int WPS> void ldsshadow()
{
    printf("lds    ds_read_b128 %s, %d wave(s) per SIMD: cycles per MFMA per SIMD with N reads per MFMA:  N=0 %5.1f  1 %5.1f  2 %5.1f  3 %5.1f\n",
           USE ? "feeding the MFMAs" : "not consumed     ", WPS, ldsshadow1<0, USE, WPS>(), ldsshadow1<1, USE, WPS>(), ldsshadow1<USE ? 1 : 2, USE, WPS>(), ldsshadow1<USE ? 1 : 3, USE, WPS>());
}
template <int NM, int G> void chunk1(const char* gname)
{
    const int chunks = NM == 48 ? 2000 : NM == 24 ? 4000 : 8000;
    const Res r = launch(chunk_k<NM, G>, 256, chunks);
    printf("chunk  NM=%2d gelu %-22s %7.1f cycles per chunk (MFMA floor %4d = %4.1f %%)  %8.2f us per 1000 chunks  clock %.2f GHz\n", NM, gname,
           r.cyc / chunks, 32 * NM, 100.0 * 32 * NM / (r.cyc / chunks), r.ms * 1e3 / chunks * 1000, r.cyc / (r.ms * 1e6));
}
template <int NM> void chunk()
{
    chunk1<NM, 0>("none"); chunk1<NM, 1>("packed f32 deg 5"); chunk1<NM, 2>("scalar f32 deg 5"); chunk1<NM, 3>("packed f16 deg 5"); chunk1<NM, 4>("packed f16 deg 4");
}

int main()
{
    if (!getenv("F16_RATE_SKIP")) {
    rate<0>("v_fma_f32"); rate<12>("v_mul_f32"); rate<11>("v_pk_fma_f32"); rate<8>("v_med3_f32"); rate<1>("v_pk_fma_f16"); rate<2>("v_pk_mul_f16");
    rate<3>("v_pk_add_f16"); rate<4>("v_pk_max_f16"); rate<5>("v_cvt_pkrtz_f16_f32"); rate<6>("v_cvt_pk_f16_f32"); rate<7>("v_cvt_pk_bf16_f32");
    rate<9>("v_fma_mix_f32"); rate<10>("v_cvt_f32_f16"); rate<13>("v_accvgpr_read_b32");
    shadow<0>("v_fma_f32"); shadow<1>("v_pk_fma_f32"); shadow<2>("v_pk_fma_f16"); shadow<3>("v_cvt_pk_f16_f32"); shadow<4>("v_med3_f32");
    }
    ldsshadow<0, 1>(); ldsshadow<1, 1>(); ldsshadow<0, 2>(); ldsshadow<1, 2>();
    if (getenv("F16_RATE_CHUNKS")) { chunk<48>(); chunk<24>(); chunk<12>(); }
    return 0;
}
