// Design microbenchmarks for the round-2 fused ConvFFN kernels (gfx950).  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
//
//  (1) mfma_ceiling : pure v_mfma_f32_32x32x16_bf16 stream; 1 or 2 waves per SIMD, NACC independent accumulators, zero or
//                     random operands.  Reports shader cycles per MFMA per SIMD (s_memtime) AND wall TF/s, so that the clock
//                     the chip holds under matrix load is visible (cycles say what the pipe does, TF/s what DVFS leaves).
//  (2) mix          : the chunk loop of the fused ConvFFN in isolation, per chunk of 32 hidden units at channel count C
//                       role pair (R):  wave G1 = GEMM1 (C/16 MFMAs into ONE accumulator, W1 fragments from LDS) + erf-GELU
//                                       of the previous chunk + P -> LDS;  wave G2 = P <- LDS, GEMM2 (C/16 MFMAs into C/32
//                                       accumulators, W2 fragments from LDS).  Two waves per SIMD, 8 waves per workgroup.
//                       symmetric (S):  every wave does GEMM1 + GELU + GEMM2 for its own 32 rows; 4 waves (one per SIMD, the
//                                       round-1 structure) or 8 waves (two per SIMD).
//                     GELU as scalar v_fma_f32 or packed v_pk_fma_f32; with or without it.  One s_barrier per chunk.
//                     Reports shader cycles per chunk per SIMD and the MFMA-pipe floor (32 cycles x MFMAs per SIMD-chunk).
//  (3) atomic       : global_atomic_pk_add_bf16 - rounding (bit-compare with round-to-nearest-even of the fp32 sum) and
//                     throughput of the epilogue-shaped access (lane = row, 4 B pieces at a row stride of 2C bytes).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DEV __device__ __forceinline__

#define GC0 3.989380888e-01f
#define GC1 -6.647037283e-02f
#define GC2 9.945140159e-03f
#define GC3 -1.168552637e-03f
#define GC4 1.084709610e-04f
#define GC5 -7.841504780e-06f
#define GC6 4.224180292e-07f
#define GC7 -1.572596130e-08f
#define GC8 3.561182860e-10f
#define GC9 -3.658831230e-12f
DEV float gelu1(float x)
{
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f), u = xc * xc;
    float q = __builtin_fmaf(GC9, u, GC8);
    q = __builtin_fmaf(q, u, GC7); q = __builtin_fmaf(q, u, GC6); q = __builtin_fmaf(q, u, GC5); q = __builtin_fmaf(q, u, GC4);
    q = __builtin_fmaf(q, u, GC3); q = __builtin_fmaf(q, u, GC2); q = __builtin_fmaf(q, u, GC1); q = __builtin_fmaf(q, u, GC0);
    return x * __builtin_fmaf(xc, q, 0.5f);
}
#define PK(c) (f32x2{c, c})
DEV f32x2 gelu2(f32x2 x)
{
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2 u = xc * xc;
    f32x2 q = __builtin_elementwise_fma(PK(GC9), u, PK(GC8));
    q = __builtin_elementwise_fma(q, u, PK(GC7)); q = __builtin_elementwise_fma(q, u, PK(GC6));
    q = __builtin_elementwise_fma(q, u, PK(GC5)); q = __builtin_elementwise_fma(q, u, PK(GC4));
    q = __builtin_elementwise_fma(q, u, PK(GC3)); q = __builtin_elementwise_fma(q, u, PK(GC2));
    q = __builtin_elementwise_fma(q, u, PK(GC1)); q = __builtin_elementwise_fma(q, u, PK(GC0));
    return x * __builtin_elementwise_fma(xc, q, PK(0.5f));
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(512) void mfma_ceiling(float* out, long long* cyc, int iters, int random)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const unsigned h = (lane * 2654435761u + j * 40503u + blockIdx.x * 97u) >> 8;
        a[j] = random ? (bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)) : (bf16)0.f;
        b[j] = random ? (bf16)(((int)((h >> 7) & 0xffff) - 32768) * (1.0f / 32768.f)) : (bf16)0.f;
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS image offsets (as ffn_fused.hip: conflict-free ds_read_b128 fragment reads)
template <int C> DEV int w1_off(int row, int slot)
{
    if constexpr (C == 384) return row * 768 + ((slot ^ (row & 15)) << 4);
    else if constexpr (C == 192) return row * 384 + ((slot ^ ((row >> 1) & 7)) << 4);
    else return row * 192 + ((slot ^ ((row >> 2) & 3)) << 4);
}
DEV int w2_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// GMODE: 0 no GELU (bias add + bf16 pack only), 1 scalar erf-GELU, 2 packed erf-GELU
template <int GMODE> DEV void gelu_pack(const f32x16& s, const float* b1, int half, bf16x8 (&p)[2])
{
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = *(const f32x4*)(b1 + 8 * q + 4 * half);
    float g[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float x0 = s[r] + bv[r >> 2][r & 3], x1 = s[r + 1] + bv[r >> 2][(r & 3) + 1];
        if constexpr (GMODE == 0) { g[r] = x0; g[r + 1] = x1; }
        else if constexpr (GMODE == 1) { g[r] = gelu1(x0); g[r + 1] = gelu1(x1); }
        else { const f32x2 y = gelu2(f32x2{x0, x1}); g[r] = y[0]; g[r + 1] = y[1]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bf16x4 lo = __builtin_convertvector(f32x4{g[8 * h], g[8 * h + 1], g[8 * h + 2], g[8 * h + 3]}, bf16x4);
        const bf16x4 hi = __builtin_convertvector(f32x4{g[8 * h + 4], g[8 * h + 5], g[8 * h + 6], g[8 * h + 7]}, bf16x4);
        p[h] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// role-pair structure: waves 0-3 = G1 of row blocks 0-3, waves 4-7 = G2 (wave w and w+4 share a SIMD)
template <int C, int GMODE>
__global__ __launch_bounds__(512) void mix_pair(const bf16* __restrict__ wsrc, float* out, long long* cyc, int iters)
{
    constexpr int KS = C / 16, NFR = C / 32, CHB = 64 * C;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [W1 2xCHB][W2 2xCHB][P 2 x 4 x 2 KB][b1 4 KB]
    char* w1r = smem;
    char* w2r = smem + 2 * CHB;
    char* pbuf = smem + 4 * CHB;
    float* lb1 = (float*)(smem + 4 * CHB + 16384);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    for (int i = tid; i < 4 * CHB / 16; i += 512) *(f32x4*)(smem + i * 16) = *(const f32x4*)((const char*)wsrc + (i * 16) % (1 << 20));
    for (int i = tid; i < 1024; i += 512) lb1[i] = 0.01f * (i & 63) - 0.3f;
    for (int i = tid; i < 16384 / 4; i += 512) ((float*)pbuf)[i] = 0.f;
    __syncthreads();
    const int pair = wave & 3;
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {                          // ---- G1
        bf16x8 afr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) afr[ks][j] = (bf16)(0.002f * ((lane * 7 + ks * 13 + j * 3) % 97) - 0.1f);
        f32x16 s_prev;
        for (int r = 0; r < 16; ++r) s_prev[r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const int ring = it & 1;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(w1r + ring * CHB + w1_off<C>(li, 2 * ks + half));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, afr[ks], s, 0, 0, 0);
            }
            bf16x8 p[2];
            gelu_pack<GMODE>(s_prev, lb1 + (it & 15) * 32, half, p);
            char* pw = pbuf + ring * 8192 + pair * 2048 + lane * 16;
            *(bf16x8*)pw = p[0];
            *(bf16x8*)(pw + 1024) = p[1];
            s_prev = s;
            __syncthreads();
        }
        float acc = 0;
        for (int r = 0; r < 16; ++r) acc += s_prev[r];
        out[blockIdx.x * 512 + tid] = acc;
    } else {                                 // ---- G2
        f32x16 o[NFR];
        for (int i = 0; i < NFR; ++i) for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const int ring = it & 1;
            const char* pr = pbuf + (ring ^ 1) * 8192 + pair * 2048 + lane * 16;
            const bf16x8 p0 = *(const bf16x8*)pr, p1 = *(const bf16x8*)(pr + 1024);
#pragma unroll
            for (int nf = 0; nf < NFR; ++nf) {
                const bf16x8 wa = *(const bf16x8*)(w2r + ring * CHB + nf * 2048 + w2_off(li, half));
                const bf16x8 wb = *(const bf16x8*)(w2r + ring * CHB + nf * 2048 + w2_off(li, 2 + half));
                o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, p0, o[nf], 0, 0, 0);
                o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, p1, o[nf], 0, 0, 0);
            }
            __syncthreads();
        }
        float acc = 0;
        for (int i = 0; i < NFR; ++i) for (int r = 0; r < 16; ++r) acc += o[i][r];
        out[blockIdx.x * 512 + tid] = acc;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// symmetric structure: every wave owns 32 rows: GEMM1(t) | GELU(t-1) | GEMM2(t-2), WAVES = 4 (one per SIMD) or 8
template <int C, int GMODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mix_sym(const bf16* __restrict__ wsrc, float* out, long long* cyc, int iters)
{
    constexpr int KS = C / 16, NFR = C / 32, CHB = 64 * C;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [W1 2xCHB][W2 2xCHB][b1 4 KB]
    char* w1r = smem;
    char* w2r = smem + 2 * CHB;
    float* lb1 = (float*)(smem + 4 * CHB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 31, half = lane >> 5;
    for (int i = tid; i < 4 * CHB / 16; i += WAVES * 64) *(f32x4*)(smem + i * 16) = *(const f32x4*)((const char*)wsrc + (i * 16) % (1 << 20));
    for (int i = tid; i < 1024; i += WAVES * 64) lb1[i] = 0.01f * (i & 63) - 0.3f;
    __syncthreads();
    bf16x8 afr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) afr[ks][j] = (bf16)(0.002f * ((lane * 7 + ks * 13 + j * 3) % 97) - 0.1f);
    f32x16 o[NFR], s_prev;
    for (int i = 0; i < NFR; ++i) for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    for (int r = 0; r < 16; ++r) s_prev[r] = 0.f;
    bf16x8 pp[2];
    for (int j = 0; j < 8; ++j) { pp[0][j] = (bf16)0.f; pp[1][j] = (bf16)0.f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int ring = it & 1;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 wf = *(const bf16x8*)(w1r + ring * CHB + w1_off<C>(li, 2 * ks + half));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, afr[ks], s, 0, 0, 0);
        }
        bf16x8 pn[2];
        gelu_pack<GMODE>(s_prev, lb1 + (it & 15) * 32, half, pn);
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) {
            const bf16x8 wa = *(const bf16x8*)(w2r + ring * CHB + nf * 2048 + w2_off(li, half));
            const bf16x8 wb = *(const bf16x8*)(w2r + ring * CHB + nf * 2048 + w2_off(li, 2 + half));
            o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, pp[0], o[nf], 0, 0, 0);
            o[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, pp[1], o[nf], 0, 0, 0);
        }
        pp[0] = pn[0]; pp[1] = pn[1];
        s_prev = s;
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < NFR; ++i) for (int r = 0; r < 16; ++r) acc += o[i][r];
    for (int r = 0; r < 16; ++r) acc += s_prev[r];
    out[blockIdx.x * WAVES * 64 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------------------
// X[row][n0 .. n0+3] += Y as two global_atomic_pk_add_bf16 per lane, lane = (row = lane & 31, half = lane >> 5), exactly the
// fused-FFN epilogue's ownership: lane holds channels nf*32 + 8q + 4*half .. +3 of its row.
DEV void atomic_pk_add_bf16(bf16* addr, unsigned v)
{
    asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(addr), "v"(v) : "memory");
}
template <int C>
__global__ __launch_bounds__(256) void atomic_epilogue(bf16* X, const bf16* Y, int M)
{
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + li;
    if (row >= M) return;
    bf16* xr = X + (size_t)row * C;
    const bf16* yr = Y + (size_t)row * C;
#pragma unroll
    for (int nf = 0; nf < C / 32; ++nf)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n0 = nf * 32 + 8 * q + 4 * half;
            const uint2 y = *(const uint2*)(yr + n0);
            atomic_pk_add_bf16(xr + n0, y.x);
            atomic_pk_add_bf16(xr + n0 + 2, y.y);
        }
}
// reference RMW with plain loads/stores (the round-1 epilogue's access shape)
template <int C>
__global__ __launch_bounds__(256) void rmw_epilogue(bf16* X, const bf16* Y, int M)
{
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + li;
    if (row >= M) return;
    bf16* xr = X + (size_t)row * C;
    const bf16* yr = Y + (size_t)row * C;
#pragma unroll
    for (int nf = 0; nf < C / 32; ++nf)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n0 = nf * 32 + 8 * q + 4 * half;
            const f32x4 a = __builtin_convertvector(*(const bf16x4*)(xr + n0), f32x4), b = __builtin_convertvector(*(const bf16x4*)(yr + n0), f32x4);
            *(bf16x4*)(xr + n0) = __builtin_convertvector(a + b, bf16x4);
        }
}

static uint16_t f2bf(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <typename F> static double time_ms(F launch, int reps = 3)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

static double avg_cycles(long long* dcyc, int n)
{
    std::vector<long long> h(n);
    CK(hipMemcpy(h.data(), dcyc, n * sizeof(long long), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < n; ++i) s += (double)h[i];
    return s / n;
}

template <int NACC> static void run_ceiling(float* out, long long* cyc, int threads, int random)
{
    const int iters = 20000 / NACC * 4;
    const double ms = time_ms([&] { hipLaunchKernelGGL(mfma_ceiling<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, random); });
    const double n_mfma = (double)iters * NACC;                       // per wave
    const double waves = 256.0 * threads / 64.0;
    const double tf = n_mfma * waves * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    const double c = avg_cycles(cyc, 256) / (n_mfma * (threads / 256.0));
    printf("mfma_ceiling waves/SIMD %d acc %2d %-6s: %7.1f TF/s  %5.1f shader cycles per MFMA per SIMD  (=> clock %.2f GHz)\n", threads / 256, NACC,
           random ? "random" : "zero", tf, c, n_mfma * (threads / 256.0) * c / (ms * 1e-3) / 1e9);
}

template <int C, int GMODE> static void run_pair(const bf16* w, float* out, long long* cyc)
{
    const int iters = 4000;
    const size_t sh = 4 * 64 * C + 16384 + 4096;
    CK(hipFuncSetAttribute((const void*)mix_pair<C, GMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const double ms = time_ms([&] { hipLaunchKernelGGL((mix_pair<C, GMODE>), dim3(256), dim3(512), sh, 0, w, out, cyc, iters); });
    const double c = avg_cycles(cyc, 256) / iters, fl = 2.0 * (C / 16) * 32;
    printf("mix pair  C=%3d gelu %d            : %7.1f cycles per chunk per SIMD (MFMA floor %4.0f = %4.1f %%), %8.2f us per 1000 chunks, clock %.2f GHz\n", C, GMODE, c, fl,
           100 * fl / c, ms * 1e3 / iters * 1000, c * iters / (ms * 1e-3) / 1e9);
}

template <int C, int GMODE, int WAVES> static void run_sym(const bf16* w, float* out, long long* cyc)
{
    const int iters = 4000;
    const size_t sh = 4 * 64 * C + 4096;
    CK(hipFuncSetAttribute((const void*)mix_sym<C, GMODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const double ms = time_ms([&] { hipLaunchKernelGGL((mix_sym<C, GMODE, WAVES>), dim3(256), dim3(WAVES * 64), sh, 0, w, out, cyc, iters); });
    const double per_simd = WAVES / 4.0;
    const double c = avg_cycles(cyc, 256) / iters, fl = 2.0 * (C / 16) * 32 * per_simd;
    printf("mix sym   C=%3d gelu %d waves %d    : %7.1f cycles per (chunk x %d waves/SIMD) (MFMA floor %4.0f = %4.1f %%), %8.2f us per 1000 iterations, clock %.2f GHz\n", C, GMODE,
           WAVES, c, (int)per_simd, fl, 100 * fl / c, ms * 1e3 / iters * 1000, c * iters / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv)
{
    const char* what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    float* out; long long* cyc; bf16* w;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&w, 1 << 20));
    {
        std::vector<uint16_t> h(1 << 19);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf(((int)(s >> 16) - 32768) * (0.05f / 32768.f)); }
        CK(hipMemcpy(w, h.data(), 1 << 20, hipMemcpyHostToDevice));
    }
    if (all || !strcmp(what, "ceiling")) {
        for (int random : {0, 1})
            for (int threads : {256, 512}) {
                run_ceiling<4>(out, cyc, threads, random);
                run_ceiling<8>(out, cyc, threads, random);
                run_ceiling<12>(out, cyc, threads, random);
            }
    }
    if (all || !strcmp(what, "mix")) {
        run_pair<384, 0>(w, out, cyc); run_pair<384, 1>(w, out, cyc); run_pair<384, 2>(w, out, cyc);
        run_sym<384, 1, 4>(w, out, cyc); run_sym<384, 2, 4>(w, out, cyc); run_sym<384, 0, 4>(w, out, cyc);
        run_pair<192, 0>(w, out, cyc); run_pair<192, 1>(w, out, cyc); run_pair<192, 2>(w, out, cyc);
        run_sym<192, 0, 8>(w, out, cyc); run_sym<192, 1, 8>(w, out, cyc); run_sym<192, 2, 8>(w, out, cyc);
        run_sym<192, 1, 4>(w, out, cyc); run_sym<192, 2, 4>(w, out, cyc);
        run_sym<96, 0, 8>(w, out, cyc); run_sym<96, 1, 8>(w, out, cyc); run_sym<96, 2, 8>(w, out, cyc);
        run_sym<96, 1, 4>(w, out, cyc);
        run_pair<96, 1>(w, out, cyc);
    }
    if (all || !strcmp(what, "atomic")) {
        constexpr int C = 384;
        const int M = 131072;
        const size_t n = (size_t)M * C;
        std::vector<uint16_t> hx(n), hy(n), got(n);
        uint32_t s = 777;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u; hx[i] = f2bf(((int)(s >> 16) - 32768) * (4.0f / 32768.f));
            s = s * 1664525u + 1013904223u; hy[i] = f2bf(((int)(s >> 16) - 32768) * (1.0f / 32768.f));
            if ((i & 1023) == 5) { hx[i] = f2bf(1.0f); hy[i] = f2bf(0.00390625f); }          // exact tie at bf16: 1 + 2^-8 -> even = 1.0
            if ((i & 1023) == 6) { hx[i] = f2bf(1.0078125f); hy[i] = f2bf(0.00390625f); }    // tie: 1.0078125 + 2^-8 -> even = 1.015625
            if ((i & 1023) == 7) { hx[i] = f2bf(1e-39f); hy[i] = f2bf(2e-39f); }              // subnormals
        }
        bf16 *dx, *dy;
        CK(hipMalloc(&dx, n * 2)); CK(hipMalloc(&dy, n * 2));
        CK(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), n * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(atomic_epilogue<C>, dim3(M / 128), dim3(256), 0, 0, dx, dy, M);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dx, n * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, bad_tie = 0, bad_sub = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint16_t want = f2bf(bf2f(hx[i]) + bf2f(hy[i]));
            if (got[i] != want) {
                ++bad;
                if ((i & 1023) == 5 || (i & 1023) == 6) ++bad_tie;
                if ((i & 1023) == 7) ++bad_sub;
                if (bad <= 5) printf("  atomic mismatch at %zu: x %g y %g got %g want %g\n", i, bf2f(hx[i]), bf2f(hy[i]), bf2f(got[i]), bf2f(want));
            }
        }
        printf("atomic pk_add_bf16 vs RNE(fp32 sum): %zu of %zu differ (ties %zu, subnormal %zu)\n", bad, n, bad_tie, bad_sub);
        const double ta = time_ms([&] { hipLaunchKernelGGL(atomic_epilogue<C>, dim3(M / 128), dim3(256), 0, 0, dx, dy, M); }, 5);
        const double tr = time_ms([&] { hipLaunchKernelGGL(rmw_epilogue<C>, dim3(M / 128), dim3(256), 0, 0, dx, dy, M); }, 5);
        printf("epilogue-shaped X += Y over %d x %d bf16 (%.0f MB each): atomics %.1f us, plain load/add/store %.1f us\n", M, C, n * 2 / 1e6, ta * 1e3, tr * 1e3);
    }
    return 0;
}
