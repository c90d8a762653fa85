// Feasibility study (round 2): depthwise 7x7 on the 16-block 4x4x4 bf16 MFMA.
//   hipcc -O3 --offload-arch=gfx950 -o dw_mfma dw_mfma.hip && ./dw_mfma
// A depthwise conv has no channel mixing, so the only matrix shape inside it is the 1-D convolution along a row: for one
// channel, out[y, x0..x0+3] += in[y+ky-3, 4-px segment] x (4x4 Toeplitz block of the 7 taps of row ky).  The 16-block MFMA
// (v_mfma_f32_4x4x4_16b_bf16: sixteen independent 4x4x4 products per instruction) takes 16 channels as its blocks; with 4-px
// segments aligned to the 4-px output tiles, three segments cover the 10 inputs of a tile and 28 of the 48 products are
// taps (58 %): 128 * 0.58 = 74 useful FMA / cycle / SIMD against 32 for v_pk_fma_f32.
// Part 1 probes the operand / result layout of the instruction, part 2 its issue rate, part 3 a sliding-window prototype
// (one wave = 16 channels x a 16*NT-px strip, marching down the rows with 7 live output rows) checked against a CPU conv.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static inline u16 f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }
static inline float bf2f(u16 h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ inline u16 d_f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }

// ---------------------------------------------------------------- 1. layout probe
__global__ void probe_kernel(const u16* a, const u16* b, float* d)
{
    const int l = threadIdx.x;
    s16x4 A, Bv;
    for (int k = 0; k < 4; ++k) { A[k] = (short)a[l * 4 + k]; Bv[k] = (short)b[l * 4 + k]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A, Bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

static void probe()
{
    std::vector<u16> a(256), b(256);
    std::vector<float> af(256), bfv(256), d(256);
    srand(3);
    for (int i = 0; i < 256; ++i) { af[i] = (float)(rand() % 9 - 4); bfv[i] = (float)(rand() % 9 - 4); a[i] = f2bf(af[i]); b[i] = f2bf(bfv[i]); }
    u16 *da, *db; float* dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    probe_kernel<<<1, 64>>>(da, db, dd);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    // candidates: block = l / 4 or l % 16, in-block index = l % 4 or l / 16; result register = row (i) or column (j)
    for (int blk_mode = 0; blk_mode < 2; ++blk_mode)
        for (int dmode = 0; dmode < 2; ++dmode) {
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int blk = blk_mode ? l % 16 : l / 4, q = blk_mode ? l / 16 : l % 4;
                    const int i = dmode ? q : r, j = dmode ? r : q;            // dmode 0: lane holds column j = q, register r = row i
                    const int la = blk_mode ? i * 16 + blk : blk * 4 + i, lb = blk_mode ? j * 16 + blk : blk * 4 + j;
                    float want = 0;
                    for (int k = 0; k < 4; ++k) want += af[la * 4 + k] * bfv[lb * 4 + k];
                    bad += want != d[l * 4 + r];
                }
            printf("probe: block=%s result-register=%s : %s (%d mismatches)\n", blk_mode ? "lane%16" : "lane/4", dmode ? "column j" : "row i",
                   bad ? "no" : "MATCH", bad);
        }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
}

// ---------------------------------------------------------------- 2. issue rate
template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, long long* cyc)
{
    s16x4 A = {(short)(0x3f80 + threadIdx.x), 0x3f80, 0x3f00, 0x4000}, Bv = {0x3f80, (short)(0x3e80 + threadIdx.x), 0x3f80, 0x3f80};
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A, Bv, acc[i], 0, 0, 0);
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
static void rate(int waves_per_simd)
{
    float* out; long long* cyc;
    const int nb = 256 * waves_per_simd;          // 4 waves per block = 1 per SIMD of a CU
    CK(hipMalloc(&out, (size_t)nb * 256 * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<NACC><<<nb, 256>>>(out, 100, cyc);
    CK(hipEventRecord(e0));
    rate_kernel<NACC><<<nb, 256>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * NACC;
    printf("rate: %2d independent accumulators, %d wave(s)/SIMD: %.2f clock64-ticks per MFMA per wave, %.1f TFMA/s useful-peak (wall %.3f ms)\n",
           NACC, waves_per_simd, c / n, n * 1024.0 * nb * 4 / (ms * 1e-3) / 1e12, ms);
    (void)hipFree(out); (void)hipFree(cyc);
}

// ---------------------------------------------------------------- 3. sliding-window dw7x7 prototype
constexpr int P = 80;                 // LDS pitch (px) of one channel's row: (P / 2) % 64 == 40 -> the 16 channels fall on 8 bank groups, 2-way = the 512-B minimum
template <int NT, int C>
__global__ __launch_bounds__(256, 2) void dw7_mfma_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int B, int H, int W, int RC, int nstrip, int nchunk)
{
    constexpr int NG = C / 16, SW = 16 * NT, IWX = SW + 8, NLD = (IWX * 2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) u16 lds_all[4][16 * P];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u16* lds = lds_all[wv];
    const int blk = lane >> 2, q = lane & 3;
    int gw = blockIdx.x * 4 + wv;
    const int g = gw % NG; gw /= NG;
    const int strip = gw % nstrip; gw /= nstrip;
    const int chunk = gw % nchunk;
    const int n = gw / nchunk;
    if (n >= B) return;
    const int c0 = g * 16, x0 = strip * SW, ylo = chunk * RC, yhi = min(H, ylo + RC);

    s16x4 bop[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = 4 * (s - 1) + k - q + 3;
                const float v = (kx >= 0 && kx < 7) ? w[(size_t)(ky * 7 + kx) * C + c0 + blk] : 0.f;
                bop[ky][s][k] = (short)d_f2bf(v);
            }
    const float bv = bias[c0 + blk];
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};

    // staging: 16-B chunk e = m*64 + lane of the input row segment: LDS column e/2, channel half e%2
    uint4 st[NLD];
    auto issue = [&](int r) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int e = m * 64 + lane, px = e >> 1, hh = e & 1, xi = x0 - 4 + px;
            const bool ok = px < IWX && xi >= 0 && xi < W && r >= 0 && r < H;
            st[m] = ok ? *(const uint4*)&x[((size_t)(n * H + r) * W + xi) * C + c0 + 8 * hh] : uint4{0, 0, 0, 0};
        }
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int e = m * 64 + lane, px = e >> 1, hh = e & 1;
            if (px < IWX) {
                u16* d = lds + (8 * hh) * P + px;
                d[0 * P] = (u16)st[m].x; d[1 * P] = (u16)(st[m].x >> 16);
                d[2 * P] = (u16)st[m].y; d[3 * P] = (u16)(st[m].y >> 16);
                d[4 * P] = (u16)st[m].z; d[5 * P] = (u16)(st[m].z >> 16);
                d[6 * P] = (u16)st[m].w; d[7 * P] = (u16)(st[m].w >> 16);
            }
        }
    };

    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);       // input rows [r_lo, r_hi)
    const int rb0 = (r_lo / 7) * 7;
    issue(r_lo);
    for (int rb = rb0; rb < r_hi; rb += 7) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int r = rb + u;                                   // r % 7 == u
            if (r >= r_lo && r < r_hi) {
                to_lds();
                issue(r + 1 < r_hi ? r + 1 : -1);
                s16x4 a[NT][3];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 3; ++s) a[t][s] = *(const s16x4*)&lds[blk * P + 16 * t + 4 * s + 4 * q];
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const int yo = r + 3 - ky;
                    constexpr int dummy = 0; (void)dummy;
                    const int sl = (u + 3 - ky + 7) % 7;
                    if (yo >= ylo && yo < yhi) {
#pragma unroll
                        for (int s = 0; s < 3; ++s)
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[sl][t] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[t][s], bop[ky][s], acc[sl][t], 0, 0, 0);
                    }
                }
            }
            // output row r - 3 is complete once input row r has been applied (or r is past the image)
            const int yo = r - 3;
            const int sl = (u + 4) % 7;                              // (u - 3) mod 7
            if (yo >= ylo && yo < yhi && r >= r_lo && r < r_hi + 0) {
                u16* yr = y + ((size_t)(n * H + yo) * W + x0) * C + c0 + blk;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xo = 16 * t + 4 * i + q;
                        if (x0 + xo < W) yr[(size_t)xo * C] = d_f2bf(acc[sl][t][i]);
                    }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};
            }
        }
    }
    // rows whose last contributing input row lies below the image (yhi + 3 > H): flush
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {
        const int sl = yo % 7;
        u16* yr = y + ((size_t)(n * H + yo) * W + x0) * C + c0 + blk;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int xo = 16 * t + 4 * i + q;
                        if (x0 + xo < W) yr[(size_t)xo * C] = d_f2bf(acc[s7][t][i]);
                    }
            }
    }
}


// ---------------------------------------------------------------- 3b. the same kernel with the instruction count taken out
// buffer addressing (per-lane byte offsets are row-independent; rows / output columns move through the scalar offset; halo
// lanes outside the image carry an out-of-range offset and read zeros), v_cvt_pk_bf16_f32, no per-store bounds checks
// (W % (16*NT) == 0), immediate LDS offsets.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int NT, int C>
__global__ __launch_bounds__(256, 2) void dw7_mfma_v2(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int B, int H, int W, int RC, int nstrip, int nchunk)
{
    constexpr int NG = C / 16, SW = 16 * NT, IWX = SW + 8, NLD = (IWX * 2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) u16 lds_all[4][16 * P];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u16* lds = lds_all[wv];
    const int blk = lane >> 2, q = lane & 3;
    int gw = blockIdx.x * 4 + wv;
    const int g = gw % NG; gw /= NG;
    const int strip = gw % nstrip; gw /= nstrip;
    const int chunk = gw % nchunk;
    const int n = gw / nchunk;
    if (n >= B) return;
    const int c0 = g * 16, x0 = strip * SW, ylo = chunk * RC, yhi = min(H, ylo + RC);
    const unsigned img_bytes = (unsigned)H * W * C * 2, row_bytes = (unsigned)W * C * 2;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);

    s16x4 bop[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = 4 * (s - 1) + k - q + 3;
                const float v = (kx >= 0 && kx < 7) ? w[(size_t)(ky * 7 + kx) * C + c0 + blk] : 0.f;
                bop[ky][s][k] = (short)d_f2bf(v);
            }
    const float bv = bias[c0 + blk];
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};

    unsigned ld_off[NLD];                       // per-lane byte offset inside a row (0x80000000: outside the image -> zeros)
    u16* ld_dst[NLD];
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int e = m * 64 + lane, px = e >> 1, hh = e & 1, xi = x0 - 4 + px;
        ld_off[m] = (px < IWX && xi >= 0 && xi < W) ? (unsigned)((xi * C + c0 + 8 * hh) * 2) : 0x80000000u;
        ld_dst[m] = lds + (8 * hh) * P + (px < IWX ? px : 0);
    }
    const bool last_partial = (NLD * 32 > IWX) && (((NLD - 1) * 64 + lane) >> 1) >= IWX;     // lanes of the last load without an LDS column
    const unsigned st_off = (unsigned)(((x0 + q) * C + c0 + blk) * 2);
    const u16* rd = lds + blk * P + 4 * q;

    u32x4 st[NLD];
    auto issue = [&](int r) {                  // rows >= H fall outside the buffer and read zeros; halo lanes carry 0x80000000
#pragma unroll
        for (int m = 0; m < NLD; ++m) st[m] = __builtin_amdgcn_raw_buffer_load_b128(rx, ld_off[m] + (unsigned)r * row_bytes, 0, 0);
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            if (m == NLD - 1 && last_partial) continue;
            u16* d = ld_dst[m];
            d[0 * P] = (u16)st[m].x; d[1 * P] = (u16)(st[m].x >> 16);
            d[2 * P] = (u16)st[m].y; d[3 * P] = (u16)(st[m].y >> 16);
            d[4 * P] = (u16)st[m].z; d[5 * P] = (u16)(st[m].z >> 16);
            d[6 * P] = (u16)st[m].w; d[7 * P] = (u16)(st[m].w >> 16);
        }
    };
    auto flush = [&](f32x4 (&a)[NT], int yo) {   // branch-free: rows above the chunk get an out-of-range offset (dropped by the buffer check)
        const unsigned vo = st_off + (unsigned)yo * row_bytes + (yo >= ylo ? 0u : 0x80000000u);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][0], a[t][1]}, bf16x2_t));
            const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][2], a[t][3]}, bf16x2_t));
            __builtin_amdgcn_raw_buffer_store_b16((u16)(p01 & 0xffffu), ry, vo, (16 * t + 0) * C * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b16((u16)(p01 >> 16), ry, vo, (16 * t + 4) * C * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b16((u16)(p23 & 0xffffu), ry, vo, (16 * t + 8) * C * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b16((u16)(p23 >> 16), ry, vo, (16 * t + 12) * C * 2, 0);
            a[t] = f32x4{bv, bv, bv, bv};
        }
    };

    // input rows [r_lo, r_hi); slot of output row yo = (yo - r_lo + 3) % 7, so the unrolled sequence always starts at u = 0
    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
    int r = r_lo;
    issue(r);
    for (;;) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            to_lds();
            issue(r + 1 < r_hi ? r + 1 : H);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                s16x4 a[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) a[t] = *(const s16x4*)&rd[16 * t + 4 * s];
#pragma unroll
                for (int ky = 6; ky >= 0; --ky)          // ky = 6 first: the slot flushed after this row gets its last update earliest
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[t]), "v"(bop[ky][s]));
            }
            if constexpr (NT == 4) asm volatile("s_nop 7" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
            else asm volatile("s_nop 7" : "+v"(acc[u][0]), "+v"(acc[u][1]));
            flush(acc[u], r - 3);                        // output row r - 3 is complete (slot (r - 3 - r_lo + 3) % 7 = u)
            if (++r >= r_hi) goto done;
        }
    }
done:
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {              // rows whose last input row lies below the image
        const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) flush(acc[s7], yo);
    }
}


// ---------------------------------------------------------------- 3c. streaming version
// v2 is latency-bound (one row = 2.3 KB per wave in flight).  Here the input rows arrive by LDS-DMA into a ring of RS raw
// rows per wave (no VGPRs, RS rows in flight, counted vmcnt), are transposed LDS -> LDS ([px][16 ch] -> [ch][px]) one row
// ahead of their use, and the finished output row goes back through LDS to leave as two 16-B-per-lane stores instead of
// sixteen 2-B-per-lane ones.  ABL: ablation bits (1 no MFMA, 2 no output path, 4 no transposition, 8 no DMA after the prologue).
template <int NT, int C, int RS, int ABL>
__global__ __launch_bounds__(256, 2) void dw7_mfma_v3(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int B, int H, int W, int RC, int nstrip, int nchunk)
{
    static_assert(NT == 4, "row = 72 px = 2 full DMA pieces + 16 lanes");
    constexpr int NG = C / 16, SW = 16 * NT, IWX = SW + 8;
    constexpr int RAWB = IWX * 32, TB = 16 * P * 2, OB = SW * 32, WB = RS * RAWB + 2 * TB + OB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* wl = smem + wv * WB;
    char* raw = wl;                                   // [RS][72 px][16 ch]
    u16* T = (u16*)(wl + RS * RAWB);                  // [2][16 ch][P]
    char* O = wl + RS * RAWB + 2 * TB;                // [64 px][16 ch]
    const int blk = lane >> 2, q = lane & 3;
    int gw = blockIdx.x * 4 + wv;
    const int g = gw % NG; gw /= NG;
    const int strip = gw % nstrip; gw /= nstrip;
    const int chunk = gw % nchunk;
    const int n = gw / nchunk;
    if (n >= B) return;
    const int c0 = g * 16, x0 = strip * SW, ylo = chunk * RC, yhi = min(H, ylo + RC);
    const unsigned img_bytes = (unsigned)H * W * C * 2, row_bytes = (unsigned)W * C * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
    const char* ximg = (const char*)(x + (size_t)n * H * W * C);

    s16x4 bop[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = 4 * (s - 1) + k - q + 3;
                const float v = (kx >= 0 && kx < 7) ? w[(size_t)(ky * 7 + kx) * C + c0 + blk] : 0.f;
                bop[ky][s][k] = (short)d_f2bf(v);
            }
    const float bv = bias[c0 + blk];
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};

    // DMA piece m covers LDS px 32m .. 32m+31 (2 lanes per px); lanes outside the image read a clamped address and are not transposed
    unsigned voff[3];
    bool okm[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int px = 32 * m + (lane >> 1), xi = x0 - 4 + px;
        okm[m] = px < IWX && xi >= 0 && xi < W;
        const int xc = min(max(xi, 0), W - 1);
        voff[m] = (unsigned)((xc * C + c0 + 8 * (lane & 1)) * 2);
    }
    {   // zero both transposed buffers once: the columns outside the image stay zero for the whole strip
        f32x4 z = {0, 0, 0, 0};
        for (int i = lane; i < 2 * TB / 16; i += 64) *(f32x4*)((char*)T + i * 16) = z;
    }
    const unsigned raw_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)raw;
    auto dma = [&](int r, int slot) {          // 3 pieces of row r -> raw[slot]; the third under exec = lanes 0..15
        const char* rb_ = ximg + (size_t)r * row_bytes;
        unsigned keep; unsigned long long ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
                     "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(raw_lds + slot * RAWB), "s"(rb_) : "memory", "scc");
    };
    auto transpose = [&](int slot, int tb) {   // raw[slot] ([px][16 ch], lane-linear 16-B chunks) -> T[tb] ([ch][px])
        if (ABL & 4) return;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m == 2 && lane >= 16) continue;
            const u32x4 v = *(const u32x4*)(raw + slot * RAWB + m * 1024 + lane * 16);
            if (okm[m]) {
                u16* d = T + tb * (16 * P) + (8 * (lane & 1)) * P + 32 * m + (lane >> 1);
                d[0 * P] = (u16)v.x; d[1 * P] = (u16)(v.x >> 16);
                d[2 * P] = (u16)v.y; d[3 * P] = (u16)(v.y >> 16);
                d[4 * P] = (u16)v.z; d[5 * P] = (u16)(v.z >> 16);
                d[6 * P] = (u16)v.w; d[7 * P] = (u16)(v.w >> 16);
            }
        }
    };
    const unsigned st_off = (unsigned)(((x0 + (lane >> 1)) * C + c0 + 8 * (lane & 1)) * 2);
    u16* Ow = (u16*)(O + q * 32 + blk * 2);
    auto flush = [&](f32x4 (&a)[NT], int yo) {
        if (!(ABL & 2)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][0], a[t][1]}, bf16x2_t));
                const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][2], a[t][3]}, bf16x2_t));
                Ow[(16 * t + 0) * 16] = (u16)p01; Ow[(16 * t + 4) * 16] = (u16)(p01 >> 16);
                Ow[(16 * t + 8) * 16] = (u16)p23; Ow[(16 * t + 12) * 16] = (u16)(p23 >> 16);
            }
            const unsigned vo = st_off + (unsigned)yo * row_bytes + (yo >= ylo ? 0u : 0x80000000u);
            typedef u16 u16x8 __attribute__((ext_vector_type(8)));      // same element type as the ds_write_b16 side (strict aliasing)
            const u32x4 o0 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + lane * 16)), o1 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + 1024 + lane * 16));
            if (!(ABL & 16)) {
                __builtin_amdgcn_raw_buffer_store_b128(o0, ry, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, ry, vo, 32 * C * 2, 0);
            } else { asm volatile("" :: "v"(o0), "v"(o1)); }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = f32x4{bv, bv, bv, bv};
    };

    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
    const u16* rd = T + blk * P + 4 * q;
    // prologue: RS rows in flight, first row transposed, its slot refilled
#pragma unroll
    for (int i = 0; i < RS; ++i) dma(min(r_lo + i, r_hi - 1), i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    transpose(0, 0);
    dma(min(r_lo + RS, r_hi - 1), 0);
    int r = r_lo, slot = 1 % RS, tb = 0;        // slot: raw slot of row r + 1; tb: T buffer of row r
    for (;;) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            // row r + 1 has landed when at most the (RS - 1) later rows' pieces and stores are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(((ABL & 32) ? 5 : 3) * (RS - 1)) : "memory");   // loads only: stores may retire ahead of older loads
            transpose(slot, tb ^ 1);
            if (!(ABL & 1)) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    s16x4 a[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) a[t] = *(const s16x4*)&rd[tb * (16 * P) + 16 * t + 4 * s];
#pragma unroll
                    for (int ky = 6; ky >= 0; --ky)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[t]), "v"(bop[ky][s]));
                }
            }
            // the compiler does not know the asm statements are MFMAs: pin the readers of this slot behind them (and behind the
            // XDL-write -> VALU-read wait states) - unpinned, the first v_cvt of tile 3 was hoisted right behind its last MFMA
            asm volatile("s_nop 7" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
            flush(acc[u], r - 3);
            if (!(ABL & 8)) dma(min(r + 1 + RS, r_hi - 1), slot);
            else { asm volatile("s_nop 0"); }
            slot = slot + 1 == RS ? 0 : slot + 1;
            tb ^= 1;
            if (++r >= r_hi) goto done;
        }
    }
done:
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {
        const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) flush(acc[s7], yo);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may be in flight when the LDS is released
}


// ---------------------------------------------------------------- 3d. workgroup-cooperative I/O
// v3 moves 32-B segments (16 channels of one pixel) per lane pair: reads alone 5 TB/s, writes alone 4.5 TB/s, both together
// 3.3 TB/s.  Here the 4 waves of a workgroup own 4 adjacent channel groups = 64 channels = one 128-B line per pixel: the
// row segment is DMAed as whole lines into a raw ring SHARED by the workgroup (each wave issues 2 of the 8 interior 1-KiB
// pieces and a quarter of the halo piece), each wave transposes its own 32-B column of every pixel, the finished output row
// is assembled in a shared [px][64 ch] buffer and leaves as whole lines.  One s_barrier per row.
template <int C, int RS, int ABL>
__global__ __launch_bounds__(256, 2) void dw7_mfma_v4(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int B, int H, int W, int RC, int nstrip, int nchunk)
{
    constexpr int NT = 4, NW = 4, CW = 64, SW = 64, IWX = 72, NCB = C / CW;
    constexpr int OPX = 144;                       // output staging: 128 B of channels + 16 B pad per pixel (4 px of a ds_write_b16 on 4 bank groups)
    constexpr int RAWB = IWX * CW * 2, OB = SW * OPX, TBY = 16 * P * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* raw = smem;                                       // [RS][64 px interior | 8 px halo][64 ch]
    char* O = smem + RS * RAWB;                             // [2][64 px][64 ch]
    u16* T = (u16*)(smem + RS * RAWB + 2 * OB + wv * TBY);  // per wave [16 ch][P]
    const int blk = lane >> 2, q = lane & 3;
    int L = blockIdx.x;
    if (!(ABL & 64)) { const int G = gridDim.x, per = G / 8; if (G % 8 == 0) L = (L % 8) * per + L / 8; }     // consecutive logical tiles share an XCD (L2): halo rows / columns
    const int cb = L % NCB; L /= NCB;
    const int strip = L % nstrip; L /= nstrip;
    const int chunk = L % nchunk;
    const int n = L / nchunk;
    const int c0 = cb * CW + wv * 16, x0 = strip * SW, ylo = chunk * RC, yhi = min(H, ylo + RC);
    const unsigned img_bytes = (unsigned)H * W * C * 2, row_bytes = (unsigned)W * C * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
    const char* ximg = (const char*)(x + (size_t)n * H * W * C);

    s16x4 bop[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = 4 * (s - 1) + k - q + 3;
                const float v = (kx >= 0 && kx < 7) ? w[(size_t)(ky * 7 + kx) * C + c0 + blk] : 0.f;
                bop[ky][s][k] = (short)d_f2bf(v);
            }
    const float bv = bias[c0 + blk];
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};

    // ---- loads: interior pieces 2 wv and 2 wv + 1 (8 px x 128 B each), halo piece lanes 16 wv .. 16 wv + 15
    // inside every 1-KiB piece the 16-B chunks are stored [consumer wave][px][half] (a wave's later ds_read_b128 of its 32-B column
    // is then bank-conflict free): DMA lane l = (consumer l >> 4, px (l >> 1) & 7, half l & 1); the global side is still 8 whole lines
    const unsigned vint = (unsigned)(((x0 + 16 * wv + ((lane >> 1) & 7)) * C + cb * CW) * 2 + (lane >> 4) * 32 + (lane & 1) * 16);
    const unsigned vst = (unsigned)(((x0 + 16 * wv + (lane >> 3)) * C + cb * CW) * 2 + (lane & 7) * 16);
    const int hp = lane >> 3, hx = hp < 4 ? x0 - 4 + hp : x0 + 60 + hp;
    const unsigned vhalo = (unsigned)((min(max(hx, 0), W - 1) * C + cb * CW) * 2 + (lane & 7) * 16);      // halo piece: plain [px][128 B]
    const unsigned long long hmask = 0xffffull << (16 * wv);
    const unsigned raw_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)raw;
    auto dma = [&](int r, int slot) {
        const char* rb0 = ximg + (size_t)r * row_bytes;
        const char* rb1 = rb0 + 8 * C * 2;
        const unsigned d0 = raw_lds + slot * RAWB + 2048 * wv, dh = raw_lds + slot * RAWB + 8192;
        unsigned keep; unsigned long long ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                     "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %8\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(vint), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb0), "s"(rb1), "s"(hmask) : "memory", "scc");
    };
    // ---- transposition: this wave's 32-B column of the 72 pixels (T column 0..3 left halo, 4..67 interior, 68..71 right halo)
    unsigned roff[3];
    bool okm[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int col = 32 * m + (lane >> 1), xi = x0 - 4 + col;
        okm[m] = col < IWX && xi >= 0 && xi < W;
        const int cc = min(col, IWX - 1);
        const int ip = cc - 4;                                    // interior pixel index
        roff[m] = (unsigned)(cc < 4 ? 8192 + cc * 128 + wv * 32 + (lane & 1) * 16 : cc >= 68 ? 8192 + (cc - 64) * 128 + wv * 32 + (lane & 1) * 16
                                    : (ip >> 3) * 1024 + wv * 256 + (ip & 7) * 32 + (lane & 1) * 16);
    }
    typedef u16 u16x8 __attribute__((ext_vector_type(8)));
    auto transpose = [&](int slot) {
        if (ABL & 4) return;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m == 2 && lane >= 16) continue;
            const u32x4 v = __builtin_bit_cast(u32x4, *(const u16x8*)(raw + slot * RAWB + roff[m]));
            if (okm[m]) {
                u16* d = T + (8 * (lane & 1)) * P + 32 * m + (lane >> 1);
                d[0 * P] = (u16)v.x; d[1 * P] = (u16)(v.x >> 16);
                d[2 * P] = (u16)v.y; d[3 * P] = (u16)(v.y >> 16);
                d[4 * P] = (u16)v.z; d[5 * P] = (u16)(v.z >> 16);
                d[6 * P] = (u16)v.w; d[7 * P] = (u16)(v.w >> 16);
            }
        }
    };
    {
        f32x4 z = {0, 0, 0, 0};
        for (int i = lane; i < TBY / 16; i += 64) *(f32x4*)((char*)T + i * 16) = z;
    }
    // ---- output: own 16 channels into the shared row buffer, then pieces 2 wv, 2 wv + 1 as whole lines
    auto stage = [&](f32x4 (&a)[NT], int ob) {
        if (!(ABL & 2)) {
            u16* Ow = (u16*)(O + ob * OB + q * OPX + wv * 32 + blk * 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][0], a[t][1]}, bf16x2_t));
                const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][2], a[t][3]}, bf16x2_t));
                Ow[(16 * t + 0) * (OPX / 2)] = (u16)p01; Ow[(16 * t + 4) * (OPX / 2)] = (u16)(p01 >> 16);
                Ow[(16 * t + 8) * (OPX / 2)] = (u16)p23; Ow[(16 * t + 12) * (OPX / 2)] = (u16)(p23 >> 16);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = f32x4{bv, bv, bv, bv};
    };
    auto store = [&](int yo, int ob) {
        if (ABL & 2) return;
        const unsigned vo = vst + (unsigned)yo * row_bytes + (yo >= ylo ? 0u : 0x80000000u);
        const u32x4 o0 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + ob * OB + (16 * wv + (lane >> 3)) * OPX + (lane & 7) * 16));
        const u32x4 o1 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + ob * OB + (16 * wv + 8 + (lane >> 3)) * OPX + (lane & 7) * 16));
        if (!(ABL & 16)) {
            __builtin_amdgcn_raw_buffer_store_b128(o0, ry, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o1, ry, vo, 8 * C * 2, 0);
        } else { asm volatile("" :: "v"(o0), "v"(o1)); }
    };

    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
    const u16* rd = T + blk * P + 4 * q;
#pragma unroll
    for (int i = 0; i < RS; ++i) dma(min(r_lo + i, r_hi - 1), (r_lo + i) % RS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    transpose(r_lo % RS);
    int r = r_lo, slot = r_lo % RS, ob = 0;        // slot: raw slot of row r
    for (;;) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (!(ABL & 1)) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    s16x4 a[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) a[t] = *(const s16x4*)&rd[16 * t + 4 * s];
#pragma unroll
                    for (int ky = 6; ky >= 0; --ky)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[t]), "v"(bop[ky][s]));
                }
            }
            asm volatile("s_nop 7" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
            stage(acc[u], ob);
            // own pieces of row r + 1 landed (loads only are counted: stores may retire ahead of older loads); LDS writes retired
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * (RS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nslot = slot + 1 == RS ? 0 : slot + 1;
            transpose(nslot);                          // row r + 1 -> T (the reads of row r are already issued: LDS is in order)
            store(r - 3, ob);
            if (!(ABL & 8)) dma(min(r + RS, r_hi - 1), slot);       // row r's slot: every wave has transposed it before this barrier
            slot = nslot;
            ob ^= 1;
            if (++r >= r_hi) goto done;
        }
    }
done:
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {              // rows whose last input row lies below the image
        const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) stage(acc[s7], ob);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        store(yo, ob);
        ob ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static void cpu_dw7(const std::vector<u16>& x, const std::vector<float>& w, const std::vector<float>& bias, std::vector<float>& out,
                    int B, int H, int W, int C)
{
    out.assign((size_t)B * H * W * C, 0.f);
    for (int n = 0; n < B; ++n)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx)
                for (int c = 0; c < C; ++c) {
                    double a = bias[c];
                    for (int ky = 0; ky < 7; ++ky)
                        for (int kx = 0; kx < 7; ++kx) {
                            const int iy = yy + ky - 3, ix = xx + kx - 3;
                            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                            a += (double)bf2f(f2bf(w[(size_t)(ky * 7 + kx) * C + c])) * bf2f(x[((size_t)(n * H + iy) * W + ix) * C + c]);
                        }
                    out[((size_t)(n * H + yy) * W + xx) * C + c] = (float)a;
                }
}

template <int NT, int C, int VER = 1, int RS = 4, int ABL = 0>
static void run_dw7(int B, int H, int W, int RC, bool check, int reps)
{
    const size_t N = (size_t)B * H * W * C;
    std::vector<u16> hx(N);
    std::vector<float> hw(49 * C), hb(C);
    srand(11);
    for (auto& v : hx) v = f2bf((rand() % 2001 - 1000) / 500.f);
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 5000.f;
    for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
    u16 *dx, *dy; float *dw, *db;
    CK(hipMalloc(&dx, N * 2)); CK(hipMalloc(&dy, N * 2)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMemcpy(dx, hx.data(), N * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, N * 2));
    const int SW = 16 * NT, nstrip = (W + SW - 1) / SW, nchunk = (H + RC - 1) / RC;
    const long long waves = (long long)B * (C / 16) * nstrip * nchunk;
    const int grid = (int)((waves + 3) / 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if constexpr (VER == 4) {
            constexpr int WBY = RS * 72 * 64 * 2 + 2 * 64 * 144 + 4 * 16 * P * 2;
            static bool once = false;
            if (!once) { CK(hipFuncSetAttribute((const void*)dw7_mfma_v4<C, RS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, WBY)); once = true; }
            dw7_mfma_v4<C, RS, ABL><<<(int)(waves / 4), 256, WBY>>>(dx, dy, dw, db, B, H, W, RC, nstrip, nchunk);
        } else if constexpr (VER == 3) {
            constexpr int WBY = RS * (16 * NT + 8) * 32 + 2 * 16 * P * 2 + 16 * NT * 32;
            static bool once = false;
            if (!once) { CK(hipFuncSetAttribute((const void*)dw7_mfma_v3<NT, C, RS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WBY)); once = true; }
            dw7_mfma_v3<NT, C, RS, ABL><<<grid, 256, 4 * WBY>>>(dx, dy, dw, db, B, H, W, RC, nstrip, nchunk);
        } else if (VER == 2) dw7_mfma_v2<NT, C><<<grid, 256>>>(dx, dy, dw, db, B, H, W, RC, nstrip, nchunk);
        else dw7_mfma_kernel<NT, C><<<grid, 256>>>(dx, dy, dw, db, B, H, W, RC, nstrip, nchunk);
    };
    launch();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<u16> hy(N);
        CK(hipMemcpy(hy.data(), dy, N * 2, hipMemcpyDeviceToHost));
        std::vector<float> ref;
        cpu_dw7(hx, hw, hb, ref, B, H, W, C);
        double maxerr = 0, maxref = 0; size_t bad = 0;
        for (size_t i = 0; i < N; ++i) {
            const double e = fabs(bf2f(hy[i]) - ref[i]);
            maxerr = fmax(maxerr, e); maxref = fmax(maxref, fabs(ref[i]));
            bad += !(e <= 0.01 * fabs(ref[i]) + 0.02);
        }
        if (bad) {
            std::vector<int> hx_(W, 0), hy_(H, 0), hc_(C, 0);
            for (size_t i = 0; i < N; ++i) {
                const double e = fabs(bf2f(hy[i]) - ref[i]);
                if (!(e <= 0.01 * fabs(ref[i]) + 0.02)) { hc_[i % C]++; hx_[(i / C) % W]++; hy_[(i / C / W) % H]++; }
            }
            printf("  bad by x:"); for (int i = 0; i < W; ++i) printf(" %d", hx_[i]); printf("\n");
            printf("  bad by y:"); for (int i = 0; i < H; ++i) printf(" %d", hy_[i]); printf("\n");
            printf("  bad by c:"); for (int i = 0; i < C; ++i) printf(" %d", hc_[i]); printf("\n");
        }
        printf("dw7 v%d check NT=%d C=%d B=%d %dx%d RC=%d: max err %.4g (max |ref| %.3g), %zu / %zu outside tolerance\n", VER, NT, C, B, H, W, RC, maxerr, maxref, bad, N);
    }
    if (reps) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("dw7 v%d RS=%d ABL=%d time  NT=%d C=%d B=%d %dx%d RC=%d: %.1f us  -> %.2f TB/s algorithmic (read + write once), %lld waves\n", VER, RS, ABL, NT, C, B, H, W, RC, us,
               2.0 * N * 2 / (us * 1e-6) / 1e12, waves);
    }
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dw); hipFree(db);
}

int main(int argc, char** argv)
{
    run_dw7<4, 64, 4>(2, 40, 64, 16, true, 0);
    run_dw7<4, 64, 4>(1, 23, 128, 32, true, 0);
    run_dw7<4, 128, 4, 3>(1, 70, 192, 64, true, 0);
    const int reps = 20;
    run_dw7<4, 384, 4>(32, 64, 64, 32, false, reps);
    run_dw7<4, 192, 4>(32, 128, 128, 32, false, reps);
    run_dw7<4, 384, 4, 4, 64>(32, 64, 64, 32, false, reps);
    run_dw7<4, 192, 4, 4, 64>(32, 128, 128, 32, false, reps);
    run_dw7<4, 384, 4>(32, 64, 64, 22, false, reps);
    run_dw7<4, 384, 4>(32, 64, 64, 16, false, reps);
    run_dw7<4, 192, 4>(32, 128, 128, 43, false, reps);
    run_dw7<4, 192, 4>(32, 128, 128, 64, false, reps);
    run_dw7<4, 192, 4, 4, 1>(32, 128, 128, 32, false, reps);
    run_dw7<4, 192, 4, 4, 2>(32, 128, 128, 32, false, reps);
    run_dw7<4, 192, 4, 4, 5>(32, 128, 128, 32, false, reps);
    return 0;
}
