// First and last ops of the FastViTHD tower.
//
//   stem[0]   MobileOneBlock dense 3x3 s2 p1, 3 -> 96, + bias, GELU          mci.py:563-574, 194-198
//             reads the caller's NCHW image (f32 / f16 / bf16) and writes NHWC bf16 - this is the only
//             layout change on the whole path; the tower's output [B, HW, 3072] *is* NHWC.
//   conv_exp  SEBlock (avg-pool -> 1x1 reduce -> ReLU -> 1x1 expand -> sigmoid -> scale) and the final
//             GELU, applied to y = dw3x3(x)+b computed by the generic depthwise kernel  mci.py:72-81, 198
//             written straight in the tower's output dtype/layout (feature_select is free:
//             mobileclip_encoder.py:60-68).
#include "fvhd_common.h"

template <typename T> FVHD_DEV float ld_as_f32(const T* p, size_t i);
template <> FVHD_DEV float ld_as_f32<float>(const float* p, size_t i) { return p[i]; }
template <> FVHD_DEV float ld_as_f32<_Float16>(const _Float16* p, size_t i) { return (float)p[i]; }
template <> FVHD_DEV float ld_as_f32<bf16>(const bf16* p, size_t i) { return (float)p[i]; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// stem[0] as an MFMA GEMM:  out[px][96] = gelu(W[96 x 27] . patch[27 x px] + b),  K padded to 32.
//   * a wave owns tiles of 32 consecutive output pixels of one output row; the im2col patch is the B operand of
//     v_mfma_f32_32x32x16_bf16 (lane = (pixel, k-half)), gathered straight from the caller's NCHW image (any of
//     f32 / f16 / bf16; rounded to bf16, the tower's compute dtype - mobileclip_encoder.py:85 casts images likewise);
//     the k axis is ordered so that a lane's 16 slots are whole (ci, ky) rows of 3 taps: half 0 = rows 0-4, half 1 = rows 5-8;
//   * the weights are the A operand (3 blocks of 32 output channels x 2 k-steps): each workgroup builds the bf16 fragment
//     image once in LDS from the fp32 [27][96] taps and every lane keeps its 6 fragments in registers;
//   * bias + exact-erf GELU in fp32 on the accumulators, then the [32 px][96 ch] bf16 tile goes through LDS so that the
//     NHWC store is 16 B per lane, fully coalesced (a tile is 6 KiB contiguous in HBM).
// (The VALU version of this kernel spent 648 FMA + 162 LDS weight reads per thread and ran at 1.1 TB/s.)
#define STEM_TPW 4      // tiles per wave
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const T* __restrict__ img, bf16* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int B, int R, long ntiles)
{
    constexpr int CO = 96;
    __shared__ __attribute__((aligned(16))) char smem[6 * 64 * 16 + 4 * 32 * CO * 2];
    bf16x8* wimg = (bf16x8*)smem;                       // [cb][s][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, half = lane >> 5;
    // tap index of k-slot q (0..15) of k-half h: rows (ci*3+ky) 0-4 -> half 0, 5-8 -> half 1; -1 = zero padding
    auto tap_of = [](int h, int q) { const int row = (h ? 5 : 0) + q / 3; return (q < (h ? 12 : 15)) ? row * 3 + q % 3 : -1; };
    for (int i = tid; i < 6 * 64; i += 256) {
        const int l = i & 63, s = (i >> 6) & 1, cb = i >> 7;
        const int ch = cb * 32 + (l & 31), h = l >> 5;
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tap_of(h, s * 8 + j);
            f[j] = (bf16)(t >= 0 ? w[t * CO + ch] : 0.0f);
        }
        wimg[i] = f;
    }
    __syncthreads();
    bf16x8 wa[3][2];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s) wa[cb][s] = wimg[(cb * 2 + s) * 64 + lane];
    f32x4 bv[3][4];                                     // bias of this lane's 48 channels: ch = 32cb + 8q + 4half + j
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[cb][q] = *(const f32x4*)(bias + cb * 32 + q * 8 + half * 4);

    const int OH = R / 2, OW = R / 2, TX = OW / 32;
    char* otile = smem + 6 * 64 * 16 + wave * (32 * CO * 2);
    for (int it = 0; it < STEM_TPW; ++it) {
        const long tile = ((long)blockIdx.x * 4 + wave) * STEM_TPW + it;
        if (tile >= ntiles) break;
        const int tx = (int)(tile % TX), oy = (int)((tile / TX) % OH), b = (int)(tile / ((long)TX * OH));
        const int ox = tx * 32 + px;
        // ---- im2col gather: this lane's 5 (half 0) or 4 (half 1) rows of 3 taps
        bf16x8 xb[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int t = tap_of(half, q);               // half is a runtime value: both variants are evaluated below
            float v = 0.0f;
            const int row = (half ? 5 : 0) + q / 3, kx = q % 3;
            const int ci = row / 3, ky = row % 3;
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            if (t >= 0 && iy >= 0 && iy < R && ix >= 0)
                v = ld_as_f32<T>(img, (((size_t)b * 3 + ci) * R + iy) * R + ix);
            xb[q >> 3][q & 7] = (bf16)v;
        }
        f32x16 acc[3];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cb][0], xb[0], z, 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cb][1], xb[1], acc[cb], 0, 0, 0);
        }
        // ---- bias + GELU, [px][96] bf16 tile into LDS (lane: pixel px, channels 32cb + 8q + 4half .. +3)
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = gelu_erf(acc[cb][4 * q + j] + bv[cb][q][j]);
                *(bf16x4*)(otile + px * (CO * 2) + (cb * 32 + q * 8 + half * 4) * 2) = f32_to_bf4(g);
            }
        // a wave's tile is private to it: the LDS round trip needs no workgroup barrier, only the wave's own ordering
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
        bf16* dst = out + (((size_t)b * OH + oy) * OW + tx * 32) * CO;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const u32x4 v = *(const u32x4*)(otile + (i * 64 + lane) * 16);
            *(u32x4*)((char*)dst + (i * 64 + lane) * 16) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // reads done before the next tile overwrites the buffer
    }
}

// img [B,3,R,R] (dtype) -> out [B,R/2,R/2,96] bf16;  R % 64 == 0 (so R/2 is a multiple of the 32-pixel tile)
extern "C" int fvhd_launch_stem_conv(hipStream_t st, const void* img, int dtype, void* out, const float* w,
                                     const float* bias, int B, int R)
{
    if (R % 64) return (int)hipErrorInvalidValue;
    const long ntiles = (long)B * (R / 2) * (R / 64);
    dim3 grid((unsigned)((ntiles + 4 * STEM_TPW - 1) / (4 * STEM_TPW))), block(256);
    if (dtype == FVHD_F32) hipLaunchKernelGGL(stem_conv_kernel<float>, grid, block, 0, st, (const float*)img, (bf16*)out, w, bias, B, R, ntiles);
    else if (dtype == FVHD_F16) hipLaunchKernelGGL(stem_conv_kernel<_Float16>, grid, block, 0, st, (const _Float16*)img, (bf16*)out, w, bias, B, R, ntiles);
    else if (dtype == FVHD_BF16) hipLaunchKernelGGL(stem_conv_kernel<bf16>, grid, block, 0, st, (const bf16*)img, (bf16*)out, w, bias, B, R, ntiles);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ---- stem[0] + stem[1] fused: dense 3x3 s2 (3 -> 96) + GELU, then depthwise 3x3 s2 + GELU ---------
// mci.py:563-586.  Unfused, stem[0] writes and stem[1] re-reads a [B, R/2, R/2, 96] bf16 tensor (1.6 GB at B = 32, R = 1024:
// the two launches were write- / read-bound on it, 0.75 + 0.48 ms); here it only ever exists as LDS tiles.
//   * a workgroup owns an 8 x 8 tile of stem[1] outputs = a 17 x 17 region of stem[0] outputs (recomputed halo: x1.13),
//     computed exactly as stem_conv_kernel does (same MFMA shapes, k order, bias, GELU, bf16 rounding - the fused result
//     is bit-identical to the two-kernel path) in 10 MFMA tiles of 32 positions, 2-3 per wave, and stored position-major
//     [289][96] bf16 in LDS; positions outside the stem[0] map are stored as zeros (stem[1]'s zero padding);
//   * then every thread computes 3 (pixel, 8-channel) units of the depthwise conv from LDS (taps fp32 in LDS, same
//     accumulation order as dwconv_tiled_kernel) and stores 16 B, a pixel row of the tile being 1.5 KiB contiguous.
#define STEMF_T 8                         // stem[1] outputs per tile side
#define STEMF_R (2 * STEMF_T + 1)         // stem[0] region side (17)
#define STEMF_NP (STEMF_R * STEMF_R)      // 289 positions
#define STEMF_IR (2 * STEMF_R + 1)        // image rows / columns under the region (35)
#define STEMF_IS 40                       // image tile row stride in LDS (elements): 4-B aligned 3-pixel runs
#define STEMF_IB (3 * STEMF_IR * STEMF_IS * 2)   // image tile bytes (8400)
#define STEMF_PS 208                      // bytes per region position in LDS: 96 bf16 + 16 B pad (a 192-B stride put the 16
                                          // lanes of a ds_write_b64 group on 4 banks: 8-way conflicts; 208 leaves 2-way)
// FULL (round 4): stem[2] - the 1x1 conv 96 -> 96 + bias + GELU (mci.py:587-598) - in the same launch.  Phase 2 leaves its 8 x 8 x 96
// tile in LDS ([64 px][96] bf16, aliasing the weight image of the kernel's start and the image tile of phase 0, both dead by then)
// instead of writing it to HBM; phase 3 runs it through 72 MFMAs (v_mfma_f32_16x16x32_bf16, W2 fragments as the A operand so that a
// lane ends with 4 consecutive output channels of one pixel), bias + GELU, and stores the tile as whole 192-B pixels.  Every wave owns
// 16 pixels of the tile from the B-operand reads to the store, so the phase needs no barrier of its own; one more barrier at the end
// of the tile keeps the next tile's phase 0 off the buffer.  The unfused path wrote and re-read a [B, R/4, R/4, 96] tensor (806 MB of
// HBM traffic per step at B = 32) and cost a GEMM launch.
template <typename T, bool FULL>
__global__ __launch_bounds__(256, 2) void stem_fused_kernel(const T* __restrict__ img, bf16* __restrict__ out,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const bf16* __restrict__ w2, const float* __restrict__ b2,
                                                            int B, int R, int ntiles)
{
    // Round 3: a workgroup walks STEMF_TPW consecutive tiles (one every gridDim.x) instead of one: the weight fragment image and the
    // stem[1] taps are built once, and the NEXT tile's 3 x 35 x 35 image pixels are loaded into registers while the current tile
    // runs its two phases (one tile per workgroup spent most of its ~15 us waiting for that gather: 32768 workgroups x 3 barriers).
    constexpr int CO = 96;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* lw1 = (float*)smem_f;                                 // [9][96] fp32 taps of stem[1]
    bf16x8* wimg = (bf16x8*)(smem_f + 9 * CO * 4);               // [cb][s][lane]: 6 KiB (read once, before the tile loop)
    bf16* itile = (bf16*)(smem_f + 9 * CO * 4 + 6 * 64 * 16);    // image tile [3][35][STEMF_IS] bf16 (zero outside the image)
    char* y1 = smem_f + 9 * CO * 4;                              // FULL: stem[1] output tile [64 px][STEMF_PS]; aliases wimg + itile
    char* reg = smem_f + 6 * 64 * 16 + 9 * CO * 4 + STEMF_IB;    // [289][96] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, half = lane >> 5;
    auto tap_of = [](int h, int q) { const int row = (h ? 5 : 0) + q / 3; return (q < (h ? 12 : 15)) ? row * 3 + q % 3 : -1; };
    for (int i = tid; i < 6 * 64; i += 256) {
        const int l = i & 63, s = (i >> 6) & 1, cb = i >> 7;
        const int ch = cb * 32 + (l & 31), h = l >> 5;
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tap_of(h, s * 8 + j);
            f[j] = (bf16)(t >= 0 ? w0[t * CO + ch] : 0.0f);
        }
        wimg[i] = f;
    }
    for (int i = tid; i < 9 * CO / 4; i += 256) *(f32x4*)&lw1[i * 4] = *(const f32x4*)&w1[i * 4];

    const int H1 = R / 2, H2 = R / 4, TXY = H2 / STEMF_T;        // stem[0] map side, stem[1] map side, tiles per side
    // 3 x 35 x 35 = 3675 image elements under a region.  Round 6: a thread's elements are (row 8 i + tid / 32, column tid % 32) for
    // i = 0 .. 13 (the 105 x 32 block) and two more for the three remaining columns - rows, channel index and the LDS address are
    // compile-time offsets from one per-thread base (the first version dealt e = i * 256 + tid and paid two divisions by 35 per element and
    // tile, twice: ~420 of a thread's ~3400 VALU instructions per tile)
    constexpr int NROW = 3 * STEMF_IR, NIT = 16;
    // ---- phase 0 (per tile, one tile ahead): the 3 x 35 x 35 image pixels under the region, rounded to bf16 (the tower's compute
    //      dtype), zero outside the image; they go into LDS so that the im2col gather below is two aligned ds_read_b32 per (channel,
    //      tap row) instead of three bounds-checked 2-byte global loads with 64-bit addressing
    // raw element values + a validity bit mask: NO dependent operation until the values are written to LDS one tile later, so the
    // loads really stay in flight across the two phases of the current tile (a select right behind the load would wait for it)
    auto elem = [&](int i, int ln, int& row, int& col) {         // element i of thread ln: (row of the [105][35] region image, column)
        if (i < 14) { row = 8 * i + ((ln >> 5) & 7); col = ln & 31; }      // (& 7: ln is opaque, the compiler needs the range to fold row / 35)
        else { const int t3 = ln / 3; row = (i == 14 ? 0 : 85) + t3; col = 32 + (ln - 3 * t3); if (i == 15 && ln >= 60) row = NROW; }
    };
    // Buffer loads through a per-image descriptor (round 6): an element outside the image gets an out-of-range offset and comes back as
    // zero bits - no validity mask to carry, no 64-bit address arithmetic, and none of the exec-masked branches hipcc put around every one
    // of the 16 conditional loads of the flat form.  The raw bits ride in registers across both phases of the current tile.
    auto load_image = [&](int tile, unsigned (&v)[NIT]) {
        int ln = tid;
        asm volatile("" : "+v"(ln));                              // opaque: keeps the per-thread address chains out of the tile loop's live set
        const int tx_ = tile % TXY, ty_ = (tile / TXY) % TXY, b_ = tile / (TXY * TXY);
        const int iy0 = 2 * (2 * ty_ * STEMF_T - 1) - 1, ix0 = 2 * (2 * tx_ * STEMF_T - 1) - 1;
        const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (size_t)b_ * 3 * R * R), 0, (unsigned)(3 * R * R * sizeof(T)), 0x00020000);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int row, col;
            elem(i, ln, row, col);
            const int ci = (row >= STEMF_IR) + (row >= 2 * STEMF_IR), r = row - ci * STEMF_IR;      // compile-time for all but two values of i
            const int iy = iy0 + r, ix = ix0 + col;
            const bool ok = (i < 13 || row < NROW) & ((unsigned)iy < (unsigned)R) & ((unsigned)ix < (unsigned)R);     // '&': selects, not branches
            const unsigned off = ok ? (unsigned)(((ci * R + iy) * R + ix) * (int)sizeof(T)) : 0xffffffffu;
            if constexpr (sizeof(T) == 4) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rimg, off, 0, 0);
            else v[i] = __builtin_amdgcn_raw_buffer_load_b16(rimg, off, 0, 0);
        }
    };
    auto bits_as_f32 = [](unsigned bits) -> float {               // raw element bits -> its value (zero bits = 0.0 in every dtype)
        if constexpr (sizeof(T) == 4) return __uint_as_float(bits);
        else { const unsigned short h = (unsigned short)bits; return ld_as_f32<T>((const T*)&h, 0); }
    };
    unsigned vimg[NIT];
    int tile = blockIdx.x;
    if (tile < ntiles) load_image(tile, vimg);
    __syncthreads();                                              // weight image + taps visible
    bf16x8 wa[3][2];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s) wa[cb][s] = wimg[(cb * 2 + s) * 64 + lane];
    f32x4 bv[3][4];
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[cb][q] = *(const f32x4*)(b0 + cb * 32 + q * 8 + half * 4);
    static_assert(64 * STEMF_PS <= 6 * 64 * 16 + STEMF_IB, "the stem[1] tile fits into the weight image + image tile it aliases");
    __syncthreads();                                              // FULL: every wave has its fragments before phase 2 of the first tile overwrites wimg

#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % TXY, ty = (tile / TXY) % TXY, b = tile / (TXY * TXY);
    const int cy0 = 2 * ty * STEMF_T - 1, cx0 = 2 * tx * STEMF_T - 1;          // stem[0] coordinates of region position (0, 0)
    {
        int ln = tid;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int row, col;
            elem(i, ln, row, col);
            if (i < 13 || row < NROW) itile[row * STEMF_IS + col] = (bf16)bits_as_f32(vimg[i]);
        }
    }
    __syncthreads();                  // image tile complete; every thread is past phase 2 of the previous tile (`reg` may be rewritten)
    if (tile + (int)gridDim.x < ntiles) load_image(tile + gridDim.x, vimg);    // in flight during both phases
    // ---- phase 1: the stem[0] region, 32 positions per MFMA tile
#pragma unroll 1
    for (int t = wave; t * 32 < STEMF_NP; t += 4) {
        const int p = t * 32 + px;
        const int ry = p / STEMF_R, rx = p - ry * STEMF_R;
        const int gy = cy0 + ry, gx = cx0 + rx;                   // this lane's stem[0] output pixel
        const bool inside = p < STEMF_NP && gy >= 0 && gy < H1 && gx >= 0 && gx < H1;
        // k-slot q = 3 i + kx of this lane's k-half <-> tap row (ci, ky) = 5 half + i, pixel 2 rx + kx of image-tile row
        // ci * 35 + 2 ry + ky: per tap row two words wa = (px0, px1), wb = (px2, -).  (half 1 has 4 rows: its fifth read
        // and, for lanes past position 288, rows past the tile land in the LDS that follows - never used.)
        unsigned ra[5], rb2[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int rr = half * 5 + i;                          // 0..8 (9 = unused)
            const int ci = rr / 3, ky = rr - ci * 3;
            const unsigned* src = (const unsigned*)(itile + (ci * STEMF_IR + 2 * ry + ky) * STEMF_IS + 2 * rx);
            ra[i] = src[0];
            rb2[i] = src[1];
        }
        u32x4 x0, x1;
        x0[0] = ra[0];
        x0[1] = (rb2[0] & 0xffffu) | (ra[1] << 16);
        x0[2] = (ra[1] >> 16) | (rb2[1] << 16);
        x0[3] = ra[2];
        x1[0] = (rb2[2] & 0xffffu) | (ra[3] << 16);
        x1[1] = (ra[3] >> 16) | (rb2[3] << 16);
        x1[2] = half ? 0u : ra[4];
        x1[3] = half ? 0u : (rb2[4] & 0xffffu);
        bf16x8 xb[2] = {__builtin_bit_cast(bf16x8, x0), __builtin_bit_cast(bf16x8, x1)};
        f32x16 acc[3];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cb][0], xb[0], z, 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cb][1], xb[1], acc[cb], 0, 0, 0);
        }
        if (p < STEMF_NP) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = inside ? gelu_erf(acc[cb][4 * q + j] + bv[cb][q][j]) : 0.0f;
                    *(bf16x4*)(reg + p * STEMF_PS + (cb * 32 + q * 8 + half * 4) * 2) = f32_to_bf4(g);
                }
        }
    }
    __syncthreads();

    // ---- phase 2: stem[1] on the LDS region; unit u = (pixel, 8-channel group), 768 units, 3 per thread
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = i * 256 + tid;
        const int cg = u % 12, pxl = u / 12;
        const int oy = pxl / STEMF_T, ox = pxl - oy * STEMF_T;
        float acc[8];
        {
            const f32x4 c0 = *(const f32x4*)(b1 + cg * 8), c1 = *(const f32x4*)(b1 + cg * 8 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[c] = c0[c]; acc[4 + c] = c1[c]; }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x8 v = bf8_to_f32(*(const bf16x8*)(reg + ((2 * oy + ky) * STEMF_R + 2 * ox + kx) * STEMF_PS + cg * 16));
                const f32x4 t0 = *(const f32x4*)&lw1[(ky * 3 + kx) * CO + cg * 8], t1 = *(const f32x4*)&lw1[(ky * 3 + kx) * CO + cg * 8 + 4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = __builtin_fmaf(t0[c], v[c], acc[c]);
                    acc[4 + c] = __builtin_fmaf(t1[c], v[4 + c], acc[4 + c]);
                }
            }
        f32x8 r;
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = gelu_erf(acc[c]);
        if constexpr (FULL) *(bf16x8*)(y1 + pxl * STEMF_PS + cg * 16) = f32_to_bf8(r);
        else *(bf16x8*)(out + (((size_t)b * H2 + ty * STEMF_T + oy) * H2 + tx * STEMF_T + ox) * CO + cg * 8) = f32_to_bf8(r);
    }
    if constexpr (FULL) {
        __syncthreads();
        // ---- phase 3: stem[2] on this wave's 16 pixels.  B operand: lane (pixel lr, k-group g) holds y1[px][32 ks + 8 g .. + 7];
        // A operand: W2 rows 16 ob + lr (output channels), the same k range, straight from the [96][96] bf16 weights (18 KB, L1 / L2 hits);
        // D: lane holds output channels 16 ob + 4 g .. + 3 of pixel lr.
        const int lr = lane & 15, g = lane >> 4;
        char* yrow = y1 + (wave * 16 + lr) * STEMF_PS;
        bf16x8 yb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) yb[ks] = *(const bf16x8*)(yrow + ks * 64 + g * 16);
        const bf16* w2l = w2 + lr * CO + g * 8;
        bf16x8 wb[2][3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wb[0][ks] = *(const bf16x8*)(w2l + ks * 32);
#pragma unroll
        for (int ob = 0; ob < 6; ++ob) {
            if (ob + 1 < 6) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) wb[(ob + 1) & 1][ks] = *(const bf16x8*)(w2l + (ob + 1) * 16 * CO + ks * 32);
            }
            // same arithmetic as the GEMM route (gemm_kernel<3, 32>: three K = 32 MFMAs from zero, then + bias, GELU, one rounding):
            // the fused result is bit-identical to it
            const f32x4 bq = *(const f32x4*)(b2 + ob * 16 + g * 4);
            f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ob & 1][ks], yb[ks], a2, 0, 0, 0);
            f32x4 gl;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) gl[jj] = gelu_erf(a2[jj] + bq[jj]);
            *(bf16x4*)(yrow + (ob * 16 + g * 4) * 2) = f32_to_bf4(gl);   // in place: this wave read its 16 rows above
        }
        // whole pixels out: chunk c = 16-B piece (c % 12) of pixel 16 wave + c / 12; a tile row of 8 pixels is 1536 B contiguous
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = i * 64 + lane, pxl = wave * 16 + c / 12, part = c % 12;
            const int oy = pxl / STEMF_T, ox = pxl - oy * STEMF_T;
            const u32x4 v = *(const u32x4*)(y1 + pxl * STEMF_PS + part * 16);
            *(u32x4*)(out + (((size_t)b * H2 + ty * STEMF_T + oy) * H2 + tx * STEMF_T + ox) * CO + part * 8) = v;
        }
        __syncthreads();                                          // the next tile's phase 0 rewrites the buffer
    }
    }   // tiles of this workgroup
}

// img [B,3,R,R] (dtype) -> out [B,R/4,R/4,96] bf16 = gelu(dw3x3s2(gelu(conv3x3s2(img) + b0)) + b1);  R % 64 == 0
// w2 (bf16 [96][96], as the GEMM packs 1x1 weights) / b2 (fp32 [96]) != NULL: stem[2] too - out = gelu(conv1x1(.) + b2)
extern "C" int fvhd_launch_stem_fused(hipStream_t st, const void* img, int dtype, void* out, const float* w0, const float* b0,
                                      const float* w1, const float* b1, const void* w2, const float* b2, int B, int R)
{
    if ((w2 == nullptr) != (b2 == nullptr)) return (int)hipErrorInvalidValue;
    const bool full = w2 != nullptr;
    if (R % 64) return (int)hipErrorInvalidValue;
    const int txy = R / 4 / STEMF_T;
    const size_t shmem = 6 * 64 * 16 + 9 * 96 * 4 + STEMF_IB + (size_t)STEMF_NP * STEMF_PS;
    const long ntiles = (long)B * txy * txy;
    if (ntiles > 0x7fffffffl) return (int)hipErrorInvalidValue;
    // STEMF_TPW tiles per workgroup, strided by the grid (tile, tile + G, ...): neighbouring workgroups work on neighbouring tiles at
    // the same time (halo pixels hit L2), the tail is at most one tile per workgroup
    constexpr int STEMF_TPW = 4;
    const long G = ntiles >= 4 * 512 * STEMF_TPW ? (ntiles + STEMF_TPW - 1) / STEMF_TPW : ntiles;
    dim3 grid((unsigned)G), block(256);
    static bool attr_set[64][6];
    int dev = 0;
    (void)hipGetDevice(&dev);
#define STEMF_LAUNCH_(TT, FF, IDX)                                                                                             \
    do {                                                                                                                       \
        if (!attr_set[dev & 63][IDX]) {                                                                                        \
            hipError_t e = hipFuncSetAttribute((const void*)stem_fused_kernel<TT, FF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)shmem);                                                                    \
            if (e != hipSuccess) return (int)e;                                                                                \
            attr_set[dev & 63][IDX] = true;                                                                                    \
        }                                                                                                                      \
        hipLaunchKernelGGL((stem_fused_kernel<TT, FF>), grid, block, shmem, st, (const TT*)img, (bf16*)out, w0, b0, w1, b1,    \
                           (const bf16*)w2, b2, B, R, (int)ntiles);                                                            \
    } while (0)
#define STEMF_LAUNCH(TT, IDX) do { if (full) STEMF_LAUNCH_(TT, true, 2 * IDX + 1); else STEMF_LAUNCH_(TT, false, 2 * IDX); } while (0)
    if (dtype == FVHD_F32) STEMF_LAUNCH(float, 0);
    else if (dtype == FVHD_F16) STEMF_LAUNCH(_Float16, 1);
    else if (dtype == FVHD_BF16) STEMF_LAUNCH(bf16, 2);
    else return (int)hipErrorInvalidValue;
#undef STEMF_LAUNCH
#undef STEMF_LAUNCH_
    return (int)hipGetLastError();
}

// ---- SE: global average pool over the T tokens of each image -------------------------------------
// y [B, T, C] bf16 -> pooled [B, C] fp32.  grid (C/128, B); a workgroup = 16 channel groups of 8 (one 16-B load each) x 16 token slices: every
// thread sums T/16 tokens (independent loads in flight instead of one dependent 2-B load per token: 75 -> ~8 us at B = 32), the slices
// are added in order through LDS.
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16* __restrict__ y, float* __restrict__ pooled, int T, int C)
{
    __shared__ float red[16][128 + 4];
    const int cg = threadIdx.x & 15, ts = threadIdx.x >> 4, b = blockIdx.y;
    const int c = blockIdx.x * 128 + cg * 8;
    f32x8 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const bf16* p = y + (size_t)b * T * C + c;
#pragma unroll 4
        for (int t = ts; t < T; t += 16) s += bf8_to_f32(*(const bf16x8*)(p + (size_t)t * C));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[ts][cg * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + (int)threadIdx.x < C) {
        float a = red[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 16; ++i) a += red[i][threadIdx.x];
        pooled[(size_t)b * C + blockIdx.x * 128 + threadIdx.x] = a / (float)T;
    }
}

// ---- SE MLP: scale[b] = sigmoid(We . relu(Wr . pooled[b] + br) + be) -----------------------------
// Wr fp32 [RD][C], We fp32 [C][RD].  Two small launches so that B*RD/4 resp. B*C/256 workgroups
// (1536 / 384 at B = 32) share the 2 x 2.4 MB of weights through L2 instead of 32 workgroups
// streaming them serially.
// hidden[b][r] = relu(Wr[r] . pooled[b] + br[r]);  grid (RD/4, B), one wave per hidden unit.
__global__ __launch_bounds__(256) void se_reduce_kernel(const float* __restrict__ pooled, const float* __restrict__ wr,
                                                        const float* __restrict__ br, float* __restrict__ hidden,
                                                        int C, int RD)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= RD) return;
    const float* wrow = wr + (size_t)r * C;
    const float* p = pooled + (size_t)b * C;
    float s = 0.f;
    for (int i = lane * 4; i < C; i += 256) {
        const f32x4 a = *(const f32x4*)(wrow + i), v = *(const f32x4*)(p + i);
        s += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
    }
    s = wave_sum(s);
    if (lane == 0) hidden[(size_t)b * RD + r] = fmaxf(s + br[r], 0.0f);
}

// scale[b][c] = sigmoid(We[c] . hidden[b] + be[c]);  grid (C/256, B), one thread per channel.
__global__ __launch_bounds__(256) void se_expand_kernel(const float* __restrict__ hidden, const float* __restrict__ we,
                                                        const float* __restrict__ be, float* __restrict__ scale,
                                                        int C, int RD)
{
    extern __shared__ float sh[];       // hidden row [RD]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < RD; i += 256) sh[i] = hidden[(size_t)b * RD + i];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* wrow = we + (size_t)c * RD;
    float s = be[c];
    for (int r = 0; r < RD; r += 4) {
        const f32x4 a = *(const f32x4*)(wrow + r);
        s += a[0] * sh[r] + a[1] * sh[r + 1] + a[2] * sh[r + 2] + a[3] * sh[r + 3];
    }
    scale[(size_t)b * C + c] = sigmoidf_fast(s);
}

// ---- out[b,t,c] = gelu(y[b,t,c] * scale[b,c]) in the caller's dtype -------------------------------
template <typename T>
__global__ __launch_bounds__(256) void se_scale_gelu_kernel(const bf16* __restrict__ y, const float* __restrict__ scale,
                                                            T* __restrict__ out, int TT, int C, long total8)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const long e = i * 8;
    const int c = (int)(e % C);
    const int b = (int)(e / ((long)TT * C));
    const f32x8 v = bf8_to_f32(*(const bf16x8*)(y + e));
    const float* sc = scale + (size_t)b * C + c;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[e + k] = (T)gelu_erf(v[k] * sc[k]);
}

extern "C" int fvhd_launch_se_head(hipStream_t st, const void* y, float* pooled, float* scale, const float* wr,
                                   const float* br, const float* we, const float* be, void* out, int out_dtype,
                                   int B, int T, int C, int RD)
{
    if (C % 8 || RD % 4 || RD > 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(se_pool_kernel, dim3((C + 127) / 128, B), dim3(256), 0, st, (const bf16*)y, pooled, T, C);
    // `pooled` is [B, C + RD]: the pooled row followed by the SE hidden row
    float* hidden = pooled + (size_t)B * C;
    hipLaunchKernelGGL(se_reduce_kernel, dim3((RD + 3) / 4, B), dim3(256), 0, st, pooled, wr, br, hidden, C, RD);
    hipLaunchKernelGGL(se_expand_kernel, dim3((C + 255) / 256, B), dim3(256), (size_t)RD * sizeof(float), st,
                       hidden, we, be, scale, C, RD);
    const long total8 = (long)B * T * C / 8;
    dim3 grid((unsigned)((total8 + 255) / 256));
    if (out_dtype == FVHD_F32) hipLaunchKernelGGL(se_scale_gelu_kernel<float>, grid, dim3(256), 0, st, (const bf16*)y, scale, (float*)out, T, C, total8);
    else if (out_dtype == FVHD_F16) hipLaunchKernelGGL(se_scale_gelu_kernel<_Float16>, grid, dim3(256), 0, st, (const bf16*)y, scale, (_Float16*)out, T, C, total8);
    else if (out_dtype == FVHD_BF16) hipLaunchKernelGGL(se_scale_gelu_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)y, scale, (bf16*)out, T, C, total8);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ---- dtype conversion of token rows (projector input when the caller's tokens are not bf16) -------
template <typename T>
__global__ __launch_bounds__(256) void cast_to_bf16_kernel(const T* __restrict__ x, bf16* __restrict__ y, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (bf16)(float)x[i];
}

extern "C" int fvhd_launch_cast_to_bf16(hipStream_t st, const void* x, int dtype, void* y, long n)
{
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == FVHD_F32) hipLaunchKernelGGL(cast_to_bf16_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (bf16*)y, n);
    else if (dtype == FVHD_F16) hipLaunchKernelGGL(cast_to_bf16_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)x, (bf16*)y, n);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
