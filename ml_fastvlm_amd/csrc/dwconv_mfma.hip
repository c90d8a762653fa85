// Depthwise 7x7 (stride 1, + folded BatchNorm bias) on the matrix cores - ConvFFN.conv / RepCPE of the reference
// (`mci.py:921`, `:992-995`; 46 launches per forward, the second-largest kernel class of the tower).
//
// A depthwise conv has no channel mixing, so a GEMM never appears across channels; the only matrix inside it is the 1-D
// convolution along a row.  For one channel and one tap row ky:
//     out[y, 4j .. 4j+3] += in[y + ky - 3, 4-px segment s] x Toep(ky, s - j)        s - j in {-1, 0, +1}
// with Toep a 4x4 block of the banded Toeplitz matrix of the 7 taps of that row.  gfx950's 16-block MFMA
// (v_mfma_f32_4x4x4_16b_bf16: sixteen independent 4x4x4 products per instruction, 2 passes) takes 16 CHANNELS as its blocks:
//     A (lane 4 b + i): 4 consecutive input pixels of channel b, segment i of a 16-px window
//     B (lane 4 b + j): column j of the Toeplitz block of channel b - built once per wave from the fp32 taps, 21 operands
//     D (lane 4 b + j, register i): output pixel 16 t + 4 i + j of channel b
// 28 of the 48 products of the three segments that feed a 4-px output tile are taps (58 %): 128 x 0.58 = 74 useful FMA per
// cycle per SIMD against 32 for v_pk_fma_f32 (the VALU kernel in dwconv.hip reaches ~35 % of that).  The taps are rounded to
// bf16 (what the reference's own bf16 weights are); accumulation is fp32; one rounding to bf16 at the store.
//
// Layout of the work (tools/ubench/dw_mfma.hip is the study this kernel came out of; measurements in profiles/):
//   * a WAVE owns 16 channels x a 64-px strip and marches down the rows of its chunk with 7 live output rows in registers
//     (112 accumulator VGPRs): every input row is read once and feeds 7 x 12 = 84 MFMAs; the row loop is unrolled 7x so the
//     slot of an output row is a compile-time register index;
//   * a WORKGROUP = 4 waves = 4 adjacent channel groups = 64 channels = one whole 128-B line per pixel: the input row
//     segment arrives by LDS-DMA as whole lines in a ring of RS raw rows shared by the workgroup (each wave issues 2 of the 8
//     interior 1-KiB pieces and a quarter of the halo piece; no VGPRs, RS - 1 rows in flight behind a counted vmcnt), each wave
//     transposes ITS 32-B column of every pixel into a private [channel][pixel] image (the A operand needs pixels of one
//     channel adjacent in a lane; NHWC has channels adjacent - this transposition, ds_read_b128 -> 8 ds_write_b16, is the price
//     of running a depthwise conv on MFMA), and the finished output row is assembled in a shared [pixel][64 ch] buffer and
//     leaves as whole lines (2 stores of 1 KiB per wave).  Per-lane 32-B segments instead (one wave = its own I/O) measured
//     3.3 TB/s of mixed traffic against 4.9 TB/s for whole lines;
//   * one s_barrier per row; columns outside the image stay zero in the transposed image, rows outside are never read.
// Hand-pinned hazards (the compiler does not know the asm statements are MFMAs): see the comments at the asm statements.
#include "fvhd_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int DWM_P = 80;        // pitch (px) of one channel row of the transposed image: (P/2) % 64 == 40 -> the 16 channels of an A read fall on 8 bank groups (2-way = the 512-B minimum)
constexpr int DWM_OPX = 144;     // output staging: 128 B of channels + 16 B pad per pixel (the 4 pixels of one ds_write_b16 on 4 bank groups)
constexpr int DWM_RS = 4;        // raw-row ring depth
constexpr int DWM_RAWB = 72 * 128, DWM_OB = 64 * DWM_OPX, DWM_TB = 16 * DWM_P * 2;
constexpr int DWM_LDS = DWM_RS * DWM_RAWB + 2 * DWM_OB + 4 * DWM_TB;

FVHD_DEV u16 f32_to_bf16_rne(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }

__global__ __launch_bounds__(256, 2) void dw7_mfma_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int H, int W, int C, int RC, int nstrip, int nchunk)
{
    constexpr int NT = 4, CW = 64, SW = 64, IWX = 72, RS = DWM_RS, P = DWM_P, OPX = DWM_OPX, RAWB = DWM_RAWB, OB = DWM_OB, TBY = DWM_TB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* raw = smem;                                       // [RS][8 interior pieces of 1 KiB | halo piece]
    char* O = smem + RS * RAWB;                             // [2][64 px][OPX]
    u16* T = (u16*)(smem + RS * RAWB + 2 * OB + wv * TBY);  // per wave: [16 ch][P px], column 0..3 left halo, 4..67 strip, 68..71 right halo
    const int blk = lane >> 2, q = lane & 3;
    const int NCB = C / CW;
    int L = blockIdx.x;
    const int cb = L % NCB; L /= NCB;
    const int strip = L % nstrip; L /= nstrip;
    const int chunk = L % nchunk;
    const int n = L / nchunk;
    const int c0 = cb * CW + wv * 16, x0 = strip * SW, ylo = chunk * RC, yhi = min(H, ylo + RC);
    const unsigned img_bytes = (unsigned)H * W * C * 2, row_bytes = (unsigned)W * C * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
    const char* ximg = (const char*)(x + (size_t)n * H * W * C);

    // ---- Toeplitz operands of this lane's channel: B[ky][s][k] = tap(ky, kx = 4 (s - 1) + k - j + 3), zero outside 0..6
    s16x4 bop[7][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = 4 * (s - 1) + k - q + 3;
                const float v = (kx >= 0 && kx < 7) ? w[(size_t)(ky * 7 + kx) * C + c0 + blk] : 0.f;
                bop[ky][s][k] = (short)f32_to_bf16_rne(v);
            }
    const float bv = bias ? bias[c0 + blk] : 0.f;
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = f32x4{bv, bv, bv, bv};

    // ---- loads.  Interior pieces 2 wv, 2 wv + 1 (8 px x 128 B each); inside a piece the 16-B chunks are stored
    // [consumer wave][px][half], so that a wave's later ds_read_b128 of its 32-B column is bank-conflict free: DMA lane
    // l = (consumer l >> 4, px (l >> 1) & 7, half l & 1) - the global side is still 8 whole lines.  Halo piece: plain
    // [4 left px | 4 right px][128 B], lanes 16 wv .. 16 wv + 15 of every wave.  Pixels outside the image read a clamped
    // (valid) address and are never transposed.
    const int ipx = x0 + 16 * wv + ((lane >> 1) & 7);
    const unsigned vint0 = (unsigned)((min(ipx, W - 1) * C + cb * CW) * 2 + (lane >> 4) * 32 + (lane & 1) * 16);
    const unsigned vint1 = (unsigned)((min(ipx + 8, W - 1) * C + cb * CW) * 2 + (lane >> 4) * 32 + (lane & 1) * 16);
    const int hp = lane >> 3, hx = hp < 4 ? x0 - 4 + hp : x0 + 60 + hp;
    const unsigned vhalo = (unsigned)((min(max(hx, 0), W - 1) * C + cb * CW) * 2 + (lane & 7) * 16);
    const unsigned long long hmask = 0xffffull << (16 * wv);
    const unsigned raw_lds = lds_addr(raw);
    auto dma = [&](int r, int slot) {
        const char* rb = ximg + (size_t)r * row_bytes;
        const unsigned d0 = raw_lds + slot * RAWB + 2048 * wv, dh = raw_lds + slot * RAWB + 8192;
        unsigned keep; unsigned long long ex;
        // M0 = LDS destination of the piece (+ lane * 16 by the hardware); s_add_u32 clobbers SCC
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
                     "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %8\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %7\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(vint0), "v"(vint1), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb), "s"(hmask) : "memory", "scc");
    };
    // ---- transposition source offsets / validity of this lane's three 16-B chunks (T column 32 m + lane / 2, channel half lane & 1)
    unsigned roff[3];
    bool okm[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int col = 32 * m + (lane >> 1), xi = x0 - 4 + col;
        okm[m] = col < IWX && xi >= 0 && xi < W;
        const int cc = min(col, IWX - 1), ip = cc - 4;
        roff[m] = (unsigned)(cc < 4 ? 8192 + cc * 128 + wv * 32 + (lane & 1) * 16 : cc >= 68 ? 8192 + (cc - 64) * 128 + wv * 32 + (lane & 1) * 16
                                    : (ip >> 3) * 1024 + wv * 256 + (ip & 7) * 32 + (lane & 1) * 16);
    }
    auto transpose = [&](int slot) {           // raw[slot] -> T; the ds_reads of the previous row's A operands are already issued (LDS is in order)
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
    // ---- output: own 16 channels into the shared row buffer; then this wave stores pixels 16 wv .. 16 wv + 15 as whole lines.
    // Pixels right of the image and rows above the chunk get an out-of-range buffer offset (dropped by the range check): no branches.
    const int spx = x0 + 16 * wv + (lane >> 3);
    const unsigned vst0 = (unsigned)((min(spx, W - 1) * C + cb * CW) * 2 + (lane & 7) * 16), oob0 = spx < W ? 0u : 0x80000000u;
    const unsigned vst1 = (unsigned)((min(spx + 8, W - 1) * C + cb * CW) * 2 + (lane & 7) * 16), oob1 = spx + 8 < W ? 0u : 0x80000000u;
    auto stage = [&](f32x4 (&a)[NT], int ob) {
        u16* Ow = (u16*)(O + ob * OB + q * OPX + wv * 32 + blk * 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][0], a[t][1]}, bf16x2_t));
            const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][2], a[t][3]}, bf16x2_t));
            Ow[(16 * t + 0) * (OPX / 2)] = (u16)p01; Ow[(16 * t + 4) * (OPX / 2)] = (u16)(p01 >> 16);
            Ow[(16 * t + 8) * (OPX / 2)] = (u16)p23; Ow[(16 * t + 12) * (OPX / 2)] = (u16)(p23 >> 16);
            a[t] = f32x4{bv, bv, bv, bv};
        }
    };
    auto store = [&](int yo, int ob) {
        const unsigned ro = (unsigned)max(yo, 0) * row_bytes, oobr = yo >= ylo ? 0u : 0x80000000u;     // valid offsets are < 2^31: the flags are OR-ed in
        // same element type as the ds_write_b16 side (strict aliasing: a u32x4 load was hoisted above the u16 stores)
        const u32x4 o0 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + ob * OB + (16 * wv + (lane >> 3)) * OPX + (lane & 7) * 16));
        const u32x4 o1 = __builtin_bit_cast(u32x4, *(const u16x8*)(O + ob * OB + (16 * wv + 8 + (lane >> 3)) * OPX + (lane & 7) * 16));
        __builtin_amdgcn_raw_buffer_store_b128(o0, ry, (vst0 + ro) | oob0 | oobr, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ry, (vst1 + ro) | oob1 | oobr, 0, 0);
    };

    // ---- rows.  Input rows [r_lo, r_hi); the slot of output row yo is (yo - r_lo + 3) % 7, so the unrolled sequence starts at u = 0.
    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
    const u16* rd = T + blk * P + 4 * q;
#pragma unroll
    for (int i = 0; i < RS; ++i) dma(min(r_lo + i, r_hi - 1), (r_lo + i) % RS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    transpose(r_lo % RS);
    int r = r_lo, slot = r_lo % RS, ob = 0;                 // slot: raw slot of row r
    for (;;) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            // every input row feeds the 7 output rows r - 3 .. r + 3, no branches: rows outside [ylo, yhi) are never stored and
            // their slot is re-initialised before its next owner's first contribution.  Order (s, ky, tile): 27 independent
            // MFMAs between two updates of one accumulator; ky = 6 first, so the slot staged after this row gets its last
            // update earliest.  "+v" ties the accumulator in place (the builtin let the allocator copy all 112 around).
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
            // pin the readers of this slot behind the MFMAs and their XDL-write -> VALU-read wait states: unpinned, the first
            // v_cvt of tile 3 was scheduled right behind its last MFMA and read stale registers
            asm volatile("s_nop 7" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
            stage(acc[u], ob);
            // own pieces of row r + 1 have landed once at most the RS - 2 later rows' pieces are outstanding (only loads are
            // counted: stores may retire ahead of older loads); own LDS writes retired; then every wave's are
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * (RS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nslot = slot + 1 == RS ? 0 : slot + 1;
            transpose(nslot);
            store(r - 3, ob);
            dma(min(r + RS, r_hi - 1), slot);               // row r's slot: every wave transposed it before this barrier
            slot = nslot;
            ob ^= 1;
            if (++r >= r_hi) goto done;
        }
    }
done:
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {     // rows whose last input row lies below the image
        const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) stage(acc[s7], ob);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        store(yo, ob);
        ob ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the LDS is released
}

}  // namespace

// 1 = this kernel takes the shape (the caller falls back to the VALU kernel otherwise)
extern "C" int fvhd_dw7_mfma_supported(int H, int W, int C)
{
    return C % 64 == 0 && W >= 64 && H >= 1 && (long long)H * W * C * 2 < (1ll << 31);
}

// x, y [B, H, W, C] bf16 (NHWC); w fp32 [49][C]; bias fp32 [C] or null
extern "C" int fvhd_launch_dw7_mfma(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C)
{
    if (!fvhd_dw7_mfma_supported(H, W, C)) return (int)hipErrorInvalidValue;
    static bool attr_set[64] = {};                           // per device (one process may drive several contexts)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)dw7_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DWM_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set[dev & 63] = true;
    }
    const int RC = 32;                                       // rows per chunk: 38 input rows per 32 output rows; measured best of 16 / 22 / 32 / 43 / 64
    const int nstrip = (W + 63) / 64, nchunk = (H + RC - 1) / RC;
    const long long grid = (long long)B * (C / 64) * nstrip * nchunk;
    if (grid <= 0 || grid > 0x7fffffffll) return (int)hipErrorInvalidValue;
    dw7_mfma_kernel<<<(int)grid, 256, DWM_LDS, st>>>((const u16*)x, (u16*)y, w, bias, H, W, C, RC, nstrip, nchunk);
    return (int)hipGetLastError();
}
