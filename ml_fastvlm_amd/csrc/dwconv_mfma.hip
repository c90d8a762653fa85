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
//   * a WORKGROUP = 4 waves = 4 adjacent channel groups = 64 channels = one whole 128-B line per pixel (6 waves = 96 channels
//     when C is a multiple of 96 only: with C = 96 the row segment of a strip is one contiguous run): the input row segment
//     arrives by LDS-DMA as whole lines in a ring of RS raw rows shared by the workgroup (each wave issues 2 of the 2 NW
//     interior 1-KiB pieces and 16 lanes of the halo pixels; no VGPRs, RS - 2 later rows in flight behind a counted vmcnt),
//     each wave transposes ITS 32-B column of every pixel into a private double-buffered [channel][pixel] image (the A operand
//     needs pixels of one channel adjacent in a lane; NHWC has channels adjacent - this transposition, ds_read_b128 -> 8
//     ds_write_b16, is the price of running a depthwise conv on MFMA), and the finished output row is assembled in a shared
//     [pixel][channel] buffer and leaves as whole lines (2 stores of 1 KiB per wave), one iteration later.  Per-lane 32-B
//     segments instead (one wave = its own I/O) measured 3.3 TB/s of mixed traffic against 4.9 TB/s for whole lines;
//   * one s_barrier per row; inside an iteration every non-MFMA instruction of the row (transposing writes, A reads, stores,
//     the next DMA, conversions, staging writes) sits behind a fixed MFMA of the 84 - one per MFMA, an 8-cycle MFMA leaves one
//     issue slot - so a wave overlaps its own LDS / memory work with its matrix work;
//   * columns outside the image stay zero in the transposed image (maps narrower than a strip run with masked columns), rows
//     outside are never read; rows per chunk shrink with the batch (dwm_rows_per_chunk) so small launches still fill the chip.
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
#ifndef DWM_RS_
#define DWM_RS_ 4
#endif
constexpr int DWM_RS = DWM_RS_;  // raw-row ring depth
// Round 5: channels 8..15 of a transposed image start DWM_TSKEW elements (64 B) later than 8 * P.  The transposing ds_write_b16 of a half-wave
// writes pixel runs of 32 B for BOTH channel halves (lane & 1): 8 rows of P = 80 apart is 1280 B = 0 mod the 128-B bank window of an LDS write -
// a 2-way conflict on every one of the 24 writes per wave-row, a third of this kernel's LDS cycles (SQ_LDS_BANK_CONFLICT 33 %, profiles/r04 / r05
// PMC summaries).  With the skew the two runs sit 64 B apart in the window; the A-operand ds_read_b64 of a half-wave touches ONE channel half
// (lanes 0-31 = channels 0-7), so its conflict-free pattern only shifts.  -DDWM_TSKEW_=0 restores the round-4 layout (A/B).
#ifndef DWM_TSKEW_
#define DWM_TSKEW_ 32
#endif
constexpr int DWM_TSKEW = DWM_TSKEW_;
// pad (bytes) behind a pixel of the output staging buffer: 16.  (Round 5 measured 32 - the four pixels' 32-B runs of a staging ds_write_b16 then
// tile the 128-B bank window instead of overlapping by half: SQ_LDS_BANK_CONFLICT of the class 11.0 % -> 15.2 %, time unchanged,
// profiles/r05_dw7_tskew_ab.log - so 16 stays; -DDWM_OPAD_=32 rebuilds the experiment.)
#ifndef DWM_OPAD_
#define DWM_OPAD_ 16
#endif
#ifndef DWM_ABL                  // timing-only ablations (wrong results; bit 0: no transposing writes, 1: no staging writes, 2: no per-row barrier,
                                 // 3: no MFMAs, 4: no output stores, 5: no LDS-DMA inside the row loop, 6: no LDS reads inside the row loop,
                                 // 7: ONE output staging buffer (racy) - with -DDWM_RS_=5 the deeper ring in the same LDS)
#define DWM_ABL 0
#endif
// NW waves per workgroup = NW adjacent channel groups = 16 NW channels: 4 (64 channels = one 128-B line per pixel; C = 192, 384, ...)
// or 6 (96 channels: with C = 96 the whole row segment of the strip is contiguous in memory)
// NT tiles of 16 px per strip: 4 (64-px strips) or, round 6, 2 (32-px strips for maps at most 32 px wide - stages 3 and 4 of the tower, where
// a 64-px strip is half / three quarters padding: every per-row cost of a wave halves with the strip)
template <int NW, int NT = 4> struct DwmCfg {
    static_assert(NT == 4 || (NT == 2 && NW == 4), "32-px strips: 64-channel workgroups only");
    static constexpr int CW = 16 * NW, PXB = CW * 2;          // channels / bytes per pixel of the workgroup's block
    static constexpr int SW = 16 * NT, IWX = SW + 8;          // strip width, strip + halo columns
    static constexpr int OPX = PXB + DWM_OPAD_;               // output staging: one pixel + pad (the 4 pixels of one ds_write_b16 on 4 disjoint bank groups)
    static constexpr int HALO = SW * PXB;                     // byte offset of the halo pixels inside a raw row
    static constexpr int RAWB = IWX * PXB, OB = SW * OPX, TB = (16 * DWM_P + DWM_TSKEW) * 2;       // TB: one transposed image (+ the skew gap between its channel halves); two per wave
    static constexpr int NLOAD = NT == 4 ? 3 : 2;             // LDS-DMA instructions per wave and row (interior pieces + halo)
    static constexpr int NM = (IWX * 2 + 63) / 64;            // transposition rounds (16-B chunks of this wave's column / 64 lanes)
    static constexpr int NOB = (DWM_ABL & 128) ? 1 : 2;
    static constexpr int LDS = DWM_RS * RAWB + NOB * OB + NW * 2 * TB;
};

FVHD_DEV u16 f32_to_bf16_rne(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }

template <int NW, bool AMAX, int NT = 4>
__global__ __launch_bounds__(64 * NW, 2) void dw7_mfma_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int H, int W, int C, int RC, int nstrip, int nchunk,
                                                              unsigned* amax)
{
    // amax (round 5; AMAX instantiation only - the plain one carries none of the reduction's instructions): max |output| (fp32, before the rounding to bf16) over the rows this launch stores and the columns of its
    // strips (for W % 64 != 0 that includes up to 63 columns right of the image, computed from the zero padding: a superset, never less than
    // the image's own maximum), as fp32 bit patterns of non-negative numbers in a row of FVHD_AMAX_SLOTS words (fvhd_common.h) - the range
    // guard of the half-precision fused ConvFFN
    using K = DwmCfg<NW, NT>;
    constexpr int CW = K::CW, PXB = K::PXB, SW = K::SW, IWX = K::IWX, NM = K::NM, RS = DWM_RS, P = DWM_P, OPX = K::OPX, RAWB = K::RAWB, OB = K::OB, TBY = K::TB, TE = TBY / 2, SK = DWM_TSKEW;   // TE: u16 elements of one transposed image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* raw = smem;                                       // [RS][64 interior px | 8 halo px][PXB]  (NW = 4: chunks permuted inside every 1-KiB piece)
    char* O = smem + RS * RAWB;                             // [2][64 px][OPX]
    u16* T = (u16*)(smem + RS * RAWB + K::NOB * OB + wv * 2 * TBY);  // per wave: [2][16 ch][P px], column 0..3 left halo, 4..67 strip, 68..71 right halo
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
    f32x4 biasq = {bv, bv, bv, bv};
    asm volatile("" : "+v"(biasq));            // one fixed register quad for the whole kernel (not re-materialised next to its readers)
    f32x4 acc[7][NT];
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[sl][t] = biasq;

    // ---- loads.  Every wave issues 2 of the 2 NW interior 1-KiB pieces and 16 lanes (256 B) of the halo pixels.
    // NW = 4: a piece is 8 px x 128 B; inside it the 16-B chunks are stored [consumer wave][px][half], so that a wave's later
    // ds_read_b128 of its 32-B column is bank-conflict free: DMA lane l = (consumer l >> 4, px (l >> 1) & 7, half l & 1) - the
    // global side is still 8 whole lines.  NW = 6: plain byte order (a piece is 1 KiB of the contiguous 192-B pixels; the column
    // read is 8-way conflicted, 12 instead of 6 cycles).  Halo: plain [4 left px | 4 right px][PXB].  Pixels outside the image read
    // a clamped (valid) address and are never transposed.
    auto goff = [&](int px, int off) { return (unsigned)((min(max(px, 0), W - 1) * C + cb * CW) * 2 + off); };
    unsigned vint0, vint1;
    if constexpr (NW == 4) {
        const int ipx = x0 + (NT == 4 ? 16 : 8) * wv + ((lane >> 1) & 7), off = (lane >> 4) * 32 + (lane & 1) * 16;
        vint0 = goff(ipx, off); vint1 = goff(ipx + 8, off);          // (NT == 2: one piece per wave, vint1 unused)
    } else {
        const int b0 = 2048 * wv + 16 * lane, b1 = b0 + 1024;
        vint0 = goff(x0 + b0 / PXB, b0 % PXB); vint1 = goff(x0 + b1 / PXB, b1 % PXB);
    }
    const int hb = 256 * wv + 16 * (lane & 15), hp = hb / PXB;
    const unsigned vhalo = goff(hp < 4 ? x0 - 4 + hp : x0 + SW - 4 + hp, hb % PXB);
    const unsigned raw_lds = lds_addr(raw);
    auto dma = [&](int r, int slot) {
        const char* rb = ximg + (size_t)r * row_bytes;
        const unsigned d0 = raw_lds + slot * RAWB + (NT == 4 ? 2048 : 1024) * wv, dh = raw_lds + slot * RAWB + K::HALO + 256 * wv;
        unsigned keep; unsigned long long ex;
        // M0 = LDS destination of the piece (+ lane * 16 by the hardware); the halo piece runs on lanes 0..15; s_add_u32 clobbers SCC
        if constexpr (NT == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                         "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
                         "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep), "=&s"(ex) : "v"(vint0), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb) : "memory", "scc");
        else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
                     "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %7\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(vint0), "v"(vint1), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb) : "memory", "scc");
    };
    // ---- transposition source offsets / validity of this lane's three 16-B chunks (T column 32 m + lane / 2, channel half lane & 1)
    unsigned roff[NM];
    bool okm[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int col = 32 * m + (lane >> 1), xi = x0 - 4 + col;
        okm[m] = col < IWX && xi >= 0 && xi < W;
        const int cc = min(col, IWX - 1), ip = cc - 4;
        const int sub = wv * 32 + (lane & 1) * 16;
        roff[m] = (unsigned)(cc < 4 ? K::HALO + cc * PXB + sub : cc >= SW + 4 ? K::HALO + (cc - SW) * PXB + sub
                                    : NW == 4 ? (ip >> 3) * 1024 + wv * 256 + (ip & 7) * 32 + (lane & 1) * 16 : ip * PXB + sub);
    }
    // transposition in two halves so that the row loop can put MFMAs between the reads and the writes.  The writes are
    // unconditional (no EXEC juggling inside the MFMA stream): lanes without a pixel of the image - halo columns outside it, lanes
    // >= 16 of the third chunk - write into the 8 spare columns 72..79 of the pitch instead, and the columns outside the image keep
    // their zeros.
    unsigned tdst[NM];                         // u16 index inside one T buffer (okm: col < IWX, i.e. not the lanes past the last chunk of the last round)
#pragma unroll
    for (int m = 0; m < NM; ++m)
        tdst[m] = (unsigned)((8 * (lane & 1)) * P + SK * (lane & 1) + (okm[m] ? 32 * m + (lane >> 1) : IWX + ((lane >> 1) & 7)));
    auto tr_read = [&](u32x4 (&v)[NM], int slot) {
#pragma unroll
        for (int m = 0; m < NM; ++m) v[m] = __builtin_bit_cast(u32x4, *(const u16x8*)(raw + slot * RAWB + roff[m]));
    };
    auto tr_write1 = [&](const u32x4 (&v)[NM], int tb, int m, int e) {      // element e (channel 8 half + e) of chunk m
        u16* d = T + tb * TE + tdst[m];
        const unsigned wv_ = e < 2 ? v[m].x : e < 4 ? v[m].y : e < 6 ? v[m].z : v[m].w;
        d[e * P] = (e & 1) ? (u16)(wv_ >> 16) : (u16)wv_;
    };
    {
        f32x4 z = {0, 0, 0, 0};
        for (int i = lane; i < 2 * TBY / 16; i += 64) *(f32x4*)((char*)T + i * 16) = z;
    }
    // ---- output: own 16 channels into the shared row buffer; then this wave stores 2 of the 2 NW 1-KiB pieces of the row (whole
    // lines).  Pixels right of the image and rows above the chunk get an out-of-range buffer offset (dropped by the range check).
    const int sb0 = (NT == 4 ? 2048 : 1024) * wv + 16 * lane, sb1 = sb0 + 1024;   // byte inside the [SW px][PXB] output row (NT == 2: one piece per wave, sb1 unused)
    const int spx0 = x0 + sb0 / PXB, spx1 = x0 + sb1 / PXB;
    const unsigned vst0 = goff(spx0, sb0 % PXB), oob0 = spx0 < W ? 0u : 0x80000000u;
    const unsigned vst1 = goff(spx1, sb1 % PXB), oob1 = spx1 < W ? 0u : 0x80000000u;
    const unsigned ord0 = (unsigned)((sb0 / PXB) * OPX + sb0 % PXB), ord1 = (unsigned)((sb1 / PXB) * OPX + sb1 % PXB);
    auto stage_cvt = [&](unsigned (&pk)[2 * NT], f32x4 (&a)[NT]) {      // 8 packed pairs (px 16 t + {0, 4}, px 16 t + {8, 12})
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            pk[2 * t] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][0], a[t][1]}, bf16x2_t));
            pk[2 * t + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a[t][2], a[t][3]}, bf16x2_t));
        }
    };
    auto stage_write1 = [&](const unsigned (&pk)[2 * NT], int ob, int j) {  // j = 0..4 NT - 1: pixel 16 (j / 4) + 4 (j % 4) + q
        u16* Ow = (u16*)(O + (ob & (K::NOB - 1)) * OB + q * OPX + wv * 32 + blk * 2);
        const unsigned v = pk[j >> 1];
        Ow[(16 * (j >> 2) + 4 * (j & 3)) * (OPX / 2)] = (j & 1) ? (u16)(v >> 16) : (u16)v;
    };
    auto stage = [&](f32x4 (&a)[NT], int ob) {
        unsigned pk[2 * NT];
        stage_cvt(pk, a);
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j) stage_write1(pk, ob, j);
    };
    auto o_read = [&](u32x4 (&o)[2], int ob) {
        // same element type as the ds_write_b16 side (strict aliasing: a u32x4 load was hoisted above the u16 stores)
        o[0] = __builtin_bit_cast(u32x4, *(const u16x8*)(O + (ob & (K::NOB - 1)) * OB + ord0));
        if constexpr (NT == 4) o[1] = __builtin_bit_cast(u32x4, *(const u16x8*)(O + (ob & (K::NOB - 1)) * OB + ord1));
    };
    auto o_store = [&](const u32x4 (&o)[2], int yo) {
        const unsigned ro = (unsigned)max(yo, 0) * row_bytes, oobr = yo >= ylo ? 0u : 0x80000000u;     // valid offsets are < 2^31: the flags are OR-ed in
        __builtin_amdgcn_raw_buffer_store_b128(o[0], ry, (vst0 + ro) | oob0 | oobr, 0, 0);
        if constexpr (NT == 4) __builtin_amdgcn_raw_buffer_store_b128(o[1], ry, (vst1 + ro) | oob1 | oobr, 0, 0);
    };

    // ---- rows.  Input rows [r_lo, r_hi); the slot of output row yo is (yo - r_lo + 3) % 7, so the unrolled sequence starts at u = 0.
    // One iteration = one input row r (already transposed in T[tb]):
    //     barrier: raw row r + 1 and the staged output row r - 4 are complete for every wave
    //     LDS reads issued up front (A operands of segment 0, this wave's column of raw row r + 1, its pieces of output row r - 4)
    //     28 MFMAs | stores + next DMA, 16 transposing writes | 28 MFMAs | 8 transposing writes | 28 MFMAs | stage output row r - 3
    // so that a wave's LDS traffic and its MFMAs overlap inside the wave (with one workgroup per CU - the 96-channel case - nothing
    // else would), and the per-row memory instructions sit behind the first MFMA group instead of in front of a stall.
    const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
    const u16* rd = T + blk * P + SK * (blk >> 3) + 4 * q;
#pragma unroll
    for (int i = 0; i < RS; ++i) dma(min(r_lo + i, r_hi - 1), (r_lo + i) % RS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        u32x4 v[NM];
        tr_read(v, r_lo % RS);
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int e = 0; e < 8; ++e) tr_write1(v, 0, m, e);
    }
    int r = r_lo, slot = r_lo % RS, ob = 0, tb = 0;         // slot: raw slot of row r; ob: staging buffer of output row r - 3; tb: T buffer of row r
    float amx = 0.f, cand = 0.f;                            // running max |output| of this lane / of the row being finished
    for (;;) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            // own pieces of row r + 1 have landed once at most the RS - 2 later rows' pieces are outstanding (only loads are
            // counted: stores may retire ahead of older loads); own staging writes of the previous iteration retired
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(K::NLOAD * (RS - 2)) : "memory");
            if (!(DWM_ABL & 4)) __builtin_amdgcn_s_barrier();
            const int nslot = slot + 1 == RS ? 0 : slot + 1;
            s16x4 a[3][NT];                                     // A operands of the three segments
            u32x4 tv[NM], ov[2];
            unsigned pk[2 * NT];
            if (DWM_ABL & 64) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { a[0][t] = s16x4{1, 2, 3, 4}; a[1][t] = a[0][t]; a[2][t] = a[0][t]; asm volatile("" : "+v"(a[0][t]), "+v"(a[1][t]), "+v"(a[2][t])); }
#pragma unroll
                for (int m = 0; m < NM; ++m) { tv[m] = u32x4{1, 2, 3, 4}; asm volatile("" : "+v"(tv[m])); }
                ov[0] = ov[1] = tv[0];
                asm volatile("" : "+v"(ov[0]), "+v"(ov[1]));
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) a[0][t] = *(const s16x4*)&rd[tb * TE + 16 * t];
            tr_read(tv, nslot);
            o_read(ov, ob ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            // 84 MFMAs in the order (segment s, tap row ky = 6..0, tile t): 27 independent MFMAs between two updates of one
            // accumulator; ky = 6 first, so the slot staged at the end gets its last update earliest.  "+v" ties the accumulator in
            // place (the builtin let the allocator copy all 112 around).  Every input row feeds the 7 output rows r - 3 .. r + 3
            // without branches: rows outside [ylo, yhi) are never stored, their slot is re-initialised before its next owner's
            // first contribution.  The row's other instructions are dealt out ONE PER MFMA behind fixed positions of the stream
            // (an 8-cycle MFMA leaves one issue slot): they overlap the matrix pipe inside the wave instead of stalling in bursts.
            // (positions for NT = 4 in the comments; NT = 2: the 42-MFMA stream with the same instructions at the scaled positions)
            constexpr int NK = 21 * NT, KS = 7 * NT, KF = 2 * KS + NT + 2;     // KF: the finished slot's last update was MFMA 2 KS .. 2 KS + NT - 1
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int s = k / KS, ky = 6 - (k % KS) / NT, t = k % NT;
                // a slot starts its life (output row r + 3: ky = 0 of segment 0) with C = the bias quad instead of being re-initialised
                // by VALU moves: the register allocator placed those moves directly in front of the MFMA that reads them, and an
                // inline-asm MFMA gets no VALU-write -> MFMA-read wait states from the compiler (wrong sums in 2 of 4 registers)
                if (DWM_ABL & 8)
                    asm volatile("" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[s][t]), "v"(bop[ky][s]));
                else if (s == 0 && ky == 0)
                    asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[s][t]), "v"(bop[ky][s]), "v"(biasq));
                else
                    asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(a[s][t]), "v"(bop[ky][s]));
                if (!(DWM_ABL & 64) && k == 2) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) a[1][tt] = *(const s16x4*)&rd[tb * TE + 16 * tt + 4];
                }
                if (!(DWM_ABL & 16) && k == 5) o_store(ov, r - 4);                 // staged in the previous iteration
                if (!(DWM_ABL & 32) && k == 8) dma(min(r + RS, r_hi - 1), slot);   // row r's slot: every wave transposed it before this iteration's barrier
                if (!(DWM_ABL & 1) && k >= 10 && k < 18) tr_write1(tv, tb ^ 1, 0, k - 10);
                if (!(DWM_ABL & 1) && k >= 18 && k < 26) tr_write1(tv, tb ^ 1, 1, k - 18);
                if (!(DWM_ABL & 64) && k == KS + 2) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) a[2][tt] = *(const s16x4*)&rd[tb * TE + 16 * tt + 8];
                }
                if constexpr (NM == 3) {
                    if (!(DWM_ABL & 1) && k >= 32 && k < 40) tr_write1(tv, tb ^ 1, 2, k - 32);
                }
                if (k == KF) {
                    // the slot's last update was MFMA 56..59: pin its readers behind the stream position (the compiler does not
                    // know the asm statements are MFMAs and once scheduled the first v_cvt right behind the last MFMA: stale registers)
                    asm volatile("" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
                    stage_cvt(pk, acc[u]);
                }
                if (!(DWM_ABL & 2) && k >= KF + 2 && k < KF + 2 + 4 * NT) stage_write1(pk, ob, k - (KF + 2));
                // max |.| of the finished row (its 16 accumulators: final since MFMA 59, pinned at k = 62), two v_max3_f32 per free slot;
                // rows above the chunk (their slots hold partial sums and are never stored) do not count
                if (AMAX && k == KF + 1) cand = 0.f;
                if (AMAX && (k == KF + 1 || (k >= NK - NT && k < NK - 1))) {
                    const int t = k == KF + 1 ? 0 : k - (NK - NT) + 1;
                    cand = __builtin_fmaxf(__builtin_fmaxf(cand, __builtin_fabsf(acc[u][t][0])), __builtin_fabsf(acc[u][t][1]));
                    cand = __builtin_fmaxf(__builtin_fmaxf(cand, __builtin_fabsf(acc[u][t][2])), __builtin_fabsf(acc[u][t][3]));
                }
                if (AMAX && k == NK - 1) amx = (r - 3 >= ylo) ? __builtin_fmaxf(amx, cand) : amx;
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
            ob ^= 1;
            tb ^= 1;
            if (++r >= r_hi) goto done;
        }
    }
done:
    {   // the row staged by the last iteration
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        u32x4 ov[2];
        o_read(ov, ob ^ 1);
        o_store(ov, r_hi - 4);
    }
    for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {     // rows whose last input row lies below the image
        const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7)
            if (s7 == sl) {
                stage(acc[s7], ob);
                if constexpr (AMAX) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) amx = __builtin_fmaxf(amx, __builtin_fabsf(acc[s7][t][i]));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        u32x4 ov[2];
        o_read(ov, ob);
        o_store(ov, yo);
        ob ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the LDS is released
    if constexpr (AMAX) {                                   // wave -> workgroup (through LDS) -> one atomic per workgroup, slot by block id
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = __builtin_fmaxf(amx, __shfl_xor(amx, o, 64));
        __syncthreads();                                    // every wave is done with the staging buffers
        float* red = (float*)smem;
        if (lane == 0) red[wv] = amx;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = red[0];
#pragma unroll
            for (int i = 1; i < NW; ++i) m = __builtin_fmaxf(m, red[i]);
            if (m > 0.f) atomicMax(amax + (blockIdx.x % FVHD_AMAX_SLOTS), __float_as_uint(m));
        }
    }
}

}  // namespace

// rows per chunk: 32 (38 input rows per 32 output rows - measured best of 16 / 22 / 32 / 43 / 64 at B = 32) while that still gives ~one
// workgroup per CU; smaller chunks for small batches (a workgroup is a serial march down its rows: 12 workgroups per image at stage 3 would
// leave a B = 8 launch on 96 of 256 CUs) - but not smaller than that needs: at B = 8 (profiles/r04_dw7_small_batch.log) 192-256 workgroups
// of the larger chunk beat 384-512 of the smaller one (C = 96: 57.9 vs 70.2 us, C = 384: 25.1 vs 28.7, C = 192: 37.0 vs 39.8; the halo rows
// are 19 % of the work at 32 rows per chunk, 38 % at 16, 75 % at 8).  0 = too few workgroups even at 8 rows -> the VALU kernel (finer tiles)
// is the better choice (B = 1, C = 96: 18.8 vs 24.4 us)
#ifdef FVHD_DEBUG_KNOBS
static int g_dwm_rc = 0;                                     // > 0: rows per chunk forced (tools/bench_ops.py dw7small)
static int g_dwm_nw = 0;                                     // 6: the 96-channel workgroup wherever C % 96 == 0 (tools/bench_ops.py dw7nw)
static int g_dwm_nt = 0;                                     // 4: 64-px strips also for maps at most 32 px wide (A/B of the 32-px strips)
extern "C" void fvhd_debug_set_dwm_rc(int rc) { g_dwm_rc = rc; }
extern "C" void fvhd_debug_set_dwm_nw(int nw) { g_dwm_nw = nw; }
extern "C" void fvhd_debug_set_dwm_nt(int nt) { g_dwm_nt = nt; }
#else
static constexpr int g_dwm_rc = 0, g_dwm_nw = 0, g_dwm_nt = 0;
#endif

// 32-px strips (round 6) for maps at most 32 px wide with 64-channel workgroups: stages 3 and 4 of the tower (B = 32: 48.9 -> ... us per launch
// at 768 @32x32, profiles/r06_dw7_strip32.log)
static bool dwm_strip32(int W, int C) { return W <= 32 && C % 64 == 0 && !(g_dwm_nw == 6 && C % 96 == 0) && g_dwm_nt != 4; }

static int dwm_rows_per_chunk(int B, int H, int W, int C)
{
    if (g_dwm_rc > 0) return g_dwm_rc;
    const int nw = (C % 64 == 0 && !(g_dwm_nw == 6 && C % 96 == 0)) ? 4 : 6;
    const int sw = dwm_strip32(W, C) ? 32 : 64;
    const long long per_row_chunk = (long long)B * (C / (16 * nw)) * ((W + sw - 1) / sw);
    if (per_row_chunk * ((H + 31) / 32) >= 192) return 32;
    if (per_row_chunk * ((H + 15) / 16) >= 192) return 16;
    if (per_row_chunk * ((H + 7) / 8) >= 192) return 8;
    return 0;
}

template <int NW, bool AMAX, int NT = 4>
static int launch_dwm(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C, unsigned* amax)
{
    static bool attr_set[64] = {};                           // per device (one process may drive several contexts)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)dw7_mfma_kernel<NW, AMAX, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, DwmCfg<NW, NT>::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set[dev & 63] = true;
    }
    const int rc_ = dwm_rows_per_chunk(B, H, W, C), RC = rc_ > 0 ? rc_ : 8;
    const int nstrip = (W + 16 * NT - 1) / (16 * NT), nchunk = (H + RC - 1) / RC;
    const long long grid = (long long)B * (C / (16 * NW)) * nstrip * nchunk;
    if (grid <= 0 || grid > 0x7fffffffll) return (int)hipErrorInvalidValue;
    dw7_mfma_kernel<NW, AMAX, NT><<<(int)grid, 64 * NW, DwmCfg<NW, NT>::LDS, st>>>((const u16*)x, (u16*)y, w, bias, H, W, C, RC, nstrip, nchunk, amax);
    return (int)hipGetLastError();
}

// 1 = this kernel takes the shape (the caller falls back to the VALU kernel otherwise); force: ignore the small-batch rule (tests)
extern "C" int fvhd_dw7_mfma_supported(int B, int H, int W, int C, int force)
{
    if (!((C % 64 == 0 || C % 96 == 0) && W >= 16 && H >= 1 && B >= 1 && (long long)H * W * C * 2 < (1ll << 31))) return 0;
    if (force) return 1;
    // narrower maps run the same 64-px strip with masked columns: still ahead of the VALU kernel down to W = 24 (B = 32, C = 768 @32x32:
    // 54 vs 62 us; 48x48: 57 vs 83; C = 1536 @24x24: 42 vs 53), behind it at W = 16 (47 vs 34: three quarters of the strip is padding)
    return (W >= 24 || (dwm_strip32(W, C) && W >= 12)) && dwm_rows_per_chunk(B, H, W, C) > 0;
}

// x, y [B, H, W, C] bf16 (NHWC); w fp32 [49][C]; bias fp32 [C] or null
extern "C" int fvhd_launch_dw7_mfma(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C, unsigned* amax)
{
    if (!fvhd_dw7_mfma_supported(B, H, W, C, 1)) return (int)hipErrorInvalidValue;
    const bool nw4 = C % 64 == 0 && !(g_dwm_nw == 6 && C % 96 == 0);
    if (dwm_strip32(W, C)) return amax ? launch_dwm<4, true, 2>(st, x, y, w, bias, B, H, W, C, amax) : launch_dwm<4, false, 2>(st, x, y, w, bias, B, H, W, C, nullptr);
    if (amax) return nw4 ? launch_dwm<4, true>(st, x, y, w, bias, B, H, W, C, amax) : launch_dwm<6, true>(st, x, y, w, bias, B, H, W, C, amax);
    return nw4 ? launch_dwm<4, false>(st, x, y, w, bias, B, H, W, C, nullptr) : launch_dwm<6, false>(st, x, y, w, bias, B, H, W, C, nullptr);
}
