// RepMixer dw3x3 -> ConvFFN dw7x7 (+ folded BatchNorm) in ONE launch, both on the matrix cores (round 6).
//
// Reference: `RepMixer.forward` inference branch (mci.py:808-811: y = dw3x3(x) + b, identity / BN / layer scale folded into the taps)
// followed by the first operator of `ConvFFN.forward` (mci.py:920-921: A = BN(dw7x7(y)), eval BatchNorm folded into taps + bias).
// The two-kernel route (dwconv.hip, dwconv_mfma.hip) moves four tensor passes through HBM (x in, y out, y in, A out); this kernel
// moves three: y is needed in HBM (the block's residual input, mci.py:1108) but is never read back for the 7x7.
//
// Structure: a workgroup = 8 waves on 64 channels x a 64-px strip x RC output rows, marching down the rows -
//   waves 0-3  PRODUCERS  (16 channels each): x rows arrive by LDS-DMA as whole 128-B lines (ring of RS raw rows, exactly the
//              scheme of dw7_mfma_kernel), each wave transposes its 32-B column into a private [channel][pixel] image, runs the
//              3x3 as a Toeplitz product on v_mfma_f32_4x4x4_16b_bf16 with the OPERAND ROLES SWAPPED (A = Toeplitz^T, B = pixels):
//                  D'[i][j] = sum_k Toep[k][i] * in[segment j][k]   ->  lane 4 b + j, register i = output pixel 4 j + i of channel b
//              i.e. a lane ends up with 4 CONSECUTIVE pixels of one channel - which is exactly the operand layout the 7x7's
//              Toeplitz product wants (lane 4 b + q: 4 consecutive pixels of segment q).  So y goes into the consumer's transposed
//              image with ONE ds_write_b64 per 16-px tile (the 7x7 kernel alone needs 24 ds_write_b16 per row for the same image),
//              is staged [pixel][channel] for its own whole-line store, and never makes the HBM round trip.
//              The 3x3 taps keep 16 mantissa bits: every tap is split hi + lo (two bf16 operands, two MFMAs).  The re-parameterised
//              centre tap is 1 + eps (mci.py:819-859) and y is the block's residual stream - a single bf16 tap would put 2^-9 of x
//              into every y.  Products of bf16 pixels with bf16 tap halves are exact in fp32; accumulation is fp32.
//              The two pixels a 4-px segment needs from its neighbours (kx = -1 of its first, kx = +1 of its last output) ride in
//              ONE extra operand {left, right, 0, 0} read with two 2-byte LDS loads: 4 MFMAs per (tap row, tile) instead of 6.
//   waves 4-7  CONSUMERS (the same 16 channels as producer wave - 4): dw7_mfma_kernel's row loop without its DMA and its
//              transposition - 84 MFMAs per input row, 7 live output rows, A staged and stored as whole lines, optional max |A|
//              for the range guard of the half-precision fused ConvFFN.  Given the same y it produces the same bits as
//              dw7_mfma_kernel (same operands, same MFMA order).
// One s_barrier per row for the whole workgroup; the consumer runs three rows behind the x row the producers are on
// (y row m is complete after x row m + 1, is written during that iteration and read after the next barrier).
// x rows outside the image are zero rows (a zeroed transposed image), y rows outside the image are never produced (the consumer
// skips them exactly as dw7_mfma_kernel does), y columns outside the image are masked to zero before they enter the 7x7's image.
// LDS: 4 raw rows 36 KB + y / A staging 2 x 18 KB + transposed x / y images 2 x 20.5 KB = 113 KB -> one workgroup (two waves
// per SIMD) per CU.
#include "fvhd_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef u16 u16x2 __attribute__((ext_vector_type(2)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u16* lds_u16p;

constexpr int FZ_P = 80;          // pitch (px) of one channel row of a transposed image (dwconv_mfma.hip: DWM_P)
constexpr int FZ_SK = 32;         // channels 8..15 start 64 B later (dwconv_mfma.hip: DWM_TSKEW)
#ifndef FZ_RS_
#define FZ_RS_ 4
#endif
constexpr int FZ_RS = FZ_RS_;     // raw-row ring depth
#ifndef FZ_SIDE_VALU              // 1: the two neighbour-pixel products of a segment (kx = -1 of its first, kx = +1 of its last output) as fp32 FMAs on
#define FZ_SIDE_VALU 1            // the VALU (exact fp32 taps) instead of a second pair of MFMAs: 30 instead of 60 producer MFMAs per row
#endif
#ifndef FZ_ABL                    // timing-only ablations (wrong results): 1 no producer MFMAs, 2 no consumer MFMAs, 4 no transposing writes,
#define FZ_ABL 0                  // 8 no staging writes, 16 no global stores, 32 no LDS-DMA inside the row loop, 64 no per-row barrier (racy)
#endif

struct FzCfg {
    static constexpr int CW = 64, PXB = 128, HALO = 64 * PXB, RAWB = 72 * PXB;
    static constexpr int TB = (16 * FZ_P + FZ_SK) * 2;                  // one [16 ch][P px] image (bytes)
    // raw x rows | transposed x (private per producer wave, 2 each) | y rows (producer -> consumer AND -> the y store, 2 per wave pair) |
    // A rows (consumer -> the A store, 2 per consumer wave): all three images have ONE layout, [wave][buffer][16 ch][P px]
    static constexpr int NTY = 3;                                        // y rows in flight per wave pair: written in iteration g, operands prefetched in g + 1, consumed in g + 2
    // The y and A images are also read by the transposing stores: a 32-lane half of a ds_read_b64_tr_b16 addresses rows {0..3, 8..11} (or
    // {4..7, 12..15}) of all FOUR waves' images at one column.  With rows 8..15 skewed by 128 B the eight rows of one image start on the
    // eight multiples of 8 dwords (mod 64 banks), and a wave-image stride of 8 bytes more than a multiple of 32 puts the four images on
    // the four even residues in between: conflict-free (the 64-B skew of the x images and plain strides: 4-way).  The 8-byte writes and the
    // operand reads of one wave see one image: unchanged, conflict-free.
    static constexpr int SKO = 64, TBO = (16 * FZ_P + SKO) * 2;          // one y / A image (bytes)
    static constexpr int WSY = NTY * TBO + 8, WSA = 2 * TBO + 8;         // wave-image strides of the y and the A region
    static constexpr int OFF_TX = FZ_RS * RAWB, OFF_TY = OFF_TX + 8 * TB, OFF_OA = OFF_TY + 4 * WSY;
    static constexpr int OFF_Z = (OFF_OA + 4 * WSA + 15) / 16 * 16;     // one all-zero image: the operand image of an x row outside the picture
    static constexpr int LDS = OFF_Z + TB;
};
typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;

// -DFZ_TRACE: cycle stamps (s_memtime) of a few row iterations of one producer and one consumer wave into a device array that
// fvhd_debug_fz_trace copies out - a timing aid (the SMEM returns share lgkmcnt with the LDS reads: results of a trace build are not to be trusted)
#ifdef FZ_TRACE
__device__ unsigned long long fz_trace_buf[2 * 4 * 16 * 8];
#define FZ_TS(i) asm volatile("s_memtime %0" : "=s"(ts[i]))
#define FZ_TS_DUMP(role, it)                                                                                       \
    if (blockIdx.x == 7 && (it) >= 9 && (it) < 25 && lane == 0) {                                                  \
        unsigned long long* d = fz_trace_buf + ((((role) * 4 + wq) * 16 + ((it) - 9)) * 8);                        \
        for (int i_ = 0; i_ < 8; ++i_) d[i_] = ts[i_];                                                            \
    }
#else
#define FZ_TS(i)
#define FZ_TS_DUMP(role, it)
#endif

FVHD_DEV u16 fz_bf16_rne(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }
FVHD_DEV float fz_bf16_f32(u16 h) { return __uint_as_float((unsigned)h << 16); }

template <bool AMAX, bool MASKALL>
__global__ __launch_bounds__(512, 2) void dw3_dw7_kernel(const u16* __restrict__ x, u16* __restrict__ y, u16* __restrict__ a,
                                                         const float* __restrict__ w3, const float* __restrict__ b3,
                                                         const float* __restrict__ w7, const float* __restrict__ b7,
                                                         int H, int W, int C, int nstrip, int rows_per_wg, int total_rows, unsigned* amax)
{
    // Work partition: the (image, strip) COLUMNS of H rows each are laid end to end and cut into equal runs of rows_per_wg output rows;
    // a GROUP of C / 64 workgroups (adjacent block ids: one per 64-channel block, each with its own taps for the whole launch) marches down
    // one run in step, one SEGMENT (= the part inside one column) after the other - so the 128-B pieces of a pixel's C channels are read
    // and written by neighbouring CUs at about the same time (DRAM pages), and with one resident workgroup per CU every CU gets the same
    // number of rows whatever B * strips is (B = 32 at C = 384: 32 columns of 64 rows over 42 groups of 6 = 49 rows each).
    using K = FzCfg;
    constexpr int NT = 4, CW = K::CW, PXB = K::PXB, SW = 64, IWX = 72, RS = FZ_RS, P = FZ_P, RAWB = K::RAWB, TBY = K::TB, TE = TBY / 2, SK = FZ_SK, SKO = K::SKO,
                  TEO = K::TBO / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = wv & 3;                                   // channel group of this wave (producer wq and consumer wq + 4 share it)
    const int blk = lane >> 2, q = lane & 3;
    const unsigned img_bytes = (unsigned)H * W * C * 2, row_bytes = (unsigned)W * C * 2;
    // C % 64 == 32 (C = 96: 192-B pixels): the last 64-channel block is half real.  Its lanes outside the C channels read the block's first
    // half a second time (valid addresses, finite values), convolve with a clamped channel's taps, and never store; the waves of those
    // channels stay out of the guard's maximum.  1.33x the work of 96 channels - still ahead of the two launches (profiles/r06_dw_mix_c96.log).
    // (C % 64 != 0: the channel blocks of a pixel share 128-B lines - 192-B pixels.  xcd_remap hands every XCD a contiguous range of logical
    // ids, so the blocks of one run of rows - consecutive ids - work behind ONE L2 instead of fetching / writing every shared line through two)
#ifndef FZ_XCD
#define FZ_XCD 1
#endif
    const int bid = ((FZ_XCD == 1 && (C % CW) != 0) || FZ_XCD == 2) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;    // (2: always - A/B)
    const int NCB = (C + CW - 1) / CW, cb = bid % NCB, c0 = cb * CW + wq * 16;
    const int chv = min(c0 + blk, C - 1);                    // the channel whose taps this lane loads
    const bool wave_real = c0 < C;
    int g0 = (bid / NCB) * rows_per_wg;
    const int g1 = min(g0 + rows_per_wg, total_rows);
    char* raw = smem;                                        // [RS][64 interior px | 8 halo px][PXB]
    // Whole-line stores of a finished row straight from its [ch][px] image (T_y for y, O_A for A), transposed by the LDS read:
    // ds_read_b64_tr_b16 gives lane l of a 16-lane group element j = element (l & 3) of the 8 bytes SOURCE lane 4 j + (l >> 2) & 3 addressed.
    // Source role of lane (group g, j = (lane >> 2) & 3, c = lane & 3): 4 consecutive pixels of channel 8 o + 4 hf + j, o = (4 g + c) & 7,
    // pixel quad 16 wq + 8 h + 4 ((4 g + c) >> 3); output role of lane (g, c = (lane >> 2) & 3, e = lane & 3): channels 8 o .. 8 o + 7
    // (two reads, hf = 0, 1) of pixel 16 wq + 8 h + 4 ((4 g + c) >> 3) + e - so one store instruction (h = 0, 1) moves 8 whole 128-B lines.
    const int ss = 4 * (lane >> 4) + (lane & 3), so = ss & 7, sj = (lane >> 2) & 3, sr = 8 * (so & 1) + sj;    // source: row inside wave (so >> 1)'s image, hf = 0
    const unsigned trsrc = (unsigned)((sr * P + SKO * (sr >> 3) + 16 * wq + 4 * (ss >> 3)) * 2);   // bytes; + wave image (so >> 1), + column offset, + 8 h px, + 4 P hf, + buffer
    const int ds_ = 4 * (lane >> 4) + ((lane >> 2) & 3), dpxr = 16 * wq + 4 * (ds_ >> 3) + (lane & 3);         // output role: strip pixel (h = 0), 16-B chunk ds_ & 7
    auto tr_row = [&](u32x4 (&o)[2], unsigned region_lds, int wave_stride, int coloff, int buf) {       // the 2 x 2 transposing reads of this wave's 16 pixels
        const unsigned a0 = region_lds + trsrc + (unsigned)((so >> 1) * wave_stride + (coloff + buf * TEO) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(size_t)(a0 + 16 * h));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(size_t)(a0 + 16 * h + 8 * P));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            o[h] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
    };
    float amx = 0.f;

    if (wv < 4) {
        // =========================================================== PRODUCER: x -> y (HBM) and y -> T_y (LDS)
        u16* T = (u16*)(smem + K::OFF_TX + wq * 2 * TBY);    // private: [2][16 ch][P px] transposed x rows
        u16* TY = (u16*)(smem + K::OFF_TY + wq * K::WSY);   // shared with consumer wq + 4: [NTY][16 ch][P px] y rows
        constexpr int ZREL = 0;                              // (placeholder: the zero image's offset is wave dependent, below)
        (void)ZREL;
        const int zrel = (K::OFF_Z - (K::OFF_TX + wq * 2 * TBY)) / 2;      // u16 index of the shared all-zero image relative to T
        {
            f32x4 z = {0, 0, 0, 0};
            for (int i = lane; i < TBY / 16; i += 64) *(f32x4*)(smem + K::OFF_Z + i * 16) = z;       // every producer writes the same zeros
        }
        // ---- transposition raw row -> T (this lane's three 16-B chunks: T column 32 m + lane / 2, channel half lane & 1)
        unsigned roff[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int col = 32 * m + (lane >> 1);
            const int cc = min(col, IWX - 1), ip = cc - 4;
            const int sub = wq * 32 + (lane & 1) * 16;
            roff[m] = (unsigned)(cc < 4 ? K::HALO + cc * PXB + sub : cc >= 68 ? K::HALO + (cc - 64) * PXB + sub
                                        : (ip >> 3) * 1024 + wq * 256 + (ip & 7) * 32 + (lane & 1) * 16);
        }
        auto tr_read = [&](u32x4 (&v)[3], int slot) {
#pragma unroll
            for (int m = 0; m < 3; ++m) v[m] = __builtin_bit_cast(u32x4, *(const u16x8*)(raw + slot * RAWB + roff[m]));
        };
        // ---- operand reads of one transposed x row: centre segment (8 B) + the two neighbour pixels (2 B each)
        const u16* rd = T + blk * P + SK * (blk >> 3) + 4 * q;
        unsigned sidebase = lds_addr(rd);
        asm volatile("" : "+v"(sidebase));
        int lofs[5], rofs[5];                                 // (t' = 0, q = 0) has no left pixel, (t' = 4, q = 3) no right one inside the row:
#pragma unroll                                                // both read a finite in-row value; the y pixels they touch are never used
        for (int t = 0; t < 5; ++t) { lofs[t] = 16 * t - ((t == 0 && q == 0) ? 0 : 1); rofs[t] = 16 * t + ((t == 4 && q == 3) ? 3 : 4); }
        u16* tyw = TY + blk * P + SKO * (blk >> 3) + 4 * q;
        const unsigned raw_lds = lds_addr(raw);
        // ---- Toeplitz^T operands of this lane (channel blk, output pixel i = q of a segment), hi + lo halves of every tap:
        //   centre  [k] = tap(ky, kx = k - q + 1)                     (input pixel k of the SAME segment)
        //   side    [0] = tap(ky, 0) for q == 0 (left neighbour's last pixel), [1] = tap(ky, 2) for q == 3 (right neighbour's first)
        s16x4 tch[3], tcl[3], tsh[3], tsl[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kx = k - q + 1;
                const float v = (kx >= 0 && kx < 3) ? w3[(size_t)(ky * 3 + kx) * C + chv] : 0.f;
                const u16 hi = fz_bf16_rne(v), lo = fz_bf16_rne(v - fz_bf16_f32(hi));
                tch[ky][k] = (short)hi; tcl[ky][k] = (short)lo;
            }
            const float vl = q == 0 ? w3[(size_t)(ky * 3 + 0) * C + chv] : 0.f, vr = q == 3 ? w3[(size_t)(ky * 3 + 2) * C + chv] : 0.f;
            const u16 lh = fz_bf16_rne(vl), ll = fz_bf16_rne(vl - fz_bf16_f32(lh)), rh = fz_bf16_rne(vr), rl = fz_bf16_rne(vr - fz_bf16_f32(rh));
            tsh[ky] = s16x4{(short)lh, (short)rh, 0, 0}; tsl[ky] = s16x4{(short)ll, (short)rl, 0, 0};
        }
        // FZ_SIDE_VALU: the neighbour taps in fp32.  In the D' layout a lane holds all four outputs of ITS segment, so every lane applies
        // tap kx = 0 to (left neighbour pixel -> its output 0) and tap kx = 2 to (right neighbour pixel -> its output 3)
        float wl[3], wr[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            wl[ky] = w3[(size_t)(ky * 3 + 0) * C + chv];
            wr[ky] = w3[(size_t)(ky * 3 + 2) * C + chv];
        }
        const float bv3 = b3 ? b3[chv] : 0.f;
        f32x4 biasq = {bv3, bv3, bv3, bv3};
        // side operands {left pixel, right pixel, 0, 0}: five fixed register pairs whose upper halves stay zero for the whole kernel
        s16x4 xs[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) { xs[t] = s16x4{0, 0, 0, 0}; asm volatile("" : "+v"(xs[t])); }

        while (g0 < g1) {
            const int col = g0 / H, ylo = g0 - col * H, yhi = min(H, ylo + (g1 - g0));
            const int n = col / nstrip, strip = col - n * nstrip, x0 = strip * SW;
            g0 += yhi - ylo;
            // y rows the 7x7 reads: [r_lo, r_hi) (all inside the image); x rows the 3x3 reads for them: r_lo - 1 .. r_hi (zero rows outside the image)
            const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3), nrow = r_hi - r_lo, NP = nrow + 2;
            const int ntail = max(0, yhi - max(ylo, r_hi - 3));      // output rows whose last input row lies below the image
            auto goff = [&](int px, int off) {                  // byte offset of (pixel, byte `off` of this block's 128); bytes beyond the C channels wrap into the block's first half
                const int o2 = (cb * CW * 2 + off < C * 2) ? off : off - 64;
                return (unsigned)((min(max(px, 0), W - 1) * C + cb * CW) * 2 + o2);
            };
            const char* ximg = (const char*)(x + (size_t)n * H * W * C);
            asm volatile("" : "+v"(biasq));
            f32x4 acc[3][5];
            float sdl[3][5], sdr[3][5];                          // FZ_SIDE_VALU: neighbour-pixel sums of the three live y rows (outputs 0 and 3 of a segment)
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
#pragma unroll
                for (int t = 0; t < 5; ++t) { acc[sl][t] = biasq; sdl[sl][t] = 0.f; sdr[sl][t] = 0.f; }
            unsigned tdst[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int col_ = 32 * m + (lane >> 1), xi = x0 - 4 + col_;
                const bool ok = col_ < IWX && xi >= 0 && xi < W && !(m == 2 && lane >= 16);
                tdst[m] = (unsigned)((8 * (lane & 1)) * P + SK * (lane & 1) + (ok ? 32 * m + (lane >> 1) : IWX + ((lane >> 1) & 7)));
            }
            auto tr_write1 = [&](const u32x4 (&v)[3], int tb, int m, int e) {
                u16* d = T + tb * TE + tdst[m];
                const unsigned wv_ = e < 2 ? v[m].x : e < 4 ? v[m].y : e < 6 ? v[m].z : v[m].w;
                d[e * P] = (e & 1) ? (u16)(wv_ >> 16) : (u16)wv_;
            };
            {   // columns outside the image are never written: they must read as zeros
                f32x4 z = {0, 0, 0, 0};
                for (int i = lane; i < 2 * TBY / 16; i += 64) *(f32x4*)((char*)T + i * 16) = z;
            }
            // ---- y: column masks (whole 4-px segments: W % 4 == 0).  Whole strips (W % 64 == 0) can only lose the left halo of tile 0
            // (strip 0) and the right halo of tile 4 (last strip); MASKALL handles any width
            unsigned msk[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) msk[t] = ((unsigned)(x0 - 4 + 16 * t + 4 * q) < (unsigned)W) ? 0xffffffffu : 0u;
            // ---- LDS-DMA of one x row (dw7_mfma_kernel's piece layout: 8 px x 128 B per 1-KiB piece, 16-B chunks stored
            // [producer wave][px][half] so that a wave's ds_read_b128 of its 32-B column is conflict-free; halo [4 left | 4 right][128 B]);
            // producer wq issues 2 of the 8 interior pieces and a quarter of the halo.  The producers issue NO stores (the consumers store
            // y and A): their vmcnt counts exactly these loads, in order
            const int ipx = x0 + 16 * wq + ((lane >> 1) & 7), ioff = (lane >> 4) * 32 + (lane & 1) * 16;
            const unsigned vint0 = goff(ipx, ioff), vint1 = goff(ipx + 8, ioff);
            const int hb = 256 * wq + 16 * (lane & 15), hp = hb / PXB;
            const unsigned vhalo = goff(hp < 4 ? x0 - 4 + hp : x0 + 60 + hp, hb % PXB);
            auto dma = [&](int pp, int slot) {                   // x row of iteration min(pp, NP - 1)
                const char* rb = ximg + (size_t)min(max(r_lo - 1 + min(pp, NP - 1), 0), H - 1) * row_bytes;
                const unsigned d0 = raw_lds + slot * RAWB + 2048 * wq, dh = raw_lds + slot * RAWB + K::HALO + 256 * wq;
                unsigned keep; unsigned long long ex;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
                             "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
                             "s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %7\n\t"
                             "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep), "=&s"(ex) : "v"(vint0), "v"(vint1), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb) : "memory", "scc");
            };

            // ---- rows.  Iteration p works on x row xr = r_lo - 1 + p (transposed in T[tb]; the shared zero image when xr is outside the
            // picture); y row m = p - 2 (image row r_lo + m) is complete after its ky = 2 contribution and leaves in the same iteration.
            // p = 0, 1 push garbage through the same path (never stored, overwritten in T_y before the consumer's first read).
            // x rows of iterations 0 .. RS - 1 up front; then row p + RS during iteration p, into the slot of row p (every producer transposed
            // it before barrier p).  Own pieces of row p + 1 have landed once at most the RS - 2 later rows' pieces are outstanding.
#pragma unroll
            for (int i = 0; i < RS; ++i) dma(i, i % RS);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // prologue barrier
            {
                u32x4 v[3];
                tr_read(v, 0);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) tr_write1(v, 0, m, e);
            }
            // operands of one transposed x row: centre segments (8 B) and the neighbour pixels (2 B each, through the opaque base: hipcc otherwise
            // merges them with the centre read into ONE misaligned ds_read_b96).  They are read one iteration AHEAD (T is private to the wave:
            // no barrier between its transposing writes and these reads), so that the MFMA stream starts right behind the row barrier.
            s16x4 xc_n[5];
            unsigned xl_n[5], xr_n[5];
            auto x_fetch = [&](int xrow, int tbuf) {
                asm volatile("" ::: "memory");                  // the u16 transposing stores above must stay above these reads
                const int ib = (xrow >= 0 && xrow < H) ? tbuf * TE : zrel;     // the shared zero image for a row outside the picture
#pragma unroll
                for (int t = 0; t < 5; ++t) xc_n[t] = *(const s16x4*)&rd[ib + 16 * t];
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    xl_n[t] = *(lds_u16p)(size_t)(sidebase + (unsigned)(2 * (ib + lofs[t])));
                    xr_n[t] = *(lds_u16p)(size_t)(sidebase + (unsigned)(2 * (ib + rofs[t])));
                }
            };
            x_fetch(r_lo - 1, 0);
            int p = 0, slot = 0, tb = 0, tyb = 0;                // tyb: T_y image of the row finished in this iteration (rotates over NTY)
#ifdef FZ_TRACE
            unsigned long long ts[8] = {};
#endif
            for (;;) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    FZ_TS(5);
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * (RS - 2)) : "memory");
                    FZ_TS(6);
                    if (!(FZ_ABL & 64)) __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    FZ_TS_DUMP(0, p);
                    FZ_TS(0);
                    const int xr = r_lo - 1 + p;
                    const int nslot = slot + 1 == RS ? 0 : slot + 1;
                    s16x4 xc[5];
                    unsigned xl[5], xr_[5];
                    u32x4 tv[3];
                    unsigned yq[5][2];
#pragma unroll
                    for (int t = 0; t < 5; ++t) { xc[t] = xc_n[t]; xl[t] = xl_n[t]; xr_[t] = xr_n[t]; }     // fetched during the previous iteration
                    __builtin_amdgcn_sched_barrier(0);
#if FZ_SIDE_VALU
                    // 30 MFMAs in the order (tap row ky = 2, 1, 0; centre-hi, centre-lo; tile): an accumulator is updated every 5th MFMA.  y row
                    // slots: ky = 0 starts row m = p (slot u, C = the bias quad), ky = 1 -> (u + 2) % 3, ky = 2 finishes (u + 1) % 3.  The
                    // neighbour pixels go through 2 x 3 fp32 FMAs per tile into side sums that join the accumulators when the row is finished.
                    float fl[5], fr[5];
#pragma unroll
                    for (int k = 0; k < 30; ++k) {
                        const int ky = 2 - k / 10, ty = (k % 10) / 5, t = k % 5;
                        const int sl = ky == 0 ? u : ky == 1 ? (u + 2) % 3 : (u + 1) % 3;
                        if (FZ_ABL & 1) {
                            asm volatile("" : "+v"(acc[sl][t]) : "v"(xc[t]), "v"(tch[ky]), "v"(tcl[ky]));
                        } else if (ky == 0 && ty == 0) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(acc[sl][t]) : "v"(tch[ky]), "v"(xc[t]), "v"(biasq));
                        } else if (ty == 0) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tch[ky]), "v"(xc[t]));
                        } else {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tcl[ky]), "v"(xc[t]));
                        }
                        if (k == 1) tr_read(tv, nslot);
                        if (!(FZ_ABL & 32) && k == 3) dma(p + RS, slot);
                        if (k < 5) {                                  // neighbour pixels of tile k as fp32 (bf16 << 16) and their three tap rows
                            fl[k] = __uint_as_float(xl[k] << 16);
                            fr[k] = __uint_as_float(xr_[k] << 16);
                            const int s2 = (u + 1) % 3, s1 = (u + 2) % 3, s0 = u;
                            sdl[s2][k] = __builtin_fmaf(wl[2], fl[k], sdl[s2][k]); sdr[s2][k] = __builtin_fmaf(wr[2], fr[k], sdr[s2][k]);
                            sdl[s1][k] = __builtin_fmaf(wl[1], fl[k], sdl[s1][k]); sdr[s1][k] = __builtin_fmaf(wr[1], fr[k], sdr[s1][k]);
                            sdl[s0][k] = wl[0] * fl[k];                         sdr[s0][k] = wr[0] * fr[k];     // the slot's new row starts here
                        }
                        if (k == 0) FZ_TS(1);
                        if (k == 10) FZ_TS(2);
                        if (k == 20) FZ_TS(3);
                        if (k == 29) FZ_TS(4);
                        if (!(FZ_ABL & 4) && k >= 5 && k < 13) tr_write1(tv, tb ^ 1, 0, k - 5);
                        if (k >= 13 && k < 18) {
                            // the finished row's last MFMA was 5 .. 9: pin its readers behind this position (asm MFMAs get no hazard padding)
                            const int tt = k - 13, fs = (u + 1) % 3;
                            asm volatile("" : "+v"(acc[fs][tt]));
                            yq[tt][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{acc[fs][tt][0] + sdl[fs][tt], acc[fs][tt][1]}, bf16x2_t));
                            yq[tt][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{acc[fs][tt][2], acc[fs][tt][3] + sdr[fs][tt]}, bf16x2_t));
                            if (MASKALL || tt == 0 || tt == 4) { yq[tt][0] &= msk[tt]; yq[tt][1] &= msk[tt]; }
                            *(u32x2*)&tyw[tyb * TEO + 16 * tt] = u32x2{yq[tt][0], yq[tt][1]};
                        }
                        if (!(FZ_ABL & 4) && k >= 13 && k < 21) tr_write1(tv, tb ^ 1, 1, k - 13);
                        if (!(FZ_ABL & 4) && k >= 21 && k < 29) tr_write1(tv, tb ^ 1, 2, k - 21);
                        if (k == 29) x_fetch(xr + 1, tb ^ 1);          // row p + 1's image is complete (its last chunk went in at k = 21 .. 28)
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
                    // 60 MFMAs in the order (tap row ky = 2, 1, 0; operand centre-hi, centre-lo, side-hi, side-lo; tile): an accumulator is
                    // updated every 5th MFMA.  y row slots: ky = 0 starts row m = p (slot u, C = the bias quad), ky = 1 -> (u + 2) % 3,
                    // ky = 2 finishes (u + 1) % 3.
#pragma unroll
                    for (int k = 0; k < 60; ++k) {
                        const int ky = 2 - k / 20, ty = (k % 20) / 5, t = k % 5;
                        const int sl = ky == 0 ? u : ky == 1 ? (u + 2) % 3 : (u + 1) % 3;
                        if (FZ_ABL & 1) {
                            asm volatile("" : "+v"(acc[sl][t]) : "v"(xc[t]), "v"(xs[t]), "v"(tch[ky]), "v"(tcl[ky]), "v"(tsh[ky]), "v"(tsl[ky]));
                        } else if (ky == 0 && ty == 0) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(acc[sl][t]) : "v"(tch[ky]), "v"(xc[t]), "v"(biasq));
                        } else if (ty == 0) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tch[ky]), "v"(xc[t]));
                        } else if (ty == 1) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tcl[ky]), "v"(xc[t]));
                        } else if (ty == 2) {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tsh[ky]), "v"(xs[t]));
                        } else {
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][t]) : "v"(tsl[ky]), "v"(xs[t]));
                        }
                        // side operand of tile k: its low word is made here - ten slots ahead of its first MFMA (10 + t): the asm MFMAs get no
                        // VALU-write -> MFMA-read wait states from the compiler
                        if (k < 5) {
                            u32x2 w2 = __builtin_bit_cast(u32x2, xs[k]);
                            w2.x = xl[k] | (xr_[k] << 16);
                            xs[k] = __builtin_bit_cast(s16x4, w2);
                            asm volatile("" : "+v"(xs[k]));
                        }
                        if (k == 1) tr_read(tv, nslot);
                        if (!(FZ_ABL & 32) && k == 6) dma(p + RS, slot);
                        if (k == 40) x_fetch(xr + 1, tb ^ 1);      // row p + 1's image is complete (chunk 2 went in at k = 28 .. 35)
                        if (k == 0) FZ_TS(1);
                        if (k == 20) FZ_TS(2);
                        if (k == 40) FZ_TS(3);
                        if (k == 59) FZ_TS(4);
                        if (!(FZ_ABL & 4) && k >= 7 && k < 15) tr_write1(tv, tb ^ 1, 0, k - 7);
                        if (!(FZ_ABL & 4) && k >= 15 && k < 23) tr_write1(tv, tb ^ 1, 1, k - 15);
                        if (k >= 23 && k < 28) {
                            // the finished row's last update was MFMA 15..19: pin its readers behind this position (asm MFMAs get no hazard padding)
                            const int tt = k - 23, fs = (u + 1) % 3;
                            asm volatile("" : "+v"(acc[fs][tt]));
                            yq[tt][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{acc[fs][tt][0], acc[fs][tt][1]}, bf16x2_t));
                            yq[tt][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{acc[fs][tt][2], acc[fs][tt][3]}, bf16x2_t));
                            if (MASKALL || tt == 0 || tt == 4) { yq[tt][0] &= msk[tt]; yq[tt][1] &= msk[tt]; }
                            *(u32x2*)&tyw[tyb * TEO + 16 * tt] = u32x2{yq[tt][0], yq[tt][1]};
                        }
                        if (!(FZ_ABL & 4) && k >= 28 && k < 36) tr_write1(tv, tb ^ 1, 2, k - 28);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                    slot = nslot;
                    tb ^= 1;
                    tyb = tyb + 1 == K::NTY ? 0 : tyb + 1;
                    if (++p >= NP) goto pdone;
                }
            }
        pdone:
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 3 + ntail; ++i) __builtin_amdgcn_s_barrier();     // global iterations NP, NP + 1 (the consumer's last rows) and the consumer's tail
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // no LDS-DMA in flight when the raw ring is re-used / the LDS released
            __builtin_amdgcn_s_barrier();                                           // end of the segment: the LDS images may be re-used
        }
    } else {
        // =========================================================== CONSUMER: T_y -> A (HBM), y (HBM); dw7_mfma_kernel's row loop
#ifndef FZ_CPRIO
#define FZ_CPRIO 0
#endif
        // the consumer carries 84 of the 144 MFMAs a SIMD issues per row and is the wave the row barrier waits for: it wins the arbitration
        if (FZ_CPRIO) __builtin_amdgcn_s_setprio(FZ_CPRIO);
        u16* OA = (u16*)(smem + K::OFF_OA + wq * K::WSA);    // this wave's [2][16 ch][P px] A rows (column = strip pixel)
        const unsigned oa_lds = lds_addr(smem + K::OFF_OA);
        const u16* TY = (const u16*)(smem + K::OFF_TY + wq * K::WSY);
        const unsigned ty_lds = lds_addr(smem + K::OFF_TY);
        s16x4 bop[7][3];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int sg = 0; sg < 3; ++sg)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int kx = 4 * (sg - 1) + k - q + 3;
                    const float v = (kx >= 0 && kx < 7) ? w7[(size_t)(ky * 7 + kx) * C + chv] : 0.f;
                    bop[ky][sg][k] = (short)fz_bf16_rne(v);
                }
        const float bv7 = b7 ? b7[chv] : 0.f;
        f32x4 biasq = {bv7, bv7, bv7, bv7};
        // operand roles swapped like the producer's (A = Toeplitz^T, B = pixels: the same register contents, the same products): lane 4 b + j
        // holds pixels 16 t + 4 j .. + 3 of channel b - one 8-byte LDS write per tile
        u16* oaw = OA + blk * P + SKO * (blk >> 3) + 4 * q;
        auto stage_cvt = [&](unsigned (&pk)[2 * NT], f32x4 (&av)[NT]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pk[2 * t] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{av[t][0], av[t][1]}, bf16x2_t));
                pk[2 * t + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{av[t][2], av[t][3]}, bf16x2_t));
            }
        };
        auto stage_write1 = [&](const unsigned (&pk)[2 * NT], int ob, int t) {
            *(u32x2*)&oaw[(ob & 1) * TEO + 16 * t] = u32x2{pk[2 * t], pk[2 * t + 1]};
        };
        auto o_read = [&](u32x4 (&o)[2], int ob) { tr_row(o, oa_lds, K::WSA, 0, ob & 1); };
        auto y_read = [&](u32x4 (&o)[2], int buf) { tr_row(o, ty_lds, K::WSY, 4, buf); };     // T column = strip pixel + 4
        const u16* rd = TY + blk * P + SKO * (blk >> 3) + 4 * q;

        while (g0 < g1) {
            const int col = g0 / H, ylo = g0 - col * H, yhi = min(H, ylo + (g1 - g0));
            const int n = col / nstrip, strip = col - n * nstrip, x0 = strip * SW;
            g0 += yhi - ylo;
            const int r_lo = max(0, ylo - 3), r_hi = min(H, yhi + 3);
            auto goff = [&](int px, int off) {                  // byte offset of (pixel, byte `off` of this block's 128); bytes beyond the C channels wrap into the block's first half
                const int o2 = (cb * CW * 2 + off < C * 2) ? off : off - 64;
                return (unsigned)((min(max(px, 0), W - 1) * C + cb * CW) * 2 + o2);
            };
            const int dpx = x0 + dpxr;
            const bool chunk_real = cb * CW * 2 + (ds_ & 7) * 16 < C * 2;       // this lane's 8 channels exist
            const unsigned vst0 = goff(dpx, (ds_ & 7) * 16), oob0 = (dpx < W && chunk_real) ? 0u : 0x80000000u;
            const unsigned vst1 = goff(dpx + 8, (ds_ & 7) * 16), oob1 = (dpx + 8 < W && chunk_real) ? 0u : 0x80000000u;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * H * W * C), 0, img_bytes, 0x00020000);
            asm volatile("" : "+v"(biasq));
            f32x4 acc[7][NT];
#pragma unroll
            for (int sl = 0; sl < 7; ++sl)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[sl][t] = biasq;
            auto o_store = [&](const u32x4 (&o)[2], int yo) {
                const unsigned ro = (FZ_ABL & 128) ? 0u : (unsigned)max(yo, 0) * row_bytes, oobr = yo >= ylo ? 0u : 0x80000000u;   // 128: every row onto row 0 (L2-resident lines)
                __builtin_amdgcn_raw_buffer_store_b128(o[0], ra, (vst0 + ro) | oob0 | oobr, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o[1], ra, (vst1 + ro) | oob1 | oobr, 0, 0);
            };
            auto y_store = [&](const u32x4 (&o)[2], int yo) {      // the RepMixer output row the 7x7 is reading: stored by its own chunk only
                const unsigned ro = (FZ_ABL & 128) ? 0u : (unsigned)yo * row_bytes, oobr = (yo >= ylo && yo < yhi) ? 0u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(o[0], ry, (vst0 + ro) | oob0 | oobr, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o[1], ry, (vst1 + ro) | oob1 | oobr, 0, 0);
            };
            __builtin_amdgcn_s_barrier();                          // the producers' prologue barrier
            __builtin_amdgcn_s_barrier();                          // global iterations 0 .. 3: y row 0 is written during iteration 2 (image 2 % NTY),
            __builtin_amdgcn_s_barrier();                          // its first operands are fetched behind barrier 3, one iteration ahead of their
            __builtin_amdgcn_s_barrier();                          // MFMAs - the consumer's stream starts right behind the row barrier too
            __builtin_amdgcn_s_barrier();
            s16x4 av0_n[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) av0_n[t] = *(const s16x4*)&rd[(2 % K::NTY) * TEO + 16 * t];
            int r = r_lo, ob = 0, tb = 2 % K::NTY;                 // tb: T_y image of y row r = (r - r_lo + 2) % NTY
            float cand = 0.f;
#ifdef FZ_TRACE
            unsigned long long ts[8] = {};
#endif
            for (;;) {
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    FZ_TS(5);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    FZ_TS(6);
                    if (!(FZ_ABL & 64)) __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    FZ_TS_DUMP(1, r - r_lo + 4);
                    FZ_TS(0);
                    s16x4 av[3][NT];
                    u32x4 ov[2], yv[2];
                    unsigned pk[2 * NT];
                    const int tbn = tb + 1 == K::NTY ? 0 : tb + 1;
#pragma unroll
                    for (int t = 0; t < NT; ++t) av[0][t] = av0_n[t];          // fetched during the previous iteration
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 84; ++k) {
                        const int s = k / 28, ky = 6 - (k % 28) / 4, t = k % 4;
                        if (FZ_ABL & 2)
                            asm volatile("" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(av[s][t]), "v"(bop[ky][s]));
                        else if (s == 0 && ky == 0)
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(acc[(u + 6 - ky) % 7][t]) : "v"(bop[ky][s]), "v"(av[s][t]), "v"(biasq));
                        else
                            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[(u + 6 - ky) % 7][t]) : "v"(bop[ky][s]), "v"(av[s][t]));
                        if (k == 1) o_read(ov, ob ^ 1);
                        if (k == 2) {
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) av[1][tt] = *(const s16x4*)&rd[tb * TEO + 16 * tt + 4];
                        }
                        if (k == 3) y_read(yv, tb);
                        if (k == 70) {                               // the next row's first operands (written by the producer one iteration ago)
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) av0_n[tt] = *(const s16x4*)&rd[tbn * TEO + 16 * tt];
                        }
                        if (!(FZ_ABL & 16) && k == 10) o_store(ov, r - 4);
                        if (!(FZ_ABL & 16) && k == 18) y_store(yv, r);
                        if (k == 0) FZ_TS(1);
                        if (k == 28) FZ_TS(2);
                        if (k == 56) FZ_TS(3);
                        if (k == 83) FZ_TS(4);
                        if (k == 30) {
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) av[2][tt] = *(const s16x4*)&rd[tb * TEO + 16 * tt + 8];
                        }
                        if (k == 62) {
                            asm volatile("" : "+v"(acc[u][0]), "+v"(acc[u][1]), "+v"(acc[u][2]), "+v"(acc[u][3]));
                            stage_cvt(pk, acc[u]);
                        }
                        if (!(FZ_ABL & 8) && k >= 64 && k < 68) stage_write1(pk, ob, k - 64);
                        if (AMAX && k == 63) cand = 0.f;
                        if (AMAX && (k == 63 || (k >= 80 && k < 83))) {
                            const int t2 = k == 63 ? 0 : k - 79;
                            cand = __builtin_fmaxf(__builtin_fmaxf(cand, __builtin_fabsf(acc[u][t2][0])), __builtin_fabsf(acc[u][t2][1]));
                            cand = __builtin_fmaxf(__builtin_fmaxf(cand, __builtin_fabsf(acc[u][t2][2])), __builtin_fabsf(acc[u][t2][3]));
                        }
                        if (AMAX && k == 83) amx = (r - 3 >= ylo) ? __builtin_fmaxf(amx, cand) : amx;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ob ^= 1;
                    tb = tbn;
                    if (++r >= r_hi) goto cdone;
                }
            }
        cdone:
            {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                u32x4 ov[2];
                o_read(ov, ob ^ 1);
                o_store(ov, r_hi - 4);
            }
            for (int yo = max(ylo, r_hi - 3); yo < yhi; ++yo) {     // rows whose last input row lies below the image (ntail of them)
                const int sl = (yo - r_lo + 3) % 7;
#pragma unroll
                for (int s7 = 0; s7 < 7; ++s7)
                    if (s7 == sl) {
                        unsigned pk[2 * NT];
                        stage_cvt(pk, acc[s7]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) stage_write1(pk, ob, j);
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // end of the segment
        }
        if constexpr (AMAX) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amx = __builtin_fmaxf(amx, __shfl_xor(amx, o, 64));
            if (lane == 0) ((float*)(smem + K::OFF_TX))[wq] = wave_real ? amx : 0.f;       // the producers' images are dead (end-of-segment barrier)
        }
    }
    if constexpr (AMAX) {                                      // workgroup maximum -> one atomic per workgroup, slot by block id
        __syncthreads();
        if (threadIdx.x == 0) {
            const float* red = (const float*)(smem + K::OFF_TX);
            const float m = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
            if (m > 0.f) atomicMax(amax + (blockIdx.x % FVHD_AMAX_SLOTS), __float_as_uint(m));
        }
    }
}

}  // namespace

// Output rows per workgroup: the columns' rows are dealt out evenly over one workgroup per CU (the kernel needs 102 KB of LDS: one
// resident workgroup per CU), but never fewer than 16 per workgroup (every segment costs 8 halo rows and a prologue).
static int fz_cu_count()
{
    static int n[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!n[dev & 63]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev & 63] = v;
    }
    return n[dev & 63];
}

#ifdef FVHD_DEBUG_KNOBS
static int g_fz_rc = 0;                                       // > 0: output rows per workgroup forced (tools/bench_ops.py dw37)
extern "C" void fvhd_debug_set_fz_rc(int rc) { g_fz_rc = rc; }
#else
static constexpr int g_fz_rc = 0;
#endif

static int fz_rows_per_wg(long long total_rows, int ncb)       // total_rows: of ONE channel block; ncb workgroups work on every run of rows
{
    if (g_fz_rc > 0) return g_fz_rc;
    const int groups = fz_cu_count() / ncb > 0 ? fz_cu_count() / ncb : 1;
    const long long per_group = (total_rows + groups - 1) / groups;
    return (int)(per_group < 16 ? 16 : per_group);
}

// 1 = this kernel takes the shape; force: ignore the launch-size rule (tests).  By itself the tower takes it from 6 output rows per CU on
// (B >= 4 at 1024^2; profiles/r06_dw_mix_small_batch.log: B = 4 / 8 / 16 / 32 -> -17 / -15 / -20 / -20 % at C = 192, -2 / -17 / -14 / -15 %
// at C = 384 against the two launches); below that a 16-row run per workgroup no longer gives every CU a workgroup and the finer tiles of
// the two-kernel route win.
extern "C" int fvhd_dw3_dw7_supported(int B, int H, int W, int C, int force)
{
    if (!(C % 32 == 0 && C >= 64 && W % 4 == 0 && W >= 16 && H >= 1 && B >= 1 && (long long)H * W * C * 2 < (1ll << 31))) return 0;
    if (force) return 1;
    const long long total = (long long)B * ((C + 63) / 64) * ((W + 63) / 64) * H;       // output rows x channel blocks
    return W >= 32 && total >= 6ll * fz_cu_count();
}

template <bool AMAX, bool MASKALL>
static int fz_launch(hipStream_t st, const void* x, void* y, void* a, const float* w3, const float* b3, const float* w7, const float* b7,
                     int B, int H, int W, int C, unsigned* amax)
{
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)dw3_dw7_kernel<AMAX, MASKALL>, hipFuncAttributeMaxDynamicSharedMemorySize, FzCfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set[dev & 63] = true;
    }
    const int nstrip = (W + 63) / 64, ncb = (C + 63) / 64;
    const long long total = (long long)B * nstrip * H;                   // output rows of one channel block
    if (total <= 0 || total > 0x7fffffffll) return (int)hipErrorInvalidValue;
    const int rpw = fz_rows_per_wg(total, ncb);
    const long long grid = ((total + rpw - 1) / rpw) * ncb;
    dw3_dw7_kernel<AMAX, MASKALL><<<(int)grid, 512, FzCfg::LDS, st>>>((const u16*)x, (u16*)y, (u16*)a, w3, b3, w7, b7, H, W, C, nstrip, rpw, (int)total, amax);
    return (int)hipGetLastError();
}

// x, y, a [B, H, W, C] bf16 (NHWC); w3 fp32 [9][C], b3 fp32 [C] or null; w7 fp32 [49][C] (BatchNorm folded), b7 fp32 [C] or null;
// amax: null or FVHD_AMAX_SLOTS words receiving max |A| (fp32 bit patterns)
extern "C" int fvhd_launch_dw3_dw7(hipStream_t st, const void* x, void* y, void* a, const float* w3, const float* b3, const float* w7,
                                   const float* b7, int B, int H, int W, int C, unsigned* amax)
{
    if (!fvhd_dw3_dw7_supported(B, H, W, C, 1)) return (int)hipErrorInvalidValue;
    const bool maskall = W % 64 != 0;
    if (amax) return maskall ? fz_launch<true, true>(st, x, y, a, w3, b3, w7, b7, B, H, W, C, amax) : fz_launch<true, false>(st, x, y, a, w3, b3, w7, b7, B, H, W, C, amax);
    return maskall ? fz_launch<false, true>(st, x, y, a, w3, b3, w7, b7, B, H, W, C, nullptr) : fz_launch<false, false>(st, x, y, a, w3, b3, w7, b7, B, H, W, C, nullptr);
}

#ifdef FZ_TRACE
extern "C" int fvhd_debug_fz_trace(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fz_trace_buf), sizeof(unsigned long long) * (n < 1024 ? n : 1024));
}
#endif
