// PatchEmbed's depthwise 7x7 / stride 2 / channel multiplier 2 (+ folded BatchNorm bias, + GELU) on the matrix cores (round 6).
//
// Reference: `ReparamLargeKernelConv.forward` inference branch (mci.py:442-451: one re-parameterised 7x7 conv, groups = C_in,
// C_out = 2 C_in, stride 2, then the activation), the first operator of `PatchEmbed` (mci.py:722-741).  The VALU kernel
// (dwconv.hip: dwconv_tiled_kernel<7, 2, 2, true, ..>) re-reads its 7 x 8 fp32 taps from LDS for every tap row of every 4-px strip
// (14 ds_read_b128 next to 13 pixel reads: the LDS pipe, not the VALU, is its limit) and runs at 0.26 of the HBM rate.
//
// Stride 2 on a Toeplitz product: split every input row into its EVEN and ODD pixels.  With window column c = input pixel
// x_in0 - 4 + c (x_in0 = 2 x first output pixel of the strip), E'[t] = column 2 t + 2 and O[t] = column 2 t + 1, output pixel j reads
//     out[j] = sum_{m = 0..3} tap(ky, 2 m) O[j + m]  +  sum_{m = 0..2} tap(ky, 2 m + 1) E'[j + m]
// i.e. two ordinary stride-1 convolutions with 4 and 3 taps.  On v_mfma_f32_4x4x4_16b_bf16 (16 blocks = 16 OUTPUT channels, i.e. 8 input
// channels, each feeding two blocks with different taps) a 16-px output tile of one tap row costs 4 MFMAs: O and E' segment q, and
// O and E' segment q + 1 (whose first 3 / 2 pixels reach into the outputs of segment q).  7 tap rows -> 28 MFMAs per tile and output
// row, 44 % of their products are taps.  Operand roles as in dwconv_fused.hip (A = Toeplitz^T, B = pixels): lane 4 b + j ends up with
// the four CONSECUTIVE output pixels 4 j .. 4 j + 3 of channel b - one 8-byte LDS write per tile into a [channel][pixel] image, read back
// transposed (ds_read_b64_tr_b16) as whole 128-B lines of 64 output channels.
//
// Work: a workgroup = 4 waves = 64 output channels (32 input channels: 64-B pieces of the input pixels, whole 128-B lines of the output)
// x a strip of 32 output pixels (72 input columns) x a chunk of RC output rows, marching down the rows.  One iteration = one PAIR of input
// rows (2 t, 2 t + 1): row 2 t carries tap rows 1, 3, 5 of output rows t + 1, t, t - 1, row 2 t + 1 tap rows 0, 2, 4, 6 of t + 2, t + 1, t,
// t - 1 - so output row t - 1 is complete, row t + 2 starts, four rows are live (32 accumulator registers).  Input pairs arrive by LDS-DMA
// (ring of DD_RSP pairs, counted vmcnt, one barrier per iteration); every wave transposes its own 16-B column (8 input channels) of the
// 72 pixels into a private [channel][E' | O][pixel] image.  Taps are rounded to bf16 (the reference's own bf16 weights; as in
// dwconv_mfma.hip), accumulation is fp32, bias rides in as the C operand of a row's first MFMA, GELU (fvhd_common.h: degree 7) in fp32.
// LDS: 3 x 9 KB raw pairs + 4 x 2.5 KB images + 4 x 2 x 1.6 KB output rows + 1.3 KB zeros = 52 KB: three workgroups per CU.
#include "fvhd_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;

#ifndef DD_RSP_
#define DD_RSP_ 3
#endif
#ifndef DD_ABL                    // timing-only ablations (wrong results): 1 no MFMAs, 2 no transposing writes, 4 no GELU, 8 no stores, 16 no DMA in the loop,
                                  // 32 no per-iteration barrier (racy), 64 no LDS reads in the loop (raw pair, output row, operands)
#define DD_ABL 0
#endif
constexpr int DD_RSP = DD_RSP_;   // raw ring depth in row PAIRS

struct DdCfg {
    static constexpr int PXB = 64;                                   // bytes of one input pixel that belong to this workgroup (32 channels)
    static constexpr int ROWB = 64 * PXB + 4 * 8 * 16;               // one raw row: 4 interior pieces [consumer wave][16 px][16 B] | halo [wave][8 px][16 B]
    static constexpr int PAIRB = 2 * ROWB;
    static constexpr int PP = 40;                                    // elements of one plane (E' or O) of one channel: 36 used + 4 dump columns
    static constexpr int CHE = 2 * PP;                               // one channel row of an image: [E' | O]
    static constexpr int TE = 8 * CHE, TB = TE * 2;                  // one transposed input row of a wave: elements / bytes
    static constexpr int YP = 48, YSK = 64;                          // output image: row pitch (px) and the skew of rows 8..15 (conflict-free transposing reads)
    static constexpr int YE = 16 * YP + YSK, YB = YE * 2;            // one [16 ch][32 px] output row image
    static constexpr int WSY = 2 * YB + 8;                           // wave stride of the output region (8 bytes beyond a multiple of 128)
    static constexpr int OFF_T = DD_RSP * PAIRB, OFF_Y = OFF_T + 4 * 2 * TB, OFF_Z = (OFF_Y + 4 * WSY + 15) / 16 * 16;
    static constexpr int LDS = OFF_Z + TB;
};

FVHD_DEV u16 dd_bf16_rne(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }

// the 56 MFMAs of one iteration, in issue order.  Groups (input row parity, tap row) -> output row: (even, 5) and (odd, 6) finish row t - 1
// (slot 0), (even, 3) / (odd, 4) row t (slot 1), (even, 1) / (odd, 2) row t + 1 (slot 2), (odd, 0) starts row t + 2 (slot 3).
// MFMAs 0..23: the EVEN input row (tap rows 5, 3, 1: six accumulators in turn) - its operand registers are free from MFMA 24 on and take the
// next pair's even row during the rest of the iteration; 24..39: the odd row's tap rows 6 and 4 alternating (the finishing row is complete
// after MFMA 38 and its epilogue runs beside the rest); 40..55: tap rows 2 and 0 alternating.
struct DdStep { int row, ky, slot, op, tile; };       // row: 0 = even input row, 1 = odd; op: 0 O centre, 1 O next, 2 E' centre, 3 E' next
constexpr DdStep dd_step(int k)
{
    if (k < 24) {
        const int op = k / 6, rem = k % 6, grp = rem >> 1, tile = rem & 1;
        return DdStep{0, 5 - 2 * grp, grp, op, tile};
    }
    const int j = (k - 24) & 15, late = (k - 24) >> 4, which = j & 1, jj = j >> 1, tile = jj & 1, op = jj >> 1;
    return late == 0 ? (which == 0 ? DdStep{1, 6, 0, op, tile} : DdStep{1, 4, 1, op, tile})
                     : (which == 0 ? DdStep{1, 2, 2, op, tile} : DdStep{1, 0, 3, op, tile});
}

template <bool ACT>
__global__ __launch_bounds__(256, 3) void dw7s2_mfma_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int H, int W, int Cin, int OH, int OW, int RC,
                                                            int nstrip, int nchunk)
{
    using K = DdCfg;
    constexpr int PXB = K::PXB, ROWB = K::ROWB, PAIRB = K::PAIRB, PP = K::PP, CHE = K::CHE, TE = K::TE, YP = K::YP, YSK = K::YSK, YE = K::YE, RSP = DD_RSP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane >> 2, q = lane & 3;
    const int Cout = 2 * Cin, NCB = Cin / 32;
    // the NCB channel blocks of one (image, chunk, strip) read the 64-B halves (thirds at C_in = 96) of the SAME 128-B lines: consecutive logical
    // ids, and xcd_remap gives every XCD a contiguous range of them - siblings share an L2 (block id b runs on XCD b % 8: with cb as the fastest
    // digit of the raw id every line was fetched into two or three L2s)
#ifndef DD_XCD
#define DD_XCD 1
#endif
    int L = DD_XCD ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cb = L % NCB; L /= NCB;
    const int strip = L % nstrip; L /= nstrip;
    const int chunk = L % nchunk;
    const int n = L / nchunk;
    const int oc0 = cb * 64 + wv * 16, xo0 = strip * 32, xi0 = strip * 64, ylo = chunk * RC, yhi = min(OH, ylo + RC);
    const unsigned row_bytes = (unsigned)W * Cin * 2, orow_bytes = (unsigned)OW * Cout * 2;
    const char* ximg = (const char*)(x + (size_t)n * H * W * Cin);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)n * OH * OW * Cout), 0, (unsigned)OH * OW * Cout * 2, 0x00020000);
    char* raw = smem;
    u16* T = (u16*)(smem + K::OFF_T + wv * 2 * K::TB);      // private: [2 rows of the pair][8 ch][E' | O][PP]
    const int zrel = (K::OFF_Z - (K::OFF_T + wv * 2 * K::TB)) / 2;     // the shared all-zero row image, as an element offset from T
    const unsigned y_lds = lds_addr(smem + K::OFF_Y);
    u16* Y = (u16*)(smem + K::OFF_Y + wv * K::WSY);         // [2][16 ch][YP px (+ skew)] finished output rows of this wave's 16 channels

    // ---- Toeplitz^T operands of this lane (channel oc0 + blk, output pixel i = q of a segment); k = pixel of the input segment
    s16x4 bop[7][4];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = k - q;
            auto tap = [&](int kx) { return w[(size_t)(ky * 7 + kx) * Cout + oc0 + blk]; };
            const float oc_ = d >= 0 ? tap(2 * d) : 0.f;                        // O segment q:      m = k - i in 0..3
            const float on_ = d <= -1 ? tap(2 * (4 + d)) : 0.f;                 // O segment q + 1:  m = 4 + k - i <= 3
            const float ec_ = (d >= 0 && d <= 2) ? tap(2 * d + 1) : 0.f;        // E' segment q:     m = k - i in 0..2
            const float en_ = d <= -2 ? tap(2 * (4 + d) + 1) : 0.f;             // E' segment q + 1: m = 4 + k - i <= 2
            bop[ky][0][k] = (short)dd_bf16_rne(oc_); bop[ky][1][k] = (short)dd_bf16_rne(on_);
            bop[ky][2][k] = (short)dd_bf16_rne(ec_); bop[ky][3][k] = (short)dd_bf16_rne(en_);
        }
    }
    const float bv = bias ? bias[oc0 + blk] : 0.f;
    f32x4 biasq = {bv, bv, bv, bv};
    asm volatile("" : "+v"(biasq));
    f32x4 acc[4][2];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) { acc[sl][0] = biasq; acc[sl][1] = biasq; }

    // ---- LDS-DMA: waves 0 and 1 bring in the pair's even / odd row - four interior pieces (16 px x 64 B each, stored [consumer wave][px][16 B]: a
    // wave's later 16-B column reads are contiguous) and the 8 halo pixels as [consumer wave][px][16 B] on 32 lanes.  Waves 2 and 3 issue the
    // stores.  So a wave's vmcnt counts ONE kind: with loads and stores in one wave "at most N outstanding" cannot tell an old store from a
    // young load, and the row barrier ended up waiting for a store or a load issued one iteration earlier (a full memory round trip per row).
    // Pixels / rows outside the image load a clamped address.
    auto goff = [&](int px, int off) { return (unsigned)((min(max(px, 0), W - 1) * Cin + cb * 32) * 2 + off); };
    unsigned vint[4];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) vint[pi] = goff(xi0 + pi * 16 + (lane & 15), (lane >> 4) * 16);
    const unsigned vhalo = goff((lane & 7) < 4 ? xi0 - 4 + (lane & 7) : xi0 + 60 + (lane & 7), ((lane >> 3) & 3) * 16);
    const unsigned raw_lds = lds_addr(raw);
    const unsigned long long lanes32 = 0xffffffffull;
    auto dma_row = [&](int t, int slot) {                   // waves 0, 1: input row 2 t + wv into its half of ring slot `slot`
        const char* rb = ximg + (size_t)min(max(2 * t + wv, 0), H - 1) * row_bytes;
        const unsigned d0 = raw_lds + slot * PAIRB + wv * ROWB, dh = d0 + 64 * PXB;
        unsigned keep; unsigned long long ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %9\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
                     "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %9\n\t"
                     "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %10\n\ts_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %9\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex)
                     : "v"(vint[0]), "v"(vint[1]), "v"(vint[2]), "v"(vint[3]), "v"(vhalo), "s"(d0), "s"(dh), "s"(rb), "s"(lanes32) : "memory", "scc");
    };
    // ---- transposition raw row -> T: round 0: lane = interior pixel (window column lane + 4); round 1: lanes 0..7 = the halo pixels
    // (columns 0..3 and 68..71).  Column c goes to plane E' index c / 2 - 1 (c even) or plane O index (c - 1) / 2 (c odd); columns outside the
    // image (and column 0, which no output reads) go to the dump columns 36..39 of their plane, so the image keeps its zeros there.
    const unsigned roff0 = (unsigned)((lane >> 4) * 1024 + wv * 256 + (lane & 15) * 16);
    const unsigned roff1 = (unsigned)(64 * PXB + wv * 128 + (lane & 7) * 16);
    unsigned tdst0, tdst1;
    {
        const int c0 = lane + 4, c1 = (lane & 7) < 4 ? (lane & 7) : 64 + (lane & 7);
        const bool ok0 = (unsigned)(xi0 - 4 + c0) < (unsigned)W, ok1 = (unsigned)(xi0 - 4 + c1) < (unsigned)W && c1 != 0 && lane < 8;
        tdst0 = (unsigned)((c0 & 1) * PP + (ok0 ? ((c0 & 1) ? (c0 - 1) / 2 : c0 / 2 - 1) : 36 + (lane & 3)));
        tdst1 = (unsigned)((c1 & 1) * PP + (ok1 ? ((c1 & 1) ? (c1 - 1) / 2 : c1 / 2 - 1) : 36 + (lane & 3)));
    }
    auto tr_read = [&](u32x4 (&v)[4], int slot) {           // [row of the pair][round]
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            v[2 * rr] = __builtin_bit_cast(u32x4, *(const u16x8*)(raw + slot * PAIRB + rr * ROWB + roff0));
            v[2 * rr + 1] = __builtin_bit_cast(u32x4, *(const u16x8*)(raw + slot * PAIRB + rr * ROWB + roff1));
        }
    };
    auto tr_write1 = [&](const u32x4 (&v)[4], int i, int e) {     // i = 2 row + round; e = channel
        u16* d = T + (i >> 1) * TE + ((i & 1) ? tdst1 : tdst0);
        const unsigned wv_ = e < 2 ? v[i].x : e < 4 ? v[i].y : e < 6 ? v[i].z : v[i].w;
        d[e * CHE] = (e & 1) ? (u16)(wv_ >> 16) : (u16)wv_;
    };
    {
        f32x4 z = {0, 0, 0, 0};
        for (int i = lane; i < 2 * K::TB / 16; i += 64) *(f32x4*)((char*)T + i * 16) = z;
        for (int i = lane; i < K::TB / 16; i += 64) *(f32x4*)(smem + K::OFF_Z + i * 16) = z;      // every wave writes the same zeros
    }
    // ---- pixel operands: lane (b, q) reads 4 pixels of input channel b / 2: plane p, index 16 tile + 4 q (+ 4 for the next segment)
    const u16* rd = T + (blk >> 1) * CHE + 4 * q;
    s16x4 opv[2][8];                                         // [row of the pair][4 tile + op]; each row is refilled as soon as its MFMAs are through
    auto op_fetch = [&](int t, int rr) {                     // operands of row rr of pair t (already transposed in T); a row outside the image: the zero image
        asm volatile("" ::: "memory");
        const int r = 2 * t + rr;
        const int ib = (r >= 0 && r < H) ? rr * TE : zrel;
#pragma unroll
        for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int op = 0; op < 4; ++op)                  // op: 0 O centre, 1 O next, 2 E' centre, 3 E' next
                opv[rr][tile * 4 + op] = *(const s16x4*)&rd[ib + ((op < 2) ? PP : 0) + 16 * tile + 4 * (op & 1)];
        asm volatile("" ::: "memory");
    };
    // ---- output: this wave's finished row goes into Y[buf] as [16 ch][32 px] (one 8-byte write per tile); one iteration later (all four
    // waves' images complete behind the barrier) waves 2 and 3 store 16 pixels x 64 channels = 16 whole lines each, read back transposed:
    // source lane (g, j, c): 4 consecutive pixels of row 8 (so & 1) + j (+ 4 for the second read) of wave (so >> 1)'s image, so = (4 g + c) & 7,
    // pixel quad 16 sw + 8 h + 4 ((4 g + c) >> 3); output lane (g, c = (lane >> 2) & 3, e = lane & 3): 16-B chunk (4 g + c) & 7 of pixel
    // 16 sw + 8 h + 4 ((4 g + c) >> 3) + e   (sw = wv - 2, h = 0, 1: the two store instructions).
    u16* yw = Y + blk * YP + YSK * (blk >> 3) + 4 * q;
    const int sw = wv & 1;
    const int ss = 4 * (lane >> 4) + (lane & 3), so = ss & 7, sj = (lane >> 2) & 3, sr = 8 * (so & 1) + sj;
    const unsigned trsrc = y_lds + (unsigned)((so >> 1) * K::WSY + (sr * YP + YSK * (sr >> 3) + 16 * sw + 4 * (ss >> 3)) * 2);
    const int ds_ = 4 * (lane >> 4) + ((lane >> 2) & 3), dpx = xo0 + 16 * sw + 4 * (ds_ >> 3) + (lane & 3);
    const unsigned vst0 = (unsigned)((min(dpx, OW - 1) * Cout + cb * 64) * 2 + (ds_ & 7) * 16), oobx0 = dpx < OW ? 0u : 0x80000000u;
    const unsigned vst1 = (unsigned)((min(dpx + 8, OW - 1) * Cout + cb * 64) * 2 + (ds_ & 7) * 16), oobx1 = dpx + 8 < OW ? 0u : 0x80000000u;
    auto y_read = [&](u32x4 (&o)[2], int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(size_t)(trsrc + buf * K::YB + 16 * h));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(size_t)(trsrc + buf * K::YB + 16 * h + 8 * YP));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            o[h] = u32x4{l2.x, l2.y, h2.x, h2.y};
        }
    };
    auto y_store = [&](const u32x4 (&o)[2], int yo) {
        const unsigned ro = (unsigned)min(max(yo, 0), OH - 1) * orow_bytes, oobr = (yo >= ylo && yo < yhi) ? 0u : 0x80000000u;
        if (!(DD_ABL & 8)) {
            __builtin_amdgcn_raw_buffer_store_b128(o[0], ry, (vst0 + ro) | oobx0 | oobr, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o[1], ry, (vst1 + ro) | oobx1 | oobr, 0, 0);
        }
    };

    // ---- rows.  Iterations t = ylo - 2 .. yhi; pair t lives in ring slot (t - t0) % RSP.
    const int t0 = ylo - 2, t1 = yhi;
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < RSP; ++i) dma_row(t0 + i, i);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        u32x4 v[4];
        tr_read(v, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) tr_write1(v, i, e);
    }
    op_fetch(t0, 0);
    op_fetch(t0, 1);
    int t = t0, slot = 0, yb = 0;
    for (;;) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // waves 0, 1: own row of pair t + 1 has landed once at most the RSP - 2 later rows' loads (5 each) are outstanding; waves 2, 3 (stores
            // only, two per iteration): never more than 2.5 iterations of stores in flight; own LDS traffic of the previous iteration retired
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(5 * (RSP - 2)) : "memory");
            if (!(DD_ABL & 32)) __builtin_amdgcn_s_barrier();
            const int nslot = slot + 1 == RSP ? 0 : slot + 1;
            u32x4 tv[4], ov[2];
            f32x4 fin[2];
            unsigned pk[4];
            if (DD_ABL & 64) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { tv[i] = u32x4{1, 2, 3, 4}; asm volatile("" : "+v"(tv[i])); }
                ov[0] = ov[1] = tv[0];
            } else {
                if (wv >= 2) y_read(ov, yb ^ 1);             // output row t - 2, staged in the previous iteration
                tr_read(tv, nslot);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 56; ++k) {
                constexpr DdStep dummy{};
                (void)dummy;
                const DdStep s = dd_step(k);
                const int sl = (u + s.slot) & 3;
                if (DD_ABL & 1)
                    asm volatile("" : "+v"(acc[sl][s.tile]) : "v"(opv[s.row][s.tile * 4 + s.op]), "v"(bop[s.ky][s.op]));
                else if (s.ky == 0 && s.op == 0)             // a row starts its life with C = the bias quad (no VALU write in front of an inline-asm MFMA)
                    asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(acc[sl][s.tile]) : "v"(bop[s.ky][s.op]), "v"(opv[s.row][s.tile * 4 + s.op]), "v"(biasq));
                else
                    asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[sl][s.tile]) : "v"(bop[s.ky][s.op]), "v"(opv[s.row][s.tile * 4 + s.op]));
                if (!(DD_ABL & 16) && k == 3 && wv < 2) dma_row(t + RSP, slot);    // pair t's slot: every wave transposed it before this iteration's barrier
                if (k == 6 && wv >= 2) y_store(ov, t - 2);
                if (!(DD_ABL & 2) && k >= 8 && k < 40) tr_write1(tv, (k - 8) >> 3, (k - 8) & 7);
                if (!(DD_ABL & 64) && k == 26) op_fetch(t + 1, 0);           // the even row's registers are free (MFMAs 0..23), its image is written (k = 8..23)
                if (k == 42) {                               // the finishing row's last update was MFMA 38: its readers stay behind this point
                    asm volatile("" : "+v"(acc[u][0]), "+v"(acc[u][1]));
                    fin[0] = acc[u][0]; fin[1] = acc[u][1];
                }
                if (ACT && !(DD_ABL & 4) && k >= 43 && k < 51) {
                    const int v = k - 43;
                    fin[v >> 2][v & 3] = gelu_erf(fin[v >> 2][v & 3]);
                }
                if (k == 52) {
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        pk[2 * tl] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{fin[tl][0], fin[tl][1]}, bf16x2_t));
                        pk[2 * tl + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{fin[tl][2], fin[tl][3]}, bf16x2_t));
                    }
                }
                if (k == 53) { *(u32x2*)&yw[yb * YE] = u32x2{pk[0], pk[1]}; *(u32x2*)&yw[yb * YE + 16] = u32x2{pk[2], pk[3]}; }
                if (!(DD_ABL & 64) && k == 55) op_fetch(t + 1, 1);   // the odd row's operands (its transposing writes are all issued: same wave, in order)
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
            yb ^= 1;
            if (++t > t1) goto done;
        }
    }
done:
    {   // the row staged by the last iteration (output row t1 - 1 = yhi - 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wv >= 2) {
            u32x4 ov[2];
            y_read(ov, yb ^ 1);
            y_store(ov, t1 - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the LDS is released
}

}  // namespace

#ifdef FVHD_DEBUG_KNOBS
static int g_dd_rc = 0, g_dd_off = 0, g_dd_narrow = 0;       // rows per chunk forced / kernel disabled / maps narrower than 24 output pixels too (tools/bench_ops.py dwdown)
extern "C" void fvhd_debug_set_dd_rc(int rc) { g_dd_rc = rc; }
extern "C" void fvhd_debug_set_dd_off(int off) { g_dd_off = off; }
extern "C" void fvhd_debug_set_dd_narrow(int on) { g_dd_narrow = on; }
#else
static constexpr int g_dd_rc = 0, g_dd_off = 0, g_dd_narrow = 0;
#endif

static int dd_cu_count()
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

// output rows per chunk: every chunk costs 3 extra iterations, so as large as still gives every CU its three workgroups
static int dd_rows_per_chunk(int B, int OH, int OW, int Cin)
{
    if (g_dd_rc > 0) return g_dd_rc;
    const long long per = (long long)B * (Cin / 32) * ((OW + 31) / 32);
    const long long want = 3ll * dd_cu_count();
    for (int rc = 64; rc > 8; rc >>= 1)
        if (per * ((OH + rc - 1) / rc) >= want) return rc;
    return 8;
}

// 1 = the matrix-core kernel takes this PatchEmbed depthwise conv (7x7, stride 2, multiplier 2): whole 64-B pieces of the input pixels
// force != 0: whether the kernel CAN take the shape (fvhd_op_dw7s2_mfma, tests); 0: whether the dispatcher of fvhd_op_dwconv picks it
extern "C" int fvhd_dw7s2_mfma_supported(int B, int H, int W, int Cin, int force)
{
    const long long OH = (H + 1) / 2, OW = (W + 1) / 2;
    if (!force) {
        if (g_dd_off) return 0;
        static int env = -1;
        if (env < 0) { const char* e = getenv("FVHD_DWDOWN_MFMA"); env = e ? atoi(e) : 1; }
        if (!env) return 0;
        if (OW < 24 && !g_dd_narrow) return 0;    // less than 3/4 of the 32-px strip: the VALU kernel's finer tiles win (B = 32, 768 @32x32: 37 vs 49 us)
    }
    return Cin % 32 == 0 && H >= 2 && W >= 2 && B >= 1 && (long long)H * W * Cin * 2 < (1ll << 31) && OH * OW * Cin * 4 < (1ll << 31);
}

// x [B, H, W, Cin] bf16 -> y [B, ceil(H/2), ceil(W/2), 2 Cin] bf16; w fp32 [49][2 Cin] (BatchNorm folded); bias fp32 [2 Cin] or null
extern "C" int fvhd_launch_dw7s2_mfma(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int Cin, int gelu)
{
    if (!fvhd_dw7s2_mfma_supported(B, H, W, Cin, 1)) return (int)hipErrorInvalidValue;
    static bool attr_set[64][2] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63][gelu ? 1 : 0]) {
        hipError_t e = gelu ? hipFuncSetAttribute((const void*)dw7s2_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DdCfg::LDS)
                            : hipFuncSetAttribute((const void*)dw7s2_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DdCfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set[dev & 63][gelu ? 1 : 0] = true;
    }
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int RC = dd_rows_per_chunk(B, OH, OW, Cin), nstrip = (OW + 31) / 32, nchunk = (OH + RC - 1) / RC;
    const long long grid = (long long)B * (Cin / 32) * nstrip * nchunk;
    if (grid <= 0 || grid > 0x7fffffffll) return (int)hipErrorInvalidValue;
    if (gelu) dw7s2_mfma_kernel<true><<<(int)grid, 256, DdCfg::LDS, st>>>((const u16*)x, (u16*)y, w, bias, H, W, Cin, OH, OW, RC, nstrip, nchunk);
    else dw7s2_mfma_kernel<false><<<(int)grid, 256, DdCfg::LDS, st>>>((const u16*)x, (u16*)y, w, bias, H, W, Cin, OH, OW, RC, nstrip, nchunk);
    return (int)hipGetLastError();
}
