// Depthwise convolutions of the FastViTHD path, NHWC bf16 activations, fp32 weights and accumulation.
//
// Covers (reference file llava/model/multimodal_encoder/mobileclip/mci.py):
//   K=3 S=1 M=1            RepMixer.reparam_conv                    (mci.py:808-811)
//   K=3 S=2 M=1 +GELU      convolutional_stem[1]                    (mci.py:575-586, 194-198)
//   K=7 S=1 M=1            ConvFFN.conv (+ eval BatchNorm folded)   (mci.py:885-907, 921)
//                          RepCPE.reparam_conv                      (mci.py:992-995)
//   K=7 S=2 M=2 +GELU      PatchEmbed.proj[0] = ReparamLargeKernelConv.lkb_reparam (mci.py:442-451)
//   K=3 S=1 M=2            FastViT.conv_exp.reparam_conv            (mci.py:1401-1411)
// where M is the channel multiplier (groups = Cin, Cout = M*Cin: output channel oc reads input
// channel oc / M, which is PyTorch's grouped-conv convention).
//
// HBM-bound op.  Layout choices for gfx950:
//   * NHWC so that the channel axis is contiguous: one lane owns 8 output channels (16 B of bf16),
//     8 or 12 consecutive lanes cover a 128 B / 192 B contiguous run of one pixel -> coalesced
//     16-B-per-lane loads and stores.
//   * a workgroup owns one channel slice (64 or 96 output channels) x a run of output strips; the
//     slice's K*K*CS fp32 taps are staged in LDS once per workgroup (<= 18.4 KiB) and read back as
//     32-B-per-lane ds_read_b128 pairs (8 distinct addresses per wave, the rest broadcast).
//   * each lane walks a strip of OWT output pixels along x and re-uses every loaded input vector
//     for up to min(K, OWT) outputs, cutting the K*K loads per output to K*(OWT*S+K-S)/OWT.
//   * blockIdx.x is remapped so each XCD (private L2) gets a contiguous range of rows: the K-1
//     halo rows a workgroup shares with its vertical neighbours are then L2 hits, not HBM re-reads.
#include "fvhd_common.h"

template <int K, int S, int MULT, bool ACT, int OWT>
__global__ __launch_bounds__(256) void dwconv_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ bias, int B, int H, int W, int Cin, int OH, int OW, int CS, int nblk_x)
{
    constexpr int PAD = K / 2;
    constexpr int NIN = (OWT - 1) * S + K;      // input columns touched by one strip
    constexpr int CI = 8 / MULT;                // input channels per lane
    extern __shared__ __attribute__((aligned(16))) float lds_w[];   // [K*K][CS]
    const int Cout = Cin * MULT;
    const int slice = blockIdx.y;

    for (int i = threadIdx.x; i < K * K * CS / 4; i += 256) {
        const int e = i * 4, tap = e / CS, c = e - tap * CS;
        *(f32x4*)&lds_w[e] = *(const f32x4*)&w[(size_t)tap * Cout + slice * CS + c];
    }
    __syncthreads();

    const int LPP = CS >> 3;                    // lanes per pixel
    const int SPB = 256 / LPP;                  // strips per block
    const int tl = threadIdx.x;
    if (tl >= SPB * LPP) return;
    const int cgl = tl % LPP, sl = tl / LPP;
    const int SX = (OW + OWT - 1) / OWT;
    const long s = (long)xcd_remap(blockIdx.x, nblk_x) * SPB + sl;
    if (s >= (long)B * OH * SX) return;
    const int sx = (int)(s % SX);
    const int oy = (int)((s / SX) % OH);
    const int b = (int)(s / ((long)SX * OH));
    const int ox0 = sx * OWT;
    const int oc0 = slice * CS + cgl * 8;
    const int ic0 = oc0 / MULT;

    float acc[OWT][8];
#pragma unroll
    for (int o = 0; o < OWT; ++o)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[o][c] = bias ? bias[oc0 + c] : 0.0f;

#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S + ky - PAD;
        if (iy < 0 || iy >= H) continue;
        float wr[K][8];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w0 = *(const f32x4*)&lds_w[(ky * K + kx) * CS + cgl * 8];
            const f32x4 w1 = *(const f32x4*)&lds_w[(ky * K + kx) * CS + cgl * 8 + 4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { wr[kx][c] = w0[c]; wr[kx][4 + c] = w1[c]; }
        }
        const bf16* row = x + ((size_t)(b * H + iy) * W) * Cin + ic0;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int ix = ox0 * S - PAD + j;
            float v[CI];
            if (ix >= 0 && ix < W) {
                if constexpr (CI == 8) {
                    const f32x8 t = bf8_to_f32(*(const bf16x8*)(row + (size_t)ix * Cin));
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = t[c];
                } else {
                    const f32x4 t = bf4_to_f32(*(const bf16x4*)(row + (size_t)ix * Cin));
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = t[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CI; ++c) v[c] = 0.0f;
            }
#pragma unroll
            for (int o = 0; o < OWT; ++o) {
                const int kx = j - o * S;
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[o][c] = __builtin_fmaf(wr[kx][c], v[c / MULT], acc[o][c]);
                }
            }
        }
    }

#pragma unroll
    for (int o = 0; o < OWT; ++o) {
        const int ox = ox0 + o;
        if (ox >= OW) break;
        f32x8 r;
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = ACT ? gelu_erf(acc[o][c]) : acc[o][c];
        *(bf16x8*)(y + ((size_t)(b * OH + oy) * OW + ox) * Cout + oc0) = f32_to_bf8(r);
    }
}

template <int K, int S, int MULT, bool ACT>
static hipError_t launch_dw(hipStream_t st, const bf16* x, bf16* y, const float* w, const float* bias,
                            int B, int H, int W, int Cin)
{
    constexpr int OWT = 4;
    const int Cout = Cin * MULT;
    const int OH = (H + 2 * (K / 2) - K) / S + 1, OW = (W + 2 * (K / 2) - K) / S + 1;
    const int CS = (Cout % 64 == 0) ? 64 : 96;
    if (Cout % CS != 0) return hipErrorInvalidValue;
    const int SPB = 256 / (CS / 8);
    const int SX = (OW + OWT - 1) / OWT;
    const long strips = (long)B * OH * SX;
    const int gx = (int)((strips + SPB - 1) / SPB);
    dim3 grid(gx, Cout / CS);
    const size_t shmem = (size_t)K * K * CS * sizeof(float);
    hipLaunchKernelGGL((dwconv_kernel<K, S, MULT, ACT, OWT>), grid, dim3(256), shmem, st,
                       x, y, w, bias, B, H, W, Cin, OH, OW, CS, gx);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// LDS-tiled stride-1 depthwise conv (K = 3 or 7, multiplier 1): the hot variant - 38 dw3x3 + 46 dw7x7
// launches per forward.  A workgroup owns a TH x 16 output tile of one CS-channel slice:
//   * the (TH+K-1) x (16+K-1) input tile (zero padded at the image border) is staged ONCE into LDS with
//     16-B-per-lane coalesced loads; the K*K re-reads per output then hit LDS (256 B/clk/CU) instead of
//     the vector L1 (64 B/clk/CU), which is what bound the direct-load kernel (708 GB/s on dw7x7);
//   * the LDS row stride is an ODD number of pixels so that vertically adjacent strips of a wave fall in
//     different 128-B bank halves;
//   * each lane computes 8 channels x a strip of 8 output pixels, re-using every LDS vector for up to K
//     outputs and every tap (fp32, read from LDS once per row) for 8 outputs;
//   * tiles are ordered (slice fastest, then x, y, image) inside one 1-D grid and XCD-remapped, so the
//     CS-channel slices of a pixel (same 128-B lines when C = 96) and the halo-sharing neighbours run on
//     the same XCD / L2.
template <int K, int CS>
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
    int B, int H, int W, int C, int tiles_x, int tiles_y, int nslices, int nwg)
{
    constexpr int PAD = K / 2;
    constexpr int LPP = CS / 8;                 // lanes per pixel
    constexpr int NSTRIP = 256 / LPP;           // strips per workgroup
    constexpr int TW = 16, OWT = 8;
    constexpr int TH = NSTRIP / (TW / OWT);     // 16 (CS=64) or 32 (CS=32)
    constexpr int IW = TW + 2 * PAD, IH = TH + 2 * PAD;
    constexpr int IWP = (IW % 2 == 0) ? IW + 1 : IW;   // odd row stride (pixels)
    constexpr int NIN = OWT + K - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tile = (bf16*)smem;                                   // [IH][IWP][CS]
    float* lw = (float*)(smem + (size_t)IH * IWP * CS * 2);     // [K*K][CS]

    const int L = xcd_remap(blockIdx.x, nwg);
    const int slice = L % nslices;
    int t = L / nslices;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int c0 = slice * CS;
    const int tid = threadIdx.x;

    for (int i = tid; i < K * K * CS / 4; i += 256) {
        const int e = i * 4, tap = e / CS, c = e - tap * CS;
        *(f32x4*)&lw[e] = *(const f32x4*)&w[(size_t)tap * C + c0 + c];
    }
    const int gy0 = ty * TH - PAD, gx0 = tx * TW - PAD;
    for (int i = tid; i < IH * IW * LPP; i += 256) {
        const int cgi = i % LPP, p = i / LPP;
        const int iy = p / IW, ix = p - iy * IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *(const u32x4*)(x + ((size_t)(b * H + gy) * W + gx) * C + c0 + cgi * 8);
        *(u32x4*)(tile + ((size_t)iy * IWP + ix) * CS + cgi * 8) = v;
    }
    __syncthreads();

    const int cg = tid % LPP, strip = tid / LPP;
    const int r = strip / (TW / OWT), xh = strip % (TW / OWT);
    const int oy = ty * TH + r, ox0 = tx * TW + xh * OWT;
    if (oy >= H || ox0 >= W) return;

    float acc[OWT][8];
#pragma unroll
    for (int o = 0; o < OWT; ++o)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[o][c] = bias ? bias[c0 + cg * 8 + c] : 0.0f;

#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {     // not unrolled: keeps the live set at acc(64) + one tap row(56) + one vector
        float wr[K][8];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w0 = *(const f32x4*)&lw[(ky * K + kx) * CS + cg * 8];
            const f32x4 w1 = *(const f32x4*)&lw[(ky * K + kx) * CS + cg * 8 + 4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { wr[kx][c] = w0[c]; wr[kx][4 + c] = w1[c]; }
        }
        const bf16* row = tile + ((size_t)(r + ky) * IWP + xh * OWT) * CS + cg * 8;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const f32x8 v = bf8_to_f32(*(const bf16x8*)(row + j * CS));
#pragma unroll
            for (int o = 0; o < OWT; ++o) {
                const int kx = j - o;
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[o][c] = __builtin_fmaf(wr[kx][c], v[c], acc[o][c]);
                }
            }
        }
    }
    bf16* yo = y + ((size_t)(b * H + oy) * W + ox0) * C + c0 + cg * 8;
#pragma unroll
    for (int o = 0; o < OWT; ++o) {
        if (ox0 + o >= W) break;
        f32x8 rr;
#pragma unroll
        for (int c = 0; c < 8; ++c) rr[c] = acc[o][c];
        *(bf16x8*)(yo + (size_t)o * C) = f32_to_bf8(rr);
    }
}

template <int K, int CS>
static hipError_t launch_dw_tiled(hipStream_t st, const bf16* x, bf16* y, const float* w, const float* bias,
                                  int B, int H, int W, int C)
{
    constexpr int PAD = K / 2, LPP = CS / 8, TW = 16, TH = (256 / LPP) / 2;
    constexpr int IW = TW + 2 * PAD, IH = TH + 2 * PAD, IWP = (IW % 2 == 0) ? IW + 1 : IW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, nslices = C / CS;
    const int nwg = B * tiles_x * tiles_y * nslices;
    const size_t shmem = (size_t)IH * IWP * CS * 2 + (size_t)K * K * CS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_tiled_kernel<K, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv_tiled_kernel<K, CS>), dim3(nwg), dim3(256), shmem, st, x, y, w, bias, B, H, W, C,
                       tiles_x, tiles_y, nslices, nwg);
    return hipGetLastError();
}

// x [B,H,W,Cin] bf16 -> y [B,OH,OW,Cin*mult] bf16; w fp32 [K*K][Cout]; bias fp32 [Cout] or null.
extern "C" int fvhd_launch_dwconv(hipStream_t st, const void* x, void* y, const float* w, const float* bias,
                                  int B, int H, int W, int Cin, int K, int stride, int mult, int gelu)
{
    const bf16* xi = (const bf16*)x;
    bf16* yo = (bf16*)y;
    hipError_t e = hipErrorInvalidValue;
    if (stride == 1 && mult == 1 && !gelu && (K == 3 || K == 7) && Cin % 32 == 0) {
        // hot variants: LDS-tiled kernel; 64-channel slices when they divide C, else 32
        if (Cin % 64 == 0) e = (K == 7) ? launch_dw_tiled<7, 64>(st, xi, yo, w, bias, B, H, W, Cin)
                                        : launch_dw_tiled<3, 64>(st, xi, yo, w, bias, B, H, W, Cin);
        else               e = (K == 7) ? launch_dw_tiled<7, 32>(st, xi, yo, w, bias, B, H, W, Cin)
                                        : launch_dw_tiled<3, 32>(st, xi, yo, w, bias, B, H, W, Cin);
    }
    else if (K == 3 && stride == 1 && mult == 1 && !gelu) e = launch_dw<3, 1, 1, false>(st, xi, yo, w, bias, B, H, W, Cin);
    else if (K == 3 && stride == 2 && mult == 1 && gelu) e = launch_dw<3, 2, 1, true>(st, xi, yo, w, bias, B, H, W, Cin);
    else if (K == 7 && stride == 1 && mult == 1 && !gelu) e = launch_dw<7, 1, 1, false>(st, xi, yo, w, bias, B, H, W, Cin);
    else if (K == 7 && stride == 2 && mult == 2 && gelu) e = launch_dw<7, 2, 2, true>(st, xi, yo, w, bias, B, H, W, Cin);
    else if (K == 3 && stride == 1 && mult == 2 && !gelu) e = launch_dw<3, 1, 2, false>(st, xi, yo, w, bias, B, H, W, Cin);
    return (int)e;
}
