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
// HBM-bound by its bytes, VALU-bound in practice for 7x7 (49 fp32 FMAs + bf16 unpacks per output).  NHWC so that the channel
// axis is contiguous: one lane owns 8 output channels (16 B of bf16).  Two kernels:
//   * dwconv_tiled_kernel (below): LDS-staged input tiles, taps in LDS, strips of 4-8 output pixels per lane - every shape;
//   * dw7_mfma_kernel (dwconv_mfma.hip): the 7x7 stride-1 case on the 16-block 4x4x4 MFMA, taken by fvhd_launch_dwconv
//     wherever the map is at least 24 px wide and the channels come in whole 128-B / 192-B pixels (and the launch fills the chip).
#include "fvhd_common.h"

// ---------------------------------------------------------------------------------------------------
// LDS-tiled depthwise conv - the hot variant (38 dw3x3 + 46 dw7x7 + 4 dw7x7/s2 + the stem's dw3x3/s2 per forward).
// A workgroup owns a TH x TW output tile of one CS-output-channel slice:
//   * the ((TH-1)S+K) x ((TW-1)S+K) input tile (zero padded at the image border) is staged ONCE into LDS.  Every
//     thread first ISSUES all of its 16-B global loads (compile-time trip count, fully unrolled: 13-18 independent
//     loads in flight per lane) and only then writes them to LDS - the round-1 version walked the tile with a
//     load -> ds_write -> next-load loop, i.e. ~15 dependent HBM round trips per tile, and ran at 1.1-1.3 TB/s;
//   * the K*K re-reads per output then hit LDS (256 B/clk/CU) instead of the vector L1 (64 B/clk/CU);
//   * the LDS row stride is an ODD number of pixels so that vertically adjacent strips of a wave fall in different
//     128-B bank halves;
//   * each lane computes 8 output channels x a strip of OWT output pixels, re-using every LDS vector for up to
//     ceil(K/S) outputs and every tap (fp32, read from LDS once per tap row) for OWT outputs;
//   * tiles are ordered (slice fastest, then x, y, image) inside one 1-D grid and XCD-remapped, so the channel slices
//     of a pixel and the halo-sharing neighbours run on the same XCD / L2: the halo re-reads (1.6-1.9x the tile) are
//     L2 hits, HBM sees every input byte about once.
template <int K, int S, int MULT, bool ACT, int CS, bool OW4 = false>
struct DwTile {
    static constexpr int PAD = K / 2;
    static constexpr int CSI = CS / MULT;            // input channels per slice
    static constexpr int LPP = CS / 8;               // lanes per output pixel (8 output channels each)
    static constexpr int LPI = CSI / 8;              // 16-B chunks per input pixel
    static constexpr int NSTRIP = 256 / LPP;         // strips per workgroup
    static constexpr int OWT = (S == 1 && !OW4) ? 8 : 4;   // output pixels per strip
    static constexpr int TW = 2 * OWT, TH = NSTRIP / 2;
    static constexpr int IW = (TW - 1) * S + K, IH = (TH - 1) * S + K;
    static constexpr int IWP = IW | 1;               // odd row stride (pixels)
    static constexpr int NIN = (OWT - 1) * S + K;    // input columns touched by one strip
    static constexpr int CI = 8 / MULT;              // input channels per lane
    static constexpr int NCHUNK = IH * IW * LPI;
    static constexpr int NLD = (NCHUNK + 255) / 256;
    static constexpr size_t TILE_B = (size_t)IH * IWP * CSI * 2;
    static constexpr size_t SHMEM = TILE_B + (size_t)K * K * CS * 4;
};

template <int K, int S, int MULT, bool ACT, int CS, bool OW4 = false, int WPE = 2, int PREF = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void dwconv_tiled_kernel(
    const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
    int B, int H, int W, int Cin, int OH, int OW, int tiles_x, int tiles_y, int nslices, int ntiles, int dbg_mode, unsigned* amax)
{
    // amax (round 5, may be null): max |output| over everything this launch stores, as fp32 bit patterns of non-negative numbers (atomicMax
    // on unsigned = numeric max) in a row of FVHD_AMAX_SLOTS words - the range guard of the half-precision fused ConvFFN reads it (fvhd_api.hip: run_ffn)
    using T = DwTile<K, S, MULT, ACT, CS, OW4>;
    // the reduction is compiled into the stride-1 / multiplier-1 / no-activation instantiations only (RepMixer 3x3, ConvFFN 7x7): the other
    // shapes never get a pointer, and their register budgets (PatchEmbed 7x7 / s2 at 168) have no room to carry it for nothing
    constexpr bool AMAXK = S == 1 && MULT == 1 && !ACT;
    constexpr int PAD = T::PAD, CSI = T::CSI, LPP = T::LPP, LPI = T::LPI, OWT = T::OWT, TW = T::TW, TH = T::TH;
    constexpr int IW = T::IW, IWP = T::IWP, NIN = T::NIN, CI = T::CI, NLD = T::NLD, IH_ = T::IH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tile = (bf16*)smem;                                   // [IH][IWP][CSI] (x2 when PREF == 2)
    float* lw = (float*)(smem + (PREF == 2 ? 2 : 1) * T::TILE_B);   // [K*K][CS]
    const int Cout = Cin * MULT;
    const int tid = threadIdx.x;
    const int G = gridDim.x;                                    // persistent workgroups; G % nslices == 0 (launcher)
    const int L0 = xcd_remap(blockIdx.x, G);
    const int slice = L0 % nslices;                             // constant over this workgroup's tiles (t = L0 + i*G)
    const int oc0 = slice * CS, ic0 = slice * CSI;

    for (int i = tid; i < K * K * CS / 4; i += 256) {           // taps of this slice: staged once per workgroup
        const int e = i * 4, tap = e / CS, c = e - tap * CS;
        *(f32x4*)&lw[e] = *(const f32x4*)&w[(size_t)tap * Cout + oc0 + c];
    }

    // per-thread staging slots (compile-time trip count): LDS destination, packed tile-relative (row, col) and channel
    // offset are tile-independent.  Loads are branch-free: the coordinates are clamped into the image (the address is
    // always valid, 32-bit element offset from a wave-uniform image base) and the result is zeroed when outside - the
    // round-1 version spent ~45 instructions of 64-bit address arithmetic plus two branches per 16-B load.
    // PREF == 2 (LDS-DMA): slot i of a lane is 16-B chunk i*256 + tid of the PADDED LDS image (a wave's 64 chunks are
    // 1 KiB contiguous - what one global_load_lds_dwordx4 writes); pad-column chunks are never read and load anything.
    constexpr bool DMA = PREF == 2;
    constexpr int NSL = DMA ? (IH_ * IWP * LPI + 255) / 256 : NLD;
    int dst[NSL], pyx[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int idx = i * 256 + tid;
        const int cgi = idx % LPI, p = idx / LPI;
        const int RW = DMA ? IWP : IW;
        const int iy = p / RW, ix = p - iy * RW;
        pyx[i] = (iy << 20) | (ix << 8) | cgi;                 // iy < 128, ix < 4096, cgi < 256
        if (DMA) dst[i] = idx < IH_ * IWP * LPI ? (ix < IW ? idx * 16 : -2) : -1;    // -2: pad column, -1: past the image
        else dst[i] = idx < T::NCHUNK ? ((iy * IWP + ix) * CSI + cgi * 8) * 2 : -1;
    }
    u32x4 v[DMA ? 1 : NLD];
    auto issue_loads = [&](int t) {                             // all 16-B loads of tile t, no waits in between
        int q = t / nslices;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int gy0 = ty * TH * S - PAD, gx0 = tx * TW * S - PAD;
        const bf16* xb = x + (size_t)b * H * W * Cin + ic0;     // wave-uniform base; per-lane offsets below fit 32 bits
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int gy = gy0 + (pyx[i] >> 20), gx = gx0 + ((pyx[i] >> 8) & 0xfff);
            const bool ok = dst[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && dbg_mode != 2;
            const unsigned off = ok ? (unsigned)((gy * W + gx) * Cin + (pyx[i] & 0xff) * 8) : 0u;
            const u32x4 ld = *(const u32x4*)(xb + off);
            v[i] = ok ? ld : u32x4{0u, 0u, 0u, 0u};
        }
    };
    // LDS-DMA staging of tile t into the tile buffer at LDS byte address `base`: no registers carry the data, so the
    // next tile's loads fly during the whole tap loop at full occupancy.  The DMA cannot zero-fill: chunks outside the
    // image load a valid dummy address and are zeroed by their owner lane after the DMA has landed (returned bit mask).
    // Address = wave-uniform SGPR image base + 32-bit per-lane byte offset (saddr form: no 64-bit VALU arithmetic);
    // the per-slot part of the offset is tile-independent (soffb), so an interior tile costs one v_add per 16-B chunk.
    const unsigned lds_tile0 = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (tid >> 6) * 1024);
    int soffb[DMA ? NSL : 1];
    unsigned vmask = 0;                                         // bit i: slot i is a real (non-pad, in-tile) chunk
    bool last_live = true;                                      // the last slot may run past the LDS image: lane masked off
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int iy = pyx[i] >> 20, ix = (pyx[i] >> 8) & 0xfff;
            const int ixc = ix < IW ? ix : IW - 1;              // pad column: re-load the last real pixel (never read)
            soffb[i] = dst[i] == -1 ? 0 : ((iy * W + ixc) * Cin + (pyx[i] & 0xff) * 8) * 2;
            vmask |= dst[i] >= 0 ? (1u << i) : 0u;
        }
        last_live = dst[NSL - 1] != -1;
    }
    auto issue_dma = [&](int t, unsigned base) -> unsigned {
        int q = t / nslices;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int gy0 = ty * TH * S - PAD, gx0 = tx * TW * S - PAD;
        const bf16* xb = x + (size_t)b * H * W * Cin + ic0;
        const int t0b = (gy0 * W + gx0) * Cin * 2;              // may be negative; t0b + soffb >= 0 for chunks inside the image
        unsigned zm = 0;
        if (gy0 >= 0 && gx0 >= 0 && gy0 + IH_ <= H && gx0 + IW <= W) {          // interior tile (wave-uniform branch)
#pragma unroll
            for (int i = 0; i < NSL; ++i)
                if (i < NSL - 1 || last_live) glds16_s(xb, (unsigned)(t0b + soffb[i]), base + i * 4096);
        } else {
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int gy = gy0 + (pyx[i] >> 20), gx = gx0 + ((pyx[i] >> 8) & 0xfff);
                const bool ok = ((vmask >> i) & 1) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                if (i < NSL - 1 || last_live) glds16_s(xb, ok ? (unsigned)(t0b + soffb[i]) : 0u, base + i * 4096);
                zm |= (((vmask >> i) & 1) && !ok) ? (1u << i) : 0u;
            }
        }
        return zm;
    };

    const int cg = tid % LPP, strip = tid / LPP;
    const int r = strip >> 1, xh = strip & 1;
    float bacc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) bacc[c] = bias ? bias[oc0 + cg * 8 + c] : 0.0f;

    // PREF 1: the next tile's loads ride in registers across the tap loop (costs NLD*4 VGPRs); PREF 0: they are issued
    // at the top of the tile and the other resident workgroups cover their latency; PREF 2: LDS-DMA into the other of
    // two LDS tile buffers during the tap loop (one barrier per tile)
    unsigned zm_cur = 0, cur = 0;
    float amx = 0.f;
    if (PREF == 1 && L0 < ntiles) issue_loads(L0);
    if (DMA && L0 < ntiles && dbg_mode != 2) zm_cur = issue_dma(L0, lds_tile0);
    for (int t = L0; t < ntiles; t += G) {
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // tile t has landed in buffer `cur`
            if (zm_cur) {                                                       // border tiles only
#pragma unroll
                for (int i = 0; i < NSL; ++i)
                    if ((zm_cur >> i) & 1) *(u32x4*)(smem + cur * T::TILE_B + (i * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
            }
            __syncthreads();                                                    // ... and every wave left buffer cur^1
            zm_cur = 0;
            if (t + G < ntiles && dbg_mode != 2) zm_cur = issue_dma(t + G, lds_tile0 + (cur ^ 1) * (unsigned)T::TILE_B);
            tile = (bf16*)(smem + cur * T::TILE_B);
            cur ^= 1;
        } else {
            if (PREF == 0) issue_loads(t);
            // ---- tile t: registers -> LDS; then the NEXT tile's loads are put in flight before the tap loop, so their
            //      HBM latency (and this tile's stores) overlap the VALU work instead of adding to it
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                if (dst[i] >= 0) *(u32x4*)(smem + dst[i]) = v[i];
            __syncthreads();
            if (PREF == 1 && t + G < ntiles) issue_loads(t + G);
        }

        int q = t / nslices;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int oy = ty * TH + r, ox0 = tx * TW + xh * OWT;
        if (oy < OH && ox0 < OW) {
            float acc[OWT][8];
#pragma unroll
            for (int o = 0; o < OWT; ++o)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[o][c] = bacc[c];
            const bf16* row = tile + ((r * S) * IWP + xh * OWT * S) * CSI + cg * CI;     // advanced by one tile row per ky
            const float* lwp = lw + cg * 8;                                                // ... and by one tap row
#pragma unroll 1
            for (int ky = 0; ky < (dbg_mode == 1 ? 1 : K); ++ky, row += IWP * CSI, lwp += K * CS) {   // not unrolled: keeps the live set at acc + one tap row + one vector
                float wr[K][8];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 w0 = *(const f32x4*)&lwp[kx * CS];
                    const f32x4 w1 = *(const f32x4*)&lwp[kx * CS + 4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { wr[kx][c] = w0[c]; wr[kx][4 + c] = w1[c]; }
                }
#pragma unroll
                for (int j = 0; j < NIN; ++j) {
                    float vv[CI];
                    if constexpr (CI == 8) {
                        const f32x8 tv = bf8_to_f32(*(const bf16x8*)(row + j * CSI));
#pragma unroll
                        for (int c = 0; c < 8; ++c) vv[c] = tv[c];
                    } else {
                        const f32x4 tv = bf4_to_f32(*(const bf16x4*)(row + j * CSI));
#pragma unroll
                        for (int c = 0; c < 4; ++c) vv[c] = tv[c];
                    }
#pragma unroll
                    for (int o = 0; o < OWT; ++o) {
                        const int kx = j - o * S;
                        if (kx >= 0 && kx < K) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) acc[o][c] = __builtin_fmaf(wr[kx][c], vv[c / MULT], acc[o][c]);
                        }
                    }
                }
            }
            bf16* yo = y + ((size_t)(b * OH + oy) * OW + ox0) * Cout + oc0 + cg * 8;
#pragma unroll
            for (int o = 0; o < OWT; ++o) {
                if (ox0 + o >= OW) break;
                f32x8 rr;
#pragma unroll
                for (int c = 0; c < 8; ++c) rr[c] = ACT ? gelu_erf(acc[o][c]) : acc[o][c];
                *(bf16x8*)(yo + (size_t)o * Cout) = f32_to_bf8(rr);
                if (AMAXK && amax) {                            // wave-uniform; v_max3_f32 with |.| modifiers: 4 VALU per 8 outputs
#pragma unroll
                    for (int c = 0; c < 8; c += 2) amx = __builtin_fmaxf(__builtin_fmaxf(amx, __builtin_fabsf(rr[c])), __builtin_fabsf(rr[c + 1]));
                }
            }
        }
        if constexpr (!DMA) __syncthreads();   // every wave is done reading the LDS tile before the next one overwrites it
    }
    if (AMAXK && amax) {                        // (wave-uniform) wave -> workgroup through LDS -> one atomic per workgroup, slot by block id
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = __builtin_fmaxf(amx, __shfl_xor(amx, o, 64));
        __syncthreads();                        // every wave is done with the tile buffers
        float* red = (float*)smem;
        if ((tid & 63) == 0) red[tid >> 6] = amx;
        __syncthreads();
        if (tid == 0) {
            const float m = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
            if (m > 0.f) atomicMax(amax + (blockIdx.x % FVHD_AMAX_SLOTS), __float_as_uint(m));
        }
    }
}

// Experiment knobs: compile-time constants in the shipped library; setters exist only in the debug build
// (`FVHD_FFN_ABLATE=1 python -m ml_fastvlm_amd.build` -> libfvhd_ablate.so, -DFVHD_DEBUG_KNOBS; tools/bench_ops.py uses it for A/B runs).
//   dw7 cfg: 0 = never the matrix-core kernel (VALU kernel with the per-C choice), 5 = the matrix-core kernel also below its
//            small-batch threshold, 1 = default dispatch, 3 = VALU: always 64-channel slices / 8-pixel strips, other = always 32 / 4
//   dw3 cfg: -1 = per-C choice, 0 = register-prefetch tiles (64|32-channel slices, 8-pixel strips), 1/2 = LDS-DMA double buffer, 32/64-channel slices
//   dw mode: 1 = stage + store only (no tap loop), 2 = no staging loads (tap loop on stale LDS)
#ifdef FVHD_DEBUG_KNOBS
static int g_dw7_cfg = 1, g_dw3_cfg = -1, g_dw_mode = 0;
extern "C" void fvhd_debug_set_dw7_cfg(int m) { g_dw7_cfg = m; }
extern "C" void fvhd_debug_set_dw3_cfg(int m) { g_dw3_cfg = m; }
extern "C" void fvhd_debug_set_dw_mode(int m) { g_dw_mode = m; }
#else
static constexpr int g_dw7_cfg = 1, g_dw3_cfg = -1, g_dw_mode = 0;
#endif

template <int K, int S, int MULT, bool ACT, int CS, bool OW4 = false, int WPE = 2, int PREF = 1>
static hipError_t launch_dw_tiled(hipStream_t st, const bf16* x, bf16* y, const float* w, const float* bias,
                                  int B, int H, int W, int Cin, unsigned* amax = nullptr)
{
    using T = DwTile<K, S, MULT, ACT, CS, OW4>;
    const int OH = (H + 2 * T::PAD - K) / S + 1, OW = (W + 2 * T::PAD - K) / S + 1;
    const int tiles_x = (OW + T::TW - 1) / T::TW, tiles_y = (OH + T::TH - 1) / T::TH, nslices = Cin * MULT / CS;
    const int ntiles = B * tiles_x * tiles_y * nslices;
    const size_t SHMEM = T::SHMEM + (PREF == 2 ? T::TILE_B : 0);
    static bool attr_set[64] = {false};      // the attribute is per device: one flag per HIP device of this process
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_tiled_kernel<K, S, MULT, ACT, CS, OW4, WPE, PREF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)SHMEM);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    // persistent workgroups: as many as fit at once (LDS-limited), a multiple of nslices (so that a workgroup's tiles
    // t = L0 + i*G all belong to one channel slice: taps staged once) and of the 8 XCDs
    const int per_cu = (int)(160 * 1024 / SHMEM) > 0 ? (int)(160 * 1024 / SHMEM) : 1;
    int G = 256 * (per_cu > 4 ? 4 : per_cu);
    const int q = nslices % 8 == 0 ? nslices : nslices * 8;      // lcm(nslices, 8) for nslices in {1,2,3,4,6,8,12,16,24,48,...}
    G = (G / q) * q;
    if (G <= 0) G = q;
    if (ntiles <= 3 * G) G = ((ntiles + nslices - 1) / nslices) * nslices;   // few rounds: one tile per workgroup (dynamic balance beats persistence)
    hipLaunchKernelGGL((dwconv_tiled_kernel<K, S, MULT, ACT, CS, OW4, WPE, PREF>), dim3(G), dim3(256), SHMEM, st, x, y, w, bias,
                       B, H, W, Cin, OH, OW, tiles_x, tiles_y, nslices, ntiles, g_dw_mode, amax);
    return hipGetLastError();
}

// x [B,H,W,Cin] bf16 -> y [B,OH,OW,Cin*mult] bf16; w fp32 [K*K][Cout]; bias fp32 [Cout] or null.
extern "C" int fvhd_dw7_mfma_supported(int B, int H, int W, int C, int force);
extern "C" int fvhd_launch_dw7_mfma(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int C, unsigned* amax);
extern "C" int fvhd_dw7s2_mfma_supported(int B, int H, int W, int Cin, int force);
extern "C" int fvhd_launch_dw7s2_mfma(hipStream_t st, const void* x, void* y, const float* w, const float* bias, int B, int H, int W, int Cin, int gelu);

// batch_invariant != 0: the kernel choice may depend on the SHAPE of one image only, never on B (bit-identical rows whatever the
// batch they travel in); 0: the fastest kernel for this B (the VALU dw7x7 below the matrix-core kernel's fill threshold)
// amax (may be null; honoured by the stride-1 7x7 kernels - the ConvFFN's depthwise conv - and by the RepMixer 3x3): see dwconv_tiled_kernel
extern "C" int fvhd_launch_dwconv(hipStream_t st, const void* x, void* y, const float* w, const float* bias,
                                  int B, int H, int W, int Cin, int K, int stride, int mult, int gelu, int flags, unsigned* amax)
{
    // flags: bit 0 = batch_invariant (below); bit 1 = the taps are NOT bf16 numbers and must be applied as they are (fp32): the tower sets it
    // for a depthwise conv whose packed taps differ from their bf16 rounding (an fp32 / fp16 checkpoint, or BatchNorm folded at load time) -
    // honoured where a matrix-core kernel would otherwise be a NEW choice of round 6 (stride 2); the stride-1 7x7 rounds as it always did
    const int batch_invariant = flags & 1;
    const bool exact_taps = (flags & 2) != 0;
    const bf16* xi = (const bf16*)x;
    bf16* yo = (bf16*)y;
    const int Cout = Cin * mult;
    hipError_t e = hipErrorInvalidValue;
    const bool c64 = Cout % 64 == 0, c32 = Cout % 32 == 0;
    // dw7x7 stride 1 (46 launches, VALU-bound): 32-channel slices, 4-pixel strips, 32x8 tiles -> 43 KB LDS and <= 170
    // VGPRs, i.e. 3 workgroups (12 waves) per CU instead of 2: a wave64 VALU instruction issues every ~4 cycles per wave
    // but executes in ~2.3, so the pipe only saturates with >= 3 waves per SIMD (tools/ubench/valu_rate.hip)
    // dw7x7 stride 1 on the matrix cores (dwconv_mfma.hip) wherever the map is at least 24 px wide and the channels
    // come in whole 128-B lines: 109 / 60 us at C = 192 / 384 (B = 32, 1024^2 input) against 246 / 118 for the VALU kernel
    // below, which stays for C = 96, narrow maps and as the comparison path (debug build: fvhd_debug_set_dw7_cfg(0)).
    if (K == 7 && stride == 1 && mult == 1 && !gelu && g_dw7_cfg != 0 && fvhd_dw7_mfma_supported(B, H, W, Cin, batch_invariant || g_dw7_cfg == 5))
        return fvhd_launch_dw7_mfma(st, x, y, w, bias, B, H, W, Cin, amax);
    if (K == 7 && stride == 1 && mult == 1 && !gelu && c32) {
        // measured per channel count (tools/bench_ops.py dw7cfg, B = 32, us): config 4 = 32-channel slices / 4-pixel strips,
        // two LDS tile buffers filled by LDS-DMA during the tap loop, 2 waves per SIMD: 445 / 246 / 118 at C = 96 / 192 / 384
        // (2 = same tile, register staging, 3 waves per SIMD: 485 / 268 / 128).  From C = 768 on there are too few tiles
        // per workgroup for the prefetch to matter: config 2 at 768 (64), 64-channel slices / 8-pixel strips at >= 1536 (34).
        const int vcfg = (g_dw7_cfg == 0 || g_dw7_cfg == 5) ? 1 : g_dw7_cfg;
        const bool wide = c64 && (vcfg == 3 || (vcfg == 1 && Cin >= 1536));
        const bool dma = vcfg == 4 || (vcfg == 1 && Cin <= 384);
        if (dma) return (int)launch_dw_tiled<7, 1, 1, false, 32, true, 2, 2>(st, xi, yo, w, bias, B, H, W, Cin, amax);
        if (wide) return (int)launch_dw_tiled<7, 1, 1, false, 64, false, 2, 0>(st, xi, yo, w, bias, B, H, W, Cin, amax);
        return (int)launch_dw_tiled<7, 1, 1, false, 32, true, 3, 0>(st, xi, yo, w, bias, B, H, W, Cin, amax);
    }
    // RepMixer dw3x3: from C = 192 on the LDS-DMA double-buffered tiles with 64-channel slices (whole 128-B lines per pixel) win
    // over the register-prefetch tiles - 99.7 -> 87.0 / 42.1 -> 37.4 us at C = 192 / 384 (B = 32), 42.0 -> 37.4 / 26.5 -> 21.9 us for
    // the half-batches the tower launches: 4.6-5.4 TB/s; at C = 96 (192-B pixels) the two tie and the register version stays.
    // Same arithmetic order: bit-identical outputs (tools/bench_ops.py dw3cfg, profiles/r02_dw3cfg.log).
    if (K == 3 && stride == 1 && mult == 1 && !gelu && c32 && g_dw3_cfg != 0) {
        const int cfg3 = g_dw3_cfg > 0 ? g_dw3_cfg : (c64 && Cin >= 192 ? 2 : 0);
        if (cfg3 == 2 && c64) return (int)launch_dw_tiled<3, 1, 1, false, 64, true, 3, 2>(st, xi, yo, w, bias, B, H, W, Cin, amax);
        if (cfg3 != 0) return (int)launch_dw_tiled<3, 1, 1, false, 32, true, 3, 2>(st, xi, yo, w, bias, B, H, W, Cin, amax);
    }
    if (K == 3 && stride == 1 && mult == 1 && !gelu && c32)
        return (int)(c64 ? launch_dw_tiled<3, 1, 1, false, 64>(st, xi, yo, w, bias, B, H, W, Cin, amax) : launch_dw_tiled<3, 1, 1, false, 32>(st, xi, yo, w, bias, B, H, W, Cin, amax));
#define DW_TILED(KK, SS, MM, AA)                                                                          \
    if (K == KK && stride == SS && mult == MM && (gelu != 0) == AA && c32) {                              \
        e = c64 ? launch_dw_tiled<KK, SS, MM, AA, 64>(st, xi, yo, w, bias, B, H, W, Cin)                  \
                : launch_dw_tiled<KK, SS, MM, AA, 32>(st, xi, yo, w, bias, B, H, W, Cin);                 \
        return (int)e;                                                                                    \
    }
    // PatchEmbed's depthwise 7x7 / stride 2 / multiplier 2 + GELU (mci.py:442-451).  Round 4 (profiles/r04_dwdown_cfg.log, B = 32): the
    // register-prefetch form of rounds 1-3 carried 8 spilled VGPRs in its tap loop at the 256-register limit; without the prefetch (loads
    // at the top of the tile, the other resident workgroups cover them) 309 -> 292 / 163 -> 149 / 42 -> 38 us at C = 96 / 192 / 768, and
    // 32-channel slices at 3 waves per SIMD 78 -> 65 us at C = 384.  Same accumulation order: identical bits.
    // Round 6: the same conv on the matrix cores (dwconv_down.hip: stride 2 as two Toeplitz products over the even and the odd input pixels)
    // wherever the input pixels come in whole 64-B pieces; a choice by shape (and by the taps being bf16 numbers: a re-parameterised bf16
    // checkpoint, mci.py:442-451 - then the kernel's bf16 operands ARE the taps) only - the same bits whatever the batch.
    // FVHD_DWDOWN_MFMA=0 in the environment keeps the VALU kernel below (A/B runs).
    if (K == 7 && stride == 2 && mult == 2 && !exact_taps && fvhd_dw7s2_mfma_supported(B, H, W, Cin, 0))
        return fvhd_launch_dw7s2_mfma(st, x, y, w, bias, B, H, W, Cin, gelu);
    if (K == 7 && stride == 2 && mult == 2 && gelu && c32) {
        if (Cin == 384 || !c64) return (int)launch_dw_tiled<7, 2, 2, true, 32, false, 3, 0>(st, xi, yo, w, bias, B, H, W, Cin);
        return (int)launch_dw_tiled<7, 2, 2, true, 64, false, 2, 0>(st, xi, yo, w, bias, B, H, W, Cin);
    }
    DW_TILED(3, 1, 1, false) DW_TILED(3, 2, 1, true) DW_TILED(3, 1, 2, false)
#undef DW_TILED
    return (int)e;
}
