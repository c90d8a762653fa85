// Image preprocessing in front of encode_images() on the GPU: what `process_images` / `expand2square`
// (`llava/mm_utils.py:154-184`) and the tower's `CLIPImageProcessor` (`mobileclip_encoder.py:45-49`) do on the CPU with Pillow -
// paste on a square canvas of the background colour, bicubic resize of the shortest edge to R (antialiased when shrinking),
// centre crop R x R, x / 255 - for one uint8 HWC RGB image already in device memory.
//
// The arithmetic is Pillow's 8-bit separable resampler (Resample.c: fixed-point coefficients with 22 fractional bits, a
// horizontal pass rounded to uint8, then a vertical pass rounded to uint8), so the result is BIT-exact, not close: the host
// (ml_fastvlm_amd/preprocess.py) computes the coefficient tables exactly like `precompute_coeffs` / `normalize_coeffs_8bpc` and
// restricts them to the crop window; the two kernels below are the two integer passes.  Byte / integer work: HBM-bound by the
// source image (a 12-MP photo is 36 MB: ~10 us at HBM speed against ~10 ms for Pillow on a CPU core).
//   pass 1: one lane per (canvas row, output column): sum of <= hk taps along x, 3 channels, -> tmp [rows][R][3] uint8
//   pass 2: one lane per output pixel: sum of <= vk taps down the rows of tmp -> clip8 -> value * (1/255) from a 256-entry
//           float table (the reference multiplies in float64 and rounds to float32: the table holds exactly those floats)
//           -> three planes [3][R][R] of the caller's dtype, x fastest (coalesced stores)
#include "fvhd_common.h"

namespace {

constexpr int PREC = 22;         // Resample.c PRECISION_BITS = 32 - 8 - 2

FVHD_DEV int clip8(int acc)
{
    const int v = acc >> PREC;   // arithmetic shift = the floor Pillow's lookup index takes
    return v < 0 ? 0 : v > 255 ? 255 : v;
}

__global__ __launch_bounds__(256) void pre_hpass_kernel(const unsigned char* __restrict__ src, int src_h, int src_w, long src_pitch,
                                                        int pad_top, int pad_left, unsigned bg, const int* __restrict__ hb,
                                                        const int* __restrict__ hc, int hk, int row0, int nrows, int R,
                                                        unsigned char* __restrict__ tmp)
{
    const int xx = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (xx >= R || j >= nrows) return;
    const int sy = row0 + j - pad_top;                     // source row of this canvas row
    const int x0 = hb[2 * xx], n = hb[2 * xx + 1];
    const int* k = hc + (size_t)xx * hk;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    const int b0 = bg & 255, b1 = (bg >> 8) & 255, b2 = (bg >> 16) & 255;
    const bool row_in = sy >= 0 && sy < src_h;
    const unsigned char* rowp = src + (size_t)(row_in ? sy : 0) * src_pitch;
    for (int x = 0; x < n; ++x) {
        const int sx = x0 + x - pad_left, kv = k[x];
        int p0 = b0, p1 = b1, p2 = b2;
        if (row_in && sx >= 0 && sx < src_w) {
            const unsigned char* p = rowp + 3 * (size_t)sx;
            p0 = p[0]; p1 = p[1]; p2 = p[2];
        }
        a0 += p0 * kv; a1 += p1 * kv; a2 += p2 * kv;
    }
    unsigned char* o = tmp + ((size_t)j * R + xx) * 3;
    o[0] = (unsigned char)clip8(a0); o[1] = (unsigned char)clip8(a1); o[2] = (unsigned char)clip8(a2);
}

template <typename T>
__global__ __launch_bounds__(256) void pre_vpass_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ vb,
                                                        const int* __restrict__ vc, int vk, int row0, int R,
                                                        const float* __restrict__ lut, T* __restrict__ out)
{
    const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    if (xx >= R) return;
    const int y0 = vb[2 * yy] - row0, n = vb[2 * yy + 1];
    const int* k = vc + (size_t)yy * vk;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < n; ++y) {
        const unsigned char* p = tmp + ((size_t)(y0 + y) * R + xx) * 3;
        const int kv = k[y];
        a0 += p[0] * kv; a1 += p[1] * kv; a2 += p[2] * kv;
    }
    const size_t plane = (size_t)R * R, o = (size_t)yy * R + xx;
    out[o] = (T)lut[clip8(a0)];
    out[plane + o] = (T)lut[clip8(a1)];
    out[2 * plane + o] = (T)lut[clip8(a2)];
}

}  // namespace

// src: uint8 HWC RGB [src_h][src_pitch bytes]; the image sits at (pad_top, pad_left) of a canvas filled with bg (0xBBGGRR);
// hb / vb: [R][2] (first canvas index, count), hc / vc: [R][hk | vk] fixed-point taps of the R cropped output columns / rows;
// tmp: [nrows][R][3] bytes of scratch for the canvas rows [row0, row0 + nrows) the vertical taps touch; lut: [256] floats;
// out: [3][R][R] of out_dtype (FVHD_F32 / F16 / BF16)
extern "C" int fvhd_launch_preprocess(hipStream_t st, const void* src, int src_h, int src_w, long src_pitch, int pad_top, int pad_left,
                                      unsigned bg, const int* hb, const int* hc, int hk, const int* vb, const int* vc, int vk, int row0,
                                      int nrows, void* tmp, const float* lut, int R, void* out, int out_dtype)
{
    if (R <= 0 || nrows <= 0 || hk <= 0 || vk <= 0 || src_h <= 0 || src_w <= 0) return (int)hipErrorInvalidValue;
    const dim3 g1((R + 255) / 256, nrows), g2((R + 255) / 256, R);
    if (g1.y > 65535u || g2.y > 65535u) return (int)hipErrorInvalidValue;      // grid.y limit: images taller than 65535 rows are not a camera format
    hipLaunchKernelGGL(pre_hpass_kernel, g1, dim3(256), 0, st, (const unsigned char*)src, src_h, src_w, src_pitch, pad_top, pad_left, bg,
                       hb, hc, hk, row0, nrows, R, (unsigned char*)tmp);
    if (out_dtype == FVHD_F32)
        hipLaunchKernelGGL(pre_vpass_kernel<float>, g2, dim3(256), 0, st, (const unsigned char*)tmp, vb, vc, vk, row0, R, lut, (float*)out);
    else if (out_dtype == FVHD_F16)
        hipLaunchKernelGGL(pre_vpass_kernel<_Float16>, g2, dim3(256), 0, st, (const unsigned char*)tmp, vb, vc, vk, row0, R, lut, (_Float16*)out);
    else if (out_dtype == FVHD_BF16)
        hipLaunchKernelGGL(pre_vpass_kernel<__bf16>, g2, dim3(256), 0, st, (const unsigned char*)tmp, vb, vc, vk, row0, R, lut, (__bf16*)out);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
