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

// One wave = 64 consecutive output pixels (along x) x 24 output channels; the 4 waves of a workgroup
// take the 4 channel quarters, so every LDS weight read is a whole-wave broadcast.
// Weights: fp32 [27][96] with k = ci*9 + ky*3 + kx.
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const T* __restrict__ img, bf16* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int B, int R)
{
    constexpr int CO = 96, CQ = 24;
    __shared__ __attribute__((aligned(16))) float lw[27 * CO];
    for (int i = threadIdx.x; i < 27 * CO; i += 256) lw[i] = w[i];
    __syncthreads();
    const int OH = R / 2, OW = R / 2;
    const int lane = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const long pix = (long)blockIdx.x * 64 + lane;
    if (pix >= (long)B * OH * OW) return;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));

    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
                in[ci * 9 + ky * 3 + kx] = (iy >= 0 && iy < R && ix >= 0 && ix < R)
                    ? ld_as_f32<T>(img, (((size_t)b * 3 + ci) * R + iy) * R + ix) : 0.0f;
            }
    float acc[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) acc[c] = bias[cq * CQ + c];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
#pragma unroll
        for (int c4 = 0; c4 < CQ / 4; ++c4) {
            const f32x4 wv = *(const f32x4*)&lw[k * CO + cq * CQ + c4 * 4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c4 * 4 + c] = __builtin_fmaf(wv[c], in[k], acc[c4 * 4 + c]);
        }
    }
    bf16* o = out + (size_t)pix * CO + cq * CQ;
#pragma unroll
    for (int c8 = 0; c8 < CQ / 8; ++c8) {
        f32x8 r;
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = gelu_erf(acc[c8 * 8 + c]);
        *(bf16x8*)(o + c8 * 8) = f32_to_bf8(r);
    }
}

// img [B,3,R,R] (dtype) -> out [B,R/2,R/2,96] bf16
extern "C" int fvhd_launch_stem_conv(hipStream_t st, const void* img, int dtype, void* out, const float* w,
                                     const float* bias, int B, int R)
{
    if (R % 2) return (int)hipErrorInvalidValue;
    const long npix = (long)B * (R / 2) * (R / 2);
    dim3 grid((unsigned)((npix + 63) / 64)), block(256);
    if (dtype == FVHD_F32) hipLaunchKernelGGL(stem_conv_kernel<float>, grid, block, 0, st, (const float*)img, (bf16*)out, w, bias, B, R);
    else if (dtype == FVHD_F16) hipLaunchKernelGGL(stem_conv_kernel<_Float16>, grid, block, 0, st, (const _Float16*)img, (bf16*)out, w, bias, B, R);
    else if (dtype == FVHD_BF16) hipLaunchKernelGGL(stem_conv_kernel<bf16>, grid, block, 0, st, (const bf16*)img, (bf16*)out, w, bias, B, R);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

// ---- SE: global average pool over the T tokens of each image -------------------------------------
// y [B, T, C] bf16 -> pooled [B, C] fp32.  grid (C/256, B), one thread per channel.
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16* __restrict__ y, float* __restrict__ pooled, int T, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    const bf16* p = y + (size_t)b * T * C + c;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += (float)p[(size_t)t * C];
    pooled[(size_t)b * C + c] = s / (float)T;
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
    hipLaunchKernelGGL(se_pool_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, (const bf16*)y, pooled, T, C);
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
