// Does a single wave's MFMA stream overlap with its own LDS fragment reads?  (the fused-FFN inner loop in isolation)
//   mode 0: MFMA only;  1: ds_read_b128 only;  2: ds_read_b128 (PF ahead) -> MFMA, fragment is the A operand;
//   3: as 2 but the read fragment is NOT consumed by the MFMA (independent);  4: as 2 with two ds_read_b64 instead of one b128
// Reports cycles per slot (one MFMA and/or one 1-KiB fragment read) per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int PF>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    const char* base = smem + lane * 16;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 bop;
    for (int j = 0; j < 8; ++j) bop[j] = (__bf16)(0.01f * (lane + j));
    bf16x8 wf[48];
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
#pragma unroll
            for (int f = 0; f < PF; ++f) wf[f] = *(const bf16x8*)(base + f * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            if (MODE == 0 || MODE == 3) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bop, bop, acc[m & 3], 0, 0, 0);
            if (MODE == 2 || MODE == 4) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[m], bop, acc[m & 3], 0, 0, 0);
            if (MODE == 1 || MODE == 3) asm volatile("" ::"v"(wf[m]));
            __builtin_amdgcn_sched_barrier(0);
            if (MODE != 0 && m + PF < 48) {
                if (MODE == 4) {
                    const bf16x4 lo = *(const bf16x4*)(base + (m + PF) * 1024), hi = *(const bf16x4*)(base + (m + PF) * 1024 + 8);
                    wf[m + PF] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                } else wf[m + PF] = *(const bf16x8*)(base + (m + PF) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int PF>
double run(int threads, int iters)
{
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipFuncSetAttribute((const void*)k<MODE, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PF>), dim3(256), dim3(threads), 49152, 0, out, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, PF>), dim3(256), dim3(threads), 49152, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(out);
    const double waves_per_simd = threads / 256.0;
    return ms * 1e-3 * 2.4e9 / ((double)iters * 48 * waves_per_simd);
}

int main()
{
    const int it = 4000;
    for (int threads : {256, 512}) {
        printf("%d waves/SIMD, cycles per slot per SIMD (2.4 GHz nominal):  mfma only %.1f | ds_read_b128 only pf3 %.1f pf8 %.1f | read->mfma pf2 %.1f pf3 %.1f pf6 %.1f pf12 %.1f | "
               "independent read+mfma pf3 %.1f | 2x ds_read_b64->mfma pf3 %.1f\n", threads / 256,
               run<0, 3>(threads, it), run<1, 3>(threads, it), run<1, 8>(threads, it), run<2, 2>(threads, it), run<2, 3>(threads, it), run<2, 6>(threads, it),
               run<2, 12>(threads, it), run<3, 3>(threads, it), run<4, 3>(threads, it));
    }
    return 0;
}
