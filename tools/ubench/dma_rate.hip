// L2 -> CU load throughput per CU: LDS-DMA (global_load_lds_dwordx4) vs plain global_load_dwordx4 -> VGPR (-> ds_write),
// source = a 3 MB buffer shared by all workgroups (L2-resident), 4 or 8 waves per CU, 1 workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>   // 0: LDS-DMA, 8 pieces in flight per wave; 1: global_load -> VGPR (8 in flight) -> ds_write_b128; 2: global_load only
__global__ void k(const char* src, float* out, int iters, int src_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    unsigned acc = 0;
    size_t off = ((size_t)blockIdx.x * 7919 * 1024) % src_bytes;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t o = (off + (size_t)(q * nw + wave) * 1024 + lane * 16) % src_bytes;
            if (MODE == 0) glds16(src + o, lds0 + (q * nw + wave) * 1024);
            else v[q] = *(const u32x4*)(src + o);
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *(u32x4*)(smem + (q * nw + wave) * 1024 + lane * 16) = v[q];
        }
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q][0];
        }
        off = (off + (size_t)8 * nw * 1024) % src_bytes;
    }
    if (MODE != 2) acc = ((unsigned*)smem)[threadIdx.x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

template <int MODE>
double run(int waves, int iters, const char* src, int src_bytes, float* out)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 160 * 1024, 0, src, out, 10, src_bytes);   // 160 KB LDS: 1 WG / CU
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 160 * 1024, 0, src, out, iters, src_bytes);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 8 * waves * 1024;
    return bytes_per_cu / (ms * 1e-3 * 2.4e9);     // B / nominal clk / CU
}

int main()
{
    const int src_bytes = 3 << 20;
    char* src; float* out;
    (void)hipMalloc(&src, src_bytes); (void)hipMemset(src, 1, src_bytes);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int waves : {4, 8, 16}) {
        printf("%2d waves/CU: LDS-DMA %.1f B/clk/CU | load->VGPR->ds_write %.1f | load->VGPR only %.1f   (2.4 GHz nominal, 3 MB L2-resident source)\n",
               waves, run<0>(waves, 2000, src, src_bytes, out), run<1>(waves, 2000, src, src_bytes, out), run<2>(waves, 2000, src, src_bytes, out));
    }
    return 0;
}
