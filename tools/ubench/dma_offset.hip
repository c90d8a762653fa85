// Does the instruction offset of global_load_lds_dwordx4 apply to the LDS destination as well as to the global source?
// One wave copies 4 KiB with ONE M0 / address setup and offset:0/1024/2048/3072; prints where the data landed.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds);
    const void* g = (const char*)src + threadIdx.x * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(lds0) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main()
{
    unsigned *src, *out, h[2048], hs[2048];
    for (int i = 0; i < 2048; ++i) hs[i] = i;
    (void)hipMalloc(&src, 8192); (void)hipMalloc(&out, 8192);
    (void)hipMemcpy(src, hs, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    (void)hipMemcpy(h, out, 8192, hipMemcpyDeviceToHost);
    int ok_linear = 1;
    for (int i = 0; i < 1024; ++i) ok_linear &= (h[i] == (unsigned)i);
    printf("LDS[0..4KiB) == src[0..4KiB): %s\n", ok_linear ? "YES (offset applies to both sides)" : "NO");
    for (int p = 0; p < 8; ++p) printf("  LDS dword %4d: %08x   %4d: %08x\n", p * 256, h[p * 256], p * 256 + 255, h[p * 256 + 255]);
    return 0;
}
