// LDS instruction costs behind the depthwise-MFMA transposition (round 2): how many CU cycles does a wave64 ds_write_b16 /
// ds_write_b32 / ds_write_b64 / ds_read_b128 / ds_read2_b64 take with the kernel's address patterns, 8 waves per CU all issuing?
//   hipcc -O3 --offload-arch=gfx950 -o lds_b16 lds_b16.hip && ./lds_b16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int P = 80;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int iters, long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* base = smem + wv * 8192;
    unsigned v = threadIdx.x * 2654435761u;
    u32x4 accv = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // transposing ds_write_b16: [ch][px] pitch P, lane = (px, half), 8 writes
            u16* d = (u16*)base + (8 * (lane & 1)) * P + (lane >> 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e * P] = (u16)(v + e);
        } else if constexpr (MODE == 1) {   // same count, lanes = consecutive px of one channel row (no half split)
            u16* d = (u16*)base + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e * P] = (u16)(v + e);
        } else if constexpr (MODE == 2) {   // ds_write_b32, lanes consecutive dwords
            unsigned* d = (unsigned*)base + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e * 80] = v + e;
        } else if constexpr (MODE == 3) {   // ds_write_b64, lanes consecutive
            u32x2* d = (u32x2*)base + lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e * 80] = u32x2{v + e, v};
        } else if constexpr (MODE == 4) {   // output staging ds_write_b16: [px][64 ch], lane = (blk, q): q*128 + blk*2
            u16* d = (u16*)(smem + (lane & 3) * 128 + wv * 32 + (lane >> 2) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e * 4 * 64] = (u16)(v + e);
        } else if constexpr (MODE == 5) {   // raw-row read: ds_read_b128 at px*128 + wv*32 + half*16 (the v4 transposition source)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const u32x4 r = *(const u32x4*)(smem + (e * 32 + (lane >> 1)) * 128 + wv * 32 + (lane & 1) * 16); accv += r; }
        } else if constexpr (MODE == 6) {   // lane-linear ds_read_b128
#pragma unroll
            for (int e = 0; e < 4; ++e) { const u32x4 r = *(const u32x4*)(base + e * 1024 + lane * 16); accv += r; }
        } else if constexpr (MODE == 7) {   // A-operand ds_read_b64: blk*P*2 + 8*q + imm
#pragma unroll
            for (int e = 0; e < 8; ++e) { const u32x2 r = *(const u32x2*)(base + (lane >> 2) * P * 2 + (lane & 3) * 8 + e * 32); accv.x += r.x; accv.y += r.y; }
        }
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = accv.x + accv.y + accv.z + accv.w + *(unsigned*)(base + lane * 4);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char* what, int per_iter)
{
    unsigned* out; long long* cyc;
    CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 4000;
    k<MODE><<<512, 256, 65536>>>(out, 10, cyc);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<MODE><<<512, 256, 65536>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // 8 waves per CU issue concurrently: CU cycles per wave-instruction = wall cycles / (instructions per wave * 8)
    printf("%-70s %6.1f ticks per instr per wave (8 waves/CU) -> %5.1f CU-ticks per wave-instruction; wall %.3f ms\n", what,
           (double)c / ((double)iters * per_iter), (double)c / ((double)iters * per_iter * 8), ms);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("ds_write_b16 transposing (px, half) -> [ch][px]", 8);
    run<1>("ds_write_b16 64 consecutive px of one channel", 8);
    run<2>("ds_write_b32 consecutive", 8);
    run<3>("ds_write_b64 consecutive", 8);
    run<4>("ds_write_b16 output staging [px][64ch]", 8);
    run<5>("ds_read_b128 raw rows, 128-B pixel stride", 4);
    run<6>("ds_read_b128 lane-linear", 4);
    run<7>("ds_read_b64 A operand pattern", 8);
    return 0;
}
