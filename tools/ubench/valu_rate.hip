// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction for v_fma_f32, v_pk_fma_f32, v_dot2c_f32_bf16,
// bf16->f32 unpack (lshl/and), v_perm_b32, at 1, 2, 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[16];
    f32x2 p[8];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; u[i] = __float_as_uint(a[i]); }
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float w = seed * 0.5f;
    const f32x2 w2 = {w, w + 1.f};
    const bf16x2 bw = {(__bf16)w, (__bf16)(w + 1.f)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], w, 1.0f);                                    // 16 independent chains
                if (MODE == 1 && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
                if (MODE == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(bw));
                if (MODE == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(0x07060302u));
                if (MODE == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 7) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (MODE == 8) asm volatile("v_cvt_f32_bf16 %0, %0" : "+v"(u[i]));
                if (MODE == 9) asm volatile("v_cvt_f32_bf16_sdwa %0, %0 src0_sel:WORD_1" : "+v"(u[i]));
                if (MODE == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (MODE == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(u[i]) : "v"(a[i]));
                if (MODE == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (MODE == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (MODE == 14) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(u[i]) : : "a0");
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + __uint_as_float(u[i]);
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks_per_cu, int iters)
{
    float* out;
    int grid = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 256>>>(out, 10, 1.0f);
    hipEventRecord(a);
    k<MODE><<<grid, 256>>>(out, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    const double n = (double)iters * 4 * (MODE == 1 ? 8 : 16);     // instructions per wave
    const double waves_per_simd = blocks_per_cu;                     // 256-thread blocks: 1 wave per SIMD each
    // cycles per instruction per SIMD (all waves of the SIMD together), assuming 2.4 GHz
    return ms * 1e-3 * 2.4e9 / (n * waves_per_simd);
}

int main()
{
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_dot2c_f32_bf16", "v_perm_b32", "v_lshlrev_b32", "v_mul_f32", "v_exp_f32",
                           "v_and_b32", "v_cvt_f32_bf16", "v_cvt_f32_bf16_sdwa", "v_max_f32", "v_cvt_pk_bf16_f32", "v_add_u32", "v_mov_b32",
                           "accvgpr_write+read(x2)"};
    const int iters = 20000;
    for (int occ : {1, 2, 4}) {
        double r[15] = {run<0>(occ, iters), run<1>(occ, iters), run<2>(occ, iters), run<3>(occ, iters), run<4>(occ, iters), run<5>(occ, iters), run<6>(occ, iters),
                        run<7>(occ, iters), run<8>(occ, iters), run<9>(occ, iters), run<10>(occ, iters), run<11>(occ, iters), run<12>(occ, iters), run<13>(occ, iters), run<14>(occ, iters)};
        printf("waves/SIMD %d:", occ);
        for (int m = 0; m < 15; ++m) printf("  %s %.2f", names[m], r[m]);
        printf("   (cycles per wave64 instruction per SIMD @2.4 GHz nominal)\n");
    }
    return 0;
}
