// VGPR-bank sensitivity of v_pk_fma_f32 / v_fma_f32 on gfx950: the same instruction count with operands drawn from the
// same or from different register banks (bank = reg % 4; 64-bit operands are even-aligned, so a pair sits in banks {0,1}
// or {2,3}), at 1-4 waves per SIMD, independent or back-to-back dependent.   hipcc --offload-arch=gfx950 -O3 pkfma_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
#define CLOB "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
// pattern P: 8 pk_fma with accumulators v[32+4i : 33+4i] (class 0) or mixed
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    // initialise v0..v95 through asm so the compiler does not touch them
    asm volatile(
        "v_mov_b32 v0, 1.0\n"
        "v_mov_b32 v1, v0\n v_mov_b32 v2, v0\n v_mov_b32 v3, v0\n v_mov_b32 v4, v0\n v_mov_b32 v5, v0\n v_mov_b32 v6, v0\n v_mov_b32 v7, v0\n"
        "v_mov_b32 v8, v0\n v_mov_b32 v9, v0\n v_mov_b32 v10, v0\n v_mov_b32 v11, v0\n v_mov_b32 v12, v0\n v_mov_b32 v13, v0\n v_mov_b32 v14, v0\n v_mov_b32 v15, v0\n"
        "v_mov_b32 v16, v0\n v_mov_b32 v17, v0\n v_mov_b32 v18, v0\n v_mov_b32 v19, v0\n v_mov_b32 v20, v0\n v_mov_b32 v21, v0\n v_mov_b32 v22, v0\n v_mov_b32 v23, v0\n"
        "v_mov_b32 v24, v0\n v_mov_b32 v25, v0\n v_mov_b32 v26, v0\n v_mov_b32 v27, v0\n v_mov_b32 v28, v0\n v_mov_b32 v29, v0\n v_mov_b32 v30, v0\n v_mov_b32 v31, v0\n"
        "v_mov_b32 v32, v0\n v_mov_b32 v33, v0\n v_mov_b32 v34, v0\n v_mov_b32 v35, v0\n v_mov_b32 v36, v0\n v_mov_b32 v37, v0\n v_mov_b32 v38, v0\n v_mov_b32 v39, v0\n"
        "v_mov_b32 v40, v0\n v_mov_b32 v41, v0\n v_mov_b32 v42, v0\n v_mov_b32 v43, v0\n v_mov_b32 v44, v0\n v_mov_b32 v45, v0\n v_mov_b32 v46, v0\n v_mov_b32 v47, v0\n"
        "v_mov_b32 v48, v0\n v_mov_b32 v49, v0\n v_mov_b32 v50, v0\n v_mov_b32 v51, v0\n v_mov_b32 v52, v0\n v_mov_b32 v53, v0\n v_mov_b32 v54, v0\n v_mov_b32 v55, v0\n"
        "v_mov_b32 v56, v0\n v_mov_b32 v57, v0\n v_mov_b32 v58, v0\n v_mov_b32 v59, v0\n v_mov_b32 v60, v0\n v_mov_b32 v61, v0\n v_mov_b32 v62, v0\n v_mov_b32 v63, v0\n"  ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)        // pk: acc class 0, A class 0, B class 0 (all pairs in banks {0,1})
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[4:5], v[8:9], v[32:33]\n v_pk_fma_f32 v[36:37], v[4:5], v[12:13], v[36:37]\n"
                "v_pk_fma_f32 v[40:41], v[4:5], v[16:17], v[40:41]\n v_pk_fma_f32 v[44:45], v[4:5], v[20:21], v[44:45]\n"
                "v_pk_fma_f32 v[48:49], v[4:5], v[24:25], v[48:49]\n v_pk_fma_f32 v[52:53], v[4:5], v[28:29], v[52:53]\n"
                "v_pk_fma_f32 v[56:57], v[4:5], v[8:9], v[56:57]\n v_pk_fma_f32 v[60:61], v[4:5], v[12:13], v[60:61]\n")  ::: CLOB);
        if (MODE == 1)        // pk: acc class 0, A class 1, B class 0
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[6:7], v[8:9], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[12:13], v[36:37]\n"
                "v_pk_fma_f32 v[40:41], v[6:7], v[16:17], v[40:41]\n v_pk_fma_f32 v[44:45], v[6:7], v[20:21], v[44:45]\n"
                "v_pk_fma_f32 v[48:49], v[6:7], v[24:25], v[48:49]\n v_pk_fma_f32 v[52:53], v[6:7], v[28:29], v[52:53]\n"
                "v_pk_fma_f32 v[56:57], v[6:7], v[8:9], v[56:57]\n v_pk_fma_f32 v[60:61], v[6:7], v[12:13], v[60:61]\n")  ::: CLOB);
        if (MODE == 2)        // pk: acc class 0, A class 1, B class 1
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[6:7], v[10:11], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[14:15], v[36:37]\n"
                "v_pk_fma_f32 v[40:41], v[6:7], v[18:19], v[40:41]\n v_pk_fma_f32 v[44:45], v[6:7], v[22:23], v[44:45]\n"
                "v_pk_fma_f32 v[48:49], v[6:7], v[26:27], v[48:49]\n v_pk_fma_f32 v[52:53], v[6:7], v[30:31], v[52:53]\n"
                "v_pk_fma_f32 v[56:57], v[6:7], v[10:11], v[56:57]\n v_pk_fma_f32 v[60:61], v[6:7], v[14:15], v[60:61]\n")  ::: CLOB);
        if (MODE == 3)        // pk: A varies every instruction too (no operand reuse), classes (0,1,1)
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[2:3], v[10:11], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[14:15], v[36:37]\n"
                "v_pk_fma_f32 v[40:41], v[2:3], v[18:19], v[40:41]\n v_pk_fma_f32 v[44:45], v[6:7], v[22:23], v[44:45]\n"
                "v_pk_fma_f32 v[48:49], v[2:3], v[26:27], v[48:49]\n v_pk_fma_f32 v[52:53], v[6:7], v[30:31], v[52:53]\n"
                "v_pk_fma_f32 v[56:57], v[2:3], v[10:11], v[56:57]\n v_pk_fma_f32 v[60:61], v[6:7], v[14:15], v[60:61]\n")  ::: CLOB);
        if (MODE == 4)        // pk: dependent chain (same accumulator back to back), classes (0,1,1)
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[6:7], v[10:11], v[32:33]\n v_pk_fma_f32 v[32:33], v[6:7], v[14:15], v[32:33]\n"
                "v_pk_fma_f32 v[32:33], v[6:7], v[18:19], v[32:33]\n v_pk_fma_f32 v[32:33], v[6:7], v[22:23], v[32:33]\n"
                "v_pk_fma_f32 v[36:37], v[6:7], v[26:27], v[36:37]\n v_pk_fma_f32 v[36:37], v[6:7], v[30:31], v[36:37]\n"
                "v_pk_fma_f32 v[36:37], v[6:7], v[10:11], v[36:37]\n v_pk_fma_f32 v[36:37], v[6:7], v[14:15], v[36:37]\n")  ::: CLOB);
        if (MODE == 5)        // scalar fma x2 (same flops as one pk): banks acc 0, a 1, b 2 (conflict free)
            asm volatile(REP8(
                "v_fma_f32 v32, v5, v10, v32\n v_fma_f32 v36, v5, v14, v36\n v_fma_f32 v40, v5, v18, v40\n v_fma_f32 v44, v5, v22, v44\n"
                "v_fma_f32 v48, v5, v26, v48\n v_fma_f32 v52, v5, v30, v52\n v_fma_f32 v56, v5, v10, v56\n v_fma_f32 v60, v5, v14, v60\n")  ::: CLOB);
        if (MODE == 6)        // scalar fma: all three operands in bank 0
            asm volatile(REP8(
                "v_fma_f32 v32, v4, v8, v32\n v_fma_f32 v36, v4, v12, v36\n v_fma_f32 v40, v4, v16, v40\n v_fma_f32 v44, v4, v20, v44\n"
                "v_fma_f32 v48, v4, v24, v48\n v_fma_f32 v52, v4, v28, v52\n v_fma_f32 v56, v4, v8, v56\n v_fma_f32 v60, v4, v12, v60\n")  ::: CLOB);
        if (MODE == 7)        // scalar fma: dst != acc (3-address), all different banks
            asm volatile(REP8(
                "v_fma_f32 v33, v5, v10, v32\n v_fma_f32 v37, v5, v14, v36\n v_fma_f32 v41, v5, v18, v40\n v_fma_f32 v45, v5, v22, v44\n"
                "v_fma_f32 v32, v5, v26, v33\n v_fma_f32 v36, v5, v30, v37\n v_fma_f32 v40, v5, v10, v41\n v_fma_f32 v44, v5, v14, v45\n")  ::: CLOB);
        if (MODE == 8)        // v_mul_u32_u24 by 65536 (candidate full-rate low-half bf16 unpack)
            asm volatile(REP8(
                "v_mul_u32_u24 v32, v5, v10\n v_mul_u32_u24 v36, v5, v14\n v_mul_u32_u24 v40, v5, v18\n v_mul_u32_u24 v44, v5, v22\n"
                "v_mul_u32_u24 v48, v5, v26\n v_mul_u32_u24 v52, v5, v30\n v_mul_u32_u24 v56, v5, v10\n v_mul_u32_u24 v60, v5, v14\n")  ::: CLOB);
        if (MODE == 9)        // v_lshlrev (reference, half rate)
            asm volatile(REP8(
                "v_lshlrev_b32 v32, 16, v10\n v_lshlrev_b32 v36, 16, v14\n v_lshlrev_b32 v40, 16, v18\n v_lshlrev_b32 v44, 16, v22\n"
                "v_lshlrev_b32 v48, 16, v26\n v_lshlrev_b32 v52, 16, v30\n v_lshlrev_b32 v56, 16, v10\n v_lshlrev_b32 v60, 16, v14\n")  ::: CLOB);
        if (MODE == 10)       // v_pk_mul_f32 with op_sel trick is not an unpack; v_mul_f32 by literal instead (full rate?)
            asm volatile(REP8(
                "v_mul_f32 v32, 0x47800000, v10\n v_mul_f32 v36, 0x47800000, v14\n v_mul_f32 v40, 0x47800000, v18\n v_mul_f32 v44, 0x47800000, v22\n"
                "v_mul_f32 v48, 0x47800000, v26\n v_mul_f32 v52, 0x47800000, v30\n v_mul_f32 v56, 0x47800000, v10\n v_mul_f32 v60, 0x47800000, v14\n")  ::: CLOB);
        if (MODE == 11)       // pk_fma interleaved with ds_read-free unpack mix like the dw7 loop: 7 pk : 2.5 lshl : 2.5 and
            asm volatile(REP8(
                "v_pk_fma_f32 v[32:33], v[6:7], v[10:11], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[14:15], v[36:37]\n v_lshlrev_b32 v20, 16, v21\n"
                "v_pk_fma_f32 v[40:41], v[6:7], v[18:19], v[40:41]\n v_and_b32 v22, 0xffff0000, v23\n v_pk_fma_f32 v[44:45], v[6:7], v[10:11], v[44:45]\n"
                "v_pk_fma_f32 v[48:49], v[6:7], v[26:27], v[48:49]\n v_lshlrev_b32 v24, 16, v25\n v_pk_fma_f32 v[52:53], v[6:7], v[30:31], v[52:53]\n"
                "v_and_b32 v28, 0xffff0000, v29\n v_pk_fma_f32 v[56:57], v[6:7], v[10:11], v[56:57]\n v_pk_fma_f32 v[60:61], v[6:7], v[14:15], v[60:61]\n")  ::: CLOB);
    }
    float s;
    asm volatile("v_add_f32 %0, v32, v36\n v_add_f32 %0, %0, v40\n v_add_f32 %0, %0, v33" : "=&v"(s) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const int NI[12] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 96};
template <int MODE>
double run(int blocks_per_cu, int iters)
{
    float* out;
    int grid = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 256>>>(out, 10);
    hipEventRecord(a);
    k<MODE><<<grid, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    return ms * 1e-3 * 2.4e9 / ((double)iters * NI[MODE] * blocks_per_cu);
}

int main()
{
    const char* names[] = {"pk(0,0,0)", "pk(0,1,0)", "pk(0,1,1)", "pk(0,1,1) A varies", "pk dependent", "fma banks 0,1,2", "fma all bank 0",
                           "fma dst!=acc", "v_mul_u32_u24", "v_lshlrev", "v_mul_f32 lit", "dw7 mix (8pk+2lshl+2and)"};
    const int iters = 20000;
    for (int occ : {1, 2, 3, 4}) {
        double r[12] = {run<0>(occ, iters), run<1>(occ, iters), run<2>(occ, iters), run<3>(occ, iters), run<4>(occ, iters), run<5>(occ, iters),
                        run<6>(occ, iters), run<7>(occ, iters), run<8>(occ, iters), run<9>(occ, iters), run<10>(occ, iters), run<11>(occ, iters)};
        printf("waves/SIMD %d:", occ);
        for (int m = 0; m < 12; ++m) printf("  [%s] %.2f", names[m], r[m]);
        printf("   (cycles per wave64 instruction per SIMD @2.4 GHz nominal)\n");
    }
    return 0;
}
