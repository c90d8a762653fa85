// Which W row an LDS row of a GEMM W tile holds, and which output columns a lane therefore ends up with (gemm.hip, round 5).
// Plain constexpr functions (host + device under hip-clang) so that tests/test_gemm_layout.py can compile them with g++ and walk the
// whole chain  tile fill -> fragment read -> MFMA D layout -> epilogue column  on the CPU.
#pragma once

#define FVHD_EPI_SWIGLU_ID 5
#define FVHD_ODT_BF16_ID 2

// -DFVHD_GEMM_GRP1 (experiments: FVHD_VARIANT_TAG=g1 FVHD_EXTRA_DEFS=-DFVHD_GEMM_GRP1 python -m ml_fastvlm_amd.build -> libfvhd_g1.so): the
// round-4 layout everywhere (identity permutation, 8-B / 4-B stores) for same-box A/B runs and a bit-for-bit comparison of the outputs
template <int NF, int EPI, int ODT> struct EpiGrp {
#ifdef FVHD_GEMM_GRP1
    static constexpr int value = 1;
#else
    static constexpr int value = (ODT != FVHD_ODT_BF16_ID || (NF % 2)) ? 1 : (EPI == FVHD_EPI_SWIGLU_ID && NF % 4 == 0) ? 4 : 2;
#endif
};
template <int GRP> constexpr int wrow_of_lds_row(int p)
{
    if constexpr (GRP == 1) return p;
    else return (p / (16 * GRP)) * (16 * GRP) + ((p >> 2) & 3) * (4 * GRP) + ((p >> 4) % GRP) * 4 + (p & 3);
}
// the same for the LDS-DMA pieces (8 LDS rows per 1-KiB piece): first W row of piece pi, and the W row offset of the lane's row rip = 0..7
template <int GRP> constexpr int wpiece_row(int pi) { return (pi / (2 * GRP)) * (16 * GRP) + (pi & 1) * (8 * GRP) + ((pi >> 1) % GRP) * 4; }
template <int GRP> constexpr int wpiece_lane_row(int rip) { return (rip >> 2) * (4 * GRP) + (rip & 3); }
// first output column (relative to the wave's block) of the 4 GRP consecutive columns lane group g holds in fragment group jb
template <int GRP> constexpr int epi_col(int jb, int g) { return jb * 16 * GRP + g * 4 * GRP; }
