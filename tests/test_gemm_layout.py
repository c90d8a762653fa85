"""CPU check of the GEMM W-tile row permutation (ml_fastvlm_amd/csrc/gemm_layout.h, round 5).

The kernels fill LDS row p of a W tile with W row sigma(p) so that, after the "swapped" MFMA (D = Wfrag x Afrag^T: lane (lr, g)
holds D rows 4 g .. 4 g + 3 = what lanes 4 g .. 4 g + 3 fed as the W operand), a lane owns 4 GRP CONSECUTIVE output columns and the
epilogue stores 16 B per lane.  The chain  tile fill (register staging / 8-row LDS-DMA pieces) -> fragment read (LDS row
16 j + lr of the wave's block) -> D layout -> epilogue column  is index arithmetic only, so it is walked here with the SAME
constexpr functions the kernels compile (g++ on the header; no GPU)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <set>
#include "gemm_layout.h"
template <int GRP> int check(int BN)
{
    int bad = 0;
    std::set<int> seen;
    for (int p = 0; p < BN; ++p) {
        const int a = wrow_of_lds_row<GRP>(p);                                  // register-staged kernel (v1)
        const int b = wpiece_row<GRP>(p >> 3) + wpiece_lane_row<GRP>(p & 7);    // LDS-DMA kernels: piece base + the lane's row
        if (a != b) { std::printf("GRP %d p %d: staged %d != piece %d\n", GRP, p, a, b); ++bad; }
        if (a / (16 * GRP) != p / (16 * GRP)) { std::printf("GRP %d p %d leaves its block\n", GRP, p); ++bad; }
        seen.insert(a);
    }
    if ((int)seen.size() != BN || *seen.begin() != 0 || *seen.rbegin() != BN - 1) { std::printf("GRP %d: not a bijection of [0, %d)\n", GRP, BN); ++bad; }
    // a wave's block = 64 columns (4 fragments) starting at LDS row / column nw
    for (int nw = 0; nw < BN; nw += 64)
        for (int jb = 0; jb < 4 / GRP; ++jb)
            for (int jl = 0; jl < GRP; ++jl)
                for (int g = 0; g < 4; ++g)
                    for (int reg = 0; reg < 4; ++reg) {
                        const int j = GRP * jb + jl;
                        const int fed_by_lane = 4 * g + reg;                     // D row 4 g + reg = W-operand row of lane lr = 4 g + reg
                        const int wrow = wrow_of_lds_row<GRP>(nw + 16 * j + fed_by_lane);
                        const int col = nw + epi_col<GRP>(jb, g) + 4 * jl + reg; // what the epilogue stores it as
                        if (wrow != col) { std::printf("GRP %d nw %d j %d g %d reg %d: holds W row %d, stored as column %d\n", GRP, nw, j, g, reg, wrow, col); ++bad; }
                    }
    return bad;
}
int main()
{
    int bad = 0;
    for (int BN : {64, 128, 256}) bad += check<1>(BN) + check<2>(BN) + check<4>(BN);
    static_assert(EpiGrp<4, 0, 2>::value == 2 && EpiGrp<4, 3, 2>::value == 2 && EpiGrp<4, 5, 2>::value == 4, "bf16 rows: pairs; SwiGLU: quads");
    static_assert(EpiGrp<3, 0, 2>::value == 1 && EpiGrp<4, 0, 0>::value == 1 && EpiGrp<4, 1, 1>::value == 1, "96-wide tile, fp32 / f16 outputs: identity");
    std::printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad != 0;
}
'''


def test_w_tile_permutation_matches_epilogue_columns():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.cpp"), os.path.join(d, "layout")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "ml_fastvlm_amd", "csrc"), src, "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
