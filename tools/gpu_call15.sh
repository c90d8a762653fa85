#!/bin/bash
# round-3 call 15: the rebuilt product library on the GEMM tests + smoke; the two-workgroups-per-CU GEMM experiment (debug library, knob 5)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm" 2>&1 | tail -2
BENCH_GEMM_VARIANTS=1,5 FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate.so timeout 200 python tools/bench_ops.py gemm 2>&1 | grep -v amdgpu.ids > gpurun_out/c15_gemm.log; cat gpurun_out/c15_gemm.log
