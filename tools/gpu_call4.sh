#!/bin/bash
# round-3 GPU call 4: full GPU suite on the current tree, bench (1024 B=32, 1536 B=16), TTFT, stem / gemm micro-benchmarks
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/c4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c4_pytest.log
timeout 300 python bench.py > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
timeout 300 python bench.py --res 1536 --batch 16 --no-cpu-baseline > gpurun_out/c4_bench_1536.json 2>> gpurun_out/c4_bench.err
timeout 300 python bench.py --ttft --steps 10 --warmup 3 > gpurun_out/c4_ttft_b8.json 2>> gpurun_out/c4_bench.err
timeout 300 python bench.py --ttft --batch 1 --steps 10 --warmup 3 > gpurun_out/c4_ttft_b1.json 2>> gpurun_out/c4_bench.err
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate.so timeout 200 python tools/bench_ops.py stem attn > gpurun_out/c4_ops.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c4_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/c4_smoke.log
grep -E "passed|failed" gpurun_out/c4_pytest.log | tail -2
