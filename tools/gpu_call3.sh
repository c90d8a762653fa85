#!/bin/bash
# round-3 GPU call 3: remaining LLM / TTFT / reference tests, GEMM tile variants on every tower + prefill shape, split-K A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_qwen2_prefill.py tests/test_gpu_ttft.py tests/test_gpu_reference.py tests/test_gpu_ops.py -m gpu -q -s > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c3_pytest.log
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate.so timeout 400 python tools/bench_ops.py gemm > gpurun_out/c3_gemm.log 2>&1
timeout 300 python bench.py --ttft --ttft-llm kernels --steps 10 --warmup 3 > gpurun_out/c3_ttft_b8_splitk.json 2>> gpurun_out/c3_ttft.err
FVHD_LLM_SPLITK=0 timeout 300 python bench.py --ttft --ttft-llm kernels --steps 10 --warmup 3 > gpurun_out/c3_ttft_b8_nosplit.json 2>> gpurun_out/c3_ttft.err
timeout 300 python bench.py --ttft --ttft-llm kernels --batch 1 --steps 10 --warmup 3 > gpurun_out/c3_ttft_b1.json 2>> gpurun_out/c3_ttft.err
grep -E "passed|failed" gpurun_out/c3_pytest.log | tail -2
