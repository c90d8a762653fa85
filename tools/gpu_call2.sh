#!/bin/bash
# round-3 GPU call 2: Qwen2 prefill kernels (tests, TTFT in every mode, kernel trace), reference e2e with prefill=True, bench with the FFN-only degree-5 GELU
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_qwen2_prefill.py tests/test_gpu_ttft.py tests/test_gpu_reference.py tests/test_gpu_ops.py -m gpu -x -q -s > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2_pytest.log
for mode in kernels kernels-graph hf-graph; do
  timeout 300 python bench.py --ttft --ttft-llm $mode --steps 10 --warmup 3 > gpurun_out/c2_ttft_b8_$mode.json 2>> gpurun_out/c2_ttft.err
done
timeout 300 python bench.py --ttft --ttft-llm kernels --batch 1 --steps 10 --warmup 3 > gpurun_out/c2_ttft_b1_kernels.json 2>> gpurun_out/c2_ttft.err
timeout 300 python bench.py --ttft --ttft-llm hf-graph --batch 1 --steps 10 --warmup 3 > gpurun_out/c2_ttft_b1_hf-graph.json 2>> gpurun_out/c2_ttft.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $GRAFT_REPO_ROOT/gpurun_out/c2_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --ttft --ttft-llm kernels --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/c2_trace.log 2>&1 )
python tools/rocpd_summary.py $(find gpurun_out/c2_trace -name "*_results.db" | head -1) > gpurun_out/c2_ttft_kernel_trace.md 2>> gpurun_out/c2_ttft.err
rm -rf gpurun_out/c2_trace
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err
timeout 600 python bench.py --ttft --ttft-llm kernels --hidden 3584 --steps 5 --warmup 2 > gpurun_out/c2_ttft_b8_h3584_kernels.json 2>> gpurun_out/c2_ttft.err
tail -3 gpurun_out/c2_pytest.log
