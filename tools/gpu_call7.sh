#!/bin/bash
# round-3 call 7: half-precision GELU microbenchmark, the new smoke() profile, the prefill tests after the rope-table change
mkdir -p gpurun_out
./tools/ubench/f16_rate > gpurun_out/c7_f16_rate.log 2>&1; echo "f16_rate rc=$?"
tail -40 gpurun_out/c7_f16_rate.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c7_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c7_smoke.log
timeout 900 python -m pytest tests/test_qwen2_prefill.py tests/test_gpu_ttft.py -m gpu -q -x > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c7_pytest.log
