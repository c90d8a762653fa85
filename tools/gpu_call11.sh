#!/bin/bash
# round-3 call 11: tests on the half-precision FFN form; the same form at C = 384 (libfvhd_f16all.so) against the default
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py tests/test_gpu_tower.py -m gpu -q > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/c11_pytest.log
for lib in libfvhd.so libfvhd_f16all.so libfvhd.so libfvhd_f16all.so; do
  echo "=== $lib" >> gpurun_out/c11_ops.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python tools/bench_ops.py ffn 2>&1 | grep "C= 384" >> gpurun_out/c11_ops.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 120 python tools/power_probe.py ffn384 2>&1 | grep "ffn384" >> gpurun_out/c11_ops.log
done
cat gpurun_out/c11_ops.log
