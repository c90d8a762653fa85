#!/bin/bash
# round-3 GPU call 5: ping-pong GEMM (correctness + timing), FFN ablation variants, power / clock probe, tests touched by the fp8 removal
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tower.py -m gpu -q -k "attention or 1536 or profile or golden_h or stem" > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c5_pytest.log
export FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate.so
timeout 500 python tools/bench_ops.py gemm > gpurun_out/c5_gemm.log 2>&1
for v in 0 1 5 3 4 2 6 7; do
  echo "=== FVHD_FFN_VARIANT=$v" >> gpurun_out/c5_ffn_variants.log
  FVHD_FFN_VARIANT=$v timeout 120 python tools/bench_ops.py ffn >> gpurun_out/c5_ffn_variants.log 2>&1
done
timeout 200 python tools/power_probe.py idle ffn384 ffn192 ffn96 gemm dw7 > gpurun_out/c5_power.log 2>&1
for v in 2 7; do FVHD_FFN_VARIANT=$v timeout 100 python tools/power_probe.py ffn384 ffn192 >> gpurun_out/c5_power.log 2>&1; done
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/c5_smi.log 2>&1
grep -E "passed|failed" gpurun_out/c5_pytest.log | tail -2
