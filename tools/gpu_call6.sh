#!/bin/bash
# round-3 GPU call 6: first-generation stagger of the fused ConvFFN (sustained-loop timing per channel count, then the whole step)
export TMPDIR=/tmp
mkdir -p gpurun_out
for us in 0 6 12 20 30 45 0; do
  echo "=== FVHD_FFN_STAGGER_US=$us" >> gpurun_out/c6_stagger.log
  FVHD_FFN_STAGGER_US=$us timeout 100 python tools/power_probe.py ffn384 ffn192 ffn96 2>&1 | grep -v amdgpu | sed 's/| power.*//' >> gpurun_out/c6_stagger.log
done
for us in 0 12 25 0 40; do
  FVHD_FFN_STAGGER_US=$us timeout 200 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/c6_bench_st$us.json 2>> gpurun_out/c6_bench.err
  python - <<PY >> gpurun_out/c6_stagger.log
import json
d=json.loads([l for l in open("gpurun_out/c6_bench_st$us.json") if l.startswith("{")][-1])
print("bench stagger $us us:", d["value"], "img/s", d["ms_per_step"], "ms  ffn", d["kernels"]["ffn_fused"]["ms_per_step"])
PY
done
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm or ffn" > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c6_pytest.log
cat gpurun_out/c6_stagger.log | tail -12
