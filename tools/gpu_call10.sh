#!/bin/bash
# round-3 call 10: half-precision GELU + f16 GEMM2 in the fused ConvFFN (C <= 192) against the bf16 form (libfvhd_bf16ffn.so)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py -m gpu -q -x > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c10_pytest.log
for lib in libfvhd.so libfvhd_bf16ffn.so libfvhd.so libfvhd_bf16ffn.so; do
  echo "=== $lib" >> gpurun_out/c10_ops.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python tools/bench_ops.py ffn 2>&1 | grep -v amdgpu.ids >> gpurun_out/c10_ops.log
done
cat gpurun_out/c10_ops.log
for lib in libfvhd.so libfvhd_bf16ffn.so libfvhd.so libfvhd_bf16ffn.so; do
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2> /dev/null | tail -1 > gpurun_out/c10_bench_${lib%.so}_$RANDOM.json
done
timeout 120 python tools/power_probe.py ffn192 ffn96 2>&1 | grep -v "amdgpu.ids\|power_probe\]" > gpurun_out/c10_power.log; cat gpurun_out/c10_power.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c10_bench_*.json')):
    try:
        d = json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), 'ffn', d['kernels']['ffn_fused']['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
