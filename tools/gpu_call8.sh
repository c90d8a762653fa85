#!/bin/bash
# round-3 call 8: A/B of the library built without packed fp32 VALU in the MFMA kernels (libfvhd_nopk.so) against the default
mkdir -p gpurun_out
for lib in libfvhd.so libfvhd_nopk.so libfvhd.so libfvhd_nopk.so; do
  echo "=== $lib" >> gpurun_out/c8_ops.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python tools/bench_ops.py ffn attn 2>&1 | grep -v amdgpu.ids >> gpurun_out/c8_ops.log
done
for lib in libfvhd.so libfvhd_nopk.so libfvhd.so libfvhd_nopk.so; do
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2> /dev/null | tail -1 > gpurun_out/c8_bench_${lib%.so}_$RANDOM.json
done
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_nopk.so timeout 300 python bench.py --ttft --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c8_ttft_nopk.json
timeout 300 python bench.py --ttft --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c8_ttft_pk.json
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_nopk.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py -m gpu -q -x > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c8_pytest.log
cat gpurun_out/c8_ops.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c8_bench_*.json')) + sorted(glob.glob('gpurun_out/c8_ttft_*.json')):
    try:
        d = json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), {k: v.get('ms_per_step') for k, v in d.get('kernels', {}).items()} if isinstance(d.get('kernels'), dict) else '')
    except Exception as e: print(f, 'ERR', e)
PY
