#!/bin/bash
# round-3 call 12: MFMA accumulators in VGPRs (libfvhd_vf.so: attention, llm, stem, gemm) against the default; power of more kernels
mkdir -p gpurun_out
for lib in libfvhd.so libfvhd_vf.so libfvhd.so libfvhd_vf.so; do
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2> /dev/null | tail -1 > gpurun_out/c12_bench_${lib%.so}_$RANDOM.json
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 300 python bench.py --ttft --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c12_ttft_${lib%.so}_$RANDOM.json
done
for lib in libfvhd.so libfvhd_vf.so; do
  echo "=== $lib" >> gpurun_out/c12_power.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/$lib timeout 200 python tools/power_probe.py attn stem gemm dw3 2>&1 | grep -v "amdgpu.ids\|power_probe\]" >> gpurun_out/c12_power.log
done
cat gpurun_out/c12_power.log
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_vf.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_qwen2_prefill.py -m gpu -q -x > gpurun_out/c12_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c12_pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c12_bench_*.json')):
    d = json.load(open(f)); k = d['kernels']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], {n: k[n]['ms_per_step'] for n in ('stem', 'gemm_fc1', 'gemm_fc2', 'gemm_1x1', 'gemm_qkv', 'gemm_proj', 'attention', 'projector')})
for f in sorted(glob.glob('gpurun_out/c12_ttft_*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], d['value'], d.get('config', {}).get('breakdown_ms') or d.get('breakdown_ms'))
PY
