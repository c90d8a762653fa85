#!/bin/bash
# round-3 call 14: softmax denominators on the matrix cores (ones fragment) in both attention kernels
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_qwen2_prefill.py tests/test_gpu_ttft.py tests/test_gpu_steps.py -m gpu -q -x > gpurun_out/c14_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c14_pytest.log
timeout 300 python tools/bench_ops.py attn 2>&1 | grep -v amdgpu.ids > gpurun_out/c14_ops.log; cat gpurun_out/c14_ops.log
timeout 100 python tools/power_probe.py attn 2>&1 | grep -v "amdgpu.ids\|power_probe\]" >> gpurun_out/c14_ops.log; tail -1 gpurun_out/c14_ops.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2> /dev/null | tail -1 > gpurun_out/c14_bench_$i.json; done
timeout 300 python bench.py --ttft --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c14_ttft_b8.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c14_bench_*.json')):
    d = json.load(open(f)); k = d['kernels']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['attention_block']['frac'], {n: k[n]['ms_per_step'] for n in ('attention', 'ffn_fused')})
d = json.load(open('gpurun_out/c14_ttft_b8.json')); print('ttft', d['value'])
PY
