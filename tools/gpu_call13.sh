#!/bin/bash
# round-3 call 13: attention kernels with the uniform mask branch / running pointers, VGPR-form accumulators: tests + bench + ttft
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_qwen2_prefill.py tests/test_gpu_ttft.py -m gpu -q -x > gpurun_out/c13_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c13_pytest.log
timeout 300 python tools/bench_ops.py attn 2>&1 | grep -v amdgpu.ids > gpurun_out/c13_ops.log; cat gpurun_out/c13_ops.log
timeout 100 python tools/power_probe.py attn 2>&1 | grep -v "amdgpu.ids\|power_probe\]" >> gpurun_out/c13_ops.log; tail -1 gpurun_out/c13_ops.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 10 2> /dev/null | tail -1 > gpurun_out/c13_bench_$i.json; done
timeout 300 python bench.py --ttft --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c13_ttft_b8.json
timeout 300 python bench.py --ttft --batch 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c13_ttft_b1.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c13_bench_*.json')):
    d = json.load(open(f)); k = d['kernels']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['attention_block']['frac'], d['conv_stage']['frac'], {n: k[n]['ms_per_step'] for n in ('stem', 'attention', 'ffn_fused')})
for f in sorted(glob.glob('gpurun_out/c13_ttft_*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], d['value'], json.dumps(d.get('config'))[:300])
PY
