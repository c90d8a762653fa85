#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'])
for k,v in d['kernels'].items(): print(k, v['ms_per_step'], v['tflops'], v['gbs'])
PY
