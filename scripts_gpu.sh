#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench.err
cat gpurun_out/bench_full.json | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d.get('cpu_baseline'))
for k,v in d['kernels'].items(): print(k, v['ms_per_step'], v['tflops'], v['gbs'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_bench.log 2>&1
FVHD_DUAL=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c_single -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
