#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
( timeout 340 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log )
tail -4 gpurun_out/pytest_gpu_final.log
timeout 100 python bench.py > gpurun_out/bench_final.log 2>&1
tail -1 gpurun_out/bench_final.log | cut -c1-400
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o r01d -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_final.log 2>&1
cd $R
ls gpurun_out/prof_final | head
timeout 60 python bench.py --res 1536 --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_1536_bf16.log 2>&1
timeout 60 python bench.py --res 1536 --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --attn-fp8 > gpurun_out/bench_1536_fp8.log 2>&1
tail -1 gpurun_out/bench_1536_bf16.log | cut -c1-200; tail -1 gpurun_out/bench_1536_fp8.log | cut -c1-200
