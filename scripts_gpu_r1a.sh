#!/bin/bash
# round-1 GPU pass A: parity tests, bench, rocprof kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1a -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -30
cat gpurun_out/pytest_gpu.log | tail -15
