#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem" > gpurun_out/pytest_stem.log 2>&1
tail -6 gpurun_out/pytest_stem.log
timeout 60 python tools/bench_ops.py stem 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_stem.log
