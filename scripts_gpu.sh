#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k gemm 2>&1 | tail -5
timeout 300 python tools/bench_ops.py gemm 2>&1 | grep -v amdgpu.ids
