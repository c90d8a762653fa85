#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for d in 0 1 2 4; do
  FVHD_DUAL=$d timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DUAL=$d', d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
