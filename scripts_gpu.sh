#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_tower.py -x -q -m gpu -k "hip_graph" > gpurun_out/pytest_graph.log 2>&1
tail -5 gpurun_out/pytest_graph.log
for B in 1 8; do for G in "" "--graph"; do
  timeout 200 python bench.py --batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-roofline $G > gpurun_out/lat_B${B}_g${#G}.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/lat_B${B}_g${#G}.log").read().strip().splitlines()[-1])
print("B=$B graph='$G'", d["ms_per_step"], "ms/step", d["value"], "img/s")
PY
done; done
