#!/bin/bash
# evidence refresh after a kernel change late in the round (no full test suite: the touched tests run in the call that validated the change):
# smoke, bench line with cpu_baseline, TTFT, power probe, kernel trace + 5 PMC passes
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 300 python bench.py > gpurun_out/${TAG}_final_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --res 1536 --batch 16 --no-cpu-baseline > gpurun_out/${TAG}_bench_1536.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --ttft --steps 10 --warmup 3 > gpurun_out/${TAG}_ttft_b8.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --ttft --batch 1 --steps 10 --warmup 3 > gpurun_out/${TAG}_ttft_b1.json 2>> gpurun_out/${TAG}_bench.err
timeout 100 python tools/power_probe.py attn 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_power_probe_attn.log
bash tools/run_pmc.sh ${TAG} > gpurun_out/${TAG}_run_pmc.log 2>&1
tail -2 gpurun_out/${TAG}_smoke.log; tail -c 600 gpurun_out/${TAG}_final_bench.json
