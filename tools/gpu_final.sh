#!/bin/bash
# round-3 evidence on the final binary: full GPU suite, smoke, bench lines (1024 B=32, 1536 B=16, H=3584), TTFT, kernel trace + 5 PMC passes
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 300 python bench.py > gpurun_out/${TAG}_final_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --res 1536 --batch 16 --no-cpu-baseline > gpurun_out/${TAG}_bench_1536.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --hidden 3584 --no-cpu-baseline > gpurun_out/${TAG}_bench_h3584.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --ttft --steps 10 --warmup 3 > gpurun_out/${TAG}_ttft_b8.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --ttft --batch 1 --steps 10 --warmup 3 > gpurun_out/${TAG}_ttft_b1.json 2>> gpurun_out/${TAG}_bench.err
timeout 200 python bench.py --batch 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_b8.json 2>> gpurun_out/${TAG}_bench.err
timeout 200 python bench.py --batch 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_b1.json 2>> gpurun_out/${TAG}_bench.err
timeout 200 python tools/power_probe.py idle ffn384 ffn192 ffn96 gemm dw7 dw3 attn stem 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_power_probe.log
bash tools/run_pmc.sh ${TAG} > gpurun_out/${TAG}_run_pmc.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_ttrace -o trace -- python $GRAFT_REPO_ROOT/bench.py --ttft --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_ttrace.log 2>&1 )
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_ttrace -name "*_results.db" | head -1) > gpurun_out/${TAG}_ttft_kernel_trace.md 2>> gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/${TAG}_ttrace
grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2; tail -2 gpurun_out/${TAG}_smoke.log
