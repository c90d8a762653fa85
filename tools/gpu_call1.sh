#!/bin/bash
# round-3 GPU call 1: full GPU test suite, per-step error table, bench line, same-box A/B of the degree-5 GELU
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/c1_dev.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
timeout 300 python -m pytest tests/test_gpu_steps.py -m gpu -q -s -k "r1024 or r256" > gpurun_out/c1_steps_g7.log 2>&1
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate_g5.so timeout 300 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tower.py -m gpu -q -s -k "r1024 or r256 or mild or bench_configuration" > gpurun_out/c1_steps_g5.log 2>&1
timeout 300 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
for lib in ablate ablate_g5 ablate ablate_g5; do
  echo "=== $lib" >> gpurun_out/c1_ops.log
  FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_$lib.so timeout 200 python tools/bench_ops.py ffn stem >> gpurun_out/c1_ops.log 2>&1
done
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate_g5.so timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c1_bench_g5.json 2>> gpurun_out/c1_bench.err
FVHD_LIB=$PWD/ml_fastvlm_amd/libfvhd_ablate.so timeout 200 python bench.py --no-cpu-baseline > gpurun_out/c1_bench_g7.json 2>> gpurun_out/c1_bench.err
tail -3 gpurun_out/c1_pytest.log
