#!/bin/bash
# rocprofv3 evidence for one round, to be run ON THE GPU BOX (through gpurun) from the repo root:
#     bash tools/run_pmc.sh r02            -> gpurun_out/r02_{trace,fetch,write,sq,grbm}/*_results.db
# The databases (~75 MB) stay on the box (gpurun merges at most 64 MiB back): they are summarised there into
#     gpurun_out/r02_pmc_summary.{md,json} and gpurun_out/r02_kernel_trace_stats_single_stream.md  -> copy those into profiles/.
# Separate passes per MI355X_MICROARCH.md "rocprofv3 PMC slots" (FETCH_SIZE = 3 TCC slots, WRITE_SIZE = 2: not in one pass);
# --kernel-trace only beside --pmc (gpurun refuses sys/runtime traces with counters).
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttft"
mkdir -p gpurun_out
run() { # name, extra rocprofv3 args...
    local name=$1; shift
    timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_${name} -o ${name} "$@" -- $CMD > gpurun_out/${TAG}_${name}.log 2>&1
    echo "${name}: rc=$? $(ls gpurun_out/${TAG}_${name} 2>/dev/null | head -3 | tr '\n' ' ')"
}
run trace
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
run grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python tools/pmc_summary.py ${TAG} gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_grbm > gpurun_out/${TAG}_pmc_summary.md
cp profiles/${TAG}_pmc_summary.json gpurun_out/ 2>/dev/null
python tools/rocpd_summary.py gpurun_out/${TAG}_trace/trace_results.db > gpurun_out/${TAG}_kernel_trace_stats_single_stream.md
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_grbm
