#!/bin/bash
# The round's GPU calls, ONE script with stages (run ON the GPU box through gpurun, from the repo root):
#     gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <stage> [tag]'
# Everything lands under gpurun_out/<tag>_*; summaries worth keeping are copied into profiles/ by hand afterwards.
set -u
STAGE=${1:-suite}
TAG=${2:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${TAG}
case "$STAGE" in
suite)      # the whole GPU suite + the driver's bench line
    timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest.log
    timeout 600 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"; cut -c1-600 ${O}_bench.json
    ;;
ffn)        # fused ConvFFN variants: correctness of the variant library, sustained time / power / energy per launch, whole step
    for lib in ${FFN_LIBS:-"" old}; do
        [ "$lib" = base ] && lib=""
        L=ml_fastvlm_amd/libfvhd${lib:+_$lib}.so
        [ -f $L ] || { echo "missing $L"; continue; }
        FVHD_LIB=$L timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "ffn" > ${O}_ffn_${lib:-base}_pytest.log 2>&1; tail -1 ${O}_ffn_${lib:-base}_pytest.log
        FVHD_LIB=$L timeout 300 python tools/power_probe.py ffn384 ffn192 ffn96 2>/dev/null | tee -a ${O}_ffn_power.log
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_${lib:-base}.json 2>/dev/null; python - <<PY
import json
d=json.load(open("${O}_bench_${lib:-base}.json")); print("${lib:-base}", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
PY
    done
    ;;
extra)      # the other BASELINE configs: TTFT at the 7B width, 1536^2 with bf16 / e4m3 attention operands
    timeout 600 python bench.py --ttft --hidden 3584 --steps 10 --warmup 2 > ${O}_ttft_h3584.json 2> ${O}_ttft_h3584.err; echo "ttft3584 rc=$?"; cut -c1-400 ${O}_ttft_h3584.json
    timeout 400 python bench.py --res 1536 --batch 16 --no-cpu-baseline --no-ttft > ${O}_bench_1536.json 2>/dev/null; echo "1536 rc=$?"; cut -c1-300 ${O}_bench_1536.json
    timeout 400 python bench.py --res 1536 --batch 16 --attn-fp8 --no-cpu-baseline --no-ttft > ${O}_bench_1536_fp8.json 2>/dev/null; echo "1536 fp8 rc=$?"; cut -c1-300 ${O}_bench_1536_fp8.json
    ;;
stem)       # the fully fused stem: op + step tests, A/B against the round-3 path (FVHD_FUSED_STEM=1) in the whole step
    timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py -m gpu -x -q -k "stem or steps" > ${O}_stem_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 ${O}_stem_pytest.log
    for v in 1 2; do
        FVHD_FUSED_STEM=$v timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_stem$v.json 2>/dev/null; python - <<PY
import json
d=json.load(open("${O}_bench_stem$v.json")); print("FVHD_FUSED_STEM=$v", d["ms_per_step"], d["value"], d["kernels"]["stem"], d["conv_stage"]["frac"])
PY
    done
    ;;
attn)       # attention variants (library builds with another ATT_QW): correctness + sustained time
    for lib in ${ATTN_LIBS:-base qw4}; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > ${O}_attn_${lib}_pytest.log 2>&1; tail -1 ${O}_attn_${lib}_pytest.log
        FVHD_LIB=$L timeout 200 python tools/power_probe.py attn 2>/dev/null | tee -a ${O}_attn_power.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py attn 2>/dev/null | tee -a ${O}_attn_ops.log
    done
    ;;
dwdown)     # PatchEmbed dw7x7/s2 tile variants (library builds with -DFVHD_DWDOWN_CFG=n): correctness + time per launch
    for lib in ${DW_LIBS:-base dd1 dd2}; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "dwconv" > ${O}_dw_${lib}_pytest.log 2>&1; tail -1 ${O}_dw_${lib}_pytest.log
        echo "--- $lib" | tee -a ${O}_dwdown.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dwraw 2>/dev/null | grep "S=2" | tee -a ${O}_dwdown.log
    done
    ;;
calib)      # FETCH_SIZE calibration for the access widths of this library (tools/ubench/fetch_calib.hip)
    hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null
    (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/fc -o fc -- /tmp/fetch_calib) > ${O}_fetch_calib_run.log 2>&1
    python - <<PY | tee ${O}_fetch_calib.log
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"].split("(")[0], r.get("Counter_Name"))].append(float(r["Counter_Value"]))
known = {"read16": 256 * 2**20, "read2": 256 * 2**20, "read2r": 21 * 3 * 1024 * 1024 * 2}
for (k, cn), v in sorted(rows.items()):
    if k not in known: continue
    avg = sum(v) / len(v)
    if cn == "FETCH_SIZE":
        print(f"{k:8s} dispatches {len(v)}  FETCH_SIZE {avg * 1024 / 1e6:9.1f} MB per dispatch  bytes of the buffer read once {known[k] / 1e6:9.1f} MB  ratio {avg * 1024 / known[k]:.3f}")
    else:
        print(f"{k:8s} dispatches {len(v)}  {cn} {avg:.4g} requests per dispatch = {known[k] / max(avg, 1):.1f} B of the buffer per request")
PY
    ;;
gemm)       # GEMM tile / ring variants through the debug library's knobs (tools/bench_ops.py gemm)
    FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so BENCH_GEMM_VARIANTS=${GEMM_VARIANTS:-3,6,7,1} timeout 600 python tools/bench_ops.py gemm 2>&1 | grep -v Warning | tee ${O}_gemm_variants.log
    ;;
gemmlib)    # GEMM source variants as whole libraries (production dispatch): op tests + per-class times in the step
    for lib in ${GEMM_LIBS:-base fp0}; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > ${O}_gemm_${lib}_pytest.log 2>&1; tail -1 ${O}_gemm_${lib}_pytest.log
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_gemm_${lib}.json 2>/dev/null; python - <<PY
import json
d=json.load(open("${O}_bench_gemm_${lib}.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm") or k=="projector"})
PY
    done
    ;;
rest)       # the GPU tests a -x run did not reach + one named file
    timeout 900 python -m pytest ${REST_TESTS:-tests/test_qwen2_prefill.py tests/test_splice.py tests/test_preprocess.py} -m gpu -x -q > ${O}_pytest_rest.log 2>&1; echo "pytest rest rc=$?"; tail -3 ${O}_pytest_rest.log
    ;;
final)      # evidence of the final binary: bench line, small batches, PMC passes + kernel trace
    timeout 600 python bench.py > ${O}_final_bench.json 2> ${O}_final_bench.err; echo "bench rc=$?"; cut -c1-300 ${O}_final_bench.json
    timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-ttft > ${O}_bench_b8.json 2>/dev/null; cut -c1-200 ${O}_bench_b8.json
    timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-ttft > ${O}_bench_b1.json 2>/dev/null; cut -c1-200 ${O}_bench_b1.json
    timeout 300 python tools/power_probe.py ffn384 ffn192 ffn96 dwmix192 dwmix384 stem attn 2>/dev/null | tee ${O}_final_power.log
    bash tools/run_pmc.sh ${TAG}
    # kernel trace of the TTFT path (encode B = 8 -> splice -> Qwen2-0.5B prefill): per-layer launch times of the prefill
    timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_ttft_trace -o ttft -- python bench.py --ttft --steps 4 --warmup 1 > gpurun_out/${TAG}_ttft_trace.log 2>&1
    python tools/rocpd_summary.py gpurun_out/${TAG}_ttft_trace/ttft_results.db > ${O}_ttft_kernel_trace.md 2>&1; rm -rf gpurun_out/${TAG}_ttft_trace; head -30 ${O}_ttft_kernel_trace.md | cut -c1-160
    ;;
r6d)        # round 6: PatchEmbed's stride-2 depthwise conv on the matrix cores: op tests, VALU vs MFMA per shape, rows-per-chunk sweep, step tests, whole step A/B
    timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dwconv" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 ${O}_pytest.log | cut -c1-400
    FVHD_DWDOWN_MFMA=0 timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep -v Warning | tee ${O}_dwdown.log
    timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep -v Warning | tee -a ${O}_dwdown.log
    [ -f ml_fastvlm_amd/libfvhd_ablate.so ] && FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so timeout 300 python tools/bench_ops.py dwdown 2>&1 | grep -v Warning | tee -a ${O}_dwdown.log
    timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tower.py -m gpu -q --maxfail=15 > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -15 ${O}_pytest_steps.log | cut -c1-300
    for v in 0 1 0 1; do
        FVHD_DWDOWN_MFMA=$v timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_dd$v.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_dd$v.json")); print("FVHD_DWDOWN_MFMA=$v", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("gemm_1x1", "stem")}, d["conv_stage"]["frac"])
PY
    done
    ;;
r6e)        # round 6: dw_down ablations (variant libraries libfvhd_dd<bits>.so / _ddr4), attention V staging A/B (libfvhd_vtr0.so = the ds_write_b16 transposition)
    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dwconv or attention" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 ${O}_pytest.log | cut -c1-400
    for lib in base dd1 dd2 dd4 dd6 dd8 dd16 ddr4; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        [ -f $L ] || continue
        echo "--- $lib" | tee -a ${O}_dwdown_abl.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep "dw_down" | tee -a ${O}_dwdown_abl.log
    done
    for lib in base vtr0 base vtr0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        echo "--- $lib" | tee -a ${O}_attn_ab.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py attn 2>/dev/null | tee -a ${O}_attn_ab.log
    done
    timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tower.py -m gpu -q --maxfail=15 > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -5 ${O}_pytest_steps.log | cut -c1-300
    for lib in base vtr0 base vtr0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("attention", "stem")}, d["conv_stage"]["frac"], d["attention_block"]["frac"])
PY
    done
    ;;
r6f)        # round 6: dw_down skeleton ablations, dw7x7 with 32-px strips (stages 3 / 4) A/B through the debug library, attention V staging A/B
    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dwconv or attention or dw7" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 ${O}_pytest.log | cut -c1-400
    for lib in base dd31 dd63 dd95 dd127 base; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        [ -f $L ] || continue
        echo "--- $lib" | tee -a ${O}_dwdown_abl.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep "dw_down" | tee -a ${O}_dwdown_abl.log
    done
    FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so timeout 300 python tools/bench_ops.py dw7s34 2>&1 | grep "dw7 " | tee ${O}_dw7_strip32.log
    for lib in base vtr0 base vtr0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        echo "--- $lib" | tee -a ${O}_attn_ab.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py attn 2>&1 | grep "attention" | tee -a ${O}_attn_ab.log
    done
    for lib in base vtr0 base vtr0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("attention", "stem")}, d["conv_stage"]["frac"], d["attention_block"]["frac"])
PY
    done
    timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tower.py -m gpu -q --maxfail=15 > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -5 ${O}_pytest_steps.log | cut -c1-300
    ;;
r6g)        # round 6: dw_down with DMA / store wave roles and per-row operand refill against the first version (libfvhd_ddold.so)
    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dwconv" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 ${O}_pytest.log | cut -c1-400
    for lib in base ddold base ddold; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        [ -f $L ] || continue
        echo "--- $lib" | tee -a ${O}_dwdown_ab.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep "dw_down" | tee -a ${O}_dwdown_ab.log
    done
    FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so timeout 300 python tools/bench_ops.py dwdown 2>&1 | grep "dw_down" | tee ${O}_dwdown_rc.log
    for lib in base ddold base ddold; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("attention", "stem")}, d["conv_stage"]["frac"], d["attention_block"]["frac"])
PY
    done
    timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --maxfail=15 > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -5 ${O}_pytest_steps.log | cut -c1-300
    ;;
r6h)        # round 6: XCD-aware block order where channel blocks share 128-B lines (dw_down: 64-B pieces; dw_mix at C = 96) against libfvhd_x0.so (raw order)
    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dwconv or dw3_dw7 or dw7" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest.log | cut -c1-400
    for lib in base x0 base x0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        echo "--- $lib" | tee -a ${O}_xcd_ab.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dwdown 2>&1 | grep "dw_down" | tee -a ${O}_xcd_ab.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dw37 2>&1 | grep -i "96\|fused" | head -6 | tee -a ${O}_xcd_ab.log
    done
    for lib in base x0 base x0; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("stem",)}, d["conv_stage"]["frac"])
PY
    done
    ;;
r6i)        # round 6: 128 x 192 tiles for the N = 192 1x1 + GELU of PatchEmbed (FVHD_GEMM_NF6=0 / 1), bits and time
    timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.log | cut -c1-300
    for v in 1 0 1 0; do
        echo "--- FVHD_GEMM_NF6=$v" | tee -a ${O}_nf6.log
        FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so FVHD_GEMM_NF6=$v BENCH_GEMM_VARIANTS=1 timeout 200 python tools/bench_ops.py gemm 2>&1 | grep "1x1" | tee -a ${O}_nf6.log
    done
    for v in 1 0 1 0; do
        FVHD_GEMM_NF6=$v timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_nf6_$v.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_nf6_$v.json")); print("FVHD_GEMM_NF6=$v", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm")})
PY
    done
    timeout 600 python -m pytest tests/test_gpu_steps.py -m gpu -q --maxfail=15 -k "bench_batch or r1024" > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -3 ${O}_pytest_steps.log | cut -c1-300
    ;;
r6j)        # round 6: XCD-remapped block order in the depthwise pair for EVERY channel count (libfvhd_fzx2.so) against the default (C % 64 != 0 only)
    for lib in base fzx2 base fzx2; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        echo "--- $lib" | tee -a ${O}_fzx2.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dw37 2>&1 | grep "dw3+dw7" | cut -c1-140 | tee -a ${O}_fzx2.log
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_fzx2.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw")}, d["conv_stage"]["frac"])
PY
    done
    ;;
r6k)        # round 6: the 32-channel block of the depthwise pair on pairs of strips (C = 96) (a patch that is not shipped: profiles/r06_dw_mix_c96_two_strips.patch; base = the patched build, fzold = the shipped kernel)
    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dw3_dw7" --maxfail=30 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 ${O}_pytest.log | cut -c1-300
    for lib in base fzold base fzold; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        echo "--- $lib" | tee -a ${O}_half2.log
        FVHD_LIB=$L timeout 200 python tools/bench_ops.py dw37 2>&1 | grep "dw3+dw7" | cut -c1-140 | tee -a ${O}_half2.log
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_$lib.json 2>/dev/null; python - <<PY | tee -a ${O}_half2.log
import json
d=json.load(open("${O}_bench_$lib.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw")}, d["conv_stage"]["frac"])
PY
    done
    timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_ffn_precision.py -m gpu -q --maxfail=15 > ${O}_pytest_steps.log 2>&1; echo "pytest steps rc=$?"; tail -4 ${O}_pytest_steps.log | cut -c1-300
    ;;
r6a)        # round 6: the fused dw3x3 -> dw7x7 kernel: op tests, then fused vs two launches (+ rows-per-chunk sweep with the debug library)
    timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "dw3_dw7" --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 ${O}_pytest.log | cut -c1-400
    timeout 300 python tools/bench_ops.py dw37 2>&1 | grep -v Warning | tee ${O}_dw37.log
    FVHD_LIB=ml_fastvlm_amd/libfvhd_ablate.so timeout 300 python tools/bench_ops.py dw37 2>&1 | grep -v Warning | tee ${O}_dw37_rc.log
    ;;
r6b)        # round 6: the fused depthwise launch inside the tower: step / tower / precision tests, then FVHD_FUSED_DW=0 / 1 on one box
    timeout 1200 python -m pytest tests/test_gpu_steps.py tests/test_gpu_tower.py tests/test_gpu_ffn_precision.py tests/test_gpu_ops.py -m gpu -q --maxfail=15 --durations=5 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 ${O}_pytest.log | cut -c1-300
    for v in 0 1 0 1; do
        FVHD_FUSED_DW=$v timeout 300 python bench.py --no-cpu-baseline --no-ttft --no-extra-configs > ${O}_bench_dw$v.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_dw$v.json")); print("FVHD_FUSED_DW=$v", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("dw") or k in ("ffn_fused", "stem")}, d["conv_stage"]["frac"])
PY
    done
    ;;
r5a)        # round 5, first call: whole GPU suite (all failures, not -x), the driver's line, GEMM layout A/B (bits + time), WRITE_SIZE of the GEMM classes
    timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -30 ${O}_pytest.log | cut -c1-300
    timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"; cut -c1-400 ${O}_bench.json; tail -3 ${O}_bench.err
    python tools/gemm_bits.py > ${O}_gemm_bits_base.json 2>${O}_gemm_bits.err; FVHD_LIB=ml_fastvlm_amd/libfvhd_g1.so python tools/gemm_bits.py > ${O}_gemm_bits_g1.json 2>>${O}_gemm_bits.err
    python tools/gemm_bits.py --diff ${O}_gemm_bits_base.json ${O}_gemm_bits_g1.json | tee ${O}_gemm_bits_diff.log
    for lib in base g1; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        FVHD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_gemm_${lib}.json 2>/dev/null; python - <<PY | tee -a ${O}_gemm_ab.log
import json
d=json.load(open("${O}_bench_gemm_${lib}.json")); print("$lib", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm") or k in ("projector", "dw7", "dw3", "ffn_fused")})
PY
    done
    CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttft"
    for lib in base; do
        [ "$lib" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$lib.so
        for pass in trace write; do
            extra=""; [ "$pass" = write ] && extra="--pmc WRITE_SIZE"
            FVHD_LIB=$L timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_${lib}_${pass} -o ${pass} $extra -- $CMD > gpurun_out/${TAG}_${lib}_${pass}.log 2>&1
        done
        python tools/pmc_summary.py ${TAG}_${lib} gpurun_out/${TAG}_${lib}_trace gpurun_out/${TAG}_${lib}_write > ${O}_${lib}_write_summary.md 2>${O}_${lib}_write_summary.err
        cp profiles/${TAG}_${lib}_pmc_summary.json gpurun_out/ 2>/dev/null; rm -f profiles/${TAG}_${lib}_pmc_summary.json
        rm -rf gpurun_out/${TAG}_${lib}_trace gpurun_out/${TAG}_${lib}_write
        grep -i "gemm\|class" ${O}_${lib}_write_summary.md | cut -c1-220
    done
    ;;
r5b)        # round 5, second call: the tests the first call failed / the kernels touched since, guard-site and persistent-GEMM A/B on one box, WRITE_SIZE again
    timeout 900 python -m pytest tests/test_gpu_ffn_precision.py tests/test_gpu_ops.py tests/test_qwen2_prefill.py tests/test_gpu_steps.py "tests/test_gpu_reference.py::test_drop_in_defaults_are_range_safe_on_a_saturating_checkpoint" -m gpu -q --maxfail=20 --durations=5 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 ${O}_pytest.log | cut -c1-300
    run_bench() { # label, env...
        local label=$1; shift
        env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_${label}.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_${label}.json")); print("${label}", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm") or k in ("projector", "dw7", "dw3", "ffn_fused")})
PY
    }
    run_bench guard_off FVHD_RANGE_GUARD=0
    run_bench guard_dw7 FVHD_GUARD_SITE=0
    run_bench guard_dw3 FVHD_GUARD_SITE=1
    run_bench persist FVHD_GUARD_SITE=1 FVHD_GEMM_PERSIST=1
    run_bench guard_off2 FVHD_RANGE_GUARD=0
    for v in "base FVHD_GEMM_PERSIST=0" "persist FVHD_GEMM_PERSIST=1" "g1 FVHD_LIB=ml_fastvlm_amd/libfvhd_g1.so"; do
        set -- $v
        env $2 timeout 300 python bench.py --ttft --steps 10 --warmup 2 > ${O}_ttft_$1.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_ttft_$1.json")); c=d["config"]; print("ttft $1", d["value"], c["encode_images_ms"], c["splice_ms"], c["prefill_first_token_ms"], c["prefill_roofline"]["frac"])
PY
    done
    CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttft"
    for pass in trace write; do
        extra=""; [ "$pass" = write ] && extra="--pmc WRITE_SIZE"
        timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_${pass} -o ${pass} $extra -- $CMD > gpurun_out/${TAG}_${pass}.log 2>&1
    done
    python tools/pmc_summary.py ${TAG}_w gpurun_out/${TAG}_trace gpurun_out/${TAG}_write > ${O}_write_summary.md 2>${O}_write_summary.err
    rm -f profiles/${TAG}_w_pmc_summary.json; rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_write
    grep -i "gemm\|class" ${O}_write_summary.md | cut -c1-200
    ;;
r5c)        # round 5, third call: the two range tests again, epilogue walk A/B (libfvhd_cm.so), the dw3 / dw7 coupling per stage (kernel traces)
    timeout 600 python -m pytest tests/test_gpu_ffn_precision.py "tests/test_gpu_reference.py::test_drop_in_defaults_are_range_safe_on_a_saturating_checkpoint" tests/test_gpu_ops.py -k "not gemm and not attention and not ffn_fused" -m gpu -q --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 ${O}_pytest.log | cut -c1-300
    run_bench() { # label, env...
        local label=$1; shift
        env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_${label}.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_${label}.json")); print("${label}", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm") or k in ("projector", "dw7", "dw3", "ffn_fused")})
PY
    }
    run_bench rowmajor FVHD_LIB=ml_fastvlm_amd/libfvhd.so
    run_bench colmajor FVHD_LIB=ml_fastvlm_amd/libfvhd_cm.so
    run_bench rowmajor2 FVHD_LIB=ml_fastvlm_amd/libfvhd.so
    run_bench colmajor2 FVHD_LIB=ml_fastvlm_amd/libfvhd_cm.so
    run_bench guard_off FVHD_RANGE_GUARD=0
    CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-ttft"
    for mode in site0 off; do
        g=1; [ "$mode" = off ] && g=0
        FVHD_RANGE_GUARD=$g timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_tr_${mode} -o tr -- $CMD > gpurun_out/${TAG}_tr_${mode}.log 2>&1
        python tools/rocpd_by_grid.py gpurun_out/${TAG}_tr_${mode}/tr_results.db dwconv_tiled dw7_mfma ffn_fused > ${O}_bygrid_${mode}.md 2>&1
        rm -rf gpurun_out/${TAG}_tr_${mode}
        echo "--- $mode"; cat ${O}_bygrid_${mode}.md | cut -c1-170
    done
    ;;
r5d)        # round 5, fourth call: amax through 64 slots + one atomic per workgroup: tests, guard on / off on one box
    timeout 600 python -m pytest tests/test_gpu_ffn_precision.py tests/test_gpu_ops.py -k "not attention and not ffn_fused and not stem" -m gpu -q --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 ${O}_pytest.log | cut -c1-300
    run_bench() { # label, env...
        local label=$1; shift
        env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_${label}.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_${label}.json")); print("${label}", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("gemm") or k in ("projector", "dw7", "dw3", "ffn_fused")})
PY
    }
    run_bench guard_on FVHD_RANGE_GUARD=1
    run_bench guard_off FVHD_RANGE_GUARD=0
    run_bench guard_on2 FVHD_RANGE_GUARD=1
    run_bench guard_off2 FVHD_RANGE_GUARD=0
    run_bench guard_dw3 FVHD_GUARD_SITE=1
    ;;
r5final)    # round 5: the whole GPU suite on the final binary, then its evidence (bench line, small batches, power, PMC passes, TTFT trace)
    timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 ${O}_pytest.log | cut -c1-300
    bash tools/gpu_run.sh final ${TAG}
    ;;
r5e)        # round 5: rotary embedding in the q|k|v projection's epilogue - tests, TTFT A/B on one box, the prefill's kernel trace
    FVHD_LLM_FUSEROPE=1 timeout 600 python -m pytest tests/test_qwen2_prefill.py tests/test_gpu_ttft.py -m gpu -q --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 ${O}_pytest.log | cut -c1-300
    for v in "unfused FVHD_LLM_FUSEROPE=0" "fused FVHD_LLM_FUSEROPE=1" "unfused2 FVHD_LLM_FUSEROPE=0" "fused2 FVHD_LLM_FUSEROPE=1"; do
        set -- $v
        env $2 timeout 300 python bench.py --ttft --steps 20 --warmup 3 > ${O}_ttft_$1.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_ttft_$1.json")); c=d["config"]; print("ttft $1", d["value"], c["encode_images_ms"], c["splice_ms"], c["prefill_first_token_ms"], c["prefill_roofline"]["frac"])
PY
    done
    for v in "unfused FVHD_LLM_FUSEROPE=0" "fused FVHD_LLM_FUSEROPE=1"; do
        set -- $v
        env $2 timeout 300 python bench.py --ttft --batch 1 --steps 20 --warmup 3 > ${O}_ttft_b1_$1.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_ttft_b1_$1.json")); c=d["config"]; print("ttft B=1 $1", d["value"], c["encode_images_ms"], c["splice_ms"], c["prefill_first_token_ms"], c["prefill_roofline"]["frac"])
PY
    done
    ;;
r5f)        # round 5: skewed channel halves of the dw7 kernel's transposed image - tests, same-box A/B against -DDWM_TSKEW_=0, LDS conflict counters
    timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_steps.py -k "dw or steps" -m gpu -q --maxfail=20 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest.log | cut -c1-300
    run_bench() { # label, env...
        local label=$1; shift
        env "$@" timeout 300 python bench.py --no-cpu-baseline --no-ttft > ${O}_bench_${label}.json 2>/dev/null; python - <<PY | tee -a ${O}_ab.log
import json
d=json.load(open("${O}_bench_${label}.json")); print("${label}", d["ms_per_step"], d["value"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k in ("dw7", "dw3", "ffn_fused", "stem", "dw_down")})
PY
    }
    run_bench ${AB_A:-skew} FVHD_LIB=ml_fastvlm_amd/libfvhd.so
    run_bench ${AB_B:-noskew} FVHD_LIB=ml_fastvlm_amd/libfvhd_${AB_LIB:-ts0}.so
    run_bench ${AB_A:-skew}2 FVHD_LIB=ml_fastvlm_amd/libfvhd.so
    run_bench ${AB_B:-noskew}2 FVHD_LIB=ml_fastvlm_amd/libfvhd_${AB_LIB:-ts0}.so
    CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-ttft"
    for pass in trace grbm; do
        extra=""; [ "$pass" = grbm ] && extra="--pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        timeout 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/${TAG}_${pass} -o ${pass} $extra -- $CMD > gpurun_out/${TAG}_${pass}.log 2>&1
    done
    python tools/pmc_summary.py ${TAG}_l gpurun_out/${TAG}_trace gpurun_out/${TAG}_grbm > ${O}_lds_summary.md 2>${O}_lds_summary.err
    rm -f profiles/${TAG}_l_pmc_summary.json; rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_grbm
    grep -i "dw\|class\|stem" ${O}_lds_summary.md | cut -c1-200
    ;;
r5g)        # round 5: dw7 prologue order (first rows in flight before the tap loads) - digests and launch times under both libraries, same box
    for rep in 1 2; do
        for lib in "" _${AB_LIB:-pre}; do
            echo "--- libfvhd${lib}.so (pass $rep)" | tee -a ${O}_dw7_bits.log
            FVHD_LIB=ml_fastvlm_amd/libfvhd${lib}.so timeout 200 python tools/dw7_bits.py --time 2>&1 | grep "digest\|time" | tee -a ${O}_dw7_bits.log > /dev/null
        done
    done
    python - <<PY
import re
txt = open("${O}_dw7_bits.log").read().split("--- ")[1:]
dig = [[l for l in t.splitlines() if l.startswith("digest")] for t in txt]
print("digests identical across libraries and passes:", all(d == dig[0] for d in dig), len(dig[0]))
tm = [[l for l in t.splitlines() if l.startswith("time")] for t in txt]
for rows in zip(*tm):
    print(rows[0][:24], " | ".join(r.split(":")[1].strip() for r in rows), " (new, pre, new, pre)")
PY
    ;;
pmc)        # rocprofv3 kernel trace + the PMC passes of the final binary
    bash tools/run_pmc.sh ${TAG}
    ;;
*)  echo "unknown stage $STAGE"; exit 2;;
esac
