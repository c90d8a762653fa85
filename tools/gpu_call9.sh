#!/bin/bash
# round-3 call 9: LDS reads beside MFMAs (ubench), power / clock telemetry of the device the kernels run on
mkdir -p gpurun_out
F16_RATE_SKIP=1 ./tools/ubench/f16_rate > gpurun_out/c9_lds_shadow.log 2>&1; echo "ubench rc=$?"; cat gpurun_out/c9_lds_shadow.log
ls /sys/bus/pci/devices/*/hwmon 2>/dev/null | head -20
timeout 120 python tools/power_probe.py idle ffn384 ffn192 ffn96 gemm dw7 > gpurun_out/c9_power.log 2>&1; echo "power rc=$?"; grep -v amdgpu.ids gpurun_out/c9_power.log
(rocm-smi --showpower --showclocks 2>&1 | head -40) > gpurun_out/c9_rocm_smi.log; head -30 gpurun_out/c9_rocm_smi.log
