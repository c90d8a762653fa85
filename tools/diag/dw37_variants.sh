# time the fused dw3 -> dw7 kernel under several library builds (FVHD_LIB), one box
for v in "$@"; do
  [ "$v" = base ] && L=ml_fastvlm_amd/libfvhd.so || L=ml_fastvlm_amd/libfvhd_$v.so
  echo "=== $v"; FVHD_LIB=$L timeout 120 python tools/bench_ops.py dw37 2>&1 | grep "^dw3+dw7\|output rows" | sed 's/rounds.*//' | cut -c1-200
done
