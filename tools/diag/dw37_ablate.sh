for v in "" abl1 abl2 abl3 abl4 abl16 abl32 abl48 abl55 abl64; do
  L=ml_fastvlm_amd/libfvhd${v:+_$v}.so
  echo "=== ${v:-base}"; FVHD_LIB=$L timeout 120 python tools/bench_ops.py dw37 2>&1 | grep "^dw3+dw7" | sed 's/rounds.*//' | cut -c1-200
done
