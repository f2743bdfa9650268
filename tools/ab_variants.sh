export TMPDIR=/tmp
for v in pw0_uni3 pw0_uni2 pw3_uni3; do
  export MI_AIRBAND_LIB=$PWD/build_variants/lib_$v.so
  echo "== $v"
  WORKLOADS="config3 config4" bash tools/gpu_check.sh --no-tests 2>&1 | grep -E "bench_|k_demod|k_channelize"
  python tools/row_cost.py 2>&1 | grep -E "AM plain|NFM \+ lowpass \+"
done
