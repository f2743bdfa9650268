export TMPDIR=/tmp
for v in ${VARIANTS}; do
  export MI_AIRBAND_LIB=$PWD/build_variants/lib_$v.so
  echo "== $v"
  bash tools/gpu_check.sh --no-tests 2>&1 | grep -E "bench_|k_demod|k_channelize"
done
