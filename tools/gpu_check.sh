#!/bin/bash
# GPU round trip used during kernel work: parity tests, then the bench on the headline and the many-stream workloads.
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" != "--no-tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log; grep -n "^E " gpurun_out/pytest_gpu.log | head -8
else
  shift
fi
show() { grep "^{" "$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(sys.argv[1], round(d['value'],1), 'MS/s', round(d['x_realtime_per_stream'],1), 'x', round(d['ms_per_step'],3), 'ms/step; frac', round(r['frac'],4), 'path', round(r['frac_path'],4), 'alu', round(r['alu']['frac'],4)); [print('  ', k, round(v['ms'],4), v['launches_per_step'], round(v['ms_per_step'],3), v['GBps'] and round(v['GBps'],1)) for k,v in d['kernels'].items()]" "$1"; }
for wl in ${WORKLOADS:-config2 am64 config4 config3}; do
  timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-10} --warmup 2 --cpu-seconds 0 "$@" > gpurun_out/bench_$wl.log 2>&1 && show gpurun_out/bench_$wl.log || { echo "bench $wl failed"; tail -5 gpurun_out/bench_$wl.log; }
done
