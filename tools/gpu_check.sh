#!/bin/bash
# GPU round trip used during kernel work: parity tests, then the headline bench (signal and noise-only captures).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log; grep -n "^E " gpurun_out/pytest_gpu.log | head -5
show() { grep "^{" "$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['x_realtime_per_stream'],1), round(d['ms_per_step'],3)); [print('  ', k, round(v['ms'],4), v['launches_per_step'], round(v['ms_per_step'],3)) for k,v in d['kernels'].items()]"; }
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 "$@" > gpurun_out/bench5.log 2>&1 && show gpurun_out/bench5.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --noise-only "$@" > gpurun_out/bench5n.log 2>&1 && show gpurun_out/bench5n.log
