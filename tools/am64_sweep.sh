#!/bin/bash
# many-stream AM workload under different settings: tools/am64_sweep.sh ENV=VAL ... (each argument one run)
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { grep "^{" "$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), 'MS/s', round(d['ms_per_step'],3), 'ms/step', ' '.join(f\"{k}={round(v['ms_per_step'],2)}\" for k,v in d['kernels'].items()))" "$2"; }
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --workload ${WL:-am64} --steps ${STEPS:-10} --warmup 2 --cpu-seconds 0 > gpurun_out/sweep.log 2>&1 && show gpurun_out/sweep.log "[$v]" || { echo "$v failed"; tail -3 gpurun_out/sweep.log; }
done
