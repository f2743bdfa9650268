#!/bin/bash
# stage 1 alone (serial stage 2, calls not overlapping): ms per launch of k_channelize on the many-stream workloads
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { grep "^{" "$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']['k_channelize']; print(sys.argv[1], 'k_channelize', round(k['ms'],3), 'ms', round(d['config']['streams_per_gpu']*d['config']['capture_seconds_per_step']*2.56e6/k['ms']/1e6,1), 'GS/s alone')" "$2"; }
for wl in ${WORKLOADS:-am64 config4}; do
  for v in "$@"; do
    env $v MI_AIRBAND_TP=0 timeout -k 10 200 python bench.py --workload $wl --steps 4 --warmup 1 --cpu-seconds 0 --no-overlap --seconds 2 > gpurun_out/s1_${wl}.log 2>&1 && show gpurun_out/s1_${wl}.log "$wl[$v]" || { echo "$wl $v failed"; tail -3 gpurun_out/s1_${wl}.log; }
  done
done
