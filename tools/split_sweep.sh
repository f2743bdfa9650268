#!/bin/bash
# pipelined serial calls: k_demod alone on the last n CUs, stage 1 on the others (MI_AIRBAND_SPLIT_CUS), many-row workloads
for wl in ${WORKLOADS:-am64 config4}; do
  for n in ${SPLITS:-0 64 48 96 128}; do
    BENCH_STREAM=side MI_AIRBAND_SPLIT_CUS=$n timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/split.log 2>&1
    echo "$wl split $n: $(grep '^{' gpurun_out/split.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['value']/1000,1), 'GS/s', round(d['ms_per_step'],3), 'ms/step;', ' '.join(f'{n} {v[\"ms\"]:.2f}' for n,v in k.items()))" 2>/dev/null || tail -2 gpurun_out/split.log)"
  done
done
