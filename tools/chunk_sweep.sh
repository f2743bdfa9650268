#!/bin/bash
# chunks per call of the time-parallel pipeline against the default, at the driver's 20 steps and at 50 (fill and drain weigh less)
for cfg in "" "MI_AIRBAND_TP_CHUNKS=1" "MI_AIRBAND_TP_CHUNKS=2" ${CHUNK_CFGS}; do
  for st in 20 50; do
    env $cfg timeout -k 10 200 python bench.py --steps $st --warmup 5 --cpu-seconds 0 > gpurun_out/chunks.log 2>&1
    echo "[$cfg] steps $st: $(grep '^{' gpurun_out/chunks.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', round(d['value']/1000,1), 'GS/s')")"
  done
done
