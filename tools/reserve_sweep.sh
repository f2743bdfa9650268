#!/bin/bash
# CUs kept free of the wide passes (MI_AIRBAND_RESERVE_CUS) against the headline step, 50 steps each
for r in ${RESERVES:-16 24 32 40 48 64}; do
  MI_AIRBAND_RESERVE_CUS=$r timeout -k 10 200 python bench.py --steps 50 --warmup 5 --cpu-seconds 0 > gpurun_out/reserve.log 2>&1
  echo "reserve $r: $(grep '^{' gpurun_out/reserve.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['ms_per_step'],3), 'ms/step;', ' '.join(f'{n} {v[\"ms\"]:.2f}' for n,v in k.items()))")"
done
