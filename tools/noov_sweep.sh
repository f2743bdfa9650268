#!/bin/bash
# isolated (non-overlapping) 64-s calls of the headline plan under different chunkings: tools/noov_sweep.sh "chunks:ratio" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for cr in "$@"; do
  c=${cr%%:*}; r=${cr##*:}
  MI_AIRBAND_TP_CHUNKS=$c MI_AIRBAND_TP_RATIO=$r timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-overlap > gpurun_out/sweep.log 2>&1
  echo "chunks $c ratio $r: $(grep "^{" gpurun_out/sweep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})")"
done
