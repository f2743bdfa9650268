#!/bin/bash
# A/B of the bench line under environment switches: tools/ab_bench.sh "<steps list>" "VAR=a VAR=b ..." (each setting is one run)
export TMPDIR=/tmp
mkdir -p gpurun_out
for st in $1; do
  for kv in $2; do
    env $kv timeout -k 10 300 python3 bench.py --steps $st --warmup 5 --cpu-seconds 0 > gpurun_out/ab.log 2>&1
    grep "^{" gpurun_out/ab.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('steps', d['steps'], '$kv', round(d['ms_per_step'], 3), 'ms/step', {k: round(v['ms'], 2) for k, v in d['kernels'].items()})"
  done
done
