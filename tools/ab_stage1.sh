#!/bin/bash
# A/B of the stage-1 variants on the 64-stream workload: level table vs arithmetic conversion, full vs pruned graph.
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { grep "^{" "$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), 'k_channelize', round(d['kernels']['k_channelize']['ms'],4))"; }
for v in "lut 0" "arith 0" "lut 1" "arith 1"; do
  set -- $v
  echo "CONV=$1 PRUNE=$2"
  MI_AIRBAND_CONV=$1 MI_AIRBAND_PRUNE=$2 timeout -k 10 200 python bench.py --workload "${WL:-am64}" --steps 5 --warmup 2 > gpurun_out/ab_$1_$2.log 2>&1 && show gpurun_out/ab_$1_$2.log || exit 1
done
