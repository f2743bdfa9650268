#!/bin/bash
# SQ counters of stage 1 alone on the many-stream AM workload (serial stage 2, calls not overlapping)
# usage: tools/sq_stage1.sh OUT.csv [ENV=VAL ...]
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
out=$1; shift
O=gpurun_out
rm -rf $O/sq1 $O/sq2
env "$@" MI_AIRBAND_TP=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --workload ${WL:-am64} --steps 3 --warmup 1 --cpu-seconds 0 --no-overlap --seconds 2 > $O/sq1.log 2>&1
env "$@" MI_AIRBAND_TP=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_VMEM --output-format csv -d $O/sq2 -- python3 bench.py --workload ${WL:-am64} --steps 3 --warmup 1 --cpu-seconds 0 --no-overlap --seconds 2 > $O/sq2.log 2>&1
python3 tools/sq_summary.py $out $O/sq1 $O/sq2 | grep -E "channelize|l64_entry"
rm -rf $O/sq1 $O/sq2
