#!/bin/bash
# SQ counters of the serial stage-2 kernel on a workload (default config3 = BASELINE configs[2]); usage: tools/sq_demod.sh OUT.csv [ENV=VAL ...]
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
out=$1; shift
O=gpurun_out
rm -rf $O/sqd1 $O/sqd2
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $O/sqd1 -- python3 bench.py --workload ${WL:-config3} --steps 3 --warmup 1 --cpu-seconds 0 > $O/sqd1.log 2>&1
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM SQ_IFETCH --output-format csv -d $O/sqd2 -- python3 bench.py --workload ${WL:-config3} --steps 3 --warmup 1 --cpu-seconds 0 > $O/sqd2.log 2>&1
python3 tools/sq_summary.py $out $O/sqd1 $O/sqd2 | grep -E "k_demod"
rm -rf $O/sqd1 $O/sqd2
