#!/bin/bash
# Regenerates profiles/${R}_*: the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains next to
# --pmc) first -- bench.py quotes `traffic` from their summary --, then the bench line, the rocprofv3 kernel trace of the same
# command, the kernel traces of the config-3 / config-4 / am64 workloads and the SQ counter passes of stage 1 on am64.
# The PMC passes force one chunk on every call (MI_AIRBAND_TP_CHUNKS=1): that is the geometry of every timed step of
# the bench (only the first, isolated warm-up call of a run uses two chunks), so the per-launch means are exact.
# Usage: tools/make_profiles.sh [round prefix, default r03] [stages: pmc bench kt kt3 kt4 ktam sq, default all]
set -e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=${1:-r03}
STAGES=${2:-"pmc bench kt kt3 kt4 ktam sq vgpr"}
O=gpurun_out
mkdir -p $O
has() { [[ " $STAGES " == *" $1 "* ]]; }
keep_stats() { find "$1" -name "*kernel_stats.csv" -exec cp {} "$2" \; ; rm -rf "$1"; }
if has pmc; then
  # headline workload + the three others: `traffic` of every bench line comes from these files (bench.py: PMC_PROFILES)
  for wl in config2 config3 config4 am64; do
    rm -rf $O/pmc_fetch $O/pmc_write
    extra="--workload $wl --steps 2 --warmup 1 --cpu-seconds 0"
    MI_AIRBAND_TP_CHUNKS=1 MI_AIRBAND_TP_RATIO=1.0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $extra > $O/pmc_fetch_$wl.log 2>&1
    MI_AIRBAND_TP_CHUNKS=1 MI_AIRBAND_TP_RATIO=1.0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $extra > $O/pmc_write_$wl.log 2>&1
    name=${R}_pmc.csv; [ $wl != config2 ] && name=${R}_${wl}_pmc.csv
    python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/$name > /dev/null
    cp $O/$name profiles/$name
    rm -rf $O/pmc_fetch $O/pmc_write
    echo "pmc $wl done"
  done
fi
if has bench; then
  python3 bench.py > $O/final_bench.log 2>&1
  grep "^{" $O/final_bench.log > $O/${R}_bench.json
  echo "bench done"
fi
if has kt; then
  rm -rf $O/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --cpu-seconds 0 > $O/bench_rocprof.log 2>&1
  grep "^{" $O/bench_rocprof.log > $O/${R}_bench_under_rocprof.json
  keep_stats $O/kt $O/${R}_kernel_stats.csv
  echo "kernel trace done"
fi
for wl in config3:kt3 config4:kt4 am64:ktam; do
  w=${wl%%:*}; st=${wl##*:}
  if has $st; then
    rm -rf $O/$st
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$st -- python3 bench.py --workload $w --steps 20 > $O/bench_${w}_rocprof.log 2>&1
    grep "^{" $O/bench_${w}_rocprof.log > $O/${R}_${w}_bench_under_rocprof.json
    keep_stats $O/$st $O/${R}_${w}_kernel_stats.csv
    echo "$w kernel trace done"
  fi
done
if has sq; then
  # stage 1 on the many-stream AM workload: where the cycles of k_channelize go (two passes of <= 8 SQ counters)
  rm -rf $O/sq1 $O/sq2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --workload am64 --steps 3 --warmup 1 > $O/sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_VMEM --output-format csv -d $O/sq2 -- python3 bench.py --workload am64 --steps 3 --warmup 1 > $O/sq2.log 2>&1
  python3 tools/sq_summary.py $O/${R}_am64_sq_counters.csv $O/sq1 $O/sq2
  rm -rf $O/sq1 $O/sq2
  echo "sq counters done"
fi
if has vgpr; then
  python3 tools/vgpr_report.py $O/${R}_vgpr.csv
  echo "vgpr report done"
fi
ls -la $O/${R}_*
