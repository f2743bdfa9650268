#!/bin/bash
# Regenerates profiles/r01_final_*: the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains next to
# --pmc) first -- bench.py quotes `traffic` from their summary --, then the bench line, the rocprofv3 kernel trace of the same
# command, and the kernel trace of the config-3 workload.
# The PMC passes force one chunk on every call (MI_AIRBAND_TP_CHUNKS=1): that is the geometry of every timed step of
# the bench (only the first, isolated warm-up call of a run uses three growing chunks), so the per-launch means are exact.
set -e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -rf gpurun_out/kt gpurun_out/kt3 gpurun_out/pmc_fetch gpurun_out/pmc_write
MI_AIRBAND_TP_CHUNKS=1 MI_AIRBAND_TP_RATIO=1.0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch done"
MI_AIRBAND_TP_CHUNKS=1 MI_AIRBAND_TP_RATIO=1.0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_write.log 2>&1
echo "pmc write done"
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r01_final_pmc.csv
cp gpurun_out/r01_final_pmc.csv profiles/r01_final_pmc.csv
python3 bench.py > gpurun_out/final_bench.log 2>&1
grep "^{" gpurun_out/final_bench.log > gpurun_out/r01_final_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --cpu-seconds 0 > gpurun_out/final_bench_rocprof.log 2>&1
grep "^{" gpurun_out/final_bench_rocprof.log > gpurun_out/r01_final_bench_under_rocprof.json
echo "kernel trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python3 bench.py --workload config3 --cpu-seconds 0 > gpurun_out/final_bench_config3_rocprof.log 2>&1
grep "^{" gpurun_out/final_bench_config3_rocprof.log > gpurun_out/r01_final_config3_bench_under_rocprof.json
find gpurun_out/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_final_kernel_stats.csv \;
find gpurun_out/kt3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_final_config3_kernel_stats.csv \;
# the raw traces are large: keep the summaries only
rm -rf gpurun_out/kt gpurun_out/kt3 gpurun_out/pmc_fetch gpurun_out/pmc_write
ls -la gpurun_out/r01_final_*
