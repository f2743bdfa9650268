#!/bin/bash
# Regenerates profiles/r01_final_*: the bench line, the rocprofv3 kernel trace of the same command, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains next to --pmc), and the kernel trace of the config-3 workload.
set -e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -rf gpurun_out/kt gpurun_out/kt3 gpurun_out/pmc_fetch gpurun_out/pmc_write
python3 bench.py > gpurun_out/final_bench.log 2>&1
grep "^{" gpurun_out/final_bench.log > gpurun_out/r01_final_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --cpu-seconds 0 > gpurun_out/final_bench_rocprof.log 2>&1
grep "^{" gpurun_out/final_bench_rocprof.log > gpurun_out/r01_final_bench_under_rocprof.json
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/pmc_write.log 2>&1
echo "pmc write done"
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r01_final_pmc.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python3 bench.py --workload config3 --cpu-seconds 0 > gpurun_out/final_bench_config3_rocprof.log 2>&1
grep "^{" gpurun_out/final_bench_config3_rocprof.log > gpurun_out/r01_final_config3_bench_under_rocprof.json
find gpurun_out/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_final_kernel_stats.csv \;
find gpurun_out/kt3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_final_config3_kernel_stats.csv \;
# the raw traces are large: keep the summaries only
rm -rf gpurun_out/kt gpurun_out/kt3 gpurun_out/pmc_fetch gpurun_out/pmc_write
ls -la gpurun_out/r01_final_*
