for st in 48 40 33; do
  for n in 0 $((st*2)) $((st*2+16)); do
    BENCH_STREAM=side MI_AIRBAND_SPLIT_CUS=$n timeout -k 10 200 python bench.py --workload am64 --streams $st --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/split.log 2>&1
    echo "am64 streams $st split $n: $(grep '^{' gpurun_out/split.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['value']/1000,1), 'GS/s', round(d['ms_per_step'],3), 'ms/step;', ' '.join(f'{n} {v[\"ms\"]:.2f}' for n,v in k.items()))" 2>/dev/null || tail -2 gpurun_out/split.log)"
  done
  BENCH_STREAM=null timeout -k 10 200 python bench.py --workload am64 --streams $st --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('am64 streams $st null stream, no split:', round(d['value']/1000,1), round(d['ms_per_step'],3))"
done
