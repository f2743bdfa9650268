#!/usr/bin/env python3
"""Timeline of overlapped serial calls in a rocprofv3 kernel trace of bench.py (many-row workloads): start / end of every stage-1 launch and
every k_demod launch relative to the first, and the idle time of each of the two between consecutive launches.
Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload am64 ...; python3 tools/serial_gaps.py DIR"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
s1 = [r for r in rows if "l64_entry" in r[2] or "k_channelize" in r[2]]
kd = [r for r in rows if "k_demod" in r[2]]
t0 = s1[0][0]
print("stage 1 launches", len(s1), " k_demod launches", len(kd))
for name, ls in (("stage 1", s1), ("k_demod", kd)):
    tail = ls[-12:]
    print(name, "last 12: start, end (ms), duration, idle before it")
    for a, b in zip(tail, tail[1:]):
        print(f"   {(b[0] - t0) / 1e6:9.3f} {(b[1] - t0) / 1e6:9.3f}   {(b[1] - b[0]) / 1e6:6.3f}   {(b[0] - a[1]) / 1e6:6.3f}")
