#!/usr/bin/env python3
"""Every kernel of a rocprofv3 kernel trace of bench.py around the steady state, in time order: start / end relative to the
start of the fourth-last core-chain launch, duration, queue.  Shows which launches of consecutive overlapped calls actually run
side by side (and which only look queued: a launch that ends exactly when another does was waiting for its registers).
Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload am64 ...; python3 tools/trace_dump.py DIR"""
import csv, glob, sys, re
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?"), r.get("Workgroup_Size","?"), r.get("Grid_Size","?")))
rows.sort()
core=[r for r in rows if "k_tp_core" in r[2]]
t0=core[-4][0]
for s,e,n,q,wg,gs in rows:
    if s>=t0-6e6 and s<=t0+7e6:
        m=re.search(r"(k_[a-z_0-9]+|l64_entry|fillBuffer)", n)
        print(f"{(s-t0)/1e6:8.3f} .. {(e-t0)/1e6:8.3f}  {(e-s)/1e6:6.3f} ms  q{q} grid {gs} wg {wg}  {m.group(1) if m else n[:40]}")
