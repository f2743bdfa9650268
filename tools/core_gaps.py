#!/usr/bin/env python3
"""Gaps of the core chain in a rocprofv3 kernel trace of bench.py: the time between the end of one k_tp_core launch and the
start of the next, and what ran meanwhile.  Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...;
python3 tools/core_gaps.py DIR"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
core = [r for r in rows if "k_tp_core" in r[2]]
print("core launches", len(core))
gaps = []
for a, b in zip(core, core[1:]):
    gaps.append((b[0] - a[1]) / 1e3)
dur = [(r[1] - r[0]) / 1e3 for r in core]
tail = core[-40:]
span = (tail[-1][1] - tail[0][0]) / 1e3
busy = sum((r[1] - r[0]) / 1e3 for r in tail)
print(f"last 40 launches: span {span:.0f} us, core busy {busy:.0f} us ({100 * busy / span:.1f} %), per step (2 launches) {span / 20:.0f} us")
print("gaps (us) of the last 40:", " ".join(f"{g:.0f}" for g in gaps[-39:]))
print("durations (us) of the last 40:", " ".join(f"{d:.0f}" for d in dur[-40:]))

# what ran around the last long gap
import re
long_gaps = [i for i, g in enumerate(gaps) if g > 200]
if long_gaps:
    i = long_gaps[-2] if len(long_gaps) > 1 else long_gaps[-1]
    t0, t1 = core[i][1], core[i + 1][0]
    print(f"\nkernels around the gap after core launch {i} (times in us relative to the end of that launch; gap = {(t1 - t0) / 1e3:.0f} us):")
    for s, e, n in rows:
        if e > t0 - 1500000 and s < t1 + 200000:
            m = re.search(r"(k_[a-z_0-9]+)", n)
            print(f"  {(s - t0) / 1e3:9.0f} .. {(e - t0) / 1e3:9.0f}  {m.group(1) if m else n[:40]}")
