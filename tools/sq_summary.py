#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes (counter_collection CSVs).

    python3 tools/sq_summary.py profiles/rNN_am64_sq_counters.csv gpurun_out/sq1 gpurun_out/sq2

One row per (kernel, counter): launches, mean per launch, and for the cycle counters the share of SQ_WAVE_CYCLES of the
same kernel (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES, MI355X_MICROARCH.md "rocprofv3 PMC slots");
SQ_LDS_BANK_CONFLICT is also given as a share of SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import re
import sys


def kernel(name):
    m = re.search(r"(k_[a-z_0-9]+|l64_entry|__amd_rocclr_[A-Za-z]+)", name)
    return m.group(1) if m else name


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[(kernel(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    rows = []
    for (k, c), m in sorted(mean.items()):
        if not (k.startswith("k_") or k.startswith("l64")):
            continue
        share = ""
        wc = mean.get((k, "SQ_WAVE_CYCLES"))
        if wc and c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            share = f"{m / wc:.4f} of SQ_WAVE_CYCLES"
        idx = mean.get((k, "SQ_LDS_IDX_ACTIVE"))
        if idx and c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT"):
            share = f"{m / idx:.4f} of SQ_LDS_IDX_ACTIVE"
        rows.append([k, c, len(acc[(k, c)]), round(m, 1), share])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "mean_per_launch", "share"])
        w.writerows(rows)
    for r in rows:
        print(*r)


if __name__ == "__main__":
    main()
