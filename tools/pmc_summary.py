#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC slots).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0
    python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/rNN_pmc.csv

Both counters are in KiB.  /opt/skills/guides/MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of
a wide streaming read, so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.  Writes a CSV with one row
per kernel: launches, mean counter per launch, corrected bytes per launch.
"""
import collections
import csv
import glob
import re
import sys


def kernel(name):
    m = re.search(r"(k_[a-z_0-9]+|l64_entry|__amd_rocclr_[A-Za-z]+)", name)
    return m.group(1) if m else name


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[kernel(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        rows.append([k, len(f) or len(w), round(fm, 2), round(wm, 2), int(2 * fm * 1024), int(wm * 1024), int(2 * fm * 1024 + wm * 1024)])
    with open(sys.argv[3], "w", newline="") as out:
        wr = csv.writer(out)
        wr.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "fetched_bytes_per_launch(2x)", "written_bytes_per_launch",
                     "hbm_bytes_per_launch"])
        wr.writerows(rows)
    for r in rows:
        print(*r)


if __name__ == "__main__":
    main()
