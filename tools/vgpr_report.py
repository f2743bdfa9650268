#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the built library, from the code objects' metadata notes.

    python tools/vgpr_report.py [out.csv]        (default: stdout)

For each object under boondock-airband_amd/csrc/build: the .hip_fatbin section is unbundled for gfx950 and `llvm-readelf --notes`
lists .vgpr_count / .agpr_count / .sgpr_count / .group_segment_fixed_size (static LDS) / .private_segment_fixed_size (scratch) per
kernel.  The plan-compiled stage-1 kernel (l64_entry, hipRTC) is not in the library: its register count is printed by the library
itself under MI_AIRBAND_DEBUG=1 (l64_jit.cpp) and can be appended with --jit "<vgprs>".
"""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::|mi::", "", o).split("(")[0] for o in out[:len(names)]]


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(ROOT, "boondock-airband_amd", "csrc", "build", "*.o"))):
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
            r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue  # a host-only object
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if line.lstrip().startswith("- .") and cur.get("name"):
                    rows.append((os.path.basename(obj), cur))
                    cur = {}
                cur[key] = val
            if cur.get("name"):
                rows.append((os.path.basename(obj), cur))
            os.remove(fat)
    names = demangle([c["name"] for _, c in rows])
    out = open(sys.argv[1], "w", newline="") if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else sys.stdout
    w = csv.writer(out)
    w.writerow(["object", "kernel", "vgprs", "agprs", "sgprs", "lds_bytes_static", "scratch_bytes", "waves_per_simd_by_vgprs"])
    for (obj, c), name in zip(rows, names):
        v = int(c.get("vgpr_count", 0)) + int(c.get("agpr_count", 0))
        waves = min(8, 512 // max(8, (v + 7) // 8 * 8)) if v else 8
        w.writerow([obj, name, c.get("vgpr_count"), c.get("agpr_count"), c.get("sgpr_count"), c.get("group_segment_fixed_size"),
                    c.get("private_segment_fixed_size"), waves])
    if "--jit" in sys.argv:
        v = int(sys.argv[sys.argv.index("--jit") + 1])
        w.writerow(["(hipRTC)", "l64_entry (plan-compiled stage 1, BASELINE configs[1] plan)", v, 0, "", "dynamic", 0, min(8, 512 // ((v + 7) // 8 * 8))])


if __name__ == "__main__":
    main()
