#!/usr/bin/env python3
"""Generates tests/golden/components_ref.npz from the REFERENCE's own compiled classes (oracle/_ref,
built by `make -C oracle ref` from /root/reference/src/{squelch,ctcss,filters,logging,generate_signal}.cpp).

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/gen_golden.py
The committed .npz holds inputs-by-seed parameters and the reference's OUTPUTS (flags, levels, filter
outputs, counters) -- data, no reference source.  tests/test_oracle_components.py replays the same seeded
inputs through the oracle and requires identical bits; that pins the oracle on machines without _ref.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import libs  # noqa: E402
from golden_inputs import CTCSS_CASES, FILTER_CASES, SQUELCH_CASES, make_audio, make_filtered, make_raw  # noqa: E402


def main():
    r = libs.ref()
    if r is None:
        raise SystemExit("oracle/_ref/libairband_ref.so unavailable (needs /root/reference)")
    out = {}
    for name, case in SQUELCH_CASES.items():
        raw = make_raw(case)
        filt = make_filtered(case, raw) if case.get("filt") else None
        audio = make_audio(case) if case.get("audio") else None
        res = r.squelch_run(raw, filt=filt, audio=audio, manual_level=case.get("manual_level", 0.0), snr_db=case.get("snr_db"),
                            ctcss_freq=case.get("ctcss_freq", 0.0), ctcss_rate=case.get("ctcss_rate", 16000))
        out[f"sq_{name}_flags"] = res["flags"]
        out[f"sq_{name}_level"] = res["level"]
        out[f"sq_{name}_noise"] = res["noise"]
        out[f"sq_{name}_signal"] = res["signal"]
        f = res["final"]
        out[f"sq_{name}_final_counts"] = np.array([f.open_count, f.flappy_count, f.ctcss_count, f.no_ctcss_count], np.uint64)
        out[f"sq_{name}_final_levels"] = np.array([f.noise_level, f.signal_level, f.squelch_level], np.float32)
    for name, case in CTCSS_CASES.items():
        x = make_audio(case)
        flags, found, nfound = r.ctcss_run(case["freq"], case["rate"], case["window"], x)
        out[f"ct_{name}_flags"] = flags
        out[f"ct_{name}_counts"] = np.array([found, nfound], np.uint64)
    for name, case in FILTER_CASES.items():
        rng = np.random.default_rng(case["seed"])
        x = rng.normal(size=case["n"]).astype(np.float32)
        y = rng.normal(size=case["n"]).astype(np.float32)
        if case["kind"] == "notch":
            out[f"fl_{name}_y"] = r.notch_run(case["freq"], 16000.0, case["q"], x)
        else:
            a, b = r.lowpass_run(case["freq"], 16000.0, x, y)
            out[f"fl_{name}_re"] = a
            out[f"fl_{name}_im"] = b
    out["tone_79_7"] = r.tone_run(8000, 79.7, 0.2, 4000)
    path = os.path.join(HERE, "components_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
