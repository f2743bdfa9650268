#!/usr/bin/env python3
"""Serial stage-2 cost per channel type: k_demod time for 8 channels of one type over 8 s (128 000 steps per channel).
Diagnostic for DESIGN.md section 6 (where the per-step time of the serial kernel goes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MI_AIRBAND_TP"] = "0"
import torch  # noqa: E402,F401
from bench import load_package  # noqa: E402
from common import gen_iq  # noqa: E402

pkg = load_package()
centre = 120_000_000
freqs = [centre - 1_000_000 + 25_000 + 250_000 * k for k in range(8)]
kinds = {
    "AM plain": lambda f: pkg.channel_cfg(f),
    "AM + notch": lambda f: pkg.channel_cfg(f, notch=1000.0),
    "AM + lowpass": lambda f: pkg.channel_cfg(f, bandwidth=8000),
    "NFM plain": lambda f: pkg.channel_cfg(f, modulation=pkg.MOD_NFM),
    "NFM + lowpass": lambda f: pkg.channel_cfg(f, modulation=pkg.MOD_NFM, bandwidth=12500),
    "NFM + lowpass + ctcss": lambda f: pkg.channel_cfg(f, modulation=pkg.MOD_NFM, bandwidth=12500, ctcss=100.0),
}
nbat = 64
for name, mk in kinds.items():
    chans = [mk(f) for f in freqs]
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=1, active=lambda k: True)
    d = pkg.Demod(dev, chans, max_batches=nbat)
    d.process([iq], nbat)
    k1, k2 = d.last_kernel_ms()
    d.close()
    print(f"{name:24s} k_demod {k2:8.2f} ms = {k2 * 1e3 / (nbat * 2000):6.3f} us per step")
