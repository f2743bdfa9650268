#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (mi_demod_process): capture in host memory -> pinned staging -> device ->
audio back to host, one synchronous call at a time (the note next to `value`)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import load_package  # noqa: E402
from common import AGC_EXTRA, WAVE_BATCH, gen_iq  # noqa: E402

pkg = load_package()
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre)
nbat = int(sys.argv[1]) if len(sys.argv) > 1 else 128  # WAVE_BATCHes per call (128 = 16 s of signal; 1 = the reference's own cadence)
calls = 6 if nbat >= 16 else 40
iq, _ = gen_iq(pkg, dev, centre, chans, nbat * calls, gate_div=1)
d = pkg.Demod(dev, chans, max_batches=nbat)
t = []
for call in range(calls):
    pos = 0 if call == 0 else (call * nbat * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
    t0 = time.perf_counter()
    d.process([iq[pos:]], nbat)
    t.append(time.perf_counter() - t0)
d.close()
per = sorted(t[1:])[len(t[1:]) // 2]
samples = nbat * WAVE_BATCH * 160
print(f"host entry: {per * 1e3:.3f} ms per {nbat / 8:.3f} s call = {samples / per / 1e9:.2f} GS/s = {samples / per / 2.56e6:.0f} x real time "
      f"({2 * samples / per / 1e9:.1f} GB/s of u8 IQ uploaded)")
