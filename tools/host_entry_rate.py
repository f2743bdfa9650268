#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entries (the note next to `value`): capture in host memory ->
device -> audio back to host.  Three ways: mi_demod_process from pageable memory (staging memcpy + upload + compute + download,
one after the other), mi_demod_process from page-locked memory (no staging copy), and mi_demod_submit / mi_demod_wait from
page-locked memory with two / three calls in flight (upload of call k+1 under the compute of call k).

    python tools/host_entry_rate.py [batches per call, default 128 = 16 s of signal; 1 = the reference's own cadence]"""
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
nbat = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = 8 if nbat >= 16 else 40
iq, _ = gen_iq(pkg, dev, centre, chans, nbat * calls, gate_div=1)
pin = pkg.PinnedBuffer(iq.size)
pin.array[:] = iq
samples = nbat * WAVE_BATCH * 160


def report(name, per):
    print(f"{name}: {per * 1e3:.3f} ms per {nbat / 8:.3f} s call = {samples / per / 1e9:.2f} GS/s = {samples / per / 2.56e6:.0f} x real time "
          f"({2 * samples / per / 1e9:.1f} GB/s of u8 IQ uploaded)")


def pos(d, call):
    return 0 if call == 0 else (call * nbat * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes


for name, src in (("mi_demod_process, pageable source", None), ("mi_demod_process, page-locked source", pin)):
    d = pkg.Demod(dev, chans, max_batches=nbat)
    t = []
    for call in range(calls):
        p = pos(d, call)
        t0 = time.perf_counter()
        if src is None:
            d.process([iq[p:]], nbat, want_stats=False)
        else:
            d.submit([pin.view(p)], nbat, want_stats=False)
            d.wait()
        t.append(time.perf_counter() - t0)
    d.close()
    report(name, sorted(t[1:])[len(t[1:]) // 2])

import numpy as np  # noqa: E402
nwave = len(chans) * (nbat * WAVE_BATCH + AGC_EXTRA)
outpin = pkg.PinnedBuffer(3 * nwave * 4)  # three page-locked audio buffers, used in turn
outs = [outpin.array[i * nwave * 4:(i + 1) * nwave * 4].view(np.float32).reshape(1, len(chans), -1) for i in range(3)]
for depth in (2, 3):  # calls kept in flight (the library has three staging slots)
    d = pkg.Demod(dev, chans, max_batches=nbat)
    for call in range(depth - 1):
        d.submit([pin.view(pos(d, call))], nbat, want_stats=False, waveout=outs[call % 3])
    t0 = None
    for call in range(depth - 1, calls):
        d.submit([pin.view(pos(d, call))], nbat, want_stats=False, waveout=outs[call % 3])
        d.wait()
        if call == depth - 1:
            t0 = time.perf_counter()  # steady state from here on
    for _ in range(depth - 1):
        d.wait()
    per = (time.perf_counter() - t0) / (calls - 1)
    d.close()
    report(f"mi_demod_submit / mi_demod_wait, page-locked source and audio buffers, {depth} calls in flight", per)
pin.free()
