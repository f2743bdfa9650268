"""Latency of the first host-entry calls of a prepared handle (mi_demod_create + mi_demod_prepare), one WAVE_BATCH each -- the
reference's cadence.  Usage: python tools/first_call_latency.py"""
import sys, time
sys.path.insert(0, 'tests')
from conftest import load_package
from common import gen_iq
import numpy as np
pkg = load_package()
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre)
iq, _ = gen_iq(pkg, dev, centre, chans, 6, gate_div=8)
d = pkg.Demod(dev, chans, nstreams=1, max_batches=1, gpu=0)
d.prepare(1)
off = 0
for i in range(5):
    t0 = time.perf_counter()
    wo, axc, _, _ = d.process([iq[off:]], 1)
    dt = time.perf_counter() - t0
    off += d.bytes_consumed(1) if i else d.bytes_consumed(1)
    print(i, round(dt*1e3, 3), "ms")
