#!/usr/bin/env python3
"""Steady-state timeline of overlapped time-parallel calls (BASELINE configs[1]): like tools/call_timeline.py, but every call's
events are read three calls later (as bench.py reads its kernel times), so the table covers the middle of a long run, not its drain.
Usage: python tools/steady_timeline.py [calls=28]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("boondock_airband_amd", os.path.join(ROOT, "boondock-airband_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
sys.modules["boondock_airband_amd"] = pkg
spec.loader.exec_module(pkg)

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 28
HOP, AGC_EXTRA, WAVE_BATCH, nbat = 160, 100, 2000, 512
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
cfg = pkg.iqgen_cfg(carriers=pkg.carriers_for(centre, chans))
nsteps = nbat * WAVE_BATCH
nbytes = ((nsteps + AGC_EXTRA) * 2 * HOP + 2 * 512 + 255) // 256 * 256
d_iq = torch.empty((1, nbytes), dtype=torch.uint8, device="cuda")
pkg.iqgen_device(cfg, 0, 1, nbytes, 0, nbytes // 2, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
wo = [torch.empty((1, 8, nsteps), dtype=torch.float32, device="cuda") for _ in range(3)]
ax = [torch.empty((1, 8, nbat), dtype=torch.uint8, device="cuda") for _ in range(3)]
h = pkg.Demod(dev, chans, nstreams=1, max_batches=nbat)
h.set_option(pkg.OPT_EARLY_INPUT, 1)
st = torch.cuda.Stream() if os.environ.get("SIDE_STREAM") else torch.cuda.current_stream()
torch.cuda.synchronize()
h.process_device(d_iq.data_ptr(), nbytes, nbat, wo[0].data_ptr(), ax[0].data_ptr(), hip_stream=st.cuda_stream)
base = d_iq.data_ptr() + AGC_EXTRA * HOP * 2
cols = [("s1", 0, 1), ("full", 11, 2), ("core", 3, 4), ("seg", 5, 12), ("scan", 10, 7), ("fix+redo", 7, 8), ("rest", 8, 9)]
rows, t_abs = [], 0.0
late = bool(os.environ.get("LATE_READ"))  # no reads while the calls are submitted: the last four calls' events after the loop
for k in range(ncalls):
    h.process_device(base, nbytes, nbat, wo[k % 3].data_ptr(), ax[k % 3].data_ptr(), hip_stream=st.cuda_stream)
    if k >= 3 and not late:  # the call three back, relative to its own core start; and the start of the next core after it
        ev = [(h.event_ms(3, 3, 0, a), h.event_ms(3, 3, 0, b)) for _, a, b in cols]
        nxt = h.event_ms(3, 2, 0, 3)
        rows.append((k - 3, t_abs, ev))
        t_abs += nxt if nxt is not None else 0.0
torch.cuda.synchronize()
if late:
    print("ms relative to the core start of the call three before the last (no event reads during the run):   " + "   ".join(f"{n:>13s}" for n, _, _ in cols))
    for age in (3, 2, 1, 0):
        ev = [(h.event_ms(3, age, 0, a), h.event_ms(3, age, 0, b)) for _, a, b in cols]
        print(f"call last-{age}:           " + "   ".join("      -      " if a is None or b is None else f"{a:6.2f}-{b:6.2f}" for a, b in ev))
    h.close()
    sys.exit(0)
print("absolute ms (core start of call 3 = 0):   " + "   ".join(f"{n:>13s}" for n, _, _ in cols))
for k, t0, ev in rows[4:]:
    print(f"call {k:3d} core starts {t0:7.2f}:           " + "   ".join("      -      " if a is None else f"{t0 + a:6.2f}-{t0 + b:6.2f}" for a, b in ev))
period = (rows[-1][1] - rows[4][1]) / (len(rows) - 5)
print(f"mean period between core starts: {period:.3f} ms")
h.close()
