#!/usr/bin/env python3
"""Timeline of consecutive overlapped time-parallel calls (BASELINE configs[1] by default), from the library's own HIP events --
no profiler in the way.  Prints, per call, when each launch began / ended relative to the start of the core chain of the call
three before the last one.  Usage: python tools/call_timeline.py [calls] [--noise-only] [--streams N] [--batches B]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("boondock_airband_amd", os.path.join(ROOT, "boondock-airband_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
sys.modules["boondock_airband_amd"] = pkg
spec.loader.exec_module(pkg)

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
noise_only = "--noise-only" in sys.argv
nstreams = int(sys.argv[sys.argv.index("--streams") + 1]) if "--streams" in sys.argv else 1
HOP, AGC_EXTRA, WAVE_BATCH = 160, 100, 2000
nbat = int(sys.argv[sys.argv.index("--batches") + 1]) if "--batches" in sys.argv else 512  # 64 s of signal per call
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
cfg = pkg.iqgen_cfg(carriers=() if noise_only else pkg.carriers_for(centre, chans))
nsteps = nbat * WAVE_BATCH
nbytes = ((nsteps + AGC_EXTRA) * 2 * HOP + 2 * 512 + 255) // 256 * 256
d_iq = torch.empty((nstreams, nbytes), dtype=torch.uint8, device="cuda")
pkg.iqgen_device(cfg, 0, nstreams, nbytes, 0, nbytes // 2, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
wo = [torch.empty((nstreams, 8, nsteps), dtype=torch.float32, device="cuda") for _ in range(3)]
ax = [torch.empty((nstreams, 8, nbat), dtype=torch.uint8, device="cuda") for _ in range(3)]
h = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=nbat)
h.set_option(pkg.OPT_EARLY_INPUT, 1)
st = torch.cuda.current_stream()
h.process_device(d_iq.data_ptr(), nbytes, nbat, wo[0].data_ptr(), ax[0].data_ptr(), hip_stream=st.cuda_stream)
base = d_iq.data_ptr() + AGC_EXTRA * HOP * 2
for k in range(ncalls):
    h.process_device(base, nbytes, nbat, wo[k % 3].data_ptr(), ax[k % 3].data_ptr(), hip_stream=st.cuda_stream)
torch.cuda.synchronize()
cols = [("s1", 0, 1), ("full", 11, 2), ("core", 3, 4), ("seg", 5, 12), ("scan", 10, 7), ("fix", 7, 8), ("rest", 8, 9)]
print("ms from the core start of the call 3 back;   " + "   ".join(f"{n:>13s}" for n, _, _ in cols))
for age in (3, 2, 1, 0):
    for chunk in range(8):
        if h.event_ms(3, age, chunk, 0) is None:
            break
        row = []
        for n, a, b in cols:
            ta, tb = h.event_ms(3, age, chunk, a), h.event_ms(3, age, chunk, b)
            row.append("      -      " if ta is None else f"{ta:6.2f}-{tb:6.2f}")
        print(f"call -{age} chunk {chunk}:                              " + "   ".join(row))
h.close()
