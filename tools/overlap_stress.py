#!/usr/bin/env python3
"""Determinism stress for the overlapped pipeline: the same sequence of device-resident calls, enqueued back to back with
MI_OPT_EARLY_INPUT, must give bit-identical audio every time (a race between a call's tail and the next call's early
stages would show up as a difference).  Compared against a run without the option.
STRESS_MODE=serial: a mixed 32-channel plan, whose calls take the serial kernel and overlap on two alternating plane sets."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from bench import load_package  # noqa: E402
from common import AGC_EXTRA, WAVE_BATCH, gen_iq  # noqa: E402

pkg = load_package()
serial = os.environ.get("STRESS_MODE") == "serial"
centre, chans = pkg.config3_channels() if serial else pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre)
calls = [6, 9, 1, 12, 3, 6, 2, 7] if serial else [64, 96, 64, 128, 32, 64]
nbat = sum(calls)
iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3, **(dict(amp_q8=1024, active=lambda k: k % 4 != 2) if serial else {}))
pad = (iq.size + 255) // 256 * 256
d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
d_iq[:iq.size] = torch.from_numpy(iq).cuda()
s = torch.cuda.current_stream().cuda_stream


def run(early):
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1 if early else 0)
    outs, flags, done = [], [], 0
    for k in calls:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
        ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
        d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
        outs.append(wo)
        flags.append(ax)
        done += k
    torch.cuda.synchronize()
    st = d.stats()
    d.close()
    return torch.cat(outs, dim=2).cpu().numpy(), torch.cat(flags, dim=2).cpu().numpy(), [(x.open_count, x.agcavgfast, x.noise_level) for x in st]


ref = run(False)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    got = run(True)
    same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    bad += 0 if same else 1
print(f"overlap stress: {bad} mismatching runs")
sys.exit(1 if bad else 0)
