"""One-off fuzz of the time-parallel path: random plain-AM plans (SNR / manual thresholds, amplification) and captures
(carriers from under the squelch level to clipping, random gate periods and phases), one 16-batch call (time-parallel) against
the same capture in 4-batch calls (serial kernel): audio, flags and statistics must be identical.  Then the same capture as
two overlapping 8-batch device calls (MI_OPT_EARLY_INPUT, two audio buffers: speculative head, chain handed over on the
device), with the submit / wait host entry (three calls in flight) as a third reading."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from common import AGC_EXTRA, WAVE_BATCH, bytes_for_batches  # noqa: E402

pkg = conftest.load_package()
bad = 0
first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    centre = 120000000
    nchan = int(rng.integers(2, 12))
    chans, carriers = [], []
    for k in range(nchan):
        f = centre - 1200000 + 60000 + k * 200000 + int(rng.integers(0, 20)) * 5000
        kw = {}
        r = rng.random()
        if r < 0.25:
            kw["squelch_threshold_dbfs"] = int(rng.integers(-55, -30))
        elif r < 0.6:
            kw["squelch_snr_db"] = float(rng.choice([0.0, 1.0, 3.0, 6.0, 12.0]))
        if rng.random() < 0.3:
            kw["ampfactor"] = float(rng.choice([0.5, 2.0, 4.0]))
        chans.append(pkg.channel_cfg(f, **kw))
        if rng.random() < 0.8:
            carriers.append((f - centre, 0, int(rng.choice([120, 250, 500, 1000, 2500, 6000])), int(rng.integers(0, 1000))))
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=int(rng.choice([8, 9, 10])))
    nbat = 16
    n = bytes_for_batches(dev, nbat) // 2
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, seed=5000 + seed, gate_samples=dev.sample_rate // int(rng.integers(2, 12)), carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, n)
    d = pkg.Demod(dev, chans, max_batches=nbat)
    wo_a, ax_a, _, _ = d.process([iq], nbat)
    path = d.last_path()[0]
    st_a = bytes(d.stats())
    d.close()
    e = pkg.Demod(dev, chans, max_batches=4)
    outs, flags = [], []
    for call in range(4):
        pos = 0 if call == 0 else (call * 4 * WAVE_BATCH + AGC_EXTRA) * e.hop_bytes
        wo, ax, _, _ = e.process([iq[pos:]], 4)
        outs.append(wo[:, :, :4 * WAVE_BATCH].copy())
        flags.append(ax.copy())
    st_b = bytes(e.stats())
    e.close()
    wo_b = np.concatenate(outs, axis=2)
    ax_b = np.concatenate(flags, axis=2)
    same = path == 1 and np.array_equal(wo_a[:, :, :nbat * WAVE_BATCH], wo_b) and np.array_equal(ax_a, ax_b) and st_a == st_b
    # overlapping device calls
    import torch
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    # (the NULL stream, or one of the caller's own: the wide passes then run on the CU-restricted streams, MI_OPT_RESERVE_CUS)
    side_ = torch.cuda.Stream() if seed % 3 == 0 else None
    s_ = side_.cuda_stream if side_ is not None else torch.cuda.current_stream().cuda_stream
    # (call sizes vary with the seed; every call has at least 8 batches, so all of them are time-parallel)
    sizes = [[8, 8], [16], [8, 8], [8, 8]][seed % 4] if nbat == 16 else [nbat]
    f = pkg.Demod(dev, chans, max_batches=max(sizes))
    f.set_option(pkg.OPT_EARLY_INPUT, 1)
    if seed % 5 == 0:
        f.set_option(pkg.OPT_SPEC_HEAD, 0)
    if seed % 7 == 0:
        f.set_option(pkg.OPT_CORE_SPLIT, 0)
    outs, flags, paths, done = [], [], [], 0
    for k in sizes:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * f.hop_bytes
        wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
        ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
        f.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s_)
        paths.append(f.last_path()[0])
        outs.append(wo)
        flags.append(ax)
        done += k
    torch.cuda.synchronize()
    st_c = bytes(f.stats())
    f.close()
    wo_c = torch.cat(outs, dim=2).cpu().numpy()
    ax_c = torch.cat(flags, dim=2).cpu().numpy()
    same = same and all(p_ == 1 for p_ in paths) and np.array_equal(wo_c, wo_b) and np.array_equal(ax_c, ax_b) and st_c == st_b
    # host entry, three calls in flight
    g = pkg.Demod(dev, chans, max_batches=8)
    for call in range(2):
        pos = 0 if call == 0 else (call * 8 * WAVE_BATCH + AGC_EXTRA) * g.hop_bytes
        g.submit([iq[pos:]], 8)
    r0, r1 = g.wait(), g.wait()
    g.close()
    wo_d = np.concatenate([r0[0][:, :, :8 * WAVE_BATCH], r1[0][:, :, :8 * WAVE_BATCH]], axis=2)
    same = same and np.array_equal(wo_d, wo_b) and np.array_equal(np.concatenate([r0[1], r1[1]], axis=2), ax_b) and bytes(r1[3]) == st_b
    if not same:
        bad += 1
        print("seed", seed, "MISMATCH (path", path, ")", flush=True)
    if seed % 100 == 99:
        print("... seed", seed, "failures so far:", bad, flush=True)
print("tp fuzz done, failures:", bad)
