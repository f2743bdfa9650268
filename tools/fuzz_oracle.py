"""One-off fuzz against the oracle: the random plans and captures of test_steady_blocks_random_plans (every channel type, odd
thresholds, carriers from under the squelch level to clipping), product vs oracle: audio, flags, raw I/Q bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from common import AGC_EXTRA, WAVE_BATCH, bytes_for_batches, oracle_run  # noqa: E402

pkg = conftest.load_package()
bad = 0
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 40)
for seed in range(first, last):
    rng = np.random.default_rng(10000 + seed)
    centre = 120000000
    nchan = int(rng.integers(3, 20))
    chans, carriers = [], []
    for k in range(nchan):
        f = centre - 1200000 + 40000 + k * 120000 + int(rng.integers(0, 20)) * 5000
        kw = {}
        nfm = rng.random() < 0.5
        if nfm:
            kw["modulation"] = pkg.MOD_NFM
        if rng.random() < 0.5:
            kw["bandwidth"] = int(rng.choice([5000, 8000, 12500]))
        if rng.random() < 0.25:
            kw["notch"] = float(rng.choice([100.0, 400.0, 1000.0]))
        if nfm and rng.random() < 0.4:
            kw["ctcss"] = float(rng.choice([100.0, 123.0, 151.4]))
        r = rng.random()
        if r < 0.2:
            kw["squelch_threshold_dbfs"] = int(rng.integers(-55, -30))
        elif r < 0.5:
            kw["squelch_snr_db"] = float(rng.choice([0.0, 3.0, 6.0, 12.0]))
        if rng.random() < 0.3:
            kw["ampfactor"] = float(rng.choice([0.5, 2.0, 4.0]))
        if rng.random() < 0.3:
            kw["has_iq_outputs"] = 1
        chans.append(pkg.channel_cfg(f, **kw))
        if rng.random() < 0.8:
            carriers.append((f - centre, int(rng.integers(0, 3)), int(rng.choice([150, 300, 600, 1200, 2500, 5000])), int(rng.integers(0, 1000))))
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=int(rng.choice([8, 9, 10])), fm_quadri=int(seed % 2))
    nbat, per_call = 8, int(rng.choice([1, 2, 4, 8]))
    n = bytes_for_batches(dev, nbat) // 2
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, seed=20000 + seed, gate_samples=dev.sample_rate // int(rng.integers(3, 9)), carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, n)
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    d = pkg.Demod(dev, chans, max_batches=per_call)
    outs, flags, zs = [], [], []
    for call in range(nbat // per_call):
        pos = 0 if call == 0 else (call * per_call * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo, ax, zo, _ = d.process([iq[pos:]], per_call, want_iq=True)
        outs.append(wo[:, :, :per_call * WAVE_BATCH].copy())
        flags.append(ax.copy())
        zs.append(zo.copy())
    d.close()
    wo = np.concatenate(outs, axis=2)
    ax = np.concatenate(flags, axis=2)
    zo = np.concatenate(zs, axis=2)
    same = nb == nbat and np.array_equal(ax[0], oaxc) and np.array_equal(wo[0], owo)
    for c, ch in enumerate(chans):
        if ch.has_iq_outputs:
            same = same and np.array_equal(zo[0, c].reshape(-1), oiq[c])
    if not same:
        bad += 1
        print("seed", seed, "MISMATCH", flush=True)
    if seed % 50 == 49:
        print("... seed", seed, "failures so far:", bad, flush=True)
print("oracle fuzz done, failures:", bad)
