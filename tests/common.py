"""Shared helpers for the parity tests: build the same configuration for the oracle and for the
product, generate the seeded synthetic IQ and compare bit for bit."""
import numpy as np

import libs
from conftest import load_package

WAVE_BATCH = 2000
AGC_EXTRA = 100


def to_oracle_cfg(dev, chans):
    """The product's and the oracle's config structs have identical fields; copy value by value."""
    odev = libs.DeviceCfg(dev.sample_rate, dev.centerfreq, dev.fft_size_log, dev.sfmt, dev.fullscale, dev.tau, dev.fm_quadri)
    ochans = [libs.ChannelCfg(c.freq, c.modulation, c.squelch_threshold_dbfs, c.has_snr_threshold, c.squelch_snr_db, c.notch_freq,
                              c.notch_q, c.ctcss_freq, c.bandwidth, c.ampfactor, c.tau, c.afc, c.has_iq_outputs) for c in chans]
    return odev, ochans


def oracle_run(dev, chans, iq, nbatches, want_iq=False):
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nb, wo, axc, iqo = od.run(iq, nbatches, want_iq=want_iq)
    od.close()
    return nb, wo, axc, iqo


def bytes_for_batches(dev, nbatches, bytes_per_sample=1):
    """Bytes a linear capture must hold so the reference's availability rule (rtl_airband.cpp:417) lets
    `nbatches` complete: the last window needs hop + 2*bps*fft_size bytes remaining."""
    hop = 2 * bytes_per_sample * int(round(dev.sample_rate / 16000))
    nfft = nbatches * WAVE_BATCH + AGC_EXTRA
    return (nfft - 1) * hop + hop + 2 * bytes_per_sample * (1 << dev.fft_size_log)


def gen_iq(pkg, dev, centre, chans, nbatches, stream=0, gate_div=4, amp_q8=3072, active=None, seed=0xA1B2C3D4):
    n = bytes_for_batches(dev, nbatches) // 2
    carriers = pkg.carriers_for(centre, chans, amp_q8=amp_q8, **({} if active is None else {"active": active}))
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, seed=seed, gate_samples=dev.sample_rate // gate_div, carriers=carriers)
    return pkg.iqgen_host(cfg, stream, 0, n), cfg


def assert_same(a, b, what):
    """Bit-exact for floats up to the sign of zero (== semantics); NaNs are not expected anywhere."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} differ, first at {first}: {a[first]!r} vs {b[first]!r}")


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))
