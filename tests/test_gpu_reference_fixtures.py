"""The reference's own vectors fed to the HIP stage 2 directly.

tests/golden/components_ref.npz holds what the reference's squelch.cpp / filters.cpp (compiled unmodified, oracle/_ref, by
tests/golden/gen_golden.py) produce on the seeded signals of tests/golden_inputs.py.  The other parity tests compare HIP with the
oracle on IQ-derived signals, and the oracle with those vectors: two links on different inputs.  Here the same seeded signals go
through mi_demod_process_planes (stage 2 over caller-supplied planes, the same kernels and paths as mi_demod_process) and the
outputs are compared with the compiled reference's -- no oracle in between.

What a channel configuration can express, and what the C ABI shows of it:
  SQUELCH_CASES  bursts, flap, dropout, ramp (default squelch), snr0 (squelch_snr_threshold = 0): the raw magnitudes are the
                 channel's wavein.  Compared: per-sample is_open (audio != 0), per-batch axcindicate, noise / signal / squelch level
                 after every WAVE_BATCH (one call per batch: the serial kernel) and at the end of one long call (the time-parallel
                 path), open_count, flappy_count.
                 Not expressible: manual (a level of exactly 0.6 is not a whole dBFS value), *_filt / snr3 (the post-filter
                 magnitude is computed from I/Q by the channel itself), ctcss_* (the tone detector hears the demodulated audio).
  FILTER_CASES   lp6250, lp2500, lp4000: the channel sits on the centre frequency (dm_dphi = 0: the derotation multiplies by 1 and 0)
                 and is kept filtering by a strong constant raw magnitude; its raw I/Q planes are the filter's input sequence and
                 the magnitude the loop writes back to wavein (rtl_airband.cpp:548) is |output|.
                 Not expressible: notch* (the notch filters the demodulated audio), CTCSS_CASES (as above).
"""
import os

import numpy as np
import pytest

from common import AGC_EXTRA, WAVE_BATCH
from golden_inputs import FILTER_CASES, SQUELCH_CASES, make_raw

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "components_ref.npz"))
CENTRE = 120000000


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def _handle(pkg, case, max_batches):
    kw = {}
    if case.get("snr_db") is not None:
        kw["squelch_snr_db"] = case["snr_db"]
    chans = [pkg.channel_cfg(CENTRE + 250000, **kw)]
    return pkg.Demod(pkg.device_cfg(centerfreq=CENTRE), chans, nstreams=1, max_batches=max_batches, gpu=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bursts", "flap", "dropout", "ramp", "snr0"])
def test_squelch_on_the_gpu_equals_the_compiled_reference(pkg, name):
    case = SQUELCH_CASES[name]
    raw = make_raw(case)
    n = raw.size
    assert n % WAVE_BATCH == 0
    nb = n // WAVE_BATCH
    flags, level, noise, signal = (GOLD[f"sq_{name}_{k}"] for k in ("flags", "level", "noise", "signal"))
    is_open = (flags & 1).astype(bool)
    counts, final_levels = GOLD[f"sq_{name}_final_counts"], GOLD[f"sq_{name}_final_levels"]
    # entry AGC_EXTRA + i of wavein is the squelch's sample of step i; the first AGC_EXTRA entries only ever serve as audio
    pad = np.full(AGC_EXTRA, 0.2, np.float32)
    seq = np.concatenate([pad, raw])

    # ---- one call per WAVE_BATCH: the serial per-channel kernel, levels checked after every batch
    d = _handle(pkg, case, 1)
    mask = np.zeros(n, bool)
    for b in range(nb):
        lo = 0 if b == 0 else AGC_EXTRA + b * WAVE_BATCH
        wo, axc, _, st = d.process_planes(seq[lo:AGC_EXTRA + (b + 1) * WAVE_BATCH][None, :], 1)
        assert d.last_path()[0] == 0
        mask[b * WAVE_BATCH:(b + 1) * WAVE_BATCH] = wo[0, 0, AGC_EXTRA:] != 0  # audio entry AGC_EXTRA + i belongs to step i of the call
        end = (b + 1) * WAVE_BATCH - 1
        got = bits([st[0].noise_level, st[0].signal_level, st[0].squelch_level])
        want = bits([noise[end], signal[end], level[end]])
        assert np.array_equal(got, want), f"{name}: levels after batch {b}: {got} vs the reference's {want}"
        assert (axc[0, 0, 0] != ord(" ")) == is_open[b * WAVE_BATCH:(b + 1) * WAVE_BATCH].any(), f"{name}: axcindicate of batch {b}"
    assert np.array_equal(mask, is_open), f"{name}: {np.count_nonzero(mask != is_open)} squelch decisions differ from the compiled reference"
    assert [st[0].open_count, st[0].flappy_count] == list(counts[:2])
    d.close()

    # ---- the same signal in one call: the time-parallel path
    d = _handle(pkg, case, nb)
    wo, axc, _, st = d.process_planes(seq[None, :], nb)
    assert d.last_path() == (1, 0), "a plain AM channel over >= 8 batches takes the time-parallel path, every segment verified"
    assert np.array_equal(wo[0, 0, AGC_EXTRA:] != 0, is_open), f"{name}: squelch decisions of the time-parallel path"
    assert np.array_equal(axc[0, 0] != ord(" "), is_open.reshape(nb, WAVE_BATCH).any(axis=1))
    assert np.array_equal(bits([st[0].noise_level, st[0].signal_level, st[0].squelch_level]), bits(final_levels))
    assert [st[0].open_count, st[0].flappy_count] == list(counts[:2])
    assert is_open.any() and not is_open.all()
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lp6250", "lp2500", "lp4000"])
def test_lowpass_on_the_gpu_equals_the_compiled_reference(pkg, name):
    case = FILTER_CASES[name]
    n = case["n"]
    nb = n // WAVE_BATCH
    rng = np.random.default_rng(case["seed"])  # the input sequence of tests/test_oracle_components.py / gen_golden.py
    x = rng.normal(size=n).astype(np.float32)
    y = rng.normal(size=n).astype(np.float32)
    want_re, want_im = GOLD[f"fl_{name}_re"], GOLD[f"fl_{name}_im"]
    # on the centre frequency the derotation is the identity; snr threshold 0 dB and a strong constant raw magnitude keep
    # has_pre_filter_signal() true from the first step (squelch.cpp:136-138), so every I/Q sample goes through the filter in order
    chans = [pkg.channel_cfg(CENTRE, bandwidth=int(2 * case["freq"]), squelch_snr_db=0.0, has_iq_outputs=1)]
    d = pkg.Demod(pkg.device_cfg(centerfreq=CENTRE), chans, nstreams=1, max_batches=nb, gpu=0)
    count = n + AGC_EXTRA
    mag = np.full((1, count), 1.0e4, np.float32)
    cplx = np.zeros((1, count, 2), np.float32)
    cplx[0, :n, 0], cplx[0, :n, 1] = x, y  # the filter takes entry i at step i (iq_in[2 * (j - AGC_EXTRA)], rtl_airband.cpp:526)
    wo, axc, iqo, st = d.process_planes(mag, nb, cplx=cplx, want_iq=True)
    got, _ = d.read_planes(0, 0, AGC_EXTRA, n)  # wavein[j] after the loop: the magnitude of the filtered sample (:548)
    want = np.sqrt(want_re * want_re + want_im * want_im)  # float32 throughout, like the loop
    assert np.array_equal(bits(got), bits(want)), f"{name}: {np.count_nonzero(bits(got) != bits(want))} of {n} filtered magnitudes differ"
    # where the squelch is open the filtered I/Q itself is emitted (rawfile output, :636-637)
    opened = (iqo[0, 0] != 0).any(axis=1)
    if opened.any():
        assert np.array_equal(bits(iqo[0, 0, opened, 0]), bits(want_re[opened])) and np.array_equal(bits(iqo[0, 0, opened, 1]), bits(want_im[opened]))
    d.close()
