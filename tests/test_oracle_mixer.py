"""The mixer restatement (src/mixer.cpp:56-98, 133-140, 190-213).  mixer.cpp cannot be built here (it needs the
reference's umbrella header with lame / shout / libconfig), so parity is unpinned by the reference; the restatement is
checked against the arithmetic written out in numpy, operation for operation."""
import numpy as np

import libs
from common import WAVE_BATCH


def numpy_mixer(inputs, wave, axc, nb):
    stereo = any(b != 0.0 for _, _, b in inputs)
    left = np.zeros(nb * WAVE_BATCH, np.float32)
    right = np.zeros(nb * WAVE_BATCH, np.float32)
    out = np.full(nb, ord(" "), np.uint8)
    for b in range(nb):
        sl = slice(b * WAVE_BATCH, (b + 1) * WAVE_BATCH)
        for row, amp, bal in inputs:
            if axc[row, b] == ord(" "):
                continue
            ampl = np.float32(min(np.float32(1.0), np.float32(1.0) - np.float32(bal)))
            ampr = np.float32(min(np.float32(1.0), np.float32(1.0) + np.float32(bal)))
            ml, mr = np.float32(amp) * ampl, np.float32(amp) * ampr
            if ml != 0:
                left[sl] = left[sl] + wave[row, sl] * ml
            if stereo and mr != 0:
                right[sl] = right[sl] + wave[row, sl] * mr
            out[b] = ord("*")
    return left, (right if stereo else None), out


def case(seed, rows=6, nb=5):
    rng = np.random.default_rng(seed)
    wave = rng.uniform(-1, 1, (rows, nb * WAVE_BATCH + 100)).astype(np.float32)
    axc = np.where(rng.random((rows, nb)) < 0.6, ord("*"), ord(" ")).astype(np.uint8)
    axc[1, 2] = ord("<")  # AFC indicators count as signal (output.cpp:564)
    axc[:, 3] = ord(" ")  # a batch nobody talks in
    return wave, axc


def test_mono_mixer_sums_in_input_order():
    wave, axc = case(1)
    inputs = [(0, 1.0, 0.0), (3, 0.5, 0.0), (1, 2.0, 0.0), (5, 0.0, 0.0)]
    l, r, out = libs.oracle_mixer(inputs, wave, axc, 5)
    el, er, eout = numpy_mixer(inputs, wave, axc, 5)
    assert r is None and er is None
    assert np.array_equal(l, el) and np.array_equal(out, eout)
    assert not l[3 * WAVE_BATCH:4 * WAVE_BATCH].any() and out[3] == ord(" ")


def test_stereo_mixer_balance():
    wave, axc = case(2)
    inputs = [(0, 1.0, -1.0), (2, 1.0, 1.0), (4, 0.7, 0.25), (1, 1.0, 0.0)]
    l, r, out = libs.oracle_mixer(inputs, wave, axc, 5)
    el, er, eout = numpy_mixer(inputs, wave, axc, 5)
    assert r is not None
    assert np.array_equal(l, el) and np.array_equal(r, er) and np.array_equal(out, eout)
    # balance -1: fully left (ampr = 0 -> skipped on the right); balance +1: fully right
    only0 = (axc[0] != ord(" ")) & (axc[2] == ord(" ")) & (axc[4] == ord(" ")) & (axc[1] == ord(" "))
    for b in np.nonzero(only0)[0]:
        sl = slice(b * WAVE_BATCH, (b + 1) * WAVE_BATCH)
        assert np.array_equal(l[sl], wave[0, sl]) and not r[sl].any()
