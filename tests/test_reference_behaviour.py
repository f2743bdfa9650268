"""The reference's own unit tests for this path, restated against the oracle (and against the compiled
reference where it exists): src/test_squelch.cpp:56-281, src/test_ctcss.cpp:122-155, src/test_filters.cpp:33-41.
They are behavioural inequalities, not vectors -- the only known-answer behaviour the reference holds
for the hot path."""
import numpy as np
import pytest

import libs

NO_SIGNAL = np.float32(0.05)  # raw_no_signal_sample, test_squelch.cpp:32
SIGNAL = np.float32(0.75)     # raw_signal_sample


def apis():
    out = [("oracle", libs.oracle())]
    if libs.ref_available():
        out.append(("reference", libs.ref()))
    return out


def noise_floor_prefix(api, **kw):
    """send_samples_for_noise_floor(): no-signal samples until noise_level <= 1.01 * 0.05 (test_squelch.cpp:40-46)."""
    res = api.squelch_run(np.full(60000, NO_SIGNAL, np.float32), **kw)
    # the C++ loop checks noise_level() BEFORE each sample: stop at the first k with noise[k-1] <= limit
    ok = np.nonzero(res["noise"] <= np.float32(1.01) * NO_SIGNAL)[0]
    assert ok.size, "noise floor never came down"
    k = int(ok[0]) + 1
    assert SIGNAL > res["level"][k - 1]
    return np.full(k, NO_SIGNAL, np.float32)


@pytest.mark.parametrize("name,api", apis())
def test_noise_floor(name, api):
    """test_squelch.cpp:56-79: starts high, decays monotonically towards (never to) the input level."""
    res = api.squelch_run(np.full(40000, NO_SIGNAL, np.float32))
    assert res["noise"][0] > 10.0 * NO_SIGNAL * 0.9  # 5.0 at construction, first update on the very first sample
    every25 = res["noise"][24::25]
    assert np.all(np.diff(every25) <= 0)
    assert res["noise"][-1] < 1.01 * NO_SIGNAL
    assert res["noise"][-1] > NO_SIGNAL


@pytest.mark.parametrize("name,api", apis())
def test_normal_operation(name, api):
    """test_squelch.cpp:81-109: opens within 500 samples of signal, stays open, closes within 100 of no-signal."""
    pre = noise_floor_prefix(api)
    seq = np.concatenate([pre, np.full(1500, SIGNAL, np.float32), np.full(100, NO_SIGNAL, np.float32)])
    f = api.squelch_run(seq)["flags"]
    is_open = (f & 1).astype(bool)
    process = (f & 2).astype(bool)
    n0 = pre.size
    assert is_open[n0:n0 + 500].any()
    first = n0 + int(np.argmax(is_open[n0:]))
    assert is_open[first:n0 + 1500].all() and process[first:n0 + 1500].all()
    assert not is_open[-1] and not process[-1]
    assert not is_open[:n0].any()


@pytest.mark.parametrize("name,api", apis())
def test_dead_spot(name, api):
    """test_squelch.cpp:111-143: a 50-sample dead spot does not close an open squelch."""
    pre = noise_floor_prefix(api)
    seq = np.concatenate([pre, np.full(1500, SIGNAL, np.float32), np.full(50, NO_SIGNAL, np.float32), np.full(1000, SIGNAL, np.float32)])
    f = api.squelch_run(seq)["flags"]
    n0 = pre.size + 1500
    assert ((f[n0:] & 1) == 1).all() and ((f[n0:] & 2) == 2).all()


@pytest.mark.parametrize("name,api", apis())
def test_should_process_audio_tracks_open(name, api):
    """test_squelch.cpp:145-165."""
    pre = noise_floor_prefix(api)
    seq = np.concatenate([pre, np.full(500, SIGNAL, np.float32), np.full(100, NO_SIGNAL, np.float32)])
    f = api.squelch_run(seq)["flags"]
    is_open = (f & 1).astype(bool)
    process = (f & 2).astype(bool)
    assert np.array_equal(is_open, process)  # without CTCSS the two coincide sample for sample
    assert is_open[pre.size:pre.size + 500].any() and not is_open[-1]


def ctcss_sequence(api, expected_tone, actual_tone, n_after=20000):
    rate = 8000.0
    pre = noise_floor_prefix(api, ctcss_freq=expected_tone, ctcss_rate=rate)
    n = pre.size + 500 + n_after
    raw = np.concatenate([pre, np.full(n - pre.size, SIGNAL, np.float32)])
    audio = api.tone_run(8000, actual_tone, 0.2, n)  # GenerateSignal + Tone::NORMAL, tones only
    res = api.squelch_run(raw, audio=audio, ctcss_freq=expected_tone, ctcss_rate=rate)
    return pre.size, res


STANDARD_TONES = [67.0, 69.3, 71.9, 74.4, 77.0, 79.7, 82.5, 85.4]  # first entries of CTCSS::standard_tones


@pytest.mark.parametrize("name,api", apis())
def test_good_ctcss(name, api):
    """test_squelch.cpp:167-203: right tone -> opens (after the fast window) and stays open; no rejections."""
    n0, res = ctcss_sequence(api, STANDARD_TONES[5], STANDARD_TONES[5])
    f = res["flags"]
    is_open, process = (f & 1).astype(bool), (f & 2).astype(bool)
    p0 = n0 + int(np.argmax(process[n0:]))
    assert process[p0] and not is_open[p0]  # audio is processed before the squelch reports open
    o0 = int(np.argmax(is_open))
    assert 0 < o0 - p0 <= 500
    assert is_open[o0:].all()
    assert res["final"].ctcss_count > 0 and res["final"].no_ctcss_count == 0


@pytest.mark.parametrize("name,api", apis())
def test_wrong_ctcss(name, api):
    """test_squelch.cpp:205-238."""
    n0, res = ctcss_sequence(api, STANDARD_TONES[7], STANDARD_TONES[0])
    f = res["flags"]
    assert not (f & 1).any()
    assert (f[n0 + 500:] & 2).all()
    assert res["final"].ctcss_count == 0 and res["final"].no_ctcss_count > 0


@pytest.mark.parametrize("name,api", apis())
def test_close_ctcss(name, api):
    """test_squelch.cpp:240-281: a tone two steps away passes the coarse fast detector, then the slow one closes it."""
    n0, res = ctcss_sequence(api, STANDARD_TONES[7], STANDARD_TONES[5])
    is_open = (res["flags"] & 1).astype(bool)
    assert is_open.any()
    o0 = int(np.argmax(is_open))
    closed_again = o0 + int(np.argmin(is_open[o0:]))
    assert closed_again - o0 <= 3000
    assert not is_open[closed_again:].any()
    assert res["final"].ctcss_count == 0 and res["final"].no_ctcss_count > 0


@pytest.mark.parametrize("name,api", apis())
def test_ctcss_each_standard_tone_detected_only_by_itself(name, api):
    """test_ctcss.cpp:122-155 (has_each_standard_tone), noise-free tones: the reference's noise generator is
    seeded from std::random_device, so its noisy variant is not reproducible (generate_signal.cpp:41-46)."""
    tones = [67.0, 69.3, 71.9, 74.4, 77.0, 79.7, 82.5, 85.4, 88.5, 91.5, 94.8, 97.4, 100.0, 103.5, 107.2, 110.9, 114.8, 118.8, 123.0,
             127.3, 131.8, 136.5, 141.3, 146.2, 150.0, 151.4, 156.7, 159.8, 162.2, 165.5, 167.9, 171.3, 173.8, 177.3, 179.9, 183.5,
             186.2, 189.9, 192.8, 196.6, 199.5, 203.5, 206.5, 210.7, 218.1, 225.7, 229.1, 233.6, 241.8, 250.3, 254.1]
    rate, slow = 8000, int(8000 * 0.4)
    for tone in tones[::5]:
        x = api.tone_run(rate, tone, 0.2, slow)
        for det in tones:
            flags, found, nfound = api.ctcss_run(det, float(rate), slow, x)
            has_tone = bool(flags[-1] & 1)
            if abs(np.float32(det) - np.float32(tone)) < 5:
                if det == tone:
                    assert has_tone, f"tone {tone} not found by its own detector"
            else:
                assert not has_tone, f"tone {tone} found by detector {det}"


def test_default_filters_are_disabled():
    """test_filters.cpp:33-41: a disabled filter passes samples through untouched."""
    o = libs.oracle()
    x = np.linspace(-1, 1, 64).astype(np.float32)
    assert np.array_equal(o.notch_run(0.0, 16000.0, 10.0, x), x)   # freq <= 0 disables (filters.cpp:31-35)
    re, im = o.lowpass_run(0.0, 16000.0, x, x[::-1].copy())
    assert np.array_equal(re, x) and np.array_equal(im, x[::-1])
