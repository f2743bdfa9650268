"""Pins the oracle's Squelch / CTCSS / NotchFilter / LowpassFilter restatements:
  - against the committed fixtures generated from the REFERENCE's own compiled sources
    (tests/golden/components_ref.npz, made by tests/golden/gen_golden.py), everywhere;
  - against the live reference build oracle/_ref when it is present (this container).
Bit-exact: flags, squelch level, noise floor, signal level, counters, filter outputs."""
import os

import numpy as np
import pytest

import libs
from golden_inputs import CTCSS_CASES, FILTER_CASES, SQUELCH_CASES, make_audio, make_filtered, make_raw

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "components_ref.npz"))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def run_squelch_case(api, case):
    raw = make_raw(case)
    filt = make_filtered(case, raw) if case.get("filt") else None
    audio = make_audio(case) if case.get("audio") else None
    return api.squelch_run(raw, filt=filt, audio=audio, manual_level=case.get("manual_level", 0.0), snr_db=case.get("snr_db"),
                           ctcss_freq=case.get("ctcss_freq", 0.0), ctcss_rate=case.get("ctcss_rate", 16000))


@pytest.mark.parametrize("name", sorted(SQUELCH_CASES))
def test_squelch_matches_reference_fixture(name):
    res = run_squelch_case(libs.oracle(), SQUELCH_CASES[name])
    for key in ("flags", "level", "noise", "signal"):
        assert np.array_equal(bits(res[key]), bits(GOLD[f"sq_{name}_{key}"])), f"{name}: {key} differs from the reference"
    f = res["final"]
    assert [f.open_count, f.flappy_count, f.ctcss_count, f.no_ctcss_count] == list(GOLD[f"sq_{name}_final_counts"])
    assert np.array_equal(bits(np.array([f.noise_level, f.signal_level, f.squelch_level], np.float32)), bits(GOLD[f"sq_{name}_final_levels"]))


def test_fixture_exercises_the_state_machine():
    """The fixtures must actually reach the interesting corners, otherwise equality proves little."""
    assert GOLD["sq_flap_final_counts"][1] > 0          # flappy_count: flap detection engaged
    assert GOLD["sq_ctcss_good_final_counts"][2] > 0    # ctcss found
    assert GOLD["sq_ctcss_wrong_final_counts"][3] > 0   # ctcss rejected
    assert (GOLD["sq_bursts_filt_flags"] & 32).any()    # signal_outside_filter seen
    states_open = (GOLD["sq_dropout_flags"] & 1).astype(bool)
    assert states_open.any() and not states_open.all()
    assert (GOLD["sq_dropout_flags"] & 16).sum() >= 2   # last_open_sample fired (incl. a low-signal abort)


@pytest.mark.parametrize("name", sorted(CTCSS_CASES))
def test_ctcss_matches_reference_fixture(name):
    case = CTCSS_CASES[name]
    flags, found, nfound = libs.oracle().ctcss_run(case["freq"], case["rate"], case["window"], make_audio(case))
    assert np.array_equal(flags, GOLD[f"ct_{name}_flags"])
    assert [found, nfound] == list(GOLD[f"ct_{name}_counts"])


@pytest.mark.parametrize("name", sorted(FILTER_CASES))
def test_filters_match_reference_fixture(name):
    case = FILTER_CASES[name]
    rng = np.random.default_rng(case["seed"])
    x = rng.normal(size=case["n"]).astype(np.float32)
    y = rng.normal(size=case["n"]).astype(np.float32)
    o = libs.oracle()
    if case["kind"] == "notch":
        assert np.array_equal(bits(o.notch_run(case["freq"], 16000.0, case["q"], x)), bits(GOLD[f"fl_{name}_y"]))
    else:
        a, b = o.lowpass_run(case["freq"], 16000.0, x, y)
        assert np.array_equal(bits(a), bits(GOLD[f"fl_{name}_re"]))
        assert np.array_equal(bits(b), bits(GOLD[f"fl_{name}_im"]))


def test_tone_generator_matches_reference_fixture():
    assert np.array_equal(bits(libs.oracle().tone_run(8000, 79.7, 0.2, 4000)), bits(GOLD["tone_79_7"]))


# ---------- live comparison with the compiled reference (only where oracle/_ref can exist) ----------

needs_ref = pytest.mark.skipif(not libs.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_squelch_random_vs_live_reference(seed):
    rng = np.random.default_rng(100 + seed)
    n = 50000
    case = dict(kind="bursts", seed=200 + seed, n=n)
    raw = make_raw(case)
    filt = (raw * rng.uniform(0.4, 1.1, n)).astype(np.float32)
    audio = make_audio(dict(seed=seed, n=n, tone=float(rng.choice([67.0, 100.0, 151.4]))))
    o, r = libs.oracle(), libs.ref()
    for kw in (dict(), dict(filt=filt), dict(audio=audio, ctcss_freq=100.0), dict(filt=filt, audio=audio, ctcss_freq=151.4),
               dict(manual_level=float(rng.uniform(0.3, 1.0))), dict(snr_db=float(rng.uniform(0.5, 15.0)), filt=filt)):
        a, b = o.squelch_run(raw, **kw), r.squelch_run(raw, **kw)
        for key in ("flags", "level", "noise", "signal"):
            assert np.array_equal(bits(a[key]), bits(b[key])), (seed, list(kw), key)
        assert a["final"].astuple() == b["final"].astuple()


@needs_ref
def test_every_standard_tone_bank_vs_live_reference():
    """Detector-bank construction (dedupe by coefficient, +-5 Hz exclusion) for every standard tone."""
    o, r = libs.oracle(), libs.ref()
    rng = np.random.default_rng(7)
    tones = np.zeros(64, np.float32)
    import ctypes as C
    reflib = r.lib
    reflib.ref_standard_tones.argtypes = [np.ctypeslib.ndpointer(np.float32), C.c_int]
    reflib.ref_standard_tones.restype = C.c_int
    nt = reflib.ref_standard_tones(tones, 64)
    assert nt == 51
    for t in tones[:nt]:
        for rate, win in ((8000, 400), (16000, 6400)):
            x = (0.2 * np.sin(2 * np.pi * float(t) * np.arange(win * 2 + 10) / rate) + 0.05 * rng.normal(size=win * 2 + 10)).astype(np.float32)
            a, b = o.ctcss_run(float(t), rate, win, x), r.ctcss_run(float(t), rate, win, x)
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (float(t), rate, win)


@needs_ref
def test_default_filters_disabled_in_reference():
    import ctypes as C
    reflib = libs.ref().lib
    reflib.ref_filters_default_disabled.restype = C.c_int
    assert reflib.ref_filters_default_disabled() == 1  # src/test_filters.cpp:33-41
