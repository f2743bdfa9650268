"""Seeded inputs for the component fixtures (shared by tests/golden/gen_golden.py, which runs them through
the compiled reference, and by the tests, which run them through the oracle)."""
import math

import numpy as np


def det_sin(cycles_per_sample, n, start=0):
    """sin(2 pi f (start + k)), k < n, from a phasor recurrence that uses only IEEE * and + (np.cumprod):
    independent of which SIMD sin() numpy dispatches to on the machine running the test."""
    d = 2.0 * math.pi * cycles_per_sample
    w = complex(math.cos(d), math.sin(d))
    ph = np.cumprod(np.full(n, w, dtype=np.complex128))
    w0 = complex(math.cos(d * (start - 1)), math.sin(d * (start - 1)))
    return (ph * w0).imag


def _bursts(rng, n, base, sigma, spans):
    """|noise| floor with rectangular signal bursts: spans = [(start, length, level)]."""
    x = np.abs(rng.normal(base, sigma, n))
    for s, l, lvl in spans:
        x[s:s + l] += lvl
    return x.astype(np.float32)


def make_raw(case):
    rng = np.random.default_rng(case["seed"])
    n = case["n"]
    kind = case["kind"]
    if kind == "bursts":
        spans = []
        s = 3000
        while s < n - 6000:
            l = int(rng.integers(150, 5000))
            spans.append((s, l, float(rng.uniform(0.3, 3.0))))
            s += l + int(rng.integers(400, 6000))
        return _bursts(rng, n, 0.2, 0.05, spans)
    if kind == "flap":  # several short openings within the 1000-sample "recent" window -> flappy detection
        spans = []
        s = 9000
        for _ in range(8):
            spans.append((s, 420, 1.5))
            s += 420 + 330
        spans.append((s + 3000, 4000, 0.75))  # a marginal signal afterwards, near the lowered threshold
        return _bursts(rng, n, 0.2, 0.03, spans)
    if kind == "dropout":  # open, then dead spots shorter/longer than low_signal_abort (88)
        x = _bursts(rng, n, 0.05, 0.005, [(9000, 12000, 0.75)])
        for s, l in ((12000, 50), (14000, 87), (16000, 95), (18000, 300)):
            x[s:s + l] = 0.05
        return x
    if kind == "ramp":  # slow fade in/out: crosses the threshold gradually, noise floor follows
        t = np.arange(n)
        env = 0.15 + 1.2 * np.clip(det_sin(1.0 / 17000.0, n), 0, None) ** 2
        return np.abs(env * (1 + 0.05 * rng.normal(size=n))).astype(np.float32)
    raise ValueError(kind)


def make_filtered(case, raw):
    rng = np.random.default_rng(case["seed"] + 1000)
    # the post-filter magnitude: mostly a bit below the raw one, sometimes far below (signal outside the filter)
    f = raw * rng.uniform(0.6, 1.05, raw.size)
    if case.get("outside"):
        s = raw.size // 2
        f[s:s + 3000] *= 0.2
    return f.astype(np.float32)


def make_audio(case):
    rng = np.random.default_rng(case["seed"] + 2000)
    n = case["n"]
    rate = case.get("ctcss_rate", case.get("rate", 16000))
    tone = case.get("tone", 100.0)
    return (0.2 * det_sin(tone / rate, n, start=1) + case.get("audio_noise", 0.02) * rng.normal(size=n)).astype(np.float32)


SQUELCH_CASES = {
    "bursts": dict(kind="bursts", seed=11, n=60000),
    "bursts_filt": dict(kind="bursts", seed=12, n=60000, filt=True, outside=True),
    "flap": dict(kind="flap", seed=13, n=40000),
    "dropout": dict(kind="dropout", seed=14, n=40000),
    "ramp": dict(kind="ramp", seed=15, n=60000),
    "manual": dict(kind="bursts", seed=16, n=40000, manual_level=0.6),
    "snr3": dict(kind="ramp", seed=17, n=40000, snr_db=3.0, filt=True),
    "snr0": dict(kind="bursts", seed=18, n=20000, snr_db=0.0),
    "ctcss_good": dict(kind="dropout", seed=19, n=40000, audio=True, ctcss_freq=100.0, tone=100.0),
    "ctcss_wrong": dict(kind="dropout", seed=20, n=40000, audio=True, ctcss_freq=100.0, tone=67.0),
    "ctcss_filt": dict(kind="bursts", seed=21, n=60000, audio=True, filt=True, ctcss_freq=123.0, tone=123.0),
}

CTCSS_CASES = {
    "t100_fast16k": dict(freq=100.0, rate=16000, window=800, tone=100.0, seed=31, n=20000),
    "t100_slow16k": dict(freq=100.0, rate=16000, window=6400, tone=100.0, seed=32, n=30000),
    "t100_vs_103": dict(freq=100.0, rate=16000, window=6400, tone=103.5, seed=33, n=30000),
    "t254_slow8k": dict(freq=254.1, rate=8000, window=3200, tone=254.1, seed=34, n=20000),
    "t67_noise": dict(freq=67.0, rate=8000, window=3200, tone=67.0, seed=35, n=20000, audio_noise=0.3),
    "nonstd": dict(freq=68.15, rate=8000, window=3200, tone=68.15, seed=36, n=20000),
}

FILTER_CASES = {
    "notch100": dict(kind="notch", freq=100.0, q=10.0, seed=41, n=4000),
    "notch1k_q2": dict(kind="notch", freq=1000.0, q=2.0, seed=42, n=4000),
    "lp6250": dict(kind="lowpass", freq=6250.0, seed=43, n=4000),
    "lp2500": dict(kind="lowpass", freq=2500.0, seed=44, n=4000),
    "lp4000": dict(kind="lowpass", freq=4000.0, seed=45, n=4000),
}
