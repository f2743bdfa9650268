"""Stage 1 of the oracle (window, level LUTs, FFT, bin formula, derotation increment) against independent
float64 references.  The reference's FFT is FFTW3f (absent, version unpinned) and its tests pin nothing at
that boundary, so the FFT is pinned by the DFT definition: a float64 numpy FFT of the same windowed input."""
import ctypes as C

import numpy as np
import pytest

import libs
from common import to_oracle_cfg
from conftest import load_package


def oracle_fft(log2n, x):
    lib = libs.oracle_lib()
    lib.ao_fft_run.argtypes = [C.c_int, libs.f32p, libs.f32p]
    lib.ao_fft_run.restype = None
    inp = np.ascontiguousarray(np.stack([x.real, x.imag], axis=1).astype(np.float32).reshape(-1))
    out = np.zeros_like(inp)
    lib.ao_fft_run(log2n, inp, out)
    return out.reshape(-1, 2)[:, 0].astype(np.float64) + 1j * out.reshape(-1, 2)[:, 1].astype(np.float64)


@pytest.mark.parametrize("log2n", [8, 9, 10, 11, 12, 13])
def test_fft_is_the_forward_dft(log2n):
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    x = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
    got = oracle_fft(log2n, x)
    want = np.fft.fft(x.astype(np.complex128))  # e^{-j...}, unnormalised = FFTW_FORWARD
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 2e-6, err
    # a pure tone lands in its bin with the right sign convention
    k = 37
    tone = np.exp(2j * np.pi * k * np.arange(n) / n).astype(np.complex64)
    spec = np.abs(oracle_fft(log2n, tone))
    assert int(np.argmax(spec)) == k and abs(spec[k] - n) < 1e-2 * n


def test_window_is_blackman7_in_double():
    for n in (256, 512, 2048):
        w = np.zeros(n, np.float32)
        libs.oracle_lib().ao_window(w, n)
        a = [np.float64(np.float32(v)) for v in (0.27105140069342, 0.43329793923448, 0.21812299954311, 0.06592544638803, 0.01081174209837,
                                                  0.00077658482522, 0.00001388721735)]
        i = np.arange(n, dtype=np.float64)
        x = a[0] - a[1] * np.cos(2 * np.pi * i / (n - 1)) + a[2] * np.cos(4 * np.pi * i / (n - 1)) - a[3] * np.cos(6 * np.pi * i / (n - 1)) + \
            a[4] * np.cos(8 * np.pi * i / (n - 1)) - a[5] * np.cos(10 * np.pi * i / (n - 1)) + a[6] * np.cos(12 * np.pi * i / (n - 1))
        assert np.abs(w.astype(np.float64) - x).max() < 1e-7
        assert w[0] < 1e-5 and abs(w[n // 2] - 1.0) < 1e-3 and np.allclose(w, w[::-1], atol=1e-6)


def test_bin_formula_quirks():
    """config.cpp:669-670: integer sample_rate/fft_size and ceil(x - 1): an on-grid frequency maps to bin x-1
    (+25 kHz -> bin 4, not 5, at 5 kHz/bin)."""
    lib = libs.oracle_lib()
    centre, rate = 120000000, 2560000
    assert lib.ao_bin_for_freq(centre + 25000, centre, rate, 512) == 4
    assert lib.ao_bin_for_freq(centre, centre, rate, 512) == 511
    assert lib.ao_bin_for_freq(centre + 1, centre, rate, 512) == 0
    assert lib.ao_bin_for_freq(centre - 1000000 + 25000, centre, rate, 512) == 316
    got = [lib.ao_bin_for_freq(centre - 1000000 + 25000 + k * 250000, centre, rate, 512) for k in range(8)]
    assert got == [316, 366, 416, 466, 4, 54, 104, 154]  # SURVEY 8(d) channel plan


def test_dm_dphi():
    lib = libs.oracle_lib()
    centre, rate = 120000000, 2560000
    # rate/WAVE_RATE is an integer (160) -> no correction; f/16000 mod 1 in 24-bit turns, negative wraps
    for off in (25000, -37500, 1200000, -1275000, 16000, 8000):
        frac = (off / 16000.0) - np.trunc(off / 16000.0)
        want = np.uint32(np.int32(int(frac * (1 << 24))))
        assert lib.ao_dm_dphi(centre + off, centre, rate) == int(want)
    # a non-integer decimation (2.4 MS/s -> 150 exactly is integer; 2.048 MS/s -> 128) keeps correction 0; 2.5 MS/s does not
    assert lib.ao_dm_dphi(centre + 100000, centre, 2500000) != lib.ao_dm_dphi(centre + 100000, centre, 2560000)


def test_levels_and_alpha():
    lib = libs.oracle_lib()
    assert abs(lib.ao_alpha_for_tau(-1) - np.exp(-1.0 / 3.2)) < 1e-7
    assert lib.ao_alpha_for_tau(0) == 0.0
    assert abs(lib.ao_alpha_for_tau(75) - np.exp(-1.0 / (16000 * 75e-6))) < 1e-7
    lvl = lib.ao_dbfs_to_level(C.c_float(-40.0), 512)
    offset = 7.54 + 10 * np.log10(256) - 2.38
    assert abs(lvl - 10 ** ((-40 - offset) / 20) * 512) < 1e-4 * lvl


def test_stage1_matches_float64_model():
    """convert x window -> FFT -> |bin| for a whole capture against a float64 numpy model of the same maths."""
    pkg = load_package()
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    from common import gen_iq
    iq, _ = gen_iq(pkg, dev, centre, chans, 1)
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nfft = 300
    mag, z = od.stage1(iq, nfft)
    w = np.zeros(512, np.float32)
    libs.oracle_lib().ao_window(w, 512)
    bins = [316, 366, 416, 466, 4, 54, 104, 154]
    x = iq.astype(np.float64).reshape(-1, 2)
    cplx = ((x[:, 0] - 127.5) / 127.5) + 1j * ((x[:, 1] - 127.5) / 127.5)
    for f in (0, 1, 150, 299):
        seg = cplx[f * 160:f * 160 + 512] * w.astype(np.float64)
        spec = np.fft.fft(seg)
        for c, b in enumerate(bins):
            assert abs(mag[c, f] - abs(spec[b])) <= 2e-5 * max(1.0, abs(spec[b]))
            assert abs(z[c, f, 0] - spec[b].real) <= 2e-5 * max(1.0, abs(spec[b]))
