"""ctypes loaders for the three shared libraries the tests talk to.

  oracle()  -> oracle/_build/libairband_oracle.so   CPU restatement (the checker)
  ref()     -> oracle/_ref/libairband_ref.so        the reference's own squelch/ctcss/filters sources
                                                    compiled in place; only where /root/reference exists
                                                    (never loaded on the GPU box: returns None there)
  product() -> boondock-airband_amd/libmi_airband.so  the C-ABI under test (include/mi_airband.h)

Nothing here reads /root/reference at run time; building _ref does (oracle/Makefile `make ref`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PKG_DIR = os.path.join(ROOT, "boondock-airband_amd")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

WAVE_RATE = 16000
WAVE_BATCH = 2000
AGC_EXTRA = 100


class SquelchCfg(C.Structure):
    _fields_ = [("manual_level", C.c_float), ("has_snr", C.c_int), ("snr_db", C.c_float), ("ctcss_freq", C.c_float),
                ("ctcss_rate", C.c_float)]


class SquelchFinal(C.Structure):
    _fields_ = [("open_count", C.c_uint64), ("flappy_count", C.c_uint64), ("ctcss_count", C.c_uint64),
                ("no_ctcss_count", C.c_uint64), ("noise_level", C.c_float), ("signal_level", C.c_float),
                ("squelch_level", C.c_float)]

    def astuple(self):
        return (self.open_count, self.flappy_count, self.ctcss_count, self.no_ctcss_count,
                np.float32(self.noise_level).tobytes(), np.float32(self.signal_level).tobytes(),
                np.float32(self.squelch_level).tobytes())


class DeviceCfg(C.Structure):
    """ao_device_cfg == mi_device_cfg field for field."""
    _fields_ = [("sample_rate", C.c_int), ("centerfreq", C.c_int), ("fft_size_log", C.c_int), ("sfmt", C.c_int),
                ("fullscale", C.c_float), ("tau", C.c_int), ("fm_quadri", C.c_int)]


class ChannelCfg(C.Structure):
    """ao_channel_cfg == mi_channel_cfg field for field."""
    _fields_ = [("freq", C.c_int), ("modulation", C.c_int), ("squelch_threshold_dbfs", C.c_int), ("has_snr_threshold", C.c_int),
                ("squelch_snr_db", C.c_float), ("notch_freq", C.c_float), ("notch_q", C.c_float), ("ctcss_freq", C.c_float),
                ("bandwidth", C.c_int), ("ampfactor", C.c_float), ("tau", C.c_int), ("afc", C.c_int), ("has_iq_outputs", C.c_int)]


def channel_cfg(freq, modulation=0, squelch_threshold_dbfs=0, squelch_snr_db=None, notch=0.0, notch_q=0.0, ctcss=0.0,
                bandwidth=0, ampfactor=1.0, tau=-1, afc=0, has_iq_outputs=0):
    return ChannelCfg(freq, modulation, squelch_threshold_dbfs, 0 if squelch_snr_db is None else 1,
                      -1.0 if squelch_snr_db is None else squelch_snr_db, notch, notch_q, ctcss, bandwidth, ampfactor, tau, afc,
                      has_iq_outputs)


def device_cfg(sample_rate=2560000, centerfreq=120000000, fft_size_log=9, sfmt=1, fullscale=127.5, tau=-1, fm_quadri=0):
    return DeviceCfg(sample_rate, centerfreq, fft_size_log, sfmt, fullscale, tau, fm_quadri)


def _make(target, cwd):
    subprocess.run(["make", "-s"] + target, cwd=cwd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _declare_component_api(lib, prefix):
    run = getattr(lib, prefix + "_squelch_run")
    run.argtypes = [C.POINTER(SquelchCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                    C.c_void_p, C.POINTER(SquelchFinal)]
    run.restype = None
    f = getattr(lib, prefix + "_ctcss_run")
    f.argtypes = [C.c_float, C.c_float, C.c_int, f32p, C.c_size_t, u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    f.restype = None
    f = getattr(lib, prefix + "_notch_run")
    f.argtypes = [C.c_float, C.c_float, C.c_float, f32p, C.c_size_t, f32p]
    f.restype = None
    f = getattr(lib, prefix + "_lowpass_run")
    f.argtypes = [C.c_float, C.c_float, f32p, f32p, C.c_size_t, f32p, f32p]
    f.restype = None
    f = getattr(lib, prefix + "_tone_run")
    f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_size_t, f32p]
    f.restype = None


class ComponentAPI:
    """Same python surface over either the oracle ('ao') or the compiled reference ('ref')."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        _declare_component_api(lib, prefix)

    def squelch_run(self, raw, filt=None, audio=None, manual_level=0.0, snr_db=None, ctcss_freq=0.0, ctcss_rate=WAVE_RATE):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        n = raw.size
        cfg = SquelchCfg(manual_level, 0 if snr_db is None else 1, 0.0 if snr_db is None else snr_db, ctcss_freq, ctcss_rate)
        flags = np.zeros(n, np.uint8)
        level = np.zeros(n, np.float32)
        noise = np.zeros(n, np.float32)
        signal = np.zeros(n, np.float32)
        fin = SquelchFinal()
        fp = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.c_void_p)
        keep = [np.ascontiguousarray(a, dtype=np.float32) if a is not None else None for a in (filt, audio)]
        getattr(self.lib, self.prefix + "_squelch_run")(
            C.byref(cfg), raw.ctypes.data_as(C.c_void_p), None if keep[0] is None else keep[0].ctypes.data_as(C.c_void_p),
            None if keep[1] is None else keep[1].ctypes.data_as(C.c_void_p), n, flags.ctypes.data_as(C.c_void_p),
            level.ctypes.data_as(C.c_void_p), noise.ctypes.data_as(C.c_void_p), signal.ctypes.data_as(C.c_void_p), C.byref(fin))
        del fp
        return dict(flags=flags, level=level, noise=noise, signal=signal, final=fin)

    def ctcss_run(self, freq, rate, window, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        flags = np.zeros(x.size, np.uint8)
        found, nfound = C.c_uint64(0), C.c_uint64(0)
        getattr(self.lib, self.prefix + "_ctcss_run")(freq, rate, window, x, x.size, flags, C.byref(found), C.byref(nfound))
        return flags, found.value, nfound.value

    def notch_run(self, freq, rate, q, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        getattr(self.lib, self.prefix + "_notch_run")(freq, rate, q, x, x.size, y)
        return y

    def lowpass_run(self, freq, rate, re, im):
        re = np.ascontiguousarray(re, dtype=np.float32)
        im = np.ascontiguousarray(im, dtype=np.float32)
        ore, oim = np.zeros_like(re), np.zeros_like(im)
        getattr(self.lib, self.prefix + "_lowpass_run")(freq, rate, re, im, re.size, ore, oim)
        return ore, oim

    def tone_run(self, sample_rate, freq, ampl, n):
        out = np.zeros(n, np.float32)
        getattr(self.lib, self.prefix + "_tone_run")(sample_rate, freq, ampl, n, out)
        return out


_cache = {}


def oracle_lib():
    if "oracle" not in _cache:
        path = os.path.join(ORACLE_DIR, "_build", "libairband_oracle.so")
        if not os.path.exists(path):
            _make([], ORACLE_DIR)
        lib = C.CDLL(path)
        lib.ao_demod_create.argtypes = [C.POINTER(DeviceCfg), C.POINTER(ChannelCfg), C.c_int]
        lib.ao_demod_create.restype = C.c_void_p
        lib.ao_demod_destroy.argtypes = [C.c_void_p]
        lib.ao_demod_destroy.restype = None
        lib.ao_demod_run.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_int, f32p, C.c_void_p, C.c_void_p]
        lib.ao_demod_run.restype = C.c_int
        lib.ao_stage1.argtypes = [C.c_void_p, u8p, C.c_size_t, f32p, C.c_void_p]
        lib.ao_stage1.restype = None
        lib.ao_mixer_run.argtypes = [C.c_void_p, C.c_int, f32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, f32p, C.c_void_p, C.c_void_p]
        lib.ao_mixer_run.restype = C.c_int
        lib.ao_afc_check.argtypes = [f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_float, C.c_ubyte]
        lib.ao_afc_check.restype = C.c_size_t
        lib.ao_demod_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ao_demod_bins.restype = None
        lib.ao_window.argtypes = [f32p, C.c_size_t]
        lib.ao_window.restype = None
        lib.ao_bin_for_freq.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t]
        lib.ao_bin_for_freq.restype = C.c_size_t
        lib.ao_dm_dphi.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.ao_dm_dphi.restype = C.c_uint32
        lib.ao_dbfs_to_level.argtypes = [C.c_float, C.c_size_t]
        lib.ao_dbfs_to_level.restype = C.c_float
        lib.ao_alpha_for_tau.argtypes = [C.c_int]
        lib.ao_alpha_for_tau.restype = C.c_float
        lib.ao_ctcss_detector_count.argtypes = [C.c_float, C.c_float, C.c_int]
        lib.ao_ctcss_detector_count.restype = C.c_int
        _cache["oracle"] = lib
    return _cache["oracle"]


def oracle():
    return ComponentAPI(oracle_lib(), "ao")


def ref_available():
    """The reference's own squelch/ctcss/filters build exists only where /root/reference does (the build container).
    On the GPU box it is never probed for or loaded: there the oracle is pinned by tests/golden/components_ref.npz."""
    if not os.path.isdir("/root/reference/src"):
        return False
    path = os.path.join(ORACLE_DIR, "_ref", "libairband_ref.so")
    if not os.path.exists(path):
        try:
            _make(["ref"], ORACLE_DIR)
        except Exception:
            return False
    return os.path.exists(path)


def ref():
    if not ref_available():
        return None
    if "ref" not in _cache:
        _cache["ref"] = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libairband_ref.so"))
    return ComponentAPI(_cache["ref"], "ref")


class OracleDemod:
    """The oracle's whole hot path over a linear capture."""

    def __init__(self, dev, chans):
        self.lib = oracle_lib()
        self.nch = len(chans)
        arr = (ChannelCfg * self.nch)(*chans)
        self.dev = dev
        self.h = self.lib.ao_demod_create(C.byref(dev), arr, self.nch)
        assert self.h

    def run(self, iq, max_batches, want_iq=False):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        wo = np.zeros((self.nch, max_batches * WAVE_BATCH), np.float32)
        axc = np.zeros((self.nch, max_batches), np.uint8)
        iqo = np.zeros((self.nch, max_batches * WAVE_BATCH * 2), np.float32) if want_iq else None
        nb = self.lib.ao_demod_run(self.h, iq, iq.size, max_batches, wo, None if iqo is None else iqo.ctypes.data_as(C.c_void_p),
                                   axc.ctypes.data_as(C.c_void_p))
        return nb, wo, axc, iqo

    def stage1(self, iq, nfft, want_iq=True):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        mag = np.zeros((self.nch, nfft), np.float32)
        iqo = np.zeros((self.nch, nfft, 2), np.float32) if want_iq else None
        self.lib.ao_stage1(self.h, iq, nfft, mag, None if iqo is None else iqo.ctypes.data_as(C.c_void_p))
        return mag, iqo

    def bins(self):
        """(dev->bins, dev->base_bins) as AFC left them."""
        cur, base = np.zeros(self.nch, np.int32), np.zeros(self.nch, np.int32)
        self.lib.ao_demod_bins(self.h, cur.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p))
        return cur, base

    def close(self):
        if self.h:
            self.lib.ao_demod_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class MixInput(C.Structure):  # ao_mix_input
    _fields_ = [("row", C.c_int), ("ampfactor", C.c_float), ("balance", C.c_float)]


def oracle_mixer(inputs, waveout, axc, nbatches):
    """ao_mixer_run over audio [rows][>= nbatches*WAVE_BATCH] and flags [rows][nbatches]: (left, right or None, axc_out)."""
    lib = oracle_lib()
    arr = (MixInput * len(inputs))(*[MixInput(int(r), float(a), float(b)) for r, a, b in inputs])
    waveout = np.ascontiguousarray(waveout, np.float32)
    axc = np.ascontiguousarray(axc, np.uint8)
    left = np.full(nbatches * WAVE_BATCH, np.nan, np.float32)
    right = np.full(nbatches * WAVE_BATCH, np.nan, np.float32)
    out = np.zeros(nbatches, np.uint8)
    stereo = lib.ao_mixer_run(arr, len(inputs), waveout.reshape(-1), waveout.shape[1], axc.ctypes.data_as(C.c_void_p), axc.shape[1], nbatches, left,
                              right.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return left, (right if stereo else None), out
