"""boondock-airband_amd -- Python view of the MI355X-native demodulate() path.

The product is `libmi_airband.so` (C ABI in include/mi_airband.h, HIP kernels in csrc/).  This module is a
thin ctypes binding used by tests/ and bench.py; it holds no DSP of its own and there is no CPU fallback:
if the shared library is missing it raises, and compute entry points fail with MI_ERR_NO_DEVICE when no
GPU is visible.

The directory name contains a hyphen (it mirrors the reference's project name), so it is loaded with
importlib under the module name `boondock_airband_amd` -- see tests/conftest.py and __graft_entry__.py.
"""
import ctypes as C
import os

import numpy as np

try:
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7.  Importing torch first
    # makes libmi_airband.so (NEEDED libamdhip64.so.7) bind to that already-loaded copy; the other order
    # loads two runtimes and the second one finds no GPU.
    import torch  # noqa: F401
except ImportError:  # the C ABI itself has no torch dependency
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_AIRBAND_LIB") or os.path.join(_HERE, "libmi_airband.so")  # (the override: A/B of two builds, tools/ab_bench.sh)

WAVE_RATE = 16000
WAVE_BATCH = 2000
MAX_IN_FLIGHT = 3  # staging slots of mi_demod_submit (mi_airband.cpp, kSlots)
AGC_EXTRA = 100

MOD_AM, MOD_NFM = 0, 1
OPT_EARLY_INPUT = 1  # MI_OPT_EARLY_INPUT
OPT_STEADY_BLOCKS = 2  # MI_OPT_STEADY_BLOCKS
OPT_TIME_PARALLEL = 3  # result-neutral tuning switches (include/mi_airband.h)
OPT_PRUNE_FFT = 4
OPT_U8_CONVERSION = 5
OPT_UNI_ROWS = 6
OPT_TP_CHUNKS = 7
OPT_TP_RATIO_PCT = 8
OPT_TP_SEG_LANES = 9
OPT_LANE_FFT = 10
OPT_LANE_FFT_JIT = 11
OPT_CORE_SPLIT = 12
OPT_SPEC_HEAD = 13
OPT_RESERVE_CUS = 15
OPT_PRE_WAVE = 14
OPT_AUDIO_WAVE = 16
OPT_MIXED_PLAN = 17
OPT_SPLIT_CUS = 18
SFMT_U8, SFMT_S8, SFMT_S16, SFMT_F32 = 1, 2, 3, 4

MI_OK, MI_ERR_INVALID, MI_ERR_NO_DEVICE, MI_ERR_NOMEM, MI_ERR_HIP, MI_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class MiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mi_airband error {code}: {msg}")
        self.code = code


class DeviceCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("centerfreq", C.c_int), ("fft_size_log", C.c_int), ("sfmt", C.c_int),
                ("fullscale", C.c_float), ("tau", C.c_int), ("fm_quadri", C.c_int)]


class ChannelCfg(C.Structure):
    _fields_ = [("freq", C.c_int), ("modulation", C.c_int), ("squelch_threshold_dbfs", C.c_int), ("has_snr_threshold", C.c_int),
                ("squelch_snr_db", C.c_float), ("notch_freq", C.c_float), ("notch_q", C.c_float), ("ctcss_freq", C.c_float),
                ("bandwidth", C.c_int), ("ampfactor", C.c_float), ("tau", C.c_int), ("afc", C.c_int), ("has_iq_outputs", C.c_int)]


class ChannelStats(C.Structure):
    _fields_ = [("noise_level", C.c_float), ("signal_level", C.c_float), ("squelch_level", C.c_float), ("agcavgfast", C.c_float),
                ("open_count", C.c_uint64), ("flappy_count", C.c_uint64), ("ctcss_count", C.c_uint64), ("no_ctcss_count", C.c_uint64),
                ("active_counter", C.c_uint64), ("squelch_state", C.c_int32), ("signal_outside_filter", C.c_int32)]


class ChannelDerived(C.Structure):
    _fields_ = [("bin", C.c_uint32), ("dm_dphi", C.c_uint32), ("needs_raw_iq", C.c_int32), ("has_iq_outputs", C.c_int32),
                ("modulation", C.c_int32), ("using_manual_level", C.c_int32), ("manual_signal_level", C.c_float),
                ("normal_signal_ratio", C.c_float), ("flappy_signal_ratio", C.c_float), ("ampfactor", C.c_float), ("alpha", C.c_float),
                ("notch_enabled", C.c_int32), ("notch_d", C.c_float * 3), ("lowpass_enabled", C.c_int32), ("lowpass_gain", C.c_float),
                ("lowpass_ycoeffs", C.c_float * 2), ("ctcss_enabled", C.c_int32), ("ctcss_fast_window", C.c_int32),
                ("ctcss_slow_window", C.c_int32), ("ctcss_fast_ndet", C.c_int32), ("ctcss_slow_ndet", C.c_int32)]


class IqGenCarrier(C.Structure):
    _fields_ = [("offset_hz", C.c_int32), ("kind", C.c_int32), ("amp_q8", C.c_int32), ("gate_phase", C.c_int32)]


class IqGenCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("seed", C.c_uint64), ("noise_q8_mul", C.c_int32), ("gate_samples", C.c_uint64),
                ("ncarriers", C.c_int32), ("carriers", IqGenCarrier * 64)]


# every symbol include/mi_airband.h declares
class MixInput(C.Structure):  # mi_mix_input
    _fields_ = [("row", C.c_int), ("ampfactor", C.c_float), ("balance", C.c_float)]


ABI_SYMBOLS = [
    "mi_last_error", "mi_device_count", "mi_demod_create", "mi_demod_destroy", "mi_demod_prepare", "mi_set_cache_dir", "mi_jit_counts", "mi_demod_bytes_needed", "mi_demod_bytes_consumed",
    "mi_demod_hop_bytes", "mi_demod_process", "mi_demod_submit", "mi_demod_wait", "mi_host_alloc", "mi_host_free", "mi_demod_process_device", "mi_demod_get_stats", "mi_demod_state_size",
    "mi_demod_get_state", "mi_demod_set_state", "mi_demod_read_planes", "mi_demod_process_planes", "mi_demod_last_path", "mi_demod_last_stage1", "mi_demod_pre_wave_timeouts", "mi_demod_tp_debug", "mi_demod_kernel_time", "mi_demod_kernel_time_prev", "mi_demod_event_ms", "mi_demod_set_option", "mi_demod_last_kernel_ms", "mi_plan_create", "mi_plan_destroy", "mi_plan_fft_size",
    "mi_plan_window", "mi_plan_twiddles", "mi_plan_levels", "mi_plan_sincos_lut", "mi_plan_channel", "mi_plan_ctcss_coeffs",
    "mi_iqgen_host", "mi_iqgen_device", "mi_mixer_create", "mi_mixer_destroy", "mi_mixer_is_stereo", "mi_mixer_process_device",
    "mi_gather_unique_id", "mi_gather_loopback_id", "mi_gather_create", "mi_gather_destroy", "mi_gather_audio", "mi_gather_stream_wait", "mi_gather_sync",
]

_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C boondock-airband_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.mi_last_error.restype = C.c_char_p
        L.mi_device_count.restype = C.c_int
        L.mi_demod_create.argtypes = [C.POINTER(DeviceCfg), C.POINTER(ChannelCfg), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.mi_demod_destroy.argtypes = [vp]
        L.mi_demod_prepare.argtypes = [vp, C.c_int]
        L.mi_set_cache_dir.argtypes = [C.c_char_p]
        L.mi_jit_counts.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi_demod_last_stage1.argtypes = [vp, C.POINTER(C.c_int)]
        L.mi_demod_submit.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
        L.mi_demod_wait.argtypes = [vp]
        L.mi_host_alloc.argtypes = [sz]
        L.mi_host_alloc.restype = vp
        L.mi_host_free.argtypes = [vp]
        L.mi_host_free.restype = None
        L.mi_gather_unique_id.argtypes = [vp]
        L.mi_gather_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(vp)]
        L.mi_gather_destroy.argtypes = [vp]
        L.mi_gather_audio.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.mi_gather_stream_wait.argtypes = [vp, vp]
        L.mi_gather_sync.argtypes = [vp]
        L.mi_demod_destroy.restype = None
        for f in (L.mi_demod_bytes_needed, L.mi_demod_bytes_consumed):
            f.argtypes = [vp, C.c_int]
            f.restype = sz
        L.mi_demod_hop_bytes.argtypes = [vp]
        L.mi_demod_hop_bytes.restype = sz
        L.mi_demod_process.argtypes = [vp, C.POINTER(vp), C.c_int, vp, vp, vp, vp]
        L.mi_demod_process_device.argtypes = [vp, vp, sz, C.c_int, vp, vp, vp, vp]
        L.mi_demod_get_stats.argtypes = [vp, vp]
        L.mi_demod_state_size.argtypes = [vp]
        L.mi_demod_state_size.restype = sz
        L.mi_demod_get_state.argtypes = [vp, vp, sz]
        L.mi_demod_set_state.argtypes = [vp, vp, sz]
        L.mi_demod_last_path.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi_demod_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.mi_demod_kernel_time_prev.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.mi_demod_event_ms.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.mi_demod_set_option.argtypes = [vp, C.c_int, C.c_int]
        L.mi_demod_tp_debug.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.POINTER(C.c_int)]
        L.mi_demod_read_planes.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.mi_demod_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mi_mixer_create.argtypes = [C.POINTER(MixInput), C.c_int, C.c_int, C.POINTER(vp)]
        L.mi_mixer_destroy.argtypes = [vp]
        L.mi_mixer_destroy.restype = None
        L.mi_mixer_is_stereo.argtypes = [vp]
        L.mi_mixer_process_device.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, vp, vp, vp]
        L.mi_plan_create.argtypes = [C.POINTER(DeviceCfg), C.POINTER(ChannelCfg), C.c_int, C.POINTER(vp)]
        L.mi_plan_destroy.argtypes = [vp]
        L.mi_plan_destroy.restype = None
        L.mi_plan_fft_size.argtypes = [vp]
        for f in (L.mi_plan_window, L.mi_plan_twiddles, L.mi_plan_levels):
            f.argtypes = [vp, vp]
        L.mi_plan_sincos_lut.argtypes = [vp, vp, vp]
        L.mi_plan_channel.argtypes = [vp, C.c_int, C.POINTER(ChannelDerived)]
        L.mi_plan_ctcss_coeffs.argtypes = [vp, C.c_int, C.c_int, vp]
        L.mi_iqgen_host.argtypes = [C.POINTER(IqGenCfg), C.c_uint32, C.c_uint64, C.c_uint64, vp]
        L.mi_iqgen_device.argtypes = [C.POINTER(IqGenCfg), C.c_uint32, C.c_uint32, sz, C.c_uint64, C.c_uint64, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error().decode())


def device_count():
    return lib().mi_device_count()


def device_cfg(sample_rate=2560000, centerfreq=120000000, fft_size_log=9, sfmt=SFMT_U8, fullscale=127.5, tau=-1, fm_quadri=0):
    return DeviceCfg(sample_rate, centerfreq, fft_size_log, sfmt, fullscale, tau, fm_quadri)


def channel_cfg(freq, modulation=MOD_AM, squelch_threshold_dbfs=0, squelch_snr_db=None, notch=0.0, notch_q=0.0, ctcss=0.0, bandwidth=0,
                ampfactor=1.0, tau=-1, afc=0, has_iq_outputs=0):
    return ChannelCfg(freq, modulation, squelch_threshold_dbfs, 0 if squelch_snr_db is None else 1,
                      -1.0 if squelch_snr_db is None else squelch_snr_db, notch, notch_q, ctcss, bandwidth, ampfactor, tau, afc,
                      has_iq_outputs)


def _chan_array(chans):
    return (ChannelCfg * len(chans))(*chans)


class Plan:
    """Host-only derived parameters (no GPU needed)."""

    def __init__(self, dev, chans):
        self._h = C.c_void_p()
        self.nch = len(chans)
        _check(lib().mi_plan_create(C.byref(dev), _chan_array(chans), self.nch, C.byref(self._h)))
        self.fft_size = lib().mi_plan_fft_size(self._h)

    def _vec(self, fn, n):
        out = np.zeros(n, np.float32)
        _check(fn(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def window(self):
        return self._vec(lib().mi_plan_window, self.fft_size)

    def twiddles(self):
        return self._vec(lib().mi_plan_twiddles, self.fft_size).reshape(-1, 2)

    def levels(self):
        return self._vec(lib().mi_plan_levels, 256)

    def sincos_lut(self):
        s, c = np.zeros(257, np.float32), np.zeros(257, np.float32)
        _check(lib().mi_plan_sincos_lut(self._h, s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
        return s, c

    def channel(self, i):
        d = ChannelDerived()
        _check(lib().mi_plan_channel(self._h, i, C.byref(d)))
        return d

    def ctcss_coeffs(self, i, slow):
        d = self.channel(i)
        n = d.ctcss_slow_ndet if slow else d.ctcss_fast_ndet
        out = np.zeros(n, np.float32)
        _check(lib().mi_plan_ctcss_coeffs(self._h, i, 1 if slow else 0, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._h:
            lib().mi_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Demod:
    """nstreams x nch channels of demodulate() state resident on one GPU."""

    def __init__(self, dev, chans, nstreams=1, max_batches=1, gpu=0):
        self._h = C.c_void_p()
        self.nch, self.nstreams, self.max_batches = len(chans), nstreams, max_batches
        self.chans = list(chans)
        _check(lib().mi_demod_create(C.byref(dev), _chan_array(chans), self.nch, nstreams, max_batches, gpu, C.byref(self._h)))

    @property
    def hop_bytes(self):
        return lib().mi_demod_hop_bytes(self._h)

    def bytes_needed(self, nbatches):
        return lib().mi_demod_bytes_needed(self._h, nbatches)

    def bytes_consumed(self, nbatches):
        return lib().mi_demod_bytes_consumed(self._h, nbatches)

    def process(self, iq_streams, nbatches, want_iq=False, want_stats=True):
        """Host-buffer entry.  iq_streams: list of uint8 arrays, one per stream, each starting at the stream's
        current position.  Returns (waveout[ns][nch][nb*2000+100], axc[ns][nch][nb], iq_out or None, stats or None)."""
        assert len(iq_streams) == self.nstreams
        need = self.bytes_needed(nbatches)
        keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in iq_streams]
        for a in keep:
            if a.size < need:
                raise ValueError(f"stream shorter than bytes_needed ({a.size} < {need})")
        ptrs = (C.c_void_p * self.nstreams)(*[a.ctypes.data for a in keep])
        # the library completes whatever was submitted before it runs this call: those results are ready for wait()
        self._done = getattr(self, "_done", []) + getattr(self, "_tickets", [])
        self._tickets = []
        n = nbatches * WAVE_BATCH
        wo = np.zeros((self.nstreams, self.nch, n + AGC_EXTRA), np.float32)
        axc = np.zeros((self.nstreams, self.nch, nbatches), np.uint8)
        iqo = np.zeros((self.nstreams, self.nch, n, 2), np.float32) if want_iq else None
        stats = (ChannelStats * (self.nstreams * self.nch))() if want_stats else None
        _check(lib().mi_demod_process(self._h, ptrs, nbatches, wo.ctypes.data_as(C.c_void_p),
                                      None if iqo is None else iqo.ctypes.data_as(C.c_void_p), axc.ctypes.data_as(C.c_void_p),
                                      None if stats is None else C.cast(stats, C.c_void_p)))
        return wo, axc, iqo, stats

    def submit(self, iq_streams, nbatches, want_iq=False, want_stats=True, waveout=None):
        """mi_demod_submit: starts a call and returns a ticket; wait() completes the oldest ticket and returns its results
        like process().  The numpy arrays of a ticket stay referenced until it has been waited for."""
        assert len(iq_streams) == self.nstreams
        need = self.bytes_needed(nbatches)
        keep = [a if isinstance(a, PinnedBuffer) else np.ascontiguousarray(a, dtype=np.uint8) for a in iq_streams]
        addr = [a.ptr if isinstance(a, PinnedBuffer) else a.ctypes.data for a in keep]
        ptrs = (C.c_void_p * self.nstreams)(*addr)
        n = nbatches * WAVE_BATCH
        # waveout: an optional caller-owned float32 array [nstreams][nch][n + AGC_EXTRA] (e.g. a view of a PinnedBuffer)
        wo = waveout if waveout is not None else np.empty((self.nstreams, self.nch, n + AGC_EXTRA), np.float32)
        assert wo.dtype == np.float32 and wo.size == self.nstreams * self.nch * (n + AGC_EXTRA) and wo.flags["C_CONTIGUOUS"]
        axc = np.zeros((self.nstreams, self.nch, nbatches), np.uint8)
        iqo = np.zeros((self.nstreams, self.nch, n, 2), np.float32) if want_iq else None
        stats = (ChannelStats * (self.nstreams * self.nch))() if want_stats else None
        _check(lib().mi_demod_submit(self._h, ptrs, nbatches, wo.ctypes.data_as(C.c_void_p),
                                     None if iqo is None else iqo.ctypes.data_as(C.c_void_p), axc.ctypes.data_as(C.c_void_p),
                                     None if stats is None else C.cast(stats, C.c_void_p)))
        if not hasattr(self, "_tickets"):
            self._tickets = []
        self._tickets.append((keep, ptrs, wo, axc, iqo, stats))
        while len(self._tickets) > MAX_IN_FLIGHT:  # a further submit completed the oldest call inside the library
            self._done = getattr(self, "_done", []) + [self._tickets.pop(0)]

    def wait(self):
        """Results (waveout, axc, iq_out, stats) of the oldest submitted call."""
        done = getattr(self, "_done", [])
        if done:
            t = done.pop(0)
            return t[2], t[3], t[4], t[5]
        _check(lib().mi_demod_wait(self._h))
        t = self._tickets.pop(0)
        return t[2], t[3], t[4], t[5]

    def process_device(self, d_iq_ptr, stream_stride, nbatches, d_waveout_ptr, d_axc_ptr, d_iq_out_ptr=None, hip_stream=None):
        """Device-resident entry: raw device pointers (ints), asynchronous on hip_stream."""
        _check(lib().mi_demod_process_device(self._h, d_iq_ptr, stream_stride, nbatches, d_waveout_ptr, d_iq_out_ptr, d_axc_ptr, hip_stream))

    def stats(self):
        st = (ChannelStats * (self.nstreams * self.nch))()
        _check(lib().mi_demod_get_stats(self._h, C.cast(st, C.c_void_p)))
        return st

    def get_state(self):
        n = lib().mi_demod_state_size(self._h)
        buf = np.zeros(n, np.uint8)
        _check(lib().mi_demod_get_state(self._h, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _check(lib().mi_demod_set_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def last_path(self):
        """(1 if the last call ran the time-parallel stage 2 else 0, rows left unverified -- always 0)."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().mi_demod_last_path(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def process_planes(self, mag, nbatches, cplx=None, want_iq=False):
        """mi_demod_process_planes (test entry): stage 2 over caller-supplied planes.  mag: [rows][count] float32; cplx:
        [iq rows][count][2] or None.  Returns (waveout [nstreams][nch][n + AGC_EXTRA], axc, iq_out or None, stats)."""
        n = nbatches * WAVE_BATCH
        mag = np.ascontiguousarray(mag, dtype=np.float32)
        wo = np.zeros((self.nstreams, self.nch, n + AGC_EXTRA), np.float32)
        axc = np.zeros((self.nstreams, self.nch, nbatches), np.uint8)
        iqo = np.zeros((self.nstreams, self.nch, n, 2), np.float32) if want_iq else None
        stats = (ChannelStats * (self.nstreams * self.nch))()
        zc = None if cplx is None else np.ascontiguousarray(cplx, dtype=np.float32)
        f = lib().mi_demod_process_planes
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(f(self._h, mag.ctypes.data_as(C.c_void_p), None if zc is None else zc.ctypes.data_as(C.c_void_p), nbatches,
                 wo.ctypes.data_as(C.c_void_p), None if iqo is None else iqo.ctypes.data_as(C.c_void_p), axc.ctypes.data_as(C.c_void_p),
                 C.cast(stats, C.c_void_p)))
        return wo, axc, iqo, list(stats)

    def prepare(self, host_slots=1):
        """mi_demod_prepare: stage-1 kernel of the plan + the staging of `host_slots` host-buffer calls, before the first batch."""
        _check(lib().mi_demod_prepare(self._h, host_slots))

    def pre_wave_timeouts(self):
        n = C.c_uint(0)
        f = lib().mi_demod_pre_wave_timeouts
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        _check(f(self._h, C.byref(n)))
        return n.value

    def last_stage1(self):
        """MI_STAGE1_* of the last call: 0 / 1 exchange kernels (full / pruned), 2 / 3 lane-resident (full graph / plan-compiled)."""
        k = C.c_int(-1)
        _check(lib().mi_demod_last_stage1(self._h, C.byref(k)))
        return k.value

    def tp_debug(self, row):
        """(core[nseg+1][4], diag[4]) of the last time-parallel call for one row."""
        nseg = C.c_int(0)
        _check(lib().mi_demod_tp_debug(self._h, row, None, 0, None, C.byref(nseg)))
        core = np.zeros((nseg.value + 1, 4), np.float32)
        diag = np.zeros(8, np.int32)
        _check(lib().mi_demod_tp_debug(self._h, row, core.ctypes.data_as(C.c_void_p), nseg.value + 1, diag.ctypes.data_as(C.c_void_p),
                                       C.byref(nseg)))
        return core, diag

    def read_planes(self, stream, ch, first, count, want_iq=False):
        mag = np.zeros(count, np.float32)
        iq = np.zeros((count, 2), np.float32) if want_iq else None
        _check(lib().mi_demod_read_planes(self._h, stream, ch, first, count, mag.ctypes.data_as(C.c_void_p),
                                          None if iq is None else iq.ctypes.data_as(C.c_void_p)))
        return mag, iq

    def set_option(self, option, value):
        _check(lib().mi_demod_set_option(self._h, option, value))

    def kernel_times(self, age=0):
        """[(kernel name, total ms, launches)] of the last call (age 0) or of the call `age` calls before it, from HIP events
        on the launch streams."""
        out = []
        i = 0
        while True:
            name, ms, n = C.c_char_p(), C.c_float(0), C.c_int(0)
            if lib().mi_demod_kernel_time_prev(self._h, age, i, C.byref(name), C.byref(ms), C.byref(n)) != MI_OK:
                break
            out.append((name.value.decode(), ms.value, n.value))
            i += 1
        return out

    def event_ms(self, ref_age, age, chunk, event):
        """(diagnostic) ms from the core chain start of the call `ref_age` back to an event of the call `age` back, or None"""
        ms = C.c_float(0)
        if lib().mi_demod_event_ms(self._h, ref_age, age, chunk, event, C.byref(ms)) != MI_OK:
            return None
        return ms.value

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().mi_demod_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().mi_demod_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def iqgen_cfg(sample_rate=2560000, seed=0xA1B2C3D4, noise_q8_mul=111, gate_samples=None, carriers=()):
    cfg = IqGenCfg()
    cfg.sample_rate = sample_rate
    cfg.seed = seed
    cfg.noise_q8_mul = noise_q8_mul
    cfg.gate_samples = sample_rate if gate_samples is None else gate_samples
    cfg.ncarriers = len(carriers)
    for i, (off, kind, amp_q8, gate_phase) in enumerate(carriers):
        cfg.carriers[i] = IqGenCarrier(off, kind, amp_q8, gate_phase)
    return cfg


def iqgen_host(cfg, stream_id, first, count):
    out = np.zeros(2 * count, np.uint8)
    _check(lib().mi_iqgen_host(C.byref(cfg), stream_id, first, count, out.ctypes.data_as(C.c_void_p)))
    return out


def iqgen_device(cfg, first_stream_id, nstreams, stream_stride, first, count, d_out_ptr, hip_stream=None):
    _check(lib().mi_iqgen_device(C.byref(cfg), first_stream_id, nstreams, stream_stride, first, count, d_out_ptr, hip_stream))


def set_cache_dir(path):
    """mi_set_cache_dir: where compiled stage-1 kernels are kept between process starts (None / "" = nowhere)."""
    _check(lib().mi_set_cache_dir(None if path is None else os.fsencode(path)))


def jit_counts():
    """(kernels compiled by this process, kernels loaded from the cache directory)"""
    a, b = C.c_int(0), C.c_int(0)
    _check(lib().mi_jit_counts(C.byref(a), C.byref(b)))
    return a.value, b.value


class Gather:
    """mi_gather_*: audio + flags of every rank to rank 0 over RCCL (the C-ABI twin of shard.AudioGather)."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib().mi_gather_unique_id(C.byref(buf)))
        return bytes(buf)

    @staticmethod
    def loopback_id(job):
        """mi_gather_loopback_id (test transport): ranks of job `job` are threads of this process."""
        buf = (C.c_char * 128)()
        f = lib().mi_gather_loopback_id
        f.argtypes = [C.c_void_p, C.c_uint64]
        _check(f(C.byref(buf), job))
        return bytes(buf)

    def __init__(self, uid, rank, world, gpu, streams_per_rank, nch, max_batches):
        self._h = C.c_void_p()
        idbuf = (C.c_char * 128).from_buffer_copy(uid) if uid is not None else None
        arr = (C.c_int * world)(*streams_per_rank)
        _check(lib().mi_gather_create(None if idbuf is None else C.byref(idbuf), rank, world, gpu, arr, nch, max_batches, C.byref(self._h)))

    def audio(self, d_waveout, d_axc, nbatches, d_all_waveout=None, d_all_axc=None, open_only=False, hip_stream=None):
        _check(lib().mi_gather_audio(self._h, d_waveout, d_axc, nbatches, 1 if open_only else 0, d_all_waveout, d_all_axc, hip_stream))

    def stream_wait(self, hip_stream=None):
        _check(lib().mi_gather_stream_wait(self._h, hip_stream))

    def sync(self):
        _check(lib().mi_gather_sync(self._h))

    def close(self):
        if self._h:
            lib().mi_gather_destroy(self._h)
            self._h = C.c_void_p()


class PinnedBuffer:
    """Page-locked host memory from mi_host_alloc, viewed as a uint8 numpy array (.array); .ptr is its address."""

    def __init__(self, nbytes, _root=None, _ptr=None):
        self._root = _root
        if _root is None:
            self._mem = lib().mi_host_alloc(nbytes)
            if not self._mem:
                raise MemoryError("mi_host_alloc failed")
            self.ptr = self._mem
        else:
            self._mem = None
            self.ptr = _ptr
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(self.ptr))

    def view(self, offset):
        """The same memory from `offset` on (a stream position inside a pinned capture)."""
        return PinnedBuffer(self.nbytes - offset, self._root or self, self.ptr + offset)

    def free(self):
        if self._mem:
            lib().mi_host_free(self._mem)
            self._mem = None


class Mixer:
    """One mixer_t (src/mixer.cpp) over device-resident audio: inputs = [(row, ampfactor, balance), ...]."""

    def __init__(self, inputs, gpu=0):
        arr = (MixInput * len(inputs))(*[MixInput(int(r), float(a), float(b)) for r, a, b in inputs])
        self._h = C.c_void_p()
        _check(lib().mi_mixer_create(arr, len(inputs), gpu, C.byref(self._h)))
        self.stereo = bool(lib().mi_mixer_is_stereo(self._h))

    def process_device(self, d_waveout, row_stride, d_axc, axc_stride, nbatches, d_left, d_right, d_axc_out, hip_stream=None):
        _check(lib().mi_mixer_process_device(self._h, d_waveout, row_stride, d_axc, axc_stride, nbatches, d_left, d_right, d_axc_out, hip_stream))

    def close(self):
        if self._h:
            lib().mi_mixer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the BASELINE.json channel plans ----

def config2_channels():
    """8 AM channels at centre -1.0 MHz + 25 kHz + k*250 kHz (bins 316,366,416,466,4,54,104,154 at fft 512)."""
    centre = 120000000
    return centre, [channel_cfg(centre - 1000000 + 25000 + k * 250000) for k in range(8)]


def config3_channels():
    """32 channels spaced 70 kHz, even = AM, odd = NFM with bandwidth 12500; every 4th NFM has ctcss 100.0,
    one has notch 100.0; fft 2048."""
    centre = 120000000
    chans = []
    nfm_idx = 0
    for k in range(32):
        f = centre - 1120000 + 35000 + k * 70000
        if k % 2 == 0:
            chans.append(channel_cfg(f))
        else:
            ct = 100.0 if nfm_idx % 4 == 0 else 0.0
            notch = 100.0 if nfm_idx == 0 else 0.0
            chans.append(channel_cfg(f, modulation=MOD_NFM, bandwidth=12500, ctcss=ct, notch=notch))
            nfm_idx += 1
    return centre, chans


def carriers_for(centre, chans, amp_q8=3072, active=lambda k: k % 2 == 0):
    """One synthetic carrier per active channel: AM for AM channels, NFM (+CTCSS where configured) otherwise."""
    out = []
    for k, c in enumerate(chans):
        if not active(k):
            continue
        kind = 0 if c.modulation == MOD_AM else (2 if c.ctcss_freq > 0 else 1)
        out.append((c.freq - centre, kind, amp_q8, 0))
    return out
