"""The C++ host mirror (boondock-airband_amd/host): the reference's input_t ring + demodulate() thread skeleton +
output-thread contract around the engine.  The replay tool feeds a capture file through circbuffer_append, the
demod thread calls the C ABI once per WAVE_BATCH, an output-thread stand-in consumes waveout[0..WAVE_BATCH) and
does the AGC_EXTRA carry -- the emitted audio must equal the oracle's (which plays the same roles on the CPU)."""
import os
import subprocess

import numpy as np
import pytest

from common import WAVE_BATCH, assert_same, gen_iq, oracle_run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "boondock-airband_amd", "host")
TOOL = os.path.join(HOST, "airband_replay")


def _build():
    subprocess.run(["make", "-s", "-C", HOST], check=True)


def _write_cfg(path, dev, chans):
    with open(path, "w") as f:
        f.write(f"{dev.sample_rate} {dev.centerfreq} {dev.fft_size_log} {dev.sfmt} {dev.tau} {dev.fm_quadri}\n")
        for c in chans:
            f.write(f"{c.freq} {c.modulation} {c.squelch_threshold_dbfs} {c.has_snr_threshold} {c.squelch_snr_db} {c.notch_freq} "
                    f"{c.notch_q} {c.ctcss_freq} {c.bandwidth} {c.ampfactor} {c.tau} {c.afc} {c.has_iq_outputs}\n")


def test_host_mirror_builds_and_links():
    _build()
    assert os.access(TOOL, os.X_OK)
    out = subprocess.run(["nm", "-D", "--undefined-only", TOOL], capture_output=True, text=True).stdout
    for sym in ("mi_demod_create", "mi_demod_prepare", "mi_demod_submit", "mi_demod_wait", "mi_demod_bytes_consumed", "mi_demod_destroy"):
        assert sym in out, f"the host mirror must go through the C ABI ({sym})"


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["config2", "options"])
def test_replay_through_ring_equals_oracle(pkg, tmp_path, case):
    _build()
    centre = 120000000
    if case == "config2":
        _, chans = pkg.config2_channels()
        dev = pkg.device_cfg(centerfreq=centre)
        kw = dict(gate_div=4)
    else:
        chans = [pkg.channel_cfg(centre + 250000, squelch_threshold_dbfs=-40), pkg.channel_cfg(centre + 500000, modulation=pkg.MOD_NFM, notch=1000.0, notch_q=5.0),
                 pkg.channel_cfg(centre - 500000, bandwidth=8000, has_iq_outputs=1), pkg.channel_cfg(centre + 750000, modulation=pkg.MOD_NFM, ctcss=100.0, bandwidth=12500)]
        dev = pkg.device_cfg(centerfreq=centre, tau=75)
        kw = dict(gate_div=2, active=lambda k: True)
    nbat = 10  # 6.4 MB of IQ: the 2.56 MB ring wraps twice
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, **kw)
    cap = tmp_path / "cap.iq"
    iq.tofile(cap)
    cfg = tmp_path / "cfg.txt"
    _write_cfg(cfg, dev, chans)
    r = subprocess.run([TOOL, str(cfg), str(cap), str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"batches={nbat} overruns=0 overflows=0" in r.stdout, r.stdout
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    flags = open(tmp_path / "out_axc.txt").read().splitlines()
    for c in range(len(chans)):
        got = np.fromfile(tmp_path / f"out_ch{c}.f32", dtype=np.float32)
        assert_same(got, owo[c], f"{case} ch{c} audio through the ring")
        assert flags[c] == bytes(oaxc[c]).decode(), f"{case} ch{c} axcindicate"
    # the byte streams of the reference's rawfile / udp_stream outputs for this capture (host/output_adapters.hpp)
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    for c, ch in enumerate(chans):
        sig = oaxc[c] != ord(" ")
        udp = np.fromfile(tmp_path / f"out_ch{c}.udp", dtype=np.float32)
        want = np.concatenate([owo[c, b * WAVE_BATCH:(b + 1) * WAVE_BATCH] for b in range(nbat) if sig[b]] or [np.zeros(0, np.float32)])
        assert_same(udp, want, f"{case} ch{c} udp_stream payloads (non-continuous: signal batches only)")
        if ch.has_iq_outputs:
            # non-continuous rawfile: signal batches plus the one trailing batch after each transmission (output.cpp:516-519,560)
            keep = [b for b in range(nbat) if sig[b] or (b > 0 and sig[b - 1])]
            raw = np.fromfile(tmp_path / f"out_ch{c}.cf32", dtype=np.float32)
            want = np.concatenate([oiq[c, 2 * b * WAVE_BATCH:2 * (b + 1) * WAVE_BATCH] for b in keep] or [np.zeros(0, np.float32)])
            assert_same(raw, want, f"{case} ch{c} rawfile cf32 stream")
            assert raw.size > 0


@pytest.mark.gpu
def test_four_devices_on_one_engine_equal_four_oracle_runs(pkg, tmp_path):
    """The reference's one-thread-many-devices loop (rtl_airband.cpp:300-306, 381-422, 1044-1078): four devices with the same
    configuration become the four streams of ONE engine, served by one submit / wait pair per turn from page-locked rings; a fifth
    device with a configuration of its own gets its own engine.  Every device's audio and flags equal its own oracle run."""
    _build()
    centre = 120000000
    _, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    other = [pkg.channel_cfg(centre + 250000, squelch_threshold_dbfs=-40), pkg.channel_cfg(centre + 500000, modulation=pkg.MOD_NFM)]
    nbat = 6  # 3.8 MB of IQ per device: every ring wraps
    caps, iqs, plans = [], [], []
    for d in range(5):
        ch = chans if d < 4 else other
        iq, _ = gen_iq(pkg, dev, centre, ch, nbat, stream=d, gate_div=4 + d, **({} if d < 4 else {"active": lambda k: True}))
        path = tmp_path / f"cap{d}.iq"
        iq.tofile(path)
        caps.append(str(path))
        iqs.append(iq)
        plans.append(ch)
    cfg_a, cfg_b = tmp_path / "a.txt", tmp_path / "b.txt"
    _write_cfg(cfg_a, dev, chans)
    _write_cfg(cfg_b, dev, other)
    cfgs = ",".join([str(cfg_a)] * 4 + [str(cfg_b)])
    r = subprocess.run([TOOL, cfgs, ",".join(caps), str(tmp_path / "out")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert f"devices=5 engines=2 batches={','.join([str(nbat)] * 5)} overruns=0 overflows=0" in r.stdout, r.stdout
    for d in range(5):
        nb, owo, oaxc, _ = oracle_run(dev, plans[d], iqs[d], nbat)
        assert nb == nbat
        flags = open(tmp_path / f"out_d{d}_axc.txt").read().splitlines()
        for c in range(len(plans[d])):
            got = np.fromfile(tmp_path / f"out_d{d}_ch{c}.f32", dtype=np.float32)
            assert_same(got, owo[c], f"device {d} ch{c} audio")
            assert flags[c] == bytes(oaxc[c]).decode(), f"device {d} ch{c} axcindicate"
    # the streams carry different signals: the devices really are different streams of the engine
    a = np.fromfile(tmp_path / "out_d0_ch0.f32", dtype=np.float32)
    b = np.fromfile(tmp_path / "out_d1_ch0.f32", dtype=np.float32)
    assert not np.array_equal(a, b)


# ---- the ring: input-helpers.cpp:37-63 is the spec ----

class RingModel:
    """The reference's circbuffer_append rule restated (input-helpers.cpp:37-63): two-piece copy at the wrap, the piece that
    lands at offset 0 mirrored (up to 2 * bytes_per_sample * fft_size bytes) behind buf_size, write position modulo buf_size,
    one overflow when the write position passes the read position; availability as rtl_airband.cpp:392-397."""

    def __init__(self, buf_size, fft_size, bps):
        self.n, self.tail = buf_size, 2 * bps * fft_size
        self.buf = bytearray([0xEE]) * (buf_size + self.tail)
        self.bufs = self.bufe = self.overflow = 0
        self.counter = 0

    def append(self, length):
        data = bytes(((self.counter + k) * 7 + 1) & 0xff for k in range(length))
        self.counter += length
        if length == 0:
            return
        space_left = self.n - self.bufe
        if space_left >= length:
            self.buf[self.bufe:self.bufe + length] = data
            if self.bufe == 0:
                m = min(length, self.tail)
                self.buf[self.n:self.n + m] = self.buf[0:m]
        else:
            self.buf[self.bufe:self.n] = data[:space_left]
            self.buf[0:length - space_left] = data[space_left:]
            m = min(length - space_left, self.tail)
            self.buf[self.n:self.n + m] = self.buf[0:m]
        old_end = self.bufe
        self.bufe = (self.bufe + length) % self.n
        if old_end < self.bufs and self.bufe >= self.bufs:
            self.overflow += 1

    def consume(self, n):
        self.bufs = (self.bufs + n) % self.n

    def fill(self):
        return self.bufe - self.bufs if self.bufe >= self.bufs else self.n - self.bufs + self.bufe

    def read(self, need):
        """What a consumer reading `need` bytes from the read position must see (the ring is circular)."""
        return bytes(self.buf[(self.bufs + k) % self.n] for k in range(need))


@pytest.mark.parametrize("ops", [
    # fill exactly to the end, start again at offset 0 (mirror refreshed), wrap in the middle of an append
    ["a1000", "a24", "a100", "c900", "a700", "a400", "r200"],
    # appends shorter and longer than the mirrored tail at the wrap; a read that ends inside the tail and one beyond it
    ["a1020", "c1000", "a10", "r30", "a200", "c30", "r190", "a900", "r64", "c1000", "a77", "r100"],
    # the writer laps the reader: overflow counted once per pass
    ["a600", "c100", "a600", "a600", "c50", "a1024", "a1"],
    # an append starting inside the mirrored region without a wrap leaves the tail as it was (the reference's rule)
    ["a1024", "c1024", "a5", "a20", "c20", "a1019", "r40"],
])
def test_ring_behaves_like_the_reference_rule(ops):
    """The host mirror's circbuffer_append / availability / linearisation against a model of input-helpers.cpp:37-63, byte
    for byte including the mirrored tail, on a 1024-byte ring with a 32-byte tail (fft 16, u8): every wrap and tail-pad edge."""
    exe = os.path.join(ROOT, "boondock-airband_amd", "host", "airband_replay")
    if not os.path.exists(exe):
        pytest.skip("airband_replay not built")
    r = subprocess.run([exe, "--ring-selftest", "1024", "16", "1"] + ops, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    m = RingModel(1024, 16, 1)
    it = iter(lines)
    for op in ops:
        n = int(op[1:])
        if op[0] == "a":
            m.append(n)
        elif op[0] == "c":
            m.consume(n)
        else:
            got = next(it)
            assert got == "r " + m.read(n).hex(), f"{op}: bytes handed to the engine differ from the circular read"
        assert next(it) == f"{m.bufs} {m.bufe} {m.overflow} {m.fill()}", f"after {op}"
    assert next(it) == bytes(m.buf).hex(), "ring + mirrored tail"
