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
    for sym in ("mi_demod_create", "mi_demod_process", "mi_demod_bytes_consumed", "mi_demod_destroy"):
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
