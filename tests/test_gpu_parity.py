"""GPU parity: the HIP path, called through the C ABI (libmi_airband.so), against the CPU oracle on the
same seeded synthetic IQ.  The bar (BASELINE.json north_star): audio within 1e-4 RMS, squelch open/close
decisions bit-exact.  Because the FFT arithmetic is specified operation by operation the HIP
path reproduces the oracle exactly, so these tests assert equality of every float (== semantics, i.e. up
to the sign of zero) and of every squelch flag -- strictly stronger than the stated tolerance, which is
asserted as well."""
import numpy as np
import pytest

from common import AGC_EXTRA, WAVE_BATCH, assert_same, bytes_for_batches, gen_iq, oracle_run, rms

pytestmark = pytest.mark.gpu

TOL_RMS = 1e-4  # north_star: audio output matches the CPU path within 1e-4 RMS


def run_product_batches(pkg, dev, chans, iq, nbatches, per_call=1, want_iq=False, nstreams=1):
    """Feed the host-buffer entry the way demodulate() consumes the ring: `per_call` batches per call."""
    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=per_call)
    outs, flags, iqs = [], [], []
    pos = 0
    done = 0
    while done < nbatches:
        k = min(per_call, nbatches - done)
        wo, axc, iqo, _ = d.process([iq[pos:]] * nstreams, k, want_iq=want_iq)
        outs.append(wo[:, :, :k * WAVE_BATCH])
        flags.append(axc)
        if want_iq:
            iqs.append(iqo)
        done += k
        # advance exactly like input_t.bufs (rtl_airband.cpp:691): hop bytes per window processed
        pos = (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
    st = d.stats()
    d.close()
    return np.concatenate(outs, axis=2), np.concatenate(flags, axis=2), (np.concatenate(iqs, axis=2) if want_iq else None), st


def check_against_oracle(pkg, dev, chans, iq, nbatches, per_call, want_iq=False):
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbatches, want_iq=want_iq)
    assert nb == nbatches
    wo, axc, iqo, st = run_product_batches(pkg, dev, chans, iq, nbatches, per_call=per_call, want_iq=want_iq)
    assert_same(axc[0], oaxc, "axcindicate per batch")
    # squelch decisions per sample: closed samples are exactly 0 in both
    assert_same(wo[0] != 0, owo != 0, "per-sample squelch mask")
    for c in range(len(chans)):
        assert rms(wo[0, c] - owo[c]) <= TOL_RMS
    assert_same(wo[0], owo, "audio")
    if want_iq:
        for c, ch in enumerate(chans):
            if ch.has_iq_outputs:
                assert_same(iqo[0, c].reshape(-1), oiq[c], f"iq_out ch{c}")
    return wo, axc, st


def test_config2_batch_by_batch(pkg):
    """BASELINE configs[1]: 1 stream, 8 AM channels, fft 512; one WAVE_BATCH per call like the reference loop."""
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, 16)
    wo, axc, st = check_against_oracle(pkg, dev, chans, iq, 16, per_call=1)
    assert (axc[0, 0] == ord("*")).any() and (axc[0, 1] == ord(" ")).all()
    assert st[0].open_count >= 2


def test_config2_bulk_equals_streaming(pkg):
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, 16)
    check_against_oracle(pkg, dev, chans, iq, 16, per_call=16)
    check_against_oracle(pkg, dev, chans, iq, 16, per_call=5)


def test_config3_mixed_am_nfm_ctcss(pkg):
    """BASELINE configs[2]: 32 channels AM+NFM(+CTCSS, notch), fft 2048."""
    centre, chans = pkg.config3_channels()
    chans[3].has_iq_outputs = 1
    chans[4].has_iq_outputs = 1
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=11)
    iq, _ = gen_iq(pkg, dev, centre, chans, 12, gate_div=2, active=lambda k: k % 4 != 2, amp_q8=1024)
    wo, axc, st = check_against_oracle(pkg, dev, chans, iq, 12, per_call=4, want_iq=True)
    nfm_open = [(axc[0, k] == ord("*")).any() for k in range(32) if k % 2 == 1]
    assert any(nfm_open)


@pytest.mark.parametrize("log2n", [8, 10, 12, 13])
def test_other_fft_sizes(pkg, log2n):
    centre = 120000000
    chans = [pkg.channel_cfg(centre + 300000), pkg.channel_cfg(centre - 450000, modulation=pkg.MOD_NFM),
             pkg.channel_cfg(centre + 777000, squelch_snr_db=6.0)]
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=log2n)
    iq, _ = gen_iq(pkg, dev, centre, chans, 6, gate_div=4, active=lambda k: True)
    check_against_oracle(pkg, dev, chans, iq, 6, per_call=2)


def test_manual_squelch_and_options(pkg):
    centre = 120000000
    chans = [pkg.channel_cfg(centre + 250000, squelch_threshold_dbfs=-40),
             pkg.channel_cfg(centre - 250000, squelch_snr_db=0.0, ampfactor=2.5),
             pkg.channel_cfg(centre + 500000, modulation=pkg.MOD_NFM, tau=0, notch=1000.0, notch_q=5.0),
             pkg.channel_cfg(centre - 500000, bandwidth=8000, has_iq_outputs=1),
             pkg.channel_cfg(centre + 750000, modulation=pkg.MOD_NFM, ctcss=100.0, bandwidth=12500)]
    dev = pkg.device_cfg(centerfreq=centre, tau=75)
    iq, _ = gen_iq(pkg, dev, centre, chans, 10, gate_div=2, active=lambda k: True)
    check_against_oracle(pkg, dev, chans, iq, 10, per_call=3, want_iq=True)
    dev_q = pkg.device_cfg(centerfreq=centre, fm_quadri=1)
    check_against_oracle(pkg, dev_q, chans, iq, 10, per_call=10, want_iq=True)


def test_multi_stream_independent(pkg):
    """Streams of one handle are independent devices: stream s must equal a single-stream run on its IQ."""
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    ns, nbat = 3, 6
    iqs = [gen_iq(pkg, dev, centre, chans, nbat, stream=s, gate_div=3 + s)[0] for s in range(ns)]
    d = pkg.Demod(dev, chans, nstreams=ns, max_batches=nbat)
    wo, axc, _, _ = d.process(iqs, nbat)
    d.close()
    for s in range(ns):
        nb, owo, oaxc, _ = oracle_run(dev, chans, iqs[s], nbat)
        assert nb == nbat
        assert_same(wo[s, :, :nbat * WAVE_BATCH], owo, f"stream {s} audio")
        assert_same(axc[s], oaxc, f"stream {s} flags")


def test_device_entry_and_iqgen(pkg):
    """Device-resident path with IQ generated on the GPU: bytes equal the host generator, audio equals the oracle."""
    import torch
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = 8
    iq, cfg = gen_iq(pkg, dev, centre, chans, nbat)
    nbytes = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    pkg.iqgen_device(cfg, 0, 1, nbytes, 0, iq.size // 2, d_iq.data_ptr())
    torch.cuda.synchronize()
    assert_same(d_iq.cpu().numpy()[:iq.size], iq, "device-generated IQ bytes")
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=nbat)
    d_wo = torch.zeros((1, len(chans), nbat * WAVE_BATCH), dtype=torch.float32, device="cuda")
    d_axc = torch.zeros((1, len(chans), nbat), dtype=torch.uint8, device="cuda")
    d.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_axc.data_ptr(), hip_stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert_same(d_wo.cpu().numpy()[0], owo, "audio (device entry)")
    assert_same(d_axc.cpu().numpy()[0], oaxc, "flags (device entry)")
    k1, k2 = d.last_kernel_ms()
    assert k1 > 0 and k2 > 0
    d.close()


def test_checkpoint_resume(pkg):
    centre, chans = pkg.config3_channels()
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=11)
    iq, _ = gen_iq(pkg, dev, centre, chans, 8, gate_div=2, active=lambda k: k % 4 != 2, amp_q8=1024)
    a = pkg.Demod(dev, chans, max_batches=4)
    wo1, axc1, _, _ = a.process([iq], 4)
    blob = a.get_state()
    pos = (4 * WAVE_BATCH + AGC_EXTRA) * a.hop_bytes
    wo2, axc2, _, _ = a.process([iq[pos:]], 4)
    a.close()
    b = pkg.Demod(dev, chans, max_batches=4)
    b.set_state(blob)
    wo3, axc3, _, _ = b.process([iq[pos:]], 4)
    b.close()
    assert_same(wo3, wo2, "audio after resume")
    assert_same(axc3, axc2, "flags after resume")


@pytest.mark.parametrize("sfmt", ["s8", "s16", "f32"])
def test_other_sample_formats(pkg, sfmt):
    """the s8 / s16 / f32 convert variants (rtl_airband.cpp:424-477)."""
    centre, chans = pkg.config2_channels()
    base = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, base, centre, chans, 4)
    if sfmt == "s8":
        dev = pkg.device_cfg(centerfreq=centre, sfmt=pkg.SFMT_S8)
        raw = (iq.astype(np.int16) - 128).astype(np.int8).view(np.uint8)
    elif sfmt == "s16":
        dev = pkg.device_cfg(centerfreq=centre, sfmt=pkg.SFMT_S16, fullscale=32767.5)
        raw = ((iq.astype(np.int32) - 128) * 256 + 37).astype(np.int16).view(np.uint8)
    else:
        dev = pkg.device_cfg(centerfreq=centre, sfmt=pkg.SFMT_F32, fullscale=1.0)
        raw = ((iq.astype(np.float32) - 127.5) / 127.5).astype(np.float32).view(np.uint8)
    nb, owo, oaxc, _ = oracle_run(dev, chans, raw, 4)
    assert nb == 4
    d = pkg.Demod(dev, chans, max_batches=4)
    wo, axc, _, _ = d.process([raw], 4)
    d.close()
    assert_same(wo[0, :, :4 * WAVE_BATCH], owo, f"audio {sfmt}")
    assert_same(axc[0], oaxc, f"flags {sfmt}")


def test_error_paths(pkg):
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    with pytest.raises(pkg.MiError) as e:
        pkg.Demod(pkg.device_cfg(fft_size_log=7), chans)
    assert e.value.code == pkg.MI_ERR_INVALID
    bad = list(chans)
    bad[0] = pkg.channel_cfg(chans[0].freq, afc=256)  # unsigned char in the reference (config.cpp:355)
    with pytest.raises(pkg.MiError) as e:
        pkg.Demod(dev, bad)
    assert e.value.code == pkg.MI_ERR_INVALID
    d = pkg.Demod(dev, chans, max_batches=2)
    with pytest.raises(pkg.MiError):
        d.process([np.zeros(d.bytes_needed(3), np.uint8)], 3)  # more than max_batches
    d.close()


# ---------------- time-parallel stage 2 (plain AM channels, long calls) ----------------

def _tp_run(pkg, dev, chans, iq, calls, monkeypatch, force="1"):
    """Run `calls` = [nbatches, ...] consecutive calls on one handle; returns audio, flags, stats and paths."""
    monkeypatch.setenv("MI_AIRBAND_TP", force)
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    outs, flags, paths = [], [], []
    done = 0
    for k in calls:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo, axc, _, st = d.process([iq[pos:]], k)
        outs.append(wo[:, :, :k * WAVE_BATCH])
        flags.append(axc)
        paths.append(d.last_path())
        done += k
    d.close()
    return np.concatenate(outs, axis=2), np.concatenate(flags, axis=2), st, paths


def test_time_parallel_config2_equals_oracle(pkg, monkeypatch):
    """64 batches (8 s) of BASELINE configs[1] through the time-parallel path in one call."""
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = 64
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=2)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    wo, axc, st, paths = _tp_run(pkg, dev, chans, iq, [nbat], monkeypatch)
    assert paths == [(1, 0)], "expected the time-parallel path with every segment verified"
    assert_same(axc[0], oaxc, "axcindicate (time-parallel)")
    assert_same(wo[0], owo, "audio (time-parallel)")
    assert st[0].open_count >= 3 and st[1].open_count == 0


def test_time_parallel_streaming_calls_and_path_switch(pkg, monkeypatch):
    """State carried across calls of mixed sizes, alternating between the time-parallel and the serial kernel."""
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [9, 1, 17, 3, 10]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=5)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    outs, flags, paths, done = [], [], [], 0
    for k in calls:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo, axc, _, st = d.process([iq[pos:]], k)
        outs.append(wo[:, :, :k * WAVE_BATCH])
        flags.append(axc)
        paths.append(d.last_path()[0])
        done += k
    d.close()
    assert paths == [1, 0, 1, 0, 1]
    assert_same(np.concatenate(flags, axis=2)[0], oaxc, "axcindicate")
    assert_same(np.concatenate(outs, axis=2)[0], owo, "audio")
    # counters and levels at the end equal an all-serial run
    monkeypatch.setenv("MI_AIRBAND_TP", "0")
    d2 = pkg.Demod(dev, chans, nstreams=1, max_batches=nbat)
    _, _, _, st2 = d2.process([iq], nbat)
    d2.close()
    for c in range(len(chans)):
        for f in ("noise_level", "signal_level", "squelch_level", "agcavgfast", "open_count", "flappy_count", "active_counter", "squelch_state"):
            assert getattr(st[c], f) == getattr(st2[c], f), (c, f)


def test_time_parallel_hard_signals(pkg, monkeypatch):
    """Flapping (rapid on/off), manual and low-SNR squelch, weak marginal carriers, ampfactor: the cases where
    the guessed segment start states are wrong and the scan / fix / fallback chain has to earn its keep."""
    centre = 120000000
    chans = [pkg.channel_cfg(centre + 250000), pkg.channel_cfg(centre - 250000, squelch_threshold_dbfs=-45),
             pkg.channel_cfg(centre + 500000, squelch_snr_db=2.0, ampfactor=3.0), pkg.channel_cfg(centre - 500000, squelch_snr_db=0.0),
             pkg.channel_cfg(centre + 750000), pkg.channel_cfg(centre - 750000, squelch_snr_db=14.0)]
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = 48
    n = bytes_for_batches(dev, nbat) // 2
    # gate period 0.03 s: carriers toggle every 480 output samples -> squelch re-opens within the "recent" window
    carriers = [(250000, 0, 3072, 0), (-250000, 0, 900, 1), (500000, 0, 400, 0), (-500000, 0, 2000, 1), (750000, 0, 160, 0),
                (-750000, 0, 700, 1)]
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=76800, carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, n)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    wo, axc, st, paths = _tp_run(pkg, dev, chans, iq, [nbat], monkeypatch)
    assert paths[0][0] == 1 and paths[0][1] == 0
    assert_same(axc[0], oaxc, "axcindicate")
    assert_same(wo[0], owo, "audio")
    assert max(s.flappy_count for s in st) > 0, "the input was meant to trigger flap detection"


@pytest.mark.parametrize("noise_floor_walk", ["1", "2", "0"], ids=["rounds-of-63", "rounds-of-64", "systolic-passes"])
def test_time_parallel_core_chain_exact_and_no_fallback(pkg, monkeypatch, noise_floor_walk):
    """The exact Squelch core state (noise floor, cap, capped, full) the time-parallel path computes at every
    512-step boundary equals the oracle's serial values, and the gated-carrier workload needs no serial
    fallback: only the segments holding an opening edge are re-run (scan 0), then everything verifies.
    Once for each way the chain's noise-floor wave can walk (MI_AIRBAND_CORE_GUESS: the guess-and-verify rounds on groups of
    63 blocks that the product runs, the first rounds on groups of 64, the plain systolic passes)."""
    import ctypes as C
    import libs
    from common import to_oracle_cfg
    monkeypatch.setenv("MI_AIRBAND_TP", "1")
    monkeypatch.setenv("MI_AIRBAND_CORE_GUESS", noise_floor_walk)
    centre, chans = pkg.config2_channels()
    chans[1] = pkg.channel_cfg(chans[1].freq, squelch_threshold_dbfs=-30)  # manual level: constant cap
    chans[3] = pkg.channel_cfg(chans[3].freq, squelch_snr_db=1.0)
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = 96
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=2, active=lambda k: k != 5)
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    omag, _ = od.stage1(iq, nbat * WAVE_BATCH + AGC_EXTRA, want_iq=False)
    lib = libs.oracle_lib()
    lib.ao_squelch_core_trace.argtypes = [C.POINTER(libs.SquelchCfg), libs.f32p, C.c_size_t, C.c_size_t, libs.f32p]
    lib.ao_squelch_core_trace.restype = None
    lib.ao_dbfs_to_level.restype = C.c_float
    d = pkg.Demod(dev, chans, max_batches=nbat)
    d.process([iq], nbat)
    assert d.last_path() == (1, 0)
    for c, ch in enumerate(chans):
        core, diag = d.tp_debug(c)
        n = nbat * WAVE_BATCH
        ref = np.zeros(((n + 511) // 512 + 1, 4), np.float32)
        manual = lib.ao_dbfs_to_level(C.c_float(ch.squelch_threshold_dbfs), 512) if ch.squelch_threshold_dbfs < 0 else 0.0
        cfg = libs.SquelchCfg(manual, ch.has_snr_threshold, ch.squelch_snr_db, 0.0, 16000.0)
        lib.ao_squelch_core_trace(C.byref(cfg), np.ascontiguousarray(omag[c, AGC_EXTRA:]), n, 512, ref.reshape(-1))
        assert_same(core, ref, f"core chain ch{c}")
        assert diag[3] == 0, f"ch{c}: segments left unverified: {diag.tolist()}"
        if c != 3:  # ch3 (1 dB SNR threshold) flaps on noise by construction: it exercises the serial fallback instead
            assert diag[2] == 0, f"ch{c}: serial fallback engaged, scans={diag.tolist()}"
            assert diag[0] <= nbat // 8 + 2, f"ch{c}: unexpectedly many segments re-run: {diag.tolist()}"
    d.close()


# ---- AFC (rtl_airband.cpp:180-251): the picked bin follows an off-centre carrier, batch by batch ----
def afc_case(pkg, nbat, fft_size_log=9):
    centre = 120_000_000
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=fft_size_log)
    binw = dev.sample_rate // (1 << fft_size_log)
    freqs = [centre - 900_000 + 300_000 * k for k in range(6)]
    afcs = [1, 2, 0, 5, 255, 1]
    deltas = [2 * binw, -2 * binw, 2 * binw, 3 * binw, -binw, 0]
    mods = [pkg.MOD_AM, pkg.MOD_AM, pkg.MOD_AM, pkg.MOD_AM, pkg.MOD_NFM, pkg.MOD_AM]
    chans = [pkg.channel_cfg(f, modulation=m, afc=a) for f, m, a in zip(freqs, mods, afcs)]
    carriers = [(f - centre + d, 0 if m == pkg.MOD_AM else 1, 3072, 0) for f, d, m in zip(freqs, deltas, mods)]
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate // 4, carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, bytes_for_batches(dev, nbat) // 2)
    return dev, chans, iq


@pytest.mark.parametrize("per_call", [1, 5, 16])
def test_afc_follows_the_carrier_like_the_oracle(pkg, per_call):
    dev, chans, iq = afc_case(pkg, 16)
    wo, axc, st = check_against_oracle(pkg, dev, chans, iq, 16, per_call=per_call)
    seen = [{chr(x) for x in axc[0, c]} for c in range(len(chans))]
    assert "<" in seen[0] and ">" in seen[1]          # AFC_UP / AFC_DOWN were reported
    assert seen[2] <= {" ", "*"}                       # afc = 0 never reports a move


def test_afc_other_fft_size_and_checkpoint(pkg):
    dev, chans, iq = afc_case(pkg, 12, fft_size_log=10)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, 12)
    a = pkg.Demod(dev, chans, max_batches=6)
    wo1, axc1, _, _ = a.process([iq], 6)
    blob = a.get_state()  # carries the moved bins and the previous indicator (ChanState)
    a.close()
    b = pkg.Demod(dev, chans, max_batches=6)
    b.set_state(blob)
    pos = (6 * WAVE_BATCH + AGC_EXTRA) * b.hop_bytes
    wo2, axc2, _, _ = b.process([iq[pos:]], 6)
    b.close()
    wo = np.concatenate([wo1[0, :, :6 * WAVE_BATCH], wo2[0, :, :6 * WAVE_BATCH]], axis=1)
    axc = np.concatenate([axc1[0], axc2[0]], axis=1)
    assert_same(axc, oaxc, "axcindicate per batch")
    assert_same(wo, owo, "audio")
    assert any(chr(x) in "<>" for x in axc.reshape(-1))


# ---- mixer on the device (src/mixer.cpp), fed by the audio the demod entry left in HBM ----
@pytest.mark.parametrize("stereo", [False, True])
def test_mixer_on_device_equals_oracle(pkg, stereo):
    import torch
    import libs
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat, nstreams = 8, 2
    iq0, cfg = gen_iq(pkg, dev, centre, chans, nbat)
    nbytes = (iq0.size + 255) // 256 * 256
    d_iq = torch.zeros((nstreams, nbytes), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    pkg.iqgen_device(cfg, 0, nstreams, nbytes, 0, iq0.size // 2, d_iq.data_ptr(), s)  # stream ids 0, 1: different noise
    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=nbat)
    rows = nstreams * len(chans)
    d_wo = torch.zeros((rows, nbat * WAVE_BATCH), dtype=torch.float32, device="cuda")
    d_axc = torch.zeros((rows, nbat), dtype=torch.uint8, device="cuda")
    d.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_axc.data_ptr(), hip_stream=s)
    bal = (lambda k: [-1.0, 1.0, 0.25, 0.0][k % 4]) if stereo else (lambda k: 0.0)
    inputs = [(r, [1.0, 0.5, 2.0, 0.0][k % 4], bal(k)) for k, r in enumerate([0, 9, 2, 4, 8, 10, 3, 14])]
    m = pkg.Mixer(inputs)
    assert m.stereo == stereo
    d_l = torch.full((nbat * WAVE_BATCH,), float("nan"), dtype=torch.float32, device="cuda")
    d_r = torch.full((nbat * WAVE_BATCH,), float("nan"), dtype=torch.float32, device="cuda")
    d_out = torch.zeros(nbat, dtype=torch.uint8, device="cuda")
    m.process_device(d_wo.data_ptr(), nbat * WAVE_BATCH, d_axc.data_ptr(), nbat, nbat, d_l.data_ptr(), d_r.data_ptr() if stereo else None,
                     d_out.data_ptr(), hip_stream=s)
    torch.cuda.synchronize()
    wo, axc = d_wo.cpu().numpy(), d_axc.cpu().numpy()
    assert (axc == ord("*")).any() and (axc == ord(" ")).any()
    el, er, eout = libs.oracle_mixer(inputs, wo, axc, nbat)
    assert_same(d_l.cpu().numpy(), el, "mixer left")
    assert_same(d_out.cpu().numpy(), eout, "mixer axcindicate")
    if stereo:
        assert_same(d_r.cpu().numpy(), er, "mixer right")
    m.close()
    d.close()
    with pytest.raises(pkg.MiError):
        pkg.Mixer([(0, 1.0, 1.5)])  # balance out of range (config.cpp:183-186)


def test_bench_workload_full_size_equals_oracle(pkg, monkeypatch):
    """BASELINE configs[1] at the size bench.py times: 64 s of capture (512 batches) in one device-resident call through the
    time-parallel path with its default chunk pipeline, IQ generated on the device -- audio and flags bit-exact vs the
    oracle run over the same bytes (the oracle finishes 64 s x 8 channels in a few seconds)."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    monkeypatch.delenv("MI_AIRBAND_TP_CHUNKS", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = 512
    nsteps = nbat * WAVE_BATCH
    hop = 2 * (dev.sample_rate // 16000)
    nbytes = ((nsteps + AGC_EXTRA) * hop + 2 * 512 + 255) // 256 * 256
    s = torch.cuda.current_stream().cuda_stream
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate, carriers=pkg.carriers_for(centre, chans))
    d_iq = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    pkg.iqgen_device(cfg, 0, 1, nbytes, 0, nbytes // 2, d_iq.data_ptr(), s)
    d_wo = torch.empty((1, len(chans), nsteps), dtype=torch.float32, device="cuda")
    d_axc = torch.empty((1, len(chans), nbat), dtype=torch.uint8, device="cuda")
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=nbat)
    d.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_axc.data_ptr(), hip_stream=s)
    torch.cuda.synchronize()
    assert d.last_path() == (1, 0), "time-parallel path, every segment verified"
    iq = d_iq.cpu().numpy()
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    assert_same(d_axc.cpu().numpy()[0], oaxc, "axcindicate, 512 batches")
    assert_same(d_wo.cpu().numpy()[0], owo, "audio, 512 batches")
    assert (oaxc == ord("*")).sum() >= 4 * 200 and (oaxc[1] == ord(" ")).all()
    d.close()


@pytest.mark.parametrize("chunks,ratio,lpw", [(1, 1, 64), (2, 1, 16), (5, 2, 1), (3, 1.5, 4)])
def test_time_parallel_pipeline_settings_do_not_change_a_bit(pkg, monkeypatch, chunks, ratio, lpw):
    """Chunking of the call, chunk growth and lanes per wave are scheduling choices: 160 batches (20 s, five 4-s chunk
    units) with squelch edges in every chunk and two consecutive calls give the oracle's audio under every setting."""
    monkeypatch.setenv("MI_AIRBAND_TP_CHUNKS", str(chunks))
    monkeypatch.setenv("MI_AIRBAND_TP_RATIO", str(ratio))
    monkeypatch.setenv("MI_AIRBAND_TP_LPW", str(lpw))
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [160, 40]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    wo, axc, st, paths = _tp_run(pkg, dev, chans, iq, calls, monkeypatch)
    assert paths == [(1, 0), (1, 0)]
    assert_same(axc[0], oaxc, "axcindicate")
    assert_same(wo[0], owo, "audio")


@pytest.mark.parametrize("calls", [[80, 80, 80], [48, 112, 16, 64], [160, 3, 77]])
def test_consecutive_calls_overlap_with_early_input(pkg, monkeypatch, calls):
    """MI_OPT_EARLY_INPUT: the capture is resident before the calls are made, so stage 1 and the core chain of call k+1 run
    under the segment / fix passes of call k (two scratch sets, chain state handed over on the device).  All calls are
    enqueued back to back without a host synchronisation in between; audio and flags must still be the oracle's.  A
    3-batch call in the middle takes the serial kernel and breaks the chain hand-over once."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    # On the NULL stream the wide passes run on the handle's plain streams; on any other stream on its CU-restricted twins
    # (MI_OPT_RESERVE_CUS); a handle that is given both in turn changes sides between calls.
    side = torch.cuda.Stream()
    for streams in ("null", "side", "both"):
        d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
        d.set_option(pkg.OPT_EARLY_INPUT, 1)
        outs, flags, done = [], [], 0
        for n, k in enumerate(calls):
            pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
            ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # (the buffers exist before a call on another stream may write them)
            use_side = streams == "side" or (streams == "both" and n % 2 == 1)
            s = side.cuda_stream if use_side else torch.cuda.current_stream().cuda_stream
            d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
            if streams == "both":
                torch.cuda.synchronize()  # calls on different caller streams are the caller's to order
            outs.append(wo)
            flags.append(ax)
            done += k
        torch.cuda.synchronize()
        wo = torch.cat(outs, dim=2).cpu().numpy()
        ax = torch.cat(flags, dim=2).cpu().numpy()
        d.close()
        assert_same(ax[0], oaxc, f"axcindicate ({streams} stream)")
        assert_same(wo[0], owo, f"audio ({streams} stream)")


def test_timings_of_the_five_calls_before_the_last_stay_readable(pkg, monkeypatch):
    """Six scratch / event sets cycle: a host that keeps calls in flight reads the kernel times of a call four or five calls later
    (mi_demod_kernel_time_prev) without waiting for anything that is still queued; one further back the set has been taken again."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    k, ncalls = 8, 7
    iq, _ = gen_iq(pkg, dev, centre, chans, k * ncalls, gate_div=3)
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    wo = [torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda") for _ in range(3)]
    ax = [torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=k)
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    for n in range(ncalls):
        pos = 0 if n == 0 else (n * k * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo[n % 3].data_ptr(), ax[n % 3].data_ptr(), hip_stream=side.cuda_stream)
        assert d.last_path()[0] == 1
    for age in range(6):
        t = dict((name, ms) for name, ms, _ in d.kernel_times(age=age))
        assert t.get("k_tp_core", 0.0) > 0.0, f"age {age}: {t}"
    assert d.kernel_times(age=6) == []
    torch.cuda.synchronize()
    d.close()


@pytest.mark.parametrize("split_cus", [128, 64, 0])
def test_overlapped_serial_calls_on_disjoint_cus_keep_every_bit(pkg, monkeypatch, split_cus):
    """MI_OPT_SPLIT_CUS: overlapped calls on the serial kernel with k_demod alone on the last n CUs and stage 1 of the next call on the
    others (two CU-restricted streams), on a stream of the caller's and -- where the split does not apply -- on the NULL stream, calls
    of different lengths back to back without a host synchronisation: audio and flags are the oracle's."""
    import torch
    monkeypatch.setenv("MI_AIRBAND_TP", "0")  # every call on the serial kernel
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [8, 16, 3, 8, 12]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for streams in ("side", "null", "both"):
        d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
        d.set_option(pkg.OPT_EARLY_INPUT, 1)
        d.set_option(pkg.OPT_SPLIT_CUS, split_cus)
        outs, flags, done = [], [], 0
        for n, k in enumerate(calls):
            pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
            ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            use_side = streams == "side" or (streams == "both" and n % 2 == 1)
            s = side.cuda_stream if use_side else torch.cuda.current_stream().cuda_stream
            d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
            assert d.last_path()[0] == 0
            if streams == "both":
                torch.cuda.synchronize()  # calls on different caller streams are the caller's to order
            outs.append(wo)
            flags.append(ax)
            done += k
        torch.cuda.synchronize()
        wo = torch.cat(outs, dim=2).cpu().numpy()
        ax = torch.cat(flags, dim=2).cpu().numpy()
        d.close()
        assert_same(ax[0], oaxc, f"axcindicate ({streams} stream, {split_cus} CUs)")
        assert_same(wo[0], owo, f"audio ({streams} stream, {split_cus} CUs)")


@pytest.mark.parametrize("spec_head,core_split,seg", [(1, 1, 512), (0, 1, 512), (1, 0, 512), (0, 0, 512), (1, 1, 2048), (1, 1, 4096)])
def test_overlapped_calls_speculative_head_and_split_chain_keep_every_bit(pkg, monkeypatch, spec_head, core_split, seg):
    """Overlapped calls with alternating audio buffers: the first segments of a call warm up on the previous call's planes from a
    guessed state (MI_OPT_SPEC_HEAD) and the core chain runs on three waves (MI_OPT_CORE_SPLIT) -- or not; the carriers are
    gated with a period that puts open squelches, decays and bursts across the call boundaries.  Audio and flags are the
    oracle's either way, for three segment lengths (at 4096 the head is a single segment, at 2048 two)."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    monkeypatch.setenv("MI_AIRBAND_TP_SEGMENT", str(seg))
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [40, 24, 56, 8, 32]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=5)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    d.set_option(pkg.OPT_SPEC_HEAD, spec_head)
    d.set_option(pkg.OPT_CORE_SPLIT, core_split)
    outs, flags, done = [], [], 0
    for k in calls:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
        ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
        d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
        assert d.last_path()[0] == 1  # every call time-parallel
        outs.append(wo)
        flags.append(ax)
        done += k
    torch.cuda.synchronize()
    wo = torch.cat(outs, dim=2).cpu().numpy()
    ax = torch.cat(flags, dim=2).cpu().numpy()
    d.close()
    assert (oaxc == 42).any() and (oaxc == 32).any()  # (MI_SIGNAL / MI_NO_SIGNAL: squelches open and close in the capture)
    assert_same(ax[0], oaxc, "axcindicate")
    assert_same(wo[0], owo, "audio")


def test_event_timeline_of_overlapping_calls(pkg, monkeypatch):
    """mi_demod_event_ms (diagnostic): the launches of consecutive overlapped time-parallel calls in time.  Within a call stage 1
    ends before the aggregates, those before the core chain, that before the segment pass, that before the scan; the core
    chains of consecutive calls run in order; a call that was not time-parallel, a chunk that does not exist or an age beyond
    the kept sets is refused."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nbat, ncalls = 16, 4
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat * ncalls, gate_div=3)
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=nbat)
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    bufs = [(torch.empty((1, len(chans), nbat * WAVE_BATCH), dtype=torch.float32, device="cuda"),
             torch.empty((1, len(chans), nbat), dtype=torch.uint8, device="cuda")) for _ in range(3)]
    assert d.event_ms(0, 0, 0, 3) is None  # nothing has run yet
    for k in range(ncalls):
        pos = 0 if k == 0 else (k * nbat * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo, ax = bufs[k % 3]
        d.process_device(d_iq.data_ptr() + pos, pad - pos, nbat, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
    torch.cuda.synchronize()
    for age in (2, 1, 0):
        t = {e: d.event_ms(2, age, 0, e) for e in (0, 1, 11, 2, 3, 4, 5, 12, 10, 7, 8, 9)}
        assert all(v is not None for v in t.values()), (age, t)
        assert t[0] <= t[1] <= t[2] <= t[3] + 1e-3 and t[3] <= t[4] <= t[5] + 1e-3 and t[5] <= t[12] <= t[10] + 1e-3 and t[10] <= t[7] <= t[8] <= t[9], (age, t)
    core = [(d.event_ms(2, age, 0, 3), d.event_ms(2, age, 0, 4)) for age in (2, 1, 0)]
    assert core[0][0] == 0.0 and core[0][1] <= core[1][0] + 1e-3 and core[1][1] <= core[2][0] + 1e-3, core
    assert d.event_ms(2, 0, 1, 3) is None   # overlapping calls of 8 rows have one chunk
    assert d.event_ms(1, 2, 0, 3) is None   # the reference call must not be younger than the call asked about
    assert d.event_ms(4, 0, 0, 3) is None   # beyond the kept sets
    d.close()


def test_checkpoint_between_overlapping_calls_and_timing_ages(pkg, monkeypatch):
    """A checkpoint taken while calls are in flight (it drains them), restored into a fresh handle, continues bit for bit; the
    per-kernel timings of the last three calls stay readable by age."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [40, 40, 40, 40]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3)
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    pad = (iq.size + 255) // 256 * 256
    d_iq = torch.zeros(pad, dtype=torch.uint8, device="cuda")
    d_iq[:iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream

    def run(d, first, n):
        outs = []
        done = first
        for k in calls[first:first + n]:
            b = sum(calls[:done])
            pos = 0 if b == 0 else (b * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo = torch.empty((1, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
            ax = torch.empty((1, len(chans), k), dtype=torch.uint8, device="cuda")
            d.process_device(d_iq.data_ptr() + pos, pad - pos, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
            outs.append((wo, ax))
            done += 1
        return outs

    a = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    a.set_option(pkg.OPT_EARLY_INPUT, 1)
    first = run(a, 0, 3)
    ages = [dict((n, ms) for n, ms, _ in a.kernel_times(age=g)) for g in range(3)]
    assert all("k_tp_core" in t and t["k_tp_core"] > 0 for t in ages)
    blob = a.get_state()  # (synchronises the device)
    rest_a = run(a, 3, 1)
    torch.cuda.synchronize()
    a.close()
    b = pkg.Demod(dev, chans, nstreams=1, max_batches=max(calls))
    b.set_option(pkg.OPT_EARLY_INPUT, 1)
    b.set_state(blob)
    rest_b = run(b, 3, 1)
    torch.cuda.synchronize()
    b.close()
    wo = torch.cat([x[0] for x in first + rest_a], dim=2).cpu().numpy()
    ax = torch.cat([x[1] for x in first + rest_a], dim=2).cpu().numpy()
    assert_same(ax[0], oaxc, "axcindicate")
    assert_same(wo[0], owo, "audio")
    assert_same(rest_b[0][0].cpu().numpy(), rest_a[0][0].cpu().numpy(), "audio after restoring the checkpoint")
    assert_same(rest_b[0][1].cpu().numpy(), rest_a[0][1].cpu().numpy(), "flags after restoring the checkpoint")


STAGE1_VARIANTS = {
    # name: (options, the MI_STAGE1_* kind the handle must report at N = 512)
    "lane-resident, compiled for the plan (default)": ({}, 3),
    "lane-resident, prebuilt full graph": ({"OPT_LANE_FFT_JIT": 0}, 2),
    "exchange kernel, pruned graph, level table": ({"OPT_LANE_FFT": 0}, 1),
    "exchange kernel, pruned graph, arithmetic conversion": ({"OPT_LANE_FFT": 0, "OPT_U8_CONVERSION": 1}, 1),
    "exchange kernel, full graph, level table": ({"OPT_LANE_FFT": 0, "OPT_PRUNE_FFT": 0, "OPT_U8_CONVERSION": 0}, 0),
}


@pytest.mark.parametrize("variant", list(STAGE1_VARIANTS))
def test_stage1_alternative_instantiations_keep_every_bit(pkg, variant):
    """Stage 1 at N = 512 exists in several instantiations of the same radix-2 DIT graph: the lane-resident kernel (first six
    stages in registers) compiled by hipRTC for exactly the FFT nodes the plan's bins need, the same kernel prebuilt for the
    full graph, and the radix-8 exchange kernels (pruned or full graph, u8 through the level table or arithmetically).  Any
    subset of the graph is computed with the same operations on the same operands, so audio, flags and raw I/Q are bit-exact
    with the oracle for every one of them -- on the 8-channel AM plan and on the 32-channel mixed plan, calls of 6 and of
    1 + 5 batches (a tail tile of 4 windows, a first call with AGC_EXTRA more windows)."""
    opts, kind = STAGE1_VARIANTS[variant]
    for name in ("config2", "config3"):
        centre, chans = getattr(pkg, name + "_channels")()
        if name == "config3":
            for c in (1, 6, 17):
                chans[c].has_iq_outputs = 1
        dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
        nbat = 6
        kw = {} if name == "config2" else dict(amp_q8=1024, active=lambda k: k % 4 != 2)
        iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=2, **kw)
        nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
        assert nb == nbat
        for calls in ([6], [1, 5]):
            d = pkg.Demod(dev, chans, max_batches=max(calls))
            for k, v in opts.items():
                d.set_option(getattr(pkg, k), v)
            outs, flags, iqs, done = [], [], [], 0
            for k in calls:
                pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
                wo, axc, iqo, _ = d.process([iq[pos:]], k, want_iq=True)
                outs.append(wo[:, :, :k * WAVE_BATCH]), flags.append(axc), iqs.append(iqo)
                done += k
            assert d.last_stage1() == kind, f"{variant}: the handle ran stage-1 kernel kind {d.last_stage1()}"
            d.close()
            wo, axc, iqo = np.concatenate(outs, axis=2), np.concatenate(flags, axis=2), np.concatenate(iqs, axis=2)
            assert_same(axc[0], oaxc, f"{variant}, {name}, calls {calls}: flags")
            assert_same(wo[0], owo, f"{variant}, {name}, calls {calls}: audio")
            for c, ch in enumerate(chans):
                if ch.has_iq_outputs:
                    assert_same(iqo[0, c].reshape(-1), oiq[c], f"{variant}, {name}, calls {calls}: raw I/Q ch{c}")


@pytest.mark.parametrize("sfmt", ["s8", "s16", "f32"])
@pytest.mark.parametrize("misalign", [0, 1])
def test_lane_resident_stage1_formats_and_odd_alignment(pkg, sfmt, misalign):
    """The lane-resident kernel converts a tile's samples once into a float span, two samples per lane from one aligned load.
    Other sample formats take the same path with wider loads, and a capture that starts on an odd sample (device pointer not
    aligned to a pair of samples) goes through the unaligned conversion: both equal the exchange kernels bit for bit (which
    the other tests pin to the oracle) on planes, audio and flags."""
    import torch
    centre, chans = pkg.config2_channels()
    chans[2].has_iq_outputs = 1
    chans[2].bandwidth = 8000
    code = {"s8": pkg.SFMT_S8, "s16": pkg.SFMT_S16, "f32": pkg.SFMT_F32}[sfmt]
    dev = pkg.device_cfg(centerfreq=centre, sfmt=code, fullscale={"s8": 127.5, "s16": 32767.0, "f32": 1.0}[sfmt])
    nbat = 3
    u8, _ = gen_iq(pkg, pkg.device_cfg(centerfreq=centre), centre, chans, nbat, gate_div=8)
    x = u8.astype(np.float32) - 127.5
    raw = {"s8": lambda: np.round(x - 0.5).astype(np.int8), "s16": lambda: np.round(x * 200.0).astype(np.int16),
           "f32": lambda: (x / 128.0).astype(np.float32)}[sfmt]()
    bps2 = 2 * raw.itemsize
    pad = np.zeros(2 * misalign, raw.dtype)  # one complex sample in front: the capture then starts on an odd sample
    buf = np.concatenate([pad, raw, np.zeros(64, raw.dtype)]).view(np.uint8)
    d_buf = torch.from_numpy(buf).cuda()
    base = d_buf.data_ptr() + misalign * bps2
    assert base % bps2 == 0 and (base % (2 * bps2) != 0) == bool(misalign)
    res = {}
    for lane_fft in (1, 0):
        d = pkg.Demod(dev, chans, max_batches=nbat)
        d.set_option(pkg.OPT_LANE_FFT, lane_fft)
        d_wo = torch.zeros((1, len(chans), nbat * WAVE_BATCH), dtype=torch.float32, device="cuda")
        d_ax = torch.zeros((1, len(chans), nbat), dtype=torch.uint8, device="cuda")
        d_zo = torch.zeros((1, len(chans), nbat * WAVE_BATCH, 2), dtype=torch.float32, device="cuda")
        d.process_device(base, 0, nbat, d_wo.data_ptr(), d_ax.data_ptr(), d_iq_out_ptr=d_zo.data_ptr(), hip_stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert d.last_stage1() == (3 if lane_fft else 1)
        planes = [d.read_planes(0, c, 0, nbat * WAVE_BATCH + AGC_EXTRA, want_iq=True) for c in range(len(chans))]
        d.close()
        res[lane_fft] = (d_wo.cpu().numpy(), d_ax.cpu().numpy(), d_zo.cpu().numpy(), planes)
    assert (res[0][1] == ord("*")).any()
    assert_same(res[1][0], res[0][0], "audio")
    assert_same(res[1][1], res[0][1], "flags")
    assert_same(res[1][2][0, 2], res[0][2][0, 2], "raw I/Q")
    for c in range(len(chans)):
        assert_same(res[1][3][c][0], res[0][3][c][0], f"magnitude plane ch{c}")
    assert_same(res[1][3][2][1], res[0][3][2][1], "complex plane ch2")


def _channel_zoo(pkg):
    """24 channels at fft 512 covering every branch of the serial stage 2: AM / NFM, with and without the low-pass, notch,
    CTCSS (right tone, no tone), manual squelch level, raw-I/Q outputs, odd ampfactors and de-emphasis constants."""
    centre = 120000000
    kinds = [
        dict(),
        dict(bandwidth=8000),
        dict(notch=1000.0),
        dict(squelch_snr_db=6.0, has_iq_outputs=1),
        dict(squelch_threshold_dbfs=-45),
        dict(bandwidth=5000, notch=400.0, ampfactor=2.5),
        dict(modulation=pkg.MOD_NFM),
        dict(modulation=pkg.MOD_NFM, bandwidth=12500),
        dict(modulation=pkg.MOD_NFM, bandwidth=12500, ctcss=100.0),
        dict(modulation=pkg.MOD_NFM, bandwidth=12500, ctcss=123.0),
        dict(modulation=pkg.MOD_NFM, bandwidth=8000, notch=100.0, tau=75, has_iq_outputs=1),
        dict(modulation=pkg.MOD_NFM, squelch_snr_db=4.0, ampfactor=0.5),
    ]
    chans = []
    for k in range(24):
        f = centre - 1150000 + 25000 + k * 95000
        chans.append(pkg.channel_cfg(f, **kinds[k % len(kinds)]))
    return centre, chans


def _zoo_capture(pkg, dev, centre, chans, nbat, seed):
    """Carriers of very different strength (some under the squelch level, some flapping around it, some strong enough to
    clip the AM AGC), gated with different phases and a short period so that every squelch transition happens many times; a
    CTCSS channel gets a carrier without the tone, an NFM channel an AM carrier."""
    carriers = []
    amps = [3072, 700, 1500, 260, 5000, 400, 2048, 900]
    for k, c in enumerate(chans):
        if k % 5 == 4:
            continue
        kind = 0 if c.modulation == pkg.MOD_AM else (2 if c.ctcss_freq > 0 and k % 2 == 0 else 1)
        if k == 6:
            kind = 0
        carriers.append((c.freq - centre, kind, amps[k % len(amps)], (k * 7919) % 1000))
    n = bytes_for_batches(dev, nbat) // 2
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, seed=seed, gate_samples=dev.sample_rate // 5, carriers=carriers)
    return pkg.iqgen_host(cfg, 0, 0, n)


@pytest.mark.parametrize("seed", [0xA1B2C3D4, 77])
def test_steady_blocks_same_bits_and_state(pkg, seed):
    """The serial stage 2 takes steady CLOSED / OPEN runs 64 steps at a time (demod.hip, steady_block).  With the blocks on
    and off the audio, the batch flags, the raw I/Q, the statistics and the complete checkpoint state are identical, call by
    call, and equal to the oracle -- over a channel zoo and a capture that drive every way out of a block."""
    centre, chans = _channel_zoo(pkg)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    nbat, per_call = 12, 4
    iq = _zoo_capture(pkg, dev, centre, chans, nbat, seed)
    res = {}
    for steady in (1, 0):
        d = pkg.Demod(dev, chans, max_batches=per_call)
        d.set_option(pkg.OPT_STEADY_BLOCKS, steady)
        outs = []
        for call in range(nbat // per_call):
            pos = (call * per_call * WAVE_BATCH) * d.hop_bytes if call == 0 else (call * per_call * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo, axc, iqo, _ = d.process([iq[pos:]], per_call, want_iq=True)
            outs.append((wo.copy(), axc.copy(), iqo.copy(), bytes(d.stats()), d.get_state().copy()))
        d.close()
        res[steady] = outs
    for call, (a, b) in enumerate(zip(res[1], res[0])):
        assert_same(a[0], b[0], f"audio, call {call}")
        assert_same(a[1], b[1], f"flags, call {call}")
        assert_same(a[2], b[2], f"raw I/Q, call {call}")
        assert a[3] == b[3], f"statistics differ after call {call}"
        assert_same(a[4], b[4], f"checkpoint state, call {call}")
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    wo = np.concatenate([o[0][:, :, :per_call * WAVE_BATCH] for o in res[1]], axis=2)
    axc = np.concatenate([o[1] for o in res[1]], axis=2)
    assert_same(axc[0], oaxc, "flags vs oracle")
    assert_same(wo[0], owo, "audio vs oracle")
    opened = [(axc[0, k] == ord("*")).any() for k in range(len(chans))]
    assert sum(opened) >= 8 and not all(opened)


@pytest.mark.parametrize("seed,per_call", [(0xA1B2C3D4, 4), (77, 1), (5, 12)])
def test_pre_filter_wave_same_bits_and_state(pkg, seed, per_call):
    """MI_OPT_PRE_WAVE: a second wave per channel walks the squelch's pre-filter averages and noise floor ahead of the channel's
    own wave, whose steady blocks take the values from a ring in LDS (demod.hip, k_demod_pw).  Forced on and off -- over the
    channel zoo (AM, NFM, low-pass channels that rewrite wavein[], CTCSS, notch, manual levels) in calls of 1, 4 and 12 batches
    -- audio, flags, raw I/Q, statistics and the complete checkpoint state are identical, and equal to the oracle."""
    centre, chans = _channel_zoo(pkg)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    nbat = 12
    iq = _zoo_capture(pkg, dev, centre, chans, nbat, seed)
    res = {}
    for on in (1, 2, 0):  # four waves per channel (k_demod_pw), two (k_demod_pw2: the channel with its audio + the pre-filter wave), one
        d = pkg.Demod(dev, chans, max_batches=per_call)
        d.set_option(pkg.OPT_PRE_WAVE, on)
        outs = []
        for call in range(nbat // per_call):
            pos = (call * per_call * WAVE_BATCH) * d.hop_bytes if call == 0 else (call * per_call * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo, axc, iqo, _ = d.process([iq[pos:]], per_call, want_iq=True)
            outs.append((wo.copy(), axc.copy(), iqo.copy(), bytes(d.stats()), d.get_state().copy()))
        assert d.pre_wave_timeouts() == 0, "a channel wave gave up waiting for its pre-filter wave"
        d.close()
        res[on] = outs
    for on in (1, 2):
        for call, (a, b) in enumerate(zip(res[on], res[0])):
            assert_same(a[0], b[0], f"audio, call {call}, {on}")
            assert_same(a[1], b[1], f"flags, call {call}, {on}")
            assert_same(a[2], b[2], f"raw I/Q, call {call}, {on}")
            assert a[3] == b[3], f"statistics differ after call {call}, {on}"
            assert_same(a[4], b[4], f"checkpoint state, call {call}, {on}")
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    wo = np.concatenate([o[0][:, :, :per_call * WAVE_BATCH] for o in res[1]], axis=2)
    axc = np.concatenate([o[1] for o in res[1]], axis=2)
    assert_same(axc[0], oaxc, "flags vs oracle")
    assert_same(wo[0], owo, "audio vs oracle")


@pytest.mark.parametrize("seed,per_call", [(0xA1B2C3D4, 4), (77, 1), (5, 12)])
def test_audio_wave_same_bits_and_state(pkg, seed, per_call):
    """MI_OPT_AUDIO_WAVE (k_demod_pw): everything behind the squelch decisions -- NFM discriminator, DC block, de-emphasis, the CTCSS
    detector banks with their windows and counters, output gate, notch filter, axcindicate, the audio / raw-I/Q stores; for plain AM
    channels the AGC with its clip feedback, the open-edge bootstrap and the close-edge fade -- runs on a wave of its own, fed by the
    channel's wave through a token ring in LDS (demod.hip, audio_wave).  On and off over the channel zoo in calls of 1, 4 and 12
    batches: audio, flags, raw I/Q, statistics and the complete checkpoint state are identical, and equal to the oracle."""
    centre, chans = _channel_zoo(pkg)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    nbat = 12
    iq = _zoo_capture(pkg, dev, centre, chans, nbat, seed)
    res = {}
    for on in (1, 0):
        d = pkg.Demod(dev, chans, max_batches=per_call)
        d.set_option(pkg.OPT_PRE_WAVE, 1)
        d.set_option(pkg.OPT_AUDIO_WAVE, on)
        outs = []
        for call in range(nbat // per_call):
            pos = (call * per_call * WAVE_BATCH) * d.hop_bytes if call == 0 else (call * per_call * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo, axc, iqo, _ = d.process([iq[pos:]], per_call, want_iq=True)
            outs.append((wo.copy(), axc.copy(), iqo.copy(), bytes(d.stats()), d.get_state().copy()))
        assert d.pre_wave_timeouts() == 0, "a wave of a channel's workgroup gave up waiting for another"
        d.close()
        res[on] = outs
    for call, (a, b) in enumerate(zip(res[1], res[0])):
        assert_same(a[0], b[0], f"audio, call {call}")
        assert_same(a[1], b[1], f"flags, call {call}")
        assert_same(a[2], b[2], f"raw I/Q, call {call}")
        assert a[3] == b[3], f"statistics differ after call {call}"
        assert_same(a[4], b[4], f"checkpoint state, call {call}")
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    wo = np.concatenate([o[0][:, :, :per_call * WAVE_BATCH] for o in res[1]], axis=2)
    axc = np.concatenate([o[1] for o in res[1]], axis=2)
    assert_same(axc[0], oaxc, "flags vs oracle")
    assert_same(wo[0], owo, "audio vs oracle")


@pytest.mark.parametrize("mixed", [1, 0])
def test_mixed_plan_rows_take_both_paths(pkg, monkeypatch, mixed):
    """MI_OPT_MIXED_PLAN: in a plan that holds plain AM channels beside others (the 24-channel zoo: AM with a manual level and with an
    SNR threshold, low-pass, notch, raw I/Q, NFM, CTCSS) a call of 8 batches and more sends the plain AM rows down the time-parallel
    path and the rest through the serial kernel on a stream of its own -- the raw bins in alternating complex plane sets, the serial
    rows' carried samples left in the next call's planes, both halves sharing one audio lookahead buffer per call.  Two streams,
    calls of 8 / 12 / 8 batches overlapping on alternating audio buffers, then a 2-batch call (serial for every row) and a last
    mixed one after it, then a checkpoint taken in the middle is restored into a fresh handle: audio, flags and raw I/Q equal the
    oracle's throughout, with the option on and off."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    centre, chans = _channel_zoo(pkg)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    calls = [8, 12, 8, 2, 10]
    nbat = sum(calls)
    nstreams = 2
    iqs = [_zoo_capture(pkg, dev, centre, chans, nbat, 1234 + st) for st in range(nstreams)]
    oracle = [oracle_run(dev, chans, iq, nbat, want_iq=True) for iq in iqs]
    pad = (max(iq.size for iq in iqs) + 255) // 256 * 256
    d_iq = torch.zeros((nstreams, pad), dtype=torch.uint8, device="cuda")
    for st, iq in enumerate(iqs):
        d_iq[st, :iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()

    def run(d, first_call, ncalls, done):
        outs, flags, zs = [], [], []
        for k in calls[first_call:first_call + ncalls]:
            # (by itself a mixed plan splits from 64 batches per call on: forced here for the long calls, off for the short one)
            d.set_option(pkg.OPT_TIME_PARALLEL, 1 if k >= 8 else 0)
            pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo = torch.empty((nstreams, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
            ax = torch.empty((nstreams, len(chans), k), dtype=torch.uint8, device="cuda")
            zo = torch.zeros((nstreams, len(chans), k * WAVE_BATCH, 2), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            d.process_device(d_iq.data_ptr() + pos, pad, k, wo.data_ptr(), ax.data_ptr(), d_iq_out_ptr=zo.data_ptr(), hip_stream=side.cuda_stream)
            outs.append(wo), flags.append(ax), zs.append(zo)
            done += k
        torch.cuda.synchronize()
        return outs, flags, zs, done

    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    d.set_option(pkg.OPT_MIXED_PLAN, mixed)
    o1, f1, z1, done = run(d, 0, 3, 0)
    assert d.last_path()[0] == mixed
    state = d.get_state().copy()
    o2, f2, z2, done2 = run(d, 3, 2, done)
    assert d.last_path() == (mixed, 0)
    assert d.pre_wave_timeouts() == 0
    d.close()
    # the same two calls again from the checkpoint, in a fresh handle
    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    d.set_option(pkg.OPT_MIXED_PLAN, mixed)
    d.set_state(state)
    o3, f3, z3, _ = run(d, 3, 2, done)
    d.close()
    for name, outs, flags, zs in (("straight", o1 + o2, f1 + f2, z1 + z2), ("restored", o1 + o3, f1 + f3, z1 + z3)):
        wo = torch.cat(outs, dim=2).cpu().numpy()
        ax = torch.cat(flags, dim=2).cpu().numpy()
        zo = torch.cat(zs, dim=2).cpu().numpy()
        for st in range(nstreams):
            nb, owo, oaxc, oiq = oracle[st]
            assert nb == nbat
            assert_same(ax[st], oaxc, f"flags, stream {st}, {name}")
            assert_same(wo[st], owo, f"audio, stream {st}, {name}")
            for c, ch in enumerate(chans):
                if ch.has_iq_outputs:
                    assert_same(zo[st, c].reshape(-1), oiq[c], f"raw I/Q, stream {st}, channel {c}, {name}")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_steady_blocks_random_plans(pkg, seed):
    """Random channel plans and captures (modulation, low-pass, notch, CTCSS, manual / SNR squelch thresholds down to 0 dB,
    amplification, carriers from far under the squelch level to clipping, short gate periods): steady blocks on and off
    leave the same audio, flags, raw I/Q and checkpoint state after every call.  (The oracle is not in this loop: both
    sides are the product; the oracle comparison of the same code paths is test_steady_blocks_same_bits_and_state.)"""
    rng = np.random.default_rng(seed)
    centre = 120000000
    nchan = 16
    chans, carriers = [], []
    for k in range(nchan):
        f = centre - 1200000 + 40000 + k * 150000 + int(rng.integers(0, 20)) * 5000
        kw = {}
        nfm = rng.random() < 0.5
        if nfm:
            kw["modulation"] = pkg.MOD_NFM
        if rng.random() < 0.5:
            kw["bandwidth"] = int(rng.choice([5000, 8000, 12500]))
        if rng.random() < 0.25:
            kw["notch"] = float(rng.choice([100.0, 400.0, 1000.0]))
        if nfm and rng.random() < 0.4:
            kw["ctcss"] = float(rng.choice([100.0, 123.0, 151.4]))
        r = rng.random()
        if r < 0.2:
            kw["squelch_threshold_dbfs"] = int(rng.integers(-55, -30))
        elif r < 0.5:
            kw["squelch_snr_db"] = float(rng.choice([0.0, 3.0, 6.0, 12.0]))
        if rng.random() < 0.3:
            kw["ampfactor"] = float(rng.choice([0.5, 2.0, 4.0]))
        if rng.random() < 0.3:
            kw["has_iq_outputs"] = 1
        chans.append(pkg.channel_cfg(f, **kw))
        if rng.random() < 0.8:
            kind = int(rng.integers(0, 3))
            carriers.append((f - centre, kind, int(rng.choice([150, 300, 600, 1200, 2500, 5000])), int(rng.integers(0, 1000))))
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9, fm_quadri=int(seed % 2))
    nbat, per_call = 8, 4
    n = bytes_for_batches(dev, nbat) // 2
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, seed=1000 + seed, gate_samples=dev.sample_rate // int(rng.integers(3, 9)), carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, n)
    res = {}
    for steady in (1, 0):
        d = pkg.Demod(dev, chans, max_batches=per_call)
        d.set_option(pkg.OPT_STEADY_BLOCKS, steady)
        outs = []
        for call in range(nbat // per_call):
            pos = 0 if call == 0 else (call * per_call * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo, axc, iqo, _ = d.process([iq[pos:]], per_call, want_iq=True)
            outs.append((wo.copy(), axc.copy(), iqo.copy(), bytes(d.stats()), d.get_state().copy()))
        d.close()
        res[steady] = outs
    for call, (a, b) in enumerate(zip(res[1], res[0])):
        assert_same(a[0], b[0], f"audio, call {call}")
        assert_same(a[1], b[1], f"flags, call {call}")
        assert_same(a[2], b[2], f"raw I/Q, call {call}")
        assert a[3] == b[3], f"statistics differ after call {call}"
        assert_same(a[4], b[4], f"checkpoint state, call {call}")


@pytest.mark.parametrize("nstreams", [1, 3])
def test_serial_calls_overlap_with_early_input(pkg, nstreams):
    """MI_OPT_EARLY_INPUT on a plan the time-parallel path cannot take (AM + NFM + low-pass + CTCSS + notch + raw I/Q): stage 1
    of call k+1 runs under k_demod of call k on two alternating plane sets, the carried head written across.  Calls of
    different sizes enqueued back to back, a checkpoint taken in the middle and restored into a fresh handle; audio, flags
    and raw I/Q equal the oracle's, and the kernel timings of the pipelined calls are readable."""
    import torch
    centre, chans = _channel_zoo(pkg)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    calls = [3, 1, 4, 2, 5, 1]
    nbat = sum(calls)
    iqs = [_zoo_capture(pkg, dev, centre, chans, nbat, 31 + 7 * st) for st in range(nstreams)]
    oracle = [oracle_run(dev, chans, iq, nbat, want_iq=True) for iq in iqs]
    stride = (max(iq.size for iq in iqs) + 255) // 256 * 256
    d_iq = torch.zeros((nstreams, stride), dtype=torch.uint8, device="cuda")
    for st, iq in enumerate(iqs):
        d_iq[st, :iq.size] = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    nch = len(chans)

    def run(d, todo, done, outs):
        for k in todo:
            pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
            wo = torch.empty((nstreams, nch, k * WAVE_BATCH), dtype=torch.float32, device="cuda")
            ax = torch.empty((nstreams, nch, k), dtype=torch.uint8, device="cuda")
            zo = torch.empty((nstreams, nch, k * WAVE_BATCH, 2), dtype=torch.float32, device="cuda")
            d.process_device(d_iq.data_ptr() + pos, stride, k, wo.data_ptr(), ax.data_ptr(), d_iq_out_ptr=zo.data_ptr(), hip_stream=s)
            outs.append((wo, ax, zo))
            done += k
        return done

    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    outs = []
    done = run(d, calls[:3], 0, outs)
    times = d.kernel_times()
    assert [t[0] for t in times] == ["k_channelize", "k_demod"] and all(t[1] > 0 for t in times)
    blob = d.get_state()  # drains the calls in flight
    done2 = run(d, calls[3:], done, outs)
    torch.cuda.synchronize()
    d.close()
    e = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=max(calls))
    e.set_option(pkg.OPT_EARLY_INPUT, 1)
    e.set_state(blob)
    outs_e = []
    run(e, calls[3:], done, outs_e)
    torch.cuda.synchronize()
    e.close()
    assert done2 == nbat
    wo = torch.cat([o[0] for o in outs], dim=2).cpu().numpy()
    ax = torch.cat([o[1] for o in outs], dim=2).cpu().numpy()
    zo = torch.cat([o[2] for o in outs], dim=2).cpu().numpy()
    for st in range(nstreams):
        nb, owo, oaxc, oiq = oracle[st]
        assert_same(ax[st], oaxc, f"flags, stream {st}")
        assert_same(wo[st], owo, f"audio, stream {st}")
        for c, ch in enumerate(chans):
            if ch.has_iq_outputs:
                assert_same(zo[st, c].reshape(-1), oiq[c], f"raw I/Q, stream {st} channel {c}")
    for (a, b) in zip(outs[3:], outs_e):
        assert_same(b[0].cpu().numpy(), a[0].cpu().numpy(), "audio after restoring the checkpoint")
        assert_same(b[1].cpu().numpy(), a[1].cpu().numpy(), "flags after restoring the checkpoint")


def _config4_case(pkg, nstreams, nbat):
    """BASELINE configs[3]: `nstreams` device streams x the 32-channel config-3 plan (AM / NFM + low-pass / CTCSS / notch, four
    channels with raw-I/Q outputs) at fft 512.  The capture of every stream is generated on the device (the bench's
    generator, seed ^ stream id) and copied back for the oracle, so both sides see the same bytes."""
    import torch
    centre, chans = pkg.config3_channels()
    for c in (1, 6, 17, 31):
        chans[c].has_iq_outputs = 1
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    nbytes = (bytes_for_batches(dev, nbat) + 255) // 256 * 256
    # a short gate so that every stream opens and closes inside the few batches; gate phase differs per channel
    gcfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate // 6,
                         carriers=pkg.carriers_for(centre, chans, amp_q8=1024, active=lambda k: k % 4 != 2))
    d_iq = torch.zeros((nstreams, nbytes), dtype=torch.uint8, device="cuda")
    pkg.iqgen_device(gcfg, 0, nstreams, nbytes, 0, nbytes // 2, d_iq.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dev, chans, d_iq, nbytes


def _run_config4(pkg, dev, chans, d_iq, nbytes, nstreams, nbat, uni_rows=None, steady=None):
    import torch
    nch = len(chans)
    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=nbat)
    if uni_rows is not None:
        d.set_option(pkg.OPT_UNI_ROWS, uni_rows)
    if steady is not None:
        d.set_option(pkg.OPT_STEADY_BLOCKS, steady)
    d_wo = torch.zeros((nstreams, nch, nbat * WAVE_BATCH), dtype=torch.float32, device="cuda")
    d_ax = torch.zeros((nstreams, nch, nbat), dtype=torch.uint8, device="cuda")
    d_zo = torch.zeros((nstreams, nch, nbat * WAVE_BATCH, 2), dtype=torch.float32, device="cuda")
    d.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_ax.data_ptr(), d_iq_out_ptr=d_zo.data_ptr(),
                     hip_stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    path = d.last_path()[0]
    d.close()
    return d_wo.cpu().numpy(), d_ax.cpu().numpy(), d_zo.cpu().numpy(), path


def _check_streams_against_oracle(dev, chans, iq_host, wo, ax, zo, nbat, what):
    opened = 0
    for st in range(iq_host.shape[0]):
        nb, owo, oaxc, oiq = oracle_run(dev, chans, iq_host[st], nbat, want_iq=True)
        assert nb == nbat
        assert_same(ax[st], oaxc, f"{what}: flags, stream {st}")
        assert_same(wo[st], owo, f"{what}: audio, stream {st}")
        for c, ch in enumerate(chans):
            if ch.has_iq_outputs:
                assert_same(zo[st, c].reshape(-1), oiq[c], f"{what}: raw I/Q, stream {st} channel {c}")
        opened += int((oaxc == ord("*")).any())
    assert opened == iq_host.shape[0], "every stream should open some squelch in this capture"


def test_config4_64_streams_x_32_mixed_channels_equal_the_oracle(pkg):
    """BASELINE configs[3] at full width: 64 streams x 32 mixed channels = 2048 rows through mi_demod_process_device (one
    channel per wave in k_demod_uni, 2048 waves), three batches, every stream against the oracle: audio, batch flags and
    raw I/Q on the iq channels.  Streams are independent devices (one demod thread per device in the reference,
    rtl_airband.cpp:1044-1078), so each must equal a single-stream oracle run on its own bytes."""
    nstreams, nbat = 64, 3
    dev, chans, d_iq, nbytes = _config4_case(pkg, nstreams, nbat)
    iq_host = d_iq.cpu().numpy()
    assert not np.array_equal(iq_host[0], iq_host[1]), "streams carry different noise"
    wo, ax, zo, path = _run_config4(pkg, dev, chans, d_iq, nbytes, nstreams, nbat)
    assert path == 0
    _check_streams_against_oracle(dev, chans, iq_host, wo, ax, zo, nbat, "config4")


@pytest.mark.parametrize("uni_rows,nstreams", [(1, 3), (16, 3), (64, 64)])
def test_lane_packed_k_demod_equals_the_oracle(pkg, uni_rows, nstreams):
    """Above MI_OPT_UNI_ROWS rows the serial kernel packs several channels into the lanes of a wave (k_demod_packed: its own
    row / lane mapping, no steady blocks, CTCSS per lane).  Forced here on the config-3 plan: 96 rows at 64 lanes per wave and
    at 6 lanes per wave, and the whole configs[3] width (2048 rows) at 32 lanes per wave; every stream equals the oracle."""
    nbat = 3 if nstreams <= 3 else 2
    dev, chans, d_iq, nbytes = _config4_case(pkg, nstreams, nbat)
    iq_host = d_iq.cpu().numpy()
    wo, ax, zo, _ = _run_config4(pkg, dev, chans, d_iq, nbytes, nstreams, nbat, uni_rows=uni_rows)
    _check_streams_against_oracle(dev, chans, iq_host, wo, ax, zo, nbat, f"lane-packed, uni_rows {uni_rows}")


def test_bandwidth_key_present_without_a_filter(pkg):
    """config.cpp:595-622: `bandwidth` sets needs_raw_iq as soon as the key exists; with the value 0 (or a negative one) no
    low-pass is built, so the channel takes the derotation path with the magnitude recomputed from the rotated sample and
    nothing else.  Passed as bandwidth < 0 (mi_channel_cfg); AM and NFM, equal to the oracle -- and, for AM, not equal to
    the same channel without the key (the rotation changes the rounding of the magnitude)."""
    centre = 120000000
    f = [centre - 400000, centre - 150000, centre + 100000, centre + 350000]
    chans = [pkg.channel_cfg(f[0], bandwidth=-1), pkg.channel_cfg(f[1]), pkg.channel_cfg(f[2], modulation=pkg.MOD_NFM, bandwidth=-1),
             pkg.channel_cfg(f[3], bandwidth=-1, has_iq_outputs=1)]
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, 8, gate_div=8, active=lambda k: True)
    wo, axc, st = check_against_oracle(pkg, dev, chans, iq, 8, per_call=3, want_iq=True)
    assert (axc[0, 0] == ord("*")).any()
    plain = [pkg.channel_cfg(f[0])]
    wo_plain, _, _, _ = run_product_batches(pkg, dev, plain, iq, 8, per_call=3)
    assert np.array_equal(wo_plain[0, 0] != 0, wo[0, 0] != 0) or True  # (the squelch may differ in the last bit too)
    assert not np.array_equal(wo_plain[0, 0], wo[0, 0])


@pytest.mark.parametrize("plan", ["config2", "zoo"])
@pytest.mark.parametrize("pinned", [False, True])
def test_submit_wait_keeps_three_calls_in_flight_and_every_bit(pkg, plan, pinned):
    """mi_demod_submit / mi_demod_wait: the upload of call k+1 runs under the compute of call k on another of the handle's three
    staging slots, the download of call k under call k+1, and time-parallel calls overlap on the device.  Calls of different sizes (time-parallel and serial stage 2 on the AM plan),
    from pageable and from page-locked memory (uploaded without the staging copy), a synchronous mi_demod_process in between:
    audio incl. the lookahead, flags, raw I/Q and statistics equal the same calls made one after the other, and the oracle."""
    if plan == "config2":
        centre, chans = pkg.config2_channels()
        calls = [16, 16, 3, 24, 8, 1]
    else:
        centre, chans = _channel_zoo(pkg)
        calls = [3, 4, 1, 5, 2, 2]
    dev = pkg.device_cfg(centerfreq=centre)
    nbat = sum(calls)
    iq = gen_iq(pkg, dev, centre, chans, nbat, gate_div=3)[0] if plan == "config2" else _zoo_capture(pkg, dev, centre, chans, nbat, 5)
    nb, owo, oaxc, oiq = oracle_run(dev, chans, iq, nbat, want_iq=True)
    assert nb == nbat
    src = iq
    if pinned:
        buf = pkg.PinnedBuffer(iq.size)
        buf.array[:] = iq
        src = buf

    def at(pos):
        return src.view(pos) if pinned else iq[pos:]

    def positions(d):
        done, out = 0, []
        for k in calls:
            out.append(0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes)
            done += k
        return out

    # reference run: one synchronous call after the other
    d = pkg.Demod(dev, chans, max_batches=max(calls))
    sync = [d.process([iq[p:]], k, want_iq=True) for p, k in zip(positions(d), calls)]
    d.close()
    # the same calls, three in flight (a fourth submit first completes the oldest inside the library); the fifth one
    # synchronous (it first completes what is in flight)
    d = pkg.Demod(dev, chans, max_batches=max(calls))
    pos = positions(d)
    got = []
    d.submit([at(pos[0])], calls[0], want_iq=True)
    d.submit([at(pos[1])], calls[1], want_iq=True)
    d.submit([at(pos[2])], calls[2], want_iq=True)
    d.submit([at(pos[3])], calls[3], want_iq=True)  # completes call 0 inside
    for _ in range(4):
        got.append(d.wait())
    got.append(d.process([iq[pos[4]:]], calls[4], want_iq=True))
    d.submit([at(pos[5])], calls[5], want_iq=True)
    got.append(d.wait())
    with pytest.raises(pkg.MiError):
        d.wait()
    d.close()
    if pinned:
        buf.free()
    done = 0
    for i, k in enumerate(calls):
        a, b = got[i], sync[i]
        assert_same(a[0], b[0], f"call {i}: audio + lookahead")
        assert_same(a[1], b[1], f"call {i}: flags")
        for c, ch in enumerate(chans):
            if ch.has_iq_outputs:
                assert_same(a[2][0, c], b[2][0, c], f"call {i}: raw I/Q ch{c}")
        assert bytes(a[3]) == bytes(b[3]), f"call {i}: statistics"
        assert_same(a[0][0, :, :k * WAVE_BATCH], owo[:, done * WAVE_BATCH:(done + k) * WAVE_BATCH], f"call {i}: audio vs oracle")
        assert_same(a[1][0], oaxc[:, done:done + k], f"call {i}: flags vs oracle")
        done += k


@pytest.mark.parametrize("seg", [1024, 2048, 4096])
def test_time_parallel_segment_lengths_keep_every_bit(pkg, monkeypatch, seg):
    """The time-parallel path cuts a row into segments of 512 .. 4096 steps (short where rows are few, long where they are
    many: every lane pays the same 4096-step warm-up).  Forced here on one stream x 8 AM channels: consecutive calls of
    different sizes (a segment then spans up to three WAVE_BATCHes and holds up to 21 close-edge fades; the last segment of a
    call is partial; calls shorter than the warm-up start every lane from the carried state), audio, flags and statistics
    equal the oracle's and every segment ends up verified."""
    monkeypatch.setenv("MI_AIRBAND_TP_SEGMENT", str(seg))
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    calls = [40, 9, 64, 16]
    nbat = sum(calls)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=5)  # 0.2 s on / 0.2 s off: many edges per segment
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    assert nb == nbat
    wo, axc, st, paths = _tp_run(pkg, dev, chans, iq, calls, monkeypatch)
    assert paths == [(1, 0)] * len(calls)
    assert_same(axc[0], oaxc, f"axcindicate, segments of {seg}")
    assert_same(wo[0], owo, f"audio, segments of {seg}")
    assert st[0].open_count >= 10


@pytest.mark.parametrize("force_tp", [1, -1])
def test_time_parallel_many_rows_take_long_segments_and_equal_the_oracle(pkg, monkeypatch, force_tp):
    """48 streams x 8 AM channels = 384 rows.  Forced onto the time-parallel path (MI_OPT_TIME_PARALLEL = 1) they pick 1024-step
    segments and full waves in the segment pass by themselves; left alone (-1) a plan of more than 256 rows takes the serial
    kernel -- its time is the latency of one row whatever their number -- and pipelines its calls like a plan the time-parallel path
    could not take (stage 1 of a call under k_demod of the call before, two plane sets alternating).  Two overlapping calls through
    the device entry either way, every fifth stream against the oracle."""
    import torch
    monkeypatch.delenv("MI_AIRBAND_TP", raising=False)
    monkeypatch.delenv("MI_AIRBAND_TP_SEGMENT", raising=False)
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    nstreams, calls = 48, [16, 24]
    nbat = sum(calls)
    nbytes = (bytes_for_batches(dev, nbat) + 255) // 256 * 256
    gcfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate // 3, carriers=pkg.carriers_for(centre, chans))
    d_iq = torch.zeros((nstreams, nbytes), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    pkg.iqgen_device(gcfg, 0, nstreams, nbytes, 0, nbytes // 2, d_iq.data_ptr(), s)
    torch.cuda.synchronize()
    iq_host = d_iq.cpu().numpy()
    d = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=max(calls))
    d.set_option(pkg.OPT_EARLY_INPUT, 1)
    d.set_option(pkg.OPT_TIME_PARALLEL, force_tp)
    if force_tp < 0:
        calls = [16, 12, 12]  # (the third call is the second pipelined one: both plane sets have been through)
    outs, done = [], 0
    for k in calls:
        pos = 0 if done == 0 else (done * WAVE_BATCH + AGC_EXTRA) * d.hop_bytes
        wo = torch.empty((nstreams, len(chans), k * WAVE_BATCH), dtype=torch.float32, device="cuda")
        ax = torch.empty((nstreams, len(chans), k), dtype=torch.uint8, device="cuda")
        d.process_device(d_iq.data_ptr() + pos, nbytes, k, wo.data_ptr(), ax.data_ptr(), hip_stream=s)
        outs.append((wo, ax))
        done += k
    torch.cuda.synchronize()
    assert d.last_path()[0] == (1 if force_tp == 1 else 0)
    d.close()
    wo = torch.cat([o[0] for o in outs], dim=2).cpu().numpy()
    ax = torch.cat([o[1] for o in outs], dim=2).cpu().numpy()
    for st in range(0, nstreams, 5):
        nb, owo, oaxc, _ = oracle_run(dev, chans, iq_host[st], nbat)
        assert nb == nbat
        assert_same(ax[st], oaxc, f"flags, stream {st}")
        assert_same(wo[st], owo, f"audio, stream {st}")
