"""AFC (rtl_airband.cpp:180-251) in the oracle.

The reference's AFC class lives in rtl_airband.cpp, which cannot be built here, and its tests hold no
vectors for it: parity unpinned by the reference.  The restatement is pinned by hand-evaluated walks of AFC::check and by
the behaviour the class exists for: an off-centre carrier pulls the picked bin to the spectral peak while the squelch is
open, flags the batch AFC_UP / AFC_DOWN, and the bin returns to its base when the channel goes quiet."""
import numpy as np

import libs
from common import WAVE_BATCH, bytes_for_batches, to_oracle_cfg
from conftest import load_package


def spectrum(sq):
    """Interleaved re,im whose squared magnitudes are `sq` (im = 0, exact squares for small integers)."""
    out = np.zeros(2 * len(sq), np.float32)
    out[0::2] = np.sqrt(np.asarray(sq, np.float64)).astype(np.float32)
    return out


def check(sq, step, base, afc):
    lib = libs.oracle_lib()
    s = spectrum(sq)
    base_value = np.float32(s[2 * base]) * np.float32(s[2 * base])
    return lib.ao_afc_check(s, len(sq), step, base, float(base_value), afc)


def test_check_walks_while_the_spectrum_stays_above_base():
    #        0  1  2   3   4    5   6   7
    sq = [1, 4, 9, 16, 64, 100, 81, 4]
    assert check(sq, +1, 3, 1) == 6    # 64, 100, 81 all clear base + threshold (48, 52.8, 58.08); 4 <= 16 stops the walk
    assert check(sq, -1, 3, 1) == 3    # 9 <= 16: no move downwards
    assert check(sq, +1, 5, 1) == 5    # already on the peak
    assert check(sq, -1, 6, 1) == 5    # from the far side


def test_check_threshold_rule():
    # base 16; first step sets threshold = (value - 16) / afc, every further step must gain at least the threshold
    # over the BASE value, and the threshold grows by 10 % per accepted step (rtl_airband.cpp:209-216)
    sq = [16, 116, 120, 400, 0, 0, 0, 0]
    assert check(sq, +1, 0, 1) == 3    # threshold 100: 120-16=104 >= 100 ok (-> 110), 400-16 >= 110 ok, then 0 <= 16
    sq = [16, 116, 110, 400, 0, 0, 0, 0]
    assert check(sq, +1, 0, 1) == 1    # 110-16 = 94 < 100: stop after the first step
    assert check(sq, +1, 0, 2) == 3    # afc = 2 halves the threshold: 94 >= 50


def test_check_edges():
    sq = [25, 16, 9, 4, 9, 16, 25, 36]
    assert check(sq, -1, 1, 1) == 0    # walks down to bin 0 and stops at the edge
    assert check(sq, +1, 6, 1) == 7    # walks up to fft_size-1 and stops at the edge
    assert check(sq, -1, 0, 1) == 0
    assert check(sq, +1, 7, 1) == 7


def run_offcentre(pkg, deltas, afcs, nbat=16, gate_div=2):
    centre = 120_000_000
    dev = pkg.device_cfg(centerfreq=centre)
    freqs = [centre - 800_000 + 400_000 * k for k in range(len(deltas))]
    chans = [pkg.channel_cfg(f, afc=a) for f, a in zip(freqs, afcs)]
    carriers = [(f - centre + d, 0, 3072, 0) for f, d in zip(freqs, deltas)]
    cfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate // gate_div, carriers=carriers)
    iq = pkg.iqgen_host(cfg, 0, 0, bytes_for_batches(dev, nbat) // 2)
    return dev, chans, iq


def test_offcentre_carrier_pulls_the_bin():
    pkg = load_package()
    # bin width 5 kHz at fft 512: carriers 10 kHz above / 10 kHz below / 10 kHz above with afc off / on frequency
    dev, chans, iq = run_offcentre(pkg, [10_000, -10_000, 10_000, 0], [1, 1, 0, 1])
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    seen = [set() for _ in chans]
    moved = [set() for _ in chans]
    hop = 2 * 160
    pos = 0
    for b in range(16):  # batch by batch, looking at the bin table in between
        nb, wo, axc, _ = od.run(iq[pos:], 1)
        assert nb == 1
        pos += (WAVE_BATCH + (100 if b == 0 else 0)) * hop
        cur, base = od.bins()
        for c in range(len(chans)):
            seen[c].add(chr(axc[c, 0]))
            moved[c].add(int(cur[c] - base[c]))
    od.close()
    # the walk runs past the peak while the bins stay `threshold` above the base bin (the window's main lobe is wide)
    up, down = moved[0] - {0}, moved[1] - {0}
    assert 0 in moved[0] and up and min(up) >= 1 and "<" in seen[0]        # AFC_UP, back to base when the gate closes
    assert 0 in moved[1] and down and max(down) <= -1 and ">" in seen[1]   # AFC_DOWN
    assert moved[2] == {0} and seen[2] <= {" ", "*"}   # afc = 0: never moves
    assert all(abs(m) <= 1 for m in moved[3])          # on frequency: at most a noise-sized step


def test_afc_raises_the_level_the_squelch_sees():
    """Same off-centre carrier on two channels: the one with AFC is picked on the peak once it has opened."""
    pkg = load_package()
    dev, chans, iq = run_offcentre(pkg, [10_000, 10_000], [1, 0], nbat=12, gate_div=2)
    # both channels at the same frequency offset need the same carrier: reuse channel 0's frequency for both
    chans = [pkg.channel_cfg(chans[0].freq, afc=1), pkg.channel_cfg(chans[0].freq, afc=0)]
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nb, wo, axc, _ = od.run(iq, 12)
    cur, base = od.bins()
    od.close()
    assert nb == 12
    assert "<" in {chr(x) for x in axc[0]} and set(axc[1].tolist()) <= {ord(" "), ord("*")}
    on = np.nonzero(axc[1] == ord("*"))[0]
    assert on.size >= 3
    b = int(on[2])  # a batch well inside the transmission
    seg = slice(b * WAVE_BATCH, (b + 1) * WAVE_BATCH)
    assert not np.array_equal(wo[0, seg], wo[1, seg])  # a different bin feeds the AM detector
