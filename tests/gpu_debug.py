"""Ad-hoc GPU diagnostics (not a test): stage-1 planes and full path vs the oracle, with diff statistics."""
import sys
import time

import numpy as np

import libs
from common import AGC_EXTRA, WAVE_BATCH, gen_iq, oracle_run, to_oracle_cfg
from conftest import load_package


def main():
    pkg = load_package()
    print("devices:", pkg.device_count())
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    centre, chans = pkg.config2_channels()
    chans[2] = pkg.channel_cfg(chans[2].freq, modulation=pkg.MOD_NFM)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=log2n)
    nbat = 4
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat)
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nfft = nbat * WAVE_BATCH + AGC_EXTRA
    omag, oiq = od.stage1(iq, nfft)
    d = pkg.Demod(dev, chans, max_batches=nbat)
    t = time.time()
    wo, axc, _, st = d.process([iq], nbat)
    print("process s:", time.time() - t, "kernel ms:", d.last_kernel_ms())
    # after the call the plane holds [carry(100) | ...]; indices AGC_EXTRA.. still hold this call's magnitudes
    for c in range(len(chans)):
        mag, z = d.read_planes(0, c, AGC_EXTRA, nfft - AGC_EXTRA, want_iq=True)
        ref = omag[c, AGC_EXTRA:]
        nd = int((mag != ref).sum())
        print(f"ch{c} stage1 mag: ndiff={nd} maxabs={np.abs(mag - ref).max():.3e} ref_rms={np.sqrt((ref**2).mean()):.4f}")
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    for c in range(len(chans)):
        a, b = wo[0, c, :nbat * WAVE_BATCH], owo[c]
        print(f"ch{c} audio: ndiff={int((a != b).sum())} maxabs={np.abs(a - b).max():.3e} mask_diff={int(((a != 0) != (b != 0)).sum())} "
              f"axc={bytes(axc[0, c]).decode()!r} oracle={bytes(oaxc[c]).decode()!r} open_count={st[c].open_count}")


if __name__ == "__main__":
    main()
