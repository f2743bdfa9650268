"""End-to-end regression fixtures for the oracle's whole path (tests/golden/oracle_e2e.json).

PARITY UNPINNED BY THE REFERENCE: these hashes were produced by the oracle itself (run
`python tests/test_oracle_e2e.py --regen` to rewrite them) because demodulate() cannot be built here (it
needs fftw3.h / lame / shout / libconfig++ / the generated config.h) and the reference's tests hold no
vectors for it.  They guard the oracle against accidental change; the reference-pinned checks are
test_oracle_components.py (Squelch/CTCSS/filters, bit-exact) and test_oracle_stage1.py (DFT definition)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from common import gen_iq, oracle_run  # noqa: E402
from conftest import load_package  # noqa: E402

PATH = os.path.join(HERE, "golden", "oracle_e2e.json")


def cases(pkg):
    centre, c2 = pkg.config2_channels()
    _, c3 = pkg.config3_channels()
    c3[3].has_iq_outputs = 1
    opts = [pkg.channel_cfg(centre + 250000, squelch_threshold_dbfs=-40), pkg.channel_cfg(centre - 250000, squelch_snr_db=0.0, ampfactor=2.5),
            pkg.channel_cfg(centre + 500000, modulation=pkg.MOD_NFM, tau=0, notch=1000.0, notch_q=5.0),
            pkg.channel_cfg(centre - 500000, bandwidth=8000, has_iq_outputs=1),
            pkg.channel_cfg(centre + 750000, modulation=pkg.MOD_NFM, ctcss=100.0, bandwidth=12500)]
    return {
        "config1_8am_fft512": (pkg.device_cfg(centerfreq=centre), c2, 8, dict(gate_div=4)),
        "config3_32ch_fft2048": (pkg.device_cfg(centerfreq=centre, fft_size_log=11), c3, 6, dict(gate_div=2, active=lambda k: k % 4 != 2, amp_q8=1024)),
        "options_fft512": (pkg.device_cfg(centerfreq=centre, tau=75), opts, 8, dict(gate_div=2, active=lambda k: True)),
        "options_quadri": (pkg.device_cfg(centerfreq=centre, fm_quadri=1), opts, 6, dict(gate_div=2, active=lambda k: True)),
    }


def compute(pkg, dev, chans, nbat, kw):
    iq, _ = gen_iq(pkg, dev, dev.centerfreq, chans, nbat, **kw)
    nb, wo, axc, iqo = oracle_run(dev, chans, iq, nbat, want_iq=True)
    assert nb == nbat
    return {"iq_sha256": hashlib.sha256(iq.tobytes()).hexdigest(), "audio_sha256": hashlib.sha256(wo.tobytes()).hexdigest(),
            "iq_out_sha256": hashlib.sha256(iqo.tobytes()).hexdigest(), "axc": [bytes(r).decode() for r in axc],
            "open_samples": [int((r != 0).sum()) for r in wo]}


def test_oracle_end_to_end_fixtures():
    pkg = load_package()
    gold = json.load(open(PATH))
    for name, (dev, chans, nbat, kw) in cases(pkg).items():
        got = compute(pkg, dev, chans, nbat, kw)
        assert got == gold[name], f"{name}: oracle output changed"
        assert any("*" in row for row in got["axc"]), f"{name}: fixture never opens a squelch"


if __name__ == "__main__":
    if "--regen" in sys.argv:
        pkg = load_package()
        out = {name: compute(pkg, *c) for name, c in cases(pkg).items()}
        json.dump(out, open(PATH, "w"), indent=1)
        print("wrote", PATH)
