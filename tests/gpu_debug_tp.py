"""Ad-hoc GPU diagnostics for the time-parallel path (not a test)."""
import ctypes as C
import os
import sys
import time

import numpy as np

import libs
from common import AGC_EXTRA, WAVE_BATCH, gen_iq, oracle_run, to_oracle_cfg
from conftest import load_package


def oracle_core_trace(raw, stride, manual_level=0.0, snr_db=None):
    lib = libs.oracle_lib()
    lib.ao_squelch_core_trace.argtypes = [C.POINTER(libs.SquelchCfg), libs.f32p, C.c_size_t, C.c_size_t, libs.f32p]
    lib.ao_squelch_core_trace.restype = None
    n = raw.size
    nk = (n + stride - 1) // stride + 1
    out = np.zeros((nk, 4), np.float32)
    cfg = libs.SquelchCfg(manual_level, 0 if snr_db is None else 1, 0.0 if snr_db is None else snr_db, 0.0, 16000.0)
    lib.ao_squelch_core_trace(C.byref(cfg), np.ascontiguousarray(raw, np.float32), n, stride, out.reshape(-1))
    return out


def main():
    os.environ["MI_AIRBAND_TP"] = "1"
    pkg = load_package()
    nbat = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=1)
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nfft = nbat * WAVE_BATCH + AGC_EXTRA
    omag, _ = od.stage1(iq, nfft, want_iq=False)
    d = pkg.Demod(dev, chans, max_batches=nbat)
    t = time.time()
    wo, axc, _, st = d.process([iq], nbat)
    print("process s:", time.time() - t, "kernel ms:", d.last_kernel_ms(), "path:", d.last_path())
    for c in range(len(chans)):
        core, diag = d.tp_debug(c)
        ref = oracle_core_trace(omag[c, AGC_EXTRA:], 512)
        bad = np.nonzero((core != ref).any(axis=1))[0]
        print(f"ch{c}: diag={diag.tolist()} nseg={core.shape[0] - 1} core mismatches={bad.size}", (bad[:5].tolist(), core[bad[0]].tolist(), ref[bad[0]].tolist()) if bad.size else "")
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, nbat)
    for c in range(len(chans)):
        a, b = wo[0, c, :nbat * WAVE_BATCH], owo[c]
        print(f"ch{c} audio ndiff={int((a != b).sum())} axc_equal={np.array_equal(axc[0, c], oaxc[c])}")


if __name__ == "__main__":
    main()
