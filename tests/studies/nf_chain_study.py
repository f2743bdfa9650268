#!/usr/bin/env python3
"""nf_chain_study.py -- TEST INFRASTRUCTURE: measurements behind the time split of the squelch core chain.

    python tests/studies/nf_chain_study.py [--seconds 64] [--noise-only]

Uses the oracle's stage 1 (oracle/, CPU) for the magnitude planes of the bench signal (BASELINE configs[1]: 8 AM channels,
fft 512, carriers gated 1 s on / 1 s off on the even channels) and tests/studies/nf_chain.c for the chain itself.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import libs  # noqa: E402
from common import bytes_for_batches  # noqa: E402
from conftest import load_package  # noqa: E402


def helper():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libnf_chain.so")
    src = os.path.join(HERE, "nf_chain.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fno-fast-math", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src], check=True)
    return C.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=64.0)
    ap.add_argument("--noise-only", action="store_true")
    args = ap.parse_args()
    pkg = load_package()
    lib = helper()
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=9)
    nbat = int(args.seconds * 8)
    n = bytes_for_batches(dev, nbat) // 2
    gcfg = pkg.iqgen_cfg(sample_rate=dev.sample_rate, gate_samples=dev.sample_rate,
                         carriers=() if args.noise_only else pkg.carriers_for(centre, chans))
    iq = pkg.iqgen_host(gcfg, 0, 0, n)
    from common import to_oracle_cfg
    odev, ochans = to_oracle_cfg(dev, chans)
    od = libs.OracleDemod(odev, ochans)
    nfft = nbat * 2000 + 100
    mag, _ = od.stage1(iq, nfft, want_iq=False)
    od.close()
    nsamp = (nfft // 16) * 16
    nblk = nsamp // 16
    ratio = np.float32(10.0 ** (9.54 / 20.0))  # squelch.cpp:100 default snr threshold
    cap_factor = np.float32(np.float32(1.5) * ratio)
    f32p = C.POINTER(C.c_float)
    print(f"signal: {args.seconds:g} s, {'noise only' if args.noise_only else 'bench carriers, 1 s on / 1 s off on even channels'}; {nblk} blocks per channel")
    for ch in range(len(chans)):
        x = np.ascontiguousarray(mag[ch, :nsamp])
        st = np.zeros((nblk, 4), np.float32)
        lib.chain_exact(x.ctypes.data_as(f32p), C.c_size_t(nsamp), C.c_float(cap_factor), C.c_float(5.0), C.c_float(0.001), C.c_float(0.001),
                        st.ctypes.data_as(f32p))
        below = float(np.mean(st[:, 1] < st[:, 0]))
        line = [f"ch{ch}: nf end {st[-1, 0]:.5f}  blocks with capped < nf: {100 * below:.2f} %"]
        # 1. meeting of speculative trajectories
        for W in (1024, 2048, 4096, 16384):
            for scale, add, tag in ((2.0, 0.0, "hi=2x"), (1.001, 0.0, "hi=+0.1%"), (1.0, 2e-8, "hi=+1ulp"), (0.999, 0.0, "lo=-0.1%")):
                seg = 4096
                met = np.full(nblk // seg + 1, -2, np.int32)
                lib.meet_blocks(x.ctypes.data_as(f32p), C.c_size_t(nsamp), C.c_float(cap_factor), st.ctypes.data_as(f32p), C.c_size_t(seg),
                                C.c_size_t(W), C.c_float(scale), C.c_float(add), met.ctypes.data_as(C.POINTER(C.c_int32)))
                m = met[1:(nblk - 1) // seg + 1]
                m = m[m != -2]
                ok_w = int(np.sum((m >= 0) & (m <= W)))
                line.append(f"  W={W:5d} blocks ({W * 16:6d} samples) {tag:9s}: met before the segment start {ok_w}/{len(m)}, within W+4096: {int(np.sum(m >= 0))}/{len(m)}")
        # 2. rounds of the guess-and-verify noise-floor walk
        for fb in (0, 2):
            hist = np.zeros(65, np.uint64)
            ev = np.zeros(3, np.uint64)
            cb = C.c_uint64(0)
            rounds = lib.guess_rounds
            rounds.restype = C.c_uint64
            r = rounds(st.ctypes.data_as(f32p), C.c_size_t(nblk), hist.ctypes.data_as(C.POINTER(C.c_uint64)),
                       ev.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int(fb), C.byref(cb))
            ng = (nblk + 63) // 64
            line.append(f"  guess/verify walk{' (classical pass after 2 short rounds)' if fb else ''}: {r / ng:.2f} rounds per group of 64 blocks; "
                        f"ended by non-self step {ev[0]}, self step off the guess {ev[1]}, group end {ev[2]}; classical blocks {cb.value}; "
                        f"groups by rounds 1..8+: {[int(h) for h in hist[1:8]]} + {int(hist[8:].sum())}")
        # 3. the walk as built: guesses from the last three settled increments
        hist = np.zeros(33, np.uint64)
        ah = np.zeros(65, np.uint64)
        lib.p3_rounds.restype = C.c_uint64
        r = lib.p3_rounds(st.ctypes.data_as(f32p), C.c_size_t(nblk), hist.ctypes.data_as(C.POINTER(C.c_uint64)), ah.ctypes.data_as(C.POINTER(C.c_uint64)))
        line.append(f"  as built (period-3 history): {r / (nblk // 64):.2f} rounds per group; groups by rounds 1..8+: {[int(h) for h in hist[1:8]]} + {int(hist[8:].sum())}; "
                    f"rounds that settled fewer than 4 blocks: {int(ah[1:4].sum())}")
        print("\n".join(line))


if __name__ == "__main__":
    main()
