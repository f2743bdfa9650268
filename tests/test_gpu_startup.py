"""Start-up behaviour of a handle on the GPU: everything slow happens in mi_demod_create / mi_demod_prepare, never in the first
processing call (the reference's ring holds 0.5 s of u8 IQ: config.cpp:799-805, overflow rule input-helpers.cpp:56-61), and the
compiled stage-1 kernel of a plan is kept on disk between process starts (include/mi_airband.h: mi_set_cache_dir)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from common import WAVE_BATCH, assert_same, gen_iq, oracle_run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, time
sys.path.insert(0, %(tests)r)
from conftest import load_package
from common import gen_iq
pkg = load_package()
pkg.set_cache_dir(sys.argv[1])
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre)
iq, _ = gen_iq(pkg, dev, centre, chans, 2, gate_div=8)
t0 = time.perf_counter()
d = pkg.Demod(dev, chans, nstreams=1, max_batches=2, gpu=0)
t_create = time.perf_counter() - t0
t0 = time.perf_counter()
d.prepare(1)
t_prepare = time.perf_counter() - t0
t0 = time.perf_counter()
wo, axc, _, _ = d.process([iq], 1)
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
wo2, axc2, _, _ = d.process([iq[d.bytes_consumed(1):]], 1)
t_second = time.perf_counter() - t0
stage1 = d.last_stage1()
d.close()
print(json.dumps(dict(create=t_create, prepare=t_prepare, first=t_first, second=t_second, counts=pkg.jit_counts(), stage1=stage1,
                      sum=float(abs(wo).sum()))))
"""


def _child(cache_dir):
    code = CHILD % {"tests": os.path.join(ROOT, "tests")}
    r = subprocess.run([sys.executable, "-c", code, str(cache_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_kernel_is_compiled_at_create_and_cached_on_disk(pkg, tmp_path):
    cache = tmp_path / "co"
    cache.mkdir()
    a = _child(cache)
    files = sorted(os.listdir(cache))
    assert a["counts"] == [1, 0], a                      # compiled once, by create
    assert a["stage1"] == 3, a                           # the plan-compiled lane kernel ran
    assert len(files) == 1 and files[0].startswith("l64_") and files[0].endswith(".co"), files
    # nothing slow is left for the first processing call: one WAVE_BATCH through the host entry, staging included
    assert a["first"] < 0.050, a
    b = _child(cache)
    assert b["counts"] == [0, 1], b                      # second start: loaded, not compiled
    assert b["stage1"] == 3 and b["sum"] == a["sum"], (a, b)
    assert b["create"] < a["create"], (a, b)
    assert sorted(os.listdir(cache)) == files
    # a damaged file is detected and replaced
    path = os.path.join(cache, files[0])
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    open(path, "wb").write(bytes(blob))
    c = _child(cache)
    assert c["counts"] == [1, 0] and c["sum"] == a["sum"], c
    # the cache switched off: compiled, nothing written
    code_off = _child("")
    assert code_off["counts"] == [1, 0] and code_off["sum"] == a["sum"], code_off


@pytest.mark.gpu
def test_prepared_handle_first_call_equals_oracle_and_is_fast(pkg):
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, 3, gate_div=8)
    d = pkg.Demod(dev, chans, nstreams=1, max_batches=1, gpu=0)
    d.prepare(1)
    t0 = time.perf_counter()
    wo, axc, _, _ = d.process([iq], 1)
    dt = time.perf_counter() - t0
    assert dt < 0.005, f"first call of a prepared handle took {dt * 1e3:.2f} ms"
    nb, owo, oaxc, _ = oracle_run(dev, chans, iq, 1)
    assert nb == 1
    assert_same(wo[0, :, :WAVE_BATCH], owo, "audio of the first call")
    assert np.array_equal(axc[0], oaxc)
    d.close()


ENV_CHILD = r"""
import sys
sys.path.insert(0, %(tests)r)
from conftest import load_package
pkg = load_package()
centre, chans = pkg.config2_channels()
d = pkg.Demod(pkg.device_cfg(centerfreq=centre), chans, nstreams=1, max_batches=2, gpu=0)
d.close()
"""


@pytest.mark.gpu
def test_tuning_variables_in_the_environment_are_named_under_debug(tmp_path):
    """MI_AIRBAND_DEBUG=1: a handle names every MI_AIRBAND_* tuning variable it found in the environment on stderr when it is created
    (they are read per handle and change what it does without another word); without MI_AIRBAND_DEBUG it stays silent."""
    code = ENV_CHILD % {"tests": os.path.join(ROOT, "tests")}
    for debug in ("1", None):
        env = dict(os.environ, MI_AIRBAND_TP="0", MI_AIRBAND_CORE_GUESS="2")
        env.pop("MI_AIRBAND_DEBUG", None)
        if debug:
            env["MI_AIRBAND_DEBUG"] = debug
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        named = [l for l in r.stderr.splitlines() if l.startswith("mi_airband: MI_AIRBAND_")]
        if debug:
            assert any("MI_AIRBAND_TP=0" in l for l in named) and any("MI_AIRBAND_CORE_GUESS=2" in l for l in named), r.stderr[-2000:]
        else:
            assert not named, named
