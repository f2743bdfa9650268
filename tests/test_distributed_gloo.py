"""N>1 host path on CPU: world_size 2, gloo.  Streams shard stream-major across ranks, each rank produces the
audio of its own streams (here with the CPU oracle standing in for the GPU data path -- this is a test of the
sharding and gather plumbing, not of the kernels), rank 0 gathers, and the result must equal a single-process
run over all streams."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
NSTREAMS, NBAT = 5, 3  # uneven over 2 ranks on purpose


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _audio_for_stream(pkg, s, with_flags=False):
    from common import gen_iq, oracle_run
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, NBAT, stream=s, gate_div=4 + s)
    nb, wo, axc, _ = oracle_run(dev, chans, iq, NBAT)
    assert nb == NBAT
    return (wo, axc) if with_flags else wo


def _worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    from conftest import load_package
    pkg = load_package()
    from boondock_airband_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.stream_range(rank, world, NSTREAMS)
    mine = [_audio_for_stream(pkg, s, with_flags=True) for s in range(lo, hi)]
    local = torch.from_numpy(np.stack([m[0] for m in mine]))
    flags = torch.from_numpy(np.stack([m[1] for m in mine]))
    got = shard.gather_audio(local, dst=0)
    if rank == 0:
        full = torch.cat(got, dim=0).numpy()
        np.save(os.path.join(outdir, "gathered.npy"), full)
    else:
        assert got is None
    # the form bench.py runs: one AudioGather reused step after step, flags included, two gathers in flight, then open-only
    g = shard.AudioGather(tuple(local.shape), NBAT, local.device, dst=0)
    h1 = g.start(local, flags)
    h2 = g.start(local * 2, flags)
    r2, r1 = h2.wait(), h1.wait()
    h3 = g.start(local, flags, open_only=True)
    r3 = h3.wait()
    if rank == 0:
        np.save(os.path.join(outdir, "g1.npy"), torch.cat(r1[0], dim=0).numpy())
        np.save(os.path.join(outdir, "g2.npy"), torch.cat(r2[0], dim=0).numpy())
        np.save(os.path.join(outdir, "f1.npy"), torch.cat(r1[1], dim=0).numpy())
        np.save(os.path.join(outdir, "g3.npy"), torch.cat(r3[0], dim=0).numpy())
        np.save(os.path.join(outdir, "f3.npy"), torch.cat(r3[1], dim=0).numpy())
    else:
        assert r1 is None and r2 is None and r3 is None
    dist.barrier()
    dist.destroy_process_group()


def test_stream_range_partitions_exactly(pkg):
    from boondock_airband_amd import shard
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 64, 512, 513):
            spans = [shard.stream_range(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard.stream_range(3, 8, 512) == (192, 256)  # BASELINE configs[4]: 64 streams per GPU
    with pytest.raises(ValueError):
        shard.stream_range(2, 2, 4)


def test_gather_to_rank0_world2(pkg, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    want = np.stack([_audio_for_stream(pkg, s) for s in range(NSTREAMS)])
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    flags = np.stack([_audio_for_stream(pkg, s, with_flags=True)[1] for s in range(NSTREAMS)])
    assert np.array_equal(np.load(tmp_path / "g1.npy"), want)
    assert np.array_equal(np.load(tmp_path / "g2.npy"), want * 2)
    assert np.array_equal(np.load(tmp_path / "f1.npy"), flags) and np.array_equal(np.load(tmp_path / "f3.npy"), flags)
    # open-only: batches whose flag is NO_SIGNAL arrive as zeros (the reference's non-continuous outputs skip them,
    # output.cpp:518,568), every other batch bit for bit
    open_blocks = np.repeat(flags != ord(" "), 2000, axis=2)
    assert open_blocks.any() and not open_blocks.all()
    assert np.array_equal(np.load(tmp_path / "g3.npy"), np.where(open_blocks, want, 0.0).astype(np.float32))
