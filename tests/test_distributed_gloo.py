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


def _audio_for_stream(pkg, s):
    from common import gen_iq, oracle_run
    centre, chans = pkg.config2_channels()
    dev = pkg.device_cfg(centerfreq=centre)
    iq, _ = gen_iq(pkg, dev, centre, chans, NBAT, stream=s, gate_div=4 + s)
    nb, wo, axc, _ = oracle_run(dev, chans, iq, NBAT)
    assert nb == NBAT
    return wo


def _worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    from conftest import load_package
    pkg = load_package()
    from boondock_airband_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.stream_range(rank, world, NSTREAMS)
    local = torch.from_numpy(np.stack([_audio_for_stream(pkg, s) for s in range(lo, hi)]))
    got = shard.gather_audio(local, dst=0)
    if rank == 0:
        full = torch.cat(got, dim=0).numpy()
        np.save(os.path.join(outdir, "gathered.npy"), full)
    else:
        assert got is None
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
