"""mi_gather_* (include/mi_airband.h): the C-ABI gather of audio + flags to rank 0 over RCCL, for a C++ host.
World size 1 runs on the one-GPU box (local copy, open-batches-only compaction / scatter kernels, stream hand-off); the
two-rank test needs two GPUs and skips otherwise -- N > 1 host logic is also covered on CPU by tests/test_distributed_gloo.py
through the Python twin (shard.AudioGather), which bench.py uses."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WAVE_BATCH = 2000


def _fake_step(rng, streams, nch, nbat):
    audio = rng.standard_normal((streams, nch, nbat * WAVE_BATCH)).astype(np.float32)
    flags = np.where(rng.random((streams, nch, nbat)) < 0.4, ord("*"), ord(" ")).astype(np.uint8)
    flags[0, 0, 0] = ord("<")  # AFC indicators count as signal too
    return audio, flags


@pytest.mark.gpu
@pytest.mark.parametrize("open_only", [False, True])
def test_gather_world1_full_and_open_only(pkg, open_only):
    import torch
    rng = np.random.default_rng(3)
    streams, nch, nbat = 5, 4, 3
    audio, flags = _fake_step(rng, streams, nch, nbat)
    g = pkg.Gather(None, 0, 1, 0, [streams], nch, nbat)
    d_a, d_f = torch.from_numpy(audio).cuda(), torch.from_numpy(flags).cuda()
    d_all = torch.full_like(d_a, 7.0)
    d_allf = torch.zeros_like(d_f)
    s = torch.cuda.current_stream().cuda_stream
    for step in range(2):  # the handle is reusable step after step
        g.audio(d_a.data_ptr(), d_f.data_ptr(), nbat, d_all.data_ptr(), d_allf.data_ptr(), open_only=open_only, hip_stream=s)
        g.stream_wait(s)
        torch.cuda.synchronize()
        want = audio if not open_only else np.where(np.repeat(flags != ord(" "), WAVE_BATCH, axis=2), audio, 0.0).astype(np.float32)
        assert np.array_equal(d_all.cpu().numpy(), want)
        assert np.array_equal(d_allf.cpu().numpy(), flags)
        d_all.fill_(9.0)
    g.sync()
    g.close()


def _rank(rank, world, tmp):
    sys.path.insert(0, HERE)
    import torch
    from conftest import load_package
    pkg = load_package()
    torch.cuda.set_device(rank)
    idfile = os.path.join(tmp, "uid")
    if rank == 0:
        uid = pkg.Gather.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        import time
        while not os.path.exists(idfile):
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    streams = [3, 2]
    nch, nbat = 4, 2
    rng = np.random.default_rng(11)
    audio, flags = _fake_step(rng, sum(streams), nch, nbat)
    lo = sum(streams[:rank])
    mine_a = torch.from_numpy(audio[lo:lo + streams[rank]]).cuda()
    mine_f = torch.from_numpy(flags[lo:lo + streams[rank]]).cuda()
    g = pkg.Gather(uid, rank, world, rank, streams, nch, nbat)
    s = torch.cuda.current_stream().cuda_stream
    for open_only in (False, True):
        d_all = torch.full((sum(streams), nch, nbat * WAVE_BATCH), 5.0, dtype=torch.float32, device="cuda") if rank == 0 else None
        d_allf = torch.zeros((sum(streams), nch, nbat), dtype=torch.uint8, device="cuda") if rank == 0 else None
        g.audio(mine_a.data_ptr(), mine_f.data_ptr(), nbat, None if d_all is None else d_all.data_ptr(), None if d_allf is None else d_allf.data_ptr(),
                open_only=open_only, hip_stream=s)
        g.sync()
        if rank == 0:
            want = audio if not open_only else np.where(np.repeat(flags != ord(" "), WAVE_BATCH, axis=2), audio, 0.0).astype(np.float32)
            assert np.array_equal(d_all.cpu().numpy(), want), f"open_only={open_only}"
            assert np.array_equal(d_allf.cpu().numpy(), flags)
    g.close()


@pytest.mark.gpu
def test_gather_two_ranks_over_rccl(pkg, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU node); world size 1 and the gloo twin are covered elsewhere")
    import torch.multiprocessing as mp
    mp.spawn(_rank, args=(2, str(tmp_path)), nprocs=2, join=True)


def test_gather_rejects_bad_geometry(pkg):
    import ctypes as C
    lib = pkg.lib()
    h = C.c_void_p()
    arr = (C.c_int * 2)(1, 1)
    assert lib.mi_gather_create(None, 0, 2, 0, arr, 4, 1, C.byref(h)) == pkg.MI_ERR_INVALID  # world > 1 needs the id
    assert lib.mi_gather_create(None, 3, 2, 0, arr, 4, 1, C.byref(h)) == pkg.MI_ERR_INVALID
    assert lib.mi_gather_create(None, 0, 1, 0, None, 4, 1, C.byref(h)) == pkg.MI_ERR_INVALID
