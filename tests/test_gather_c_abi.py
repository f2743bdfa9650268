"""mi_gather_* (include/mi_airband.h): the C-ABI gather of audio + flags to rank 0 over RCCL, for a C++ host.
World size 1 runs on the one-GPU box (local copy, open-batches-only compaction / scatter kernels, stream hand-off).  Every world > 1
branch -- grouped sends / receives, the two-phase open-batches-only protocol with its block counts and landing offsets, uneven and
empty ranks -- runs there too, through the loopback transport (mi_gather_loopback_id: the ranks are threads of one process, the
transfers device-to-device copies ordered by events, the call sequence RCCL's).  The RCCL transport itself (ncclCommInitRank with
the id by value, ncclSend / ncclRecv) needs two GPUs: that test skips on the one-GPU box and is the only part left unpinned there.
N > 1 host logic is also covered on CPU by tests/test_distributed_gloo.py through the Python twin (shard.AudioGather)."""
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


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [[3, 2], [2, 0, 3, 1], [0, 4, 1]])
def test_gather_many_ranks_through_the_loopback_transport(pkg, streams):
    """Ranks as threads on the one GPU: full and open-only gathers, two steps each, uneven and empty ranks (rank 0 included)."""
    import threading
    import torch
    world, nch, nbat = len(streams), 3, 4
    rng = np.random.default_rng(17 + world)
    job = int(rng.integers(1, 1 << 40))
    uid = pkg.Gather.loopback_id(job)
    steps = [_fake_step(rng, sum(streams), nch, nbat) for _ in range(2)]
    for _, fl in steps:
        fl[:, :, :] = np.where(rng.random(fl.shape) < 0.4, ord("*"), ord(" "))
    steps[1][1][sum(streams[:2]):sum(streams[:2]) + 1] = ord(" ")  # a rank-2 stream (if any) with nothing open in the second step
    errors, results = [], {}

    def run(rank):
        try:
            lo = sum(streams[:rank])
            g = pkg.Gather(uid, rank, world, 0, streams, nch, nbat)
            st = torch.cuda.Stream()
            for open_only in (False, True):
                for k, (audio, flags) in enumerate(steps):
                    mine_a = torch.from_numpy(np.ascontiguousarray(audio[lo:lo + streams[rank]])).cuda()
                    mine_f = torch.from_numpy(np.ascontiguousarray(flags[lo:lo + streams[rank]])).cuda()
                    d_all = torch.full((sum(streams), nch, nbat * WAVE_BATCH), 5.0, dtype=torch.float32, device="cuda") if rank == 0 else None
                    d_allf = torch.zeros((sum(streams), nch, nbat), dtype=torch.uint8, device="cuda") if rank == 0 else None
                    torch.cuda.synchronize()
                    g.audio(mine_a.data_ptr() if streams[rank] else None, mine_f.data_ptr() if streams[rank] else None, nbat,
                            None if d_all is None else d_all.data_ptr(), None if d_allf is None else d_allf.data_ptr(), open_only=open_only,
                            hip_stream=st.cuda_stream)
                    g.stream_wait(st.cuda_stream)
                    g.sync()
                    st.synchronize()
                    if rank == 0:
                        results[(open_only, k)] = (d_all.cpu().numpy(), d_allf.cpu().numpy())
            g.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for open_only in (False, True):
        for k, (audio, flags) in enumerate(steps):
            got_a, got_f = results[(open_only, k)]
            want = audio if not open_only else np.where(np.repeat(flags != ord(" "), WAVE_BATCH, axis=2), audio, 0.0).astype(np.float32)
            assert np.array_equal(got_f, flags), (open_only, k)
            assert np.array_equal(got_a, want), (open_only, k)


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
