"""Multi-GPU host logic: independent device streams shard across ranks (one process per GPU); there is no
exchange step inside the path (the reference runs one demod thread per device, rtl_airband.cpp:1044-1078).
The only collective is the gather of decimated audio (+ the per-batch axcindicate flags) to rank 0, where the
reference's output / mixer threads live (output.cpp:899-961; BASELINE.json north_star), issued through
torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests.  bench.py and
tests/test_distributed_gloo.py both go through this module; the C-ABI twin for a C++ host is mi_gather_*
(include/mi_airband.h, csrc/gather.cpp)."""

WAVE_BATCH = 2000
NO_SIGNAL = 0x20  # ' ' (enum status, boondock_airband.h:101)


def stream_range(rank, world, nstreams):
    """Stream-major contiguous partition: rank r owns [lo, hi).  Per-channel state never migrates."""
    if world < 1 or not (0 <= rank < world) or nstreams < 0:
        raise ValueError("bad rank/world/nstreams")
    base, extra = divmod(nstreams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class AudioGather:
    """Gather of each rank's audio [streams_local][nch][nbatches*WAVE_BATCH] f32 and flags [streams_local][nch][nbatches] u8
    to `dst`, reusable step after step: the (possibly uneven) per-rank shapes are exchanged once, here.  Point-to-point under
    the hood (isend / irecv; RCCL's gather is grouped send/recv as well), so rank-0 ingress over xGMI is the bound, not HBM
    (SURVEY 8e).

    start() posts the transfers and returns a handle; handle.wait() returns ([audio per rank], [flags per rank]) on dst and
    None elsewhere, so a caller can compute the next step meanwhile.

    open_only=True sends only the (row, batch) blocks whose flag is not NO_SIGNAL -- what the reference's non-continuous
    outputs consume (output.cpp:518,568 skip NO_SIGNAL batches) -- and dst rebuilds full-size tensors with the skipped
    batches zero.  The flags always travel whole (they are 1/8000 of the audio), and they tell dst how many blocks follow."""

    def __init__(self, audio_shape, nbatches, device, dst=0, group=None):
        import torch
        import torch.distributed as dist

        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        s, c, n = (int(v) for v in audio_shape)
        if n != nbatches * WAVE_BATCH:
            raise ValueError("audio rows must hold nbatches * WAVE_BATCH samples")
        self.nbatches = nbatches
        mine = torch.tensor([s, c, n], dtype=torch.int64, device=device)
        shapes = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(shapes, mine, group=group)
        self.shapes = [tuple(int(v) for v in t.tolist()) for t in shapes]

    class _Handle:
        def __init__(self, owner, reqs, audio, flags, open_only, keep):
            self.o, self.reqs, self.audio, self.flags, self.open_only, self.keep = owner, reqs, audio, flags, open_only, keep
            self.done = False

        def is_completed(self):
            return all(r.is_completed() for r in self.reqs)

        def wait(self):
            o = self.o
            for r in self.reqs:
                r.wait()
            self.reqs = []
            if o.rank != o.dst:
                self.keep = None
                return None
            if self.open_only and not self.done:
                torch = o.torch
                ops, parts = [], {}
                for r in range(o.world):
                    if r == o.dst:
                        continue
                    idx = (self.flags[r].reshape(-1) != NO_SIGNAL).nonzero().reshape(-1)
                    if idx.numel():
                        buf = torch.empty((idx.numel(), WAVE_BATCH), dtype=self.audio[r].dtype, device=o.device)
                        ops.append(o.dist.P2POp(o.dist.irecv, buf, r, o.group))
                        parts[r] = (idx, buf)
                for q in (o.dist.batch_isend_irecv(ops) if ops else []):
                    q.wait()
                for r, (idx, buf) in parts.items():
                    self.audio[r].zero_()
                    self.audio[r].view(-1, WAVE_BATCH)[idx] = buf
                for r in range(o.world):
                    if r != o.dst and r not in parts:
                        self.audio[r].zero_()
            self.done = True
            return self.audio, self.flags

    def start(self, audio, flags, open_only=False, out=None):
        """audio / flags: this rank's tensors.  `out`: optional (audio list, flags list) on dst to receive into."""
        torch, dist = self.torch, self.dist
        reqs, keep = [], []
        if self.rank == self.dst:
            if out is None:
                out = ([torch.empty(sh, dtype=audio.dtype, device=self.device) for sh in self.shapes],
                       [torch.empty((sh[0], sh[1], self.nbatches), dtype=flags.dtype, device=self.device) for sh in self.shapes])
            oa, of = out
            oa[self.dst].copy_(audio)
            of[self.dst].copy_(flags)
            if open_only:
                closed = (flags.reshape(-1) == NO_SIGNAL).nonzero().reshape(-1)
                if closed.numel():
                    oa[self.dst].view(-1, WAVE_BATCH)[closed] = 0
            ops = []
            for r in range(self.world):
                if r == self.dst:
                    continue
                ops.append(dist.P2POp(dist.irecv, of[r], r, self.group))
                if not open_only:
                    ops.append(dist.P2POp(dist.irecv, oa[r], r, self.group))
            # one group: the transfers from all peers run side by side (ncclGroupStart/End under "nccl")
            reqs = dist.batch_isend_irecv(ops) if ops else []
            return self._Handle(self, reqs, oa, of, open_only, keep)
        flags = flags.contiguous()
        ops = [dist.P2POp(dist.isend, flags, self.dst, self.group)]
        keep.append(flags)
        payload = None
        if open_only:
            idx = (flags.reshape(-1) != NO_SIGNAL).nonzero().reshape(-1)  # (synchronises with the producer of `flags`)
            if idx.numel():
                payload = audio.reshape(-1, WAVE_BATCH)[idx].contiguous()
        else:
            payload = audio.contiguous()
        if payload is not None:
            keep.append(payload)
        if open_only:
            # dst posts the receive of the payload only after it has seen the flags: a group of its own on both sides
            reqs = dist.batch_isend_irecv(ops)
            if payload is not None:
                reqs += dist.batch_isend_irecv([dist.P2POp(dist.isend, payload, self.dst, self.group)])
        else:
            reqs = dist.batch_isend_irecv(ops + [dist.P2POp(dist.isend, payload, self.dst, self.group)])
        return self._Handle(self, reqs, None, None, open_only, keep)


def gather_audio(local, flags=None, dst=0, group=None, open_only=False):
    """One-shot, blocking form of AudioGather.  Returns the list of per-rank audio tensors on dst (and the flags list too
    when `flags` is given), None elsewhere."""
    import torch

    nb = local.shape[2] // WAVE_BATCH
    fl = flags if flags is not None else torch.full((local.shape[0], local.shape[1], nb), 0x2A, dtype=torch.uint8, device=local.device)
    got = AudioGather(tuple(local.shape), nb, local.device, dst=dst, group=group).start(local, fl, open_only=open_only).wait()
    if got is None:
        return None
    return got if flags is not None else got[0]
