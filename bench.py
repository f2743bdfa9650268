#!/usr/bin/env python3
"""bench.py -- throughput of the demodulate() hot path on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE configs[1] -- per GPU ONE device stream of synthetic 2.56 MS/s u8 IQ,
8 AM channels, fft_size 512 (SURVEY 8d channel plan and signal recipe).  A "step" is one pass of the whole
path (channelize kernel + demod kernel, through the C ABI's device-resident entry) over `--seconds` of
capture already resident in HBM.  With N GPUs each rank owns its own stream (independent dongles, weak
scaling) and the decimated audio of every rank is gathered to rank 0 over RCCL inside the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the kernel that dominates the step, from HIP events the
library records on its launch stream around each kernel; `cpu_baseline` is the CPU oracle (a port of the
reference path, oracle/airband_oracle.c) timed on one host core over a bounded sample of the same workload.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_PROFILE = "r01_final_pmc.csv"
SAMPLE_RATE = 2560000
WAVE_BATCH = 2000
AGC_EXTRA = 100


def load_package():
    name = "boondock_airband_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "boondock-airband_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(pkg, dev, centre, chans, seconds):
    """The oracle (port of the reference's CPU path) on one host core, free-running from memory."""
    from common import gen_iq, oracle_run
    nbat = int(seconds * 8)
    iq, _ = gen_iq(pkg, dev, centre, chans, nbat, gate_div=1)
    t0 = time.perf_counter()
    nb, _, _, _ = oracle_run(dev, chans, iq, nbat)
    dt = time.perf_counter() - t0
    samples = nb * WAVE_BATCH * (SAMPLE_RATE // 16000)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": samples / dt / 1e6, "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{nb} WAVE_BATCHes ({nb / 8:.1f} s of the same 1-stream 8-AM-channel fft512 capture), oracle/airband_oracle.c "
                      f"(own radix-2 FFT, FFTW3f absent), 1 thread = the reference's one-demod-thread-per-device model; host CPU: {model}",
            "x_realtime": samples / dt / SAMPLE_RATE}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=64.0, help="capture length per stream per step (HBM-resident)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="length of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--noise-only", action="store_true", help="diagnostic: no carriers in the synthetic capture (squelch never opens)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of audio to rank 0 (N>1)")
    ap.add_argument("--diag", action="store_true", help="print the time-parallel path's per-channel counters to stderr")
    ap.add_argument("--no-overlap", action="store_true", help="do not let consecutive steps overlap (MI_OPT_EARLY_INPUT off)")
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "am64"], default="config2",
                    help="config2 = BASELINE configs[1], the bench line (default); config3 = 1 stream x 32 mixed AM/NFM/CTCSS channels at fft 2048; "
                         "config4 = 64 streams x the config-3 plan at fft 512; am64 = 64 streams x the config-2 plan (extra measurements, not the driver's line)")
    ap.add_argument("--streams", type=int, default=0, help="with --workload am64: this many streams instead of 64")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    pkg = load_package()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    nstreams, fft_log = 1, 9
    if args.workload == "config2":
        centre, chans = pkg.config2_channels()
    elif args.workload == "am64":
        centre, chans = pkg.config2_channels()
        nstreams = args.streams or 64  # (--streams: the stream-count sweep of DESIGN.md section 6)
        if args.seconds == 64.0:
            args.seconds = 8.0
        args.cpu_seconds = 0.0
    else:
        centre, chans = pkg.config3_channels()
        nstreams, fft_log = (1, 11) if args.workload == "config3" else (64, 9)
        if args.seconds == 64.0:
            args.seconds = 8.0 if args.workload == "config3" else 2.0
        args.cpu_seconds = 0.0
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=fft_log)
    nch = len(chans)
    nbat = max(1, int(round(args.seconds * 8)))
    nsteps = nbat * WAVE_BATCH
    hop = 2 * (SAMPLE_RATE // 16000)
    nbytes = ((nsteps + AGC_EXTRA) * hop + 2 * (1 << fft_log) + 255) // 256 * 256
    stream = torch.cuda.current_stream()

    # synthetic capture of this rank's stream, generated on the device (same integer recipe as the host generator)
    amp = {} if args.workload == "config2" else {"amp_q8": 1024}  # 16 carriers: keep the sum inside the u8 range
    gcfg = pkg.iqgen_cfg(sample_rate=SAMPLE_RATE, gate_samples=SAMPLE_RATE, carriers=() if args.noise_only else pkg.carriers_for(centre, chans, **amp))
    d_iq = torch.empty((nstreams, nbytes), dtype=torch.uint8, device="cuda")
    pkg.iqgen_device(gcfg, rank * nstreams, nstreams, nbytes, 0, nbytes // 2, d_iq.data_ptr(), stream.cuda_stream)
    # The audio of a step is gathered to rank 0 while the next step computes: two output buffers, the gather of buffer b
    # (RCCL, asynchronous on its own stream) is waited for by the stream only when buffer b is written again.
    gathering = (world > 1 or os.environ.get("BENCH_FORCE_GATHER") == "1") and not args.no_gather
    if gathering and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    nbuf = 3  # consecutive steps write different audio buffers (the segment passes of step k+1 may run under the tail of step k), three deep like the timing reads
    d_wos = [torch.empty((nstreams, nch, nsteps), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    d_wo = d_wos[0]
    d_axc = torch.empty((nstreams, nch, nbat), dtype=torch.uint8, device="cuda")
    gather_lists = [[torch.empty_like(d_wo) for _ in range(world)] if (gathering and rank == 0) else None for _ in range(nbuf)]
    pending = [None] * nbuf

    h = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=nbat, gpu=local_rank)
    # The capture is resident before the timed region starts: let the library read it without waiting for the previous
    # step's tail on the stream, so consecutive steps overlap (stage 1 + core chain of step k+1 under the segment / fix
    # passes of step k).  --no-overlap times every step in isolation.
    if not args.no_overlap:
        h.set_option(pkg.OPT_EARLY_INPUT, 1)
    # prime: the handle's first call consumes AGC_EXTRA extra windows (waveend starts at 0 in the reference)
    h.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_axc.data_ptr(), hip_stream=stream.cuda_stream)
    base = d_iq.data_ptr() + AGC_EXTRA * hop
    kms = {}  # kernel name -> [total ms over the timed steps, launches]

    nstep = [0]
    TIMING_AGE = 3
    timed_hist = [False] * TIMING_AGE  # was the step one / two / three steps ago a timed one

    def add_times(times):
        for name, ms, launches in times:  # HIP events around the launches, on the launch streams
            acc = kms.setdefault(name, [0.0, 0])
            acc[0] += ms
            acc[1] += launches

    def step(timed):
        b = nstep[0] % nbuf
        nstep[0] += 1
        if pending[b] is not None:
            # The gather of two steps ago still reads this buffer.  With MI_OPT_EARLY_INPUT the library may write a call's
            # audio before the stream reaches the call, so a stream-side wait is not enough: the host waits (the gather
            # started when that step finished, a whole step ago, so this normally returns at once).
            while not pending[b].is_completed():
                time.sleep(0)
            pending[b].wait()
            pending[b] = None
        h.process_device(base, nbytes - AGC_EXTRA * hop, nbat, d_wos[b].data_ptr(), d_axc.data_ptr(), hip_stream=stream.cuda_stream)
        if gathering:
            pending[b] = dist.gather(d_wos[b], gather_lists[b], dst=0, async_op=True)
        # the timings of a step are read three steps later, so that reading them does not drain the pipeline (the tail of a
        # call ends about one call after its core chain: with a depth of two the next call's front started late now and then)
        if timed_hist[-1]:
            add_times(h.kernel_times(age=TIMING_AGE))
        timed_hist.pop()
        timed_hist.insert(0, timed)

    for _ in range(args.warmup):
        step(False)
    for b in range(nbuf):
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    for age in range(TIMING_AGE - 1, -1, -1):  # the steps whose timings have not been read yet, oldest first
        if timed_hist[age]:
            add_times(h.kernel_times(age=age))  # (age 0 synchronises with the end of the last step)
        timed_hist[age] = False
    for b in range(nbuf):  # every gather of the timed steps completes inside the timed region
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    samples_per_step_per_gpu = nsteps * (SAMPLE_RATE // 16000) * nstreams
    total_samples = samples_per_step_per_gpu * args.steps * world
    value = total_samples / dt / 1e6  # MS/s, whole job

    if args.diag and rank == 0 and h.last_path()[0] == 1:
        for c in range(nch):
            print(f"diag ch{c}: {h.tp_debug(c)[1].tolist()}", file=sys.stderr)
    if rank == 0:
        # algorithmic HBM bytes per complex input sample, per kernel (DESIGN.md "Kernels"); hop = 160 samples
        hopn = SAMPLE_RATE // 16000
        per_sample = {
            "k_channelize": 2.0 + 4.0 * nch / hopn,    # u8 I+Q read once + |bin| written        (2.2 B/sample @ 8 ch)
            "k_demod": 8.0 * nch / hopn,               # |bin| read + audio written
            "k_tp_full": 5.0 * nch / hopn,             # |bin| read + 16 B of block aggregates per 16 steps
            "k_tp_core": 5.0 * nch / hopn,             # aggregates + raw samples read
            "k_tp_seg": 8.0 * nch / hopn,              # |bin| read + audio written
        }
        # A long call is processed in chunks: a kernel runs `launches_per_step` times per step, each launch over
        # 1/launches_per_step of the samples.  ms / bytes below are PER LAUNCH (what rocprofv3's average shows);
        # ms_per_step is their sum over the step (kernels of different chunks and steps overlap on several streams).
        kernels = {}
        for name, (tot, launches) in kms.items():
            per_step = launches / args.steps
            ms = tot / launches
            nbytes = samples_per_step_per_gpu * per_sample.get(name.split("#")[0], 0.0) / per_step
            kernels[name] = {"ms": ms, "launches_per_step": per_step, "ms_per_step": tot / args.steps, "algorithmic_bytes": nbytes,
                             "GBps": (nbytes / (ms * 1e-3) / 1e9) if ms > 0 and nbytes > 0 else None}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        achieved = kernels[dom]["GBps"]
        # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the bench):
        # only quoted when the launch geometry is the one that was profiled (default --seconds, default chunking).
        traffic, traffic_src = {}, None
        pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if os.path.exists(pmc) and nbat == 512 and args.workload == "config2" and not args.noise_only and "MI_AIRBAND_TP_CHUNKS" not in os.environ:
            import csv
            for row in csv.DictReader(open(pmc)):
                traffic[row["kernel"]] = int(row["hbm_bytes_per_launch"])
            traffic_src = f"profiles/{PMC_PROFILE}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command, FETCH_SIZE x2 (gfx950)"
        for name, k in kernels.items():
            base = name.split("#")[0]
            k["traffic"] = traffic.get(base, traffic.get(base + "9p"))  # k_channelize9p: the pruned N = 512 instantiation
        out = {
            "metric": f"IQ MS/s processed (x real-time) @ {nch}ch fft_size={1 << fft_log}",
            "value": value,
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": {"config2": "BASELINE configs[1]: 1 device stream per GPU @ 2.56 MS/s u8 IQ, 8 AM channels, fft_size=512",
                                    "config3": "BASELINE configs[2]: 1 device stream, 32 channels mixed AM+NFM + CTCSS, fft_size=2048",
                                    "config4": "BASELINE configs[3]: 64 device streams x 32 mixed channels per GPU, fft_size=512",
                                    "am64": "64 device streams x 8 AM channels per GPU, fft_size=512"}[args.workload],
                       "streams_per_gpu": nstreams, "channels": nch, "fft_size": 1 << fft_log, "capture_seconds_per_step": nbat / 8.0,
                       "audio_gather_to_rank0": bool(gathering), "gather_overlaps_next_step": bool(gathering),
                       "stage2_path": "time-parallel" if h.last_path()[0] == 1 else "serial"},
            "x_realtime_per_stream": value / world / nstreams / (SAMPLE_RATE / 1e6),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": kernels[dom]["traffic"],
                         "traffic_source": traffic_src,
                         "note": "k_tp_core is the per-channel serial recurrence (one wave per channel): its time is set by VALU issue "
                                 "latency of a lone wave, not by HBM; k_channelize is the kernel that streams the capture (see kernels)"},
            "kernels": kernels,
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(pkg, dev, centre, chans, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    h.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
