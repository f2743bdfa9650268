#!/usr/bin/env python3
"""bench.py -- throughput of the demodulate() hot path on MI355X (BASELINE.json metric).

Workload of the JSON line (config.workload): BASELINE configs[1] -- per GPU ONE device stream of synthetic 2.56 MS/s u8 IQ,
8 AM channels, fft_size 512 (SURVEY 8d channel plan and signal recipe).  A "step" is one pass of the whole path (stage 1
+ stage 2, through the C ABI's device-resident entry) over `--seconds` of capture already resident in HBM.  With N GPUs the
streams shard stream-major over the ranks (boondock-airband_amd/shard.py: independent dongles, weak scaling, no data-path
collective) and the decimated audio + batch flags of every rank are gathered to rank 0 over RCCL inside the timed region.
With N > 1 the same line also carries `config5`: BASELINE configs[4] (64 streams x 32 mixed channels per GPU, fft 512)
measured with the full gather, with the open-batches-only gather and without the gather.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline`: the kernel with the largest time per step, from HIP events the library records
on its launch streams around each kernel; `achieved` = SURVEY 8(d) algorithmic bytes per sample x the samples one launch
covers / that kernel's mean launch duration (`frac_path`: the same bytes / the step's wall time; `alu_frac`: algorithmic
flops / step time / 157.3 TFLOP/s fp32 vector).  `cpu_baseline` is the CPU oracle (a port of the reference path,
oracle/airband_oracle.c) timed on one host core over a bounded sample of the same workload.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3  # same guide: peak fp32 vector
PMC_PROFILES = {"config2": "r03_pmc.csv", "config3": "r03_config3_pmc.csv", "config4": "r03_config4_pmc.csv", "am64": "r03_am64_pmc.csv"}
SAMPLE_RATE = 2560000
WAVE_BATCH = 2000
AGC_EXTRA = 100
HOP = SAMPLE_RATE // 16000  # complex samples per output sample

WORKLOADS = {
    "config2": "BASELINE configs[1]: 1 device stream per GPU @ 2.56 MS/s u8 IQ, 8 AM channels, fft_size=512",
    "config3": "BASELINE configs[2]: 1 device stream, 32 channels mixed AM+NFM + CTCSS, fft_size=2048",
    "config4": "BASELINE configs[3]: 64 device streams x 32 mixed channels per GPU, fft_size=512",
    "config5": "BASELINE configs[4]: 64 device streams x 32 mixed channels per GPU (512 streams on 8 GPUs), fft_size=512, gather to rank 0",
    "am64": "64 device streams x 8 AM channels per GPU, fft_size=512",
}


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
    samples = nb * WAVE_BATCH * HOP
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": samples / dt / 1e6, "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{nb} WAVE_BATCHes ({nb / 8:.1f} s of the same 1-stream 8-AM-channel fft512 capture), oracle/airband_oracle.c "
                      f"(own radix-2 FFT, FFTW3f absent), 1 thread = the reference's one-demod-thread-per-device model; host CPU: {model}",
            "x_realtime": samples / dt / SAMPLE_RATE}


def workload_shape(pkg, workload, streams, seconds):
    """(centre, channels, streams per GPU, fft_size_log, capture seconds per step)"""
    if workload == "config2":
        centre, chans = pkg.config2_channels()
        return centre, chans, 1, 9, seconds or 64.0
    if workload == "am64":
        centre, chans = pkg.config2_channels()
        return centre, chans, streams or 64, 9, seconds or 8.0
    centre, chans = pkg.config3_channels()
    if workload == "config3":
        return centre, chans, 1, 11, seconds or 8.0
    return centre, chans, streams or 64, 9, seconds or 2.0  # config4 / config5


def run_workload(pkg, args, workload, steps, warmup, gather_mode, rank, world, local_rank):
    """Times `steps` steps of one workload.  gather_mode: None | "full" | "open" (open batches only)."""
    import torch
    import torch.distributed as dist
    from boondock_airband_amd import shard

    centre, chans, nstreams, fft_log, seconds = workload_shape(pkg, workload, args.streams, args.seconds)
    dev = pkg.device_cfg(centerfreq=centre, fft_size_log=fft_log)
    nch = len(chans)
    n_iq = sum(1 for c in chans if c.has_iq_outputs)
    nbat = max(1, int(round(seconds * 8)))
    nsteps = nbat * WAVE_BATCH
    hop = 2 * HOP
    nbytes = ((nsteps + AGC_EXTRA) * hop + 2 * (1 << fft_log) + 255) // 256 * 256
    # the library's calls go to a stream of their own, not the NULL stream: its CU-restricted streams (MI_OPT_RESERVE_CUS) are blocking
    # ones and would synchronise with the NULL stream
    # (only where those streams are used -- plans of up to 64 rows on the time-parallel path: elsewhere a second user stream only takes a
    # hardware queue away from the library's own, BENCH_STREAM=null|side forces one)
    # (... and plans of 449 .. 512 rows on the serial kernel, whose stage 1 and k_demod take disjoint CUs: MI_OPT_SPLIT_CUS)
    # (config3: the 16 plain AM rows of its mixed plan take the time-parallel path; with the wide passes off 32 CUs the serial kernel of the
    #  other 16 rows finds free CUs as well: k_demod 1.73 -> 1.42 ms per call, 11.6 -> 14.0 GS/s)
    want_side = {"null": False, "side": True}.get(os.environ.get("BENCH_STREAM", ""),
                                                  (nstreams * nch <= 64 and workload in ("config2", "config3", "am64")) or
                                                  (workload == "am64" and 448 < nstreams * nch <= 512))
    stream = torch.cuda.Stream() if want_side else torch.cuda.current_stream()
    torch.cuda.set_stream(stream)

    # this rank's streams of the job (stream-major partition, the one the gloo test covers); their ids seed the generator
    lo, hi = shard.stream_range(rank, world, nstreams * world)
    assert hi - lo == nstreams
    amp = {} if workload in ("config2", "am64") else {"amp_q8": 1024}  # many carriers: keep the sum inside the u8 range
    gcfg = pkg.iqgen_cfg(sample_rate=SAMPLE_RATE, gate_samples=SAMPLE_RATE,
                         carriers=() if args.noise_only else pkg.carriers_for(centre, chans, **amp))
    d_iq = torch.empty((nstreams, nbytes), dtype=torch.uint8, device="cuda")
    pkg.iqgen_device(gcfg, lo, nstreams, nbytes, 0, nbytes // 2, d_iq.data_ptr(), stream.cuda_stream)

    # The audio of a step is gathered to rank 0 while the next steps compute: three output buffers, and the host waits for
    # the gather of buffer b only when buffer b is about to be written again.
    gathering = gather_mode is not None and dist.is_initialized()
    # (consecutive steps write different audio buffers: the segment passes of step k+1 may run under the tail of step k.  Five of them:
    #  with three the host waited for the gather of step k - 3 before it could submit step k, i.e. for the end of that step's tail, and the
    #  pipeline was three calls deep -- the same bound the timing reads at age 3 put on it)
    nbuf = int(os.environ.get("BENCH_NBUF", "5"))
    d_wos = [torch.empty((nstreams, nch, nsteps), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    d_axcs = [torch.empty((nstreams, nch, nbat), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    gath = shard.AudioGather((nstreams, nch, nsteps), nbat, torch.device("cuda", local_rank), dst=0) if gathering else None
    outs = [None] * nbuf
    if gathering and rank == 0:
        outs = [([torch.empty(sh, dtype=torch.float32, device="cuda") for sh in gath.shapes],
                 [torch.empty((sh[0], sh[1], nbat), dtype=torch.uint8, device="cuda") for sh in gath.shapes]) for _ in range(nbuf)]
    pending = [None] * nbuf

    h = pkg.Demod(dev, chans, nstreams=nstreams, max_batches=nbat, gpu=local_rank)
    # The capture is resident before the timed region starts: let the library read it without waiting for the previous
    # step's tail on the stream, so consecutive steps overlap.  --no-overlap times every step in isolation.
    if not args.no_overlap:
        h.set_option(pkg.OPT_EARLY_INPUT, 1)
    # prime: the handle's first call consumes AGC_EXTRA extra windows (waveend starts at 0 in the reference)
    h.process_device(d_iq.data_ptr(), nbytes, nbat, d_wos[0].data_ptr(), d_axcs[0].data_ptr(), hip_stream=stream.cuda_stream)
    base = d_iq.data_ptr() + AGC_EXTRA * hop
    kms = {}  # kernel name -> [total ms, launches, steps whose timings were read]

    nstep = [0]
    TIMING_AGE = int(os.environ.get("BENCH_TIMING_AGE", "4"))
    timed_hist = [False] * TIMING_AGE  # was the step one / two / three steps ago a timed one

    def add_times(times):
        for name, ms, launches in times:  # HIP events around the launches, on the launch streams
            acc = kms.setdefault(name, [0.0, 0, 0])
            acc[0] += ms
            acc[1] += launches
            acc[2] += 1

    def step(timed):
        b = nstep[0] % nbuf
        nstep[0] += 1
        if pending[b] is not None:
            # The gather of three steps ago still reads this buffer.  With MI_OPT_EARLY_INPUT the library may write a call's
            # audio before the stream reaches the call, so a stream-side wait is not enough: the host waits.
            pending[b].wait()
            pending[b] = None
        # every stream sits `nbytes` after the previous one (the priming call used the same stride)
        h.process_device(base, nbytes, nbat, d_wos[b].data_ptr(), d_axcs[b].data_ptr(), hip_stream=stream.cuda_stream)
        if gathering:
            pending[b] = gath.start(d_wos[b], d_axcs[b], open_only=(gather_mode == "open"), out=outs[b])
        # the timings of a step are read three steps later, so that reading them does not drain the pipeline; calls that
        # reuse an event set (plain serial calls) have none by then and are skipped -- every read counts one step
        if timed_hist[-1] and not os.environ.get("BENCH_LATE_TIMES"):
            add_times(h.kernel_times(age=TIMING_AGE))
        timed_hist.pop()
        timed_hist.insert(0, timed)

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(warmup):
        step(False)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    for age in range(TIMING_AGE - 1, -1, -1):  # the steps whose timings have not been read yet, oldest first
        if timed_hist[age]:
            add_times(h.kernel_times(age=age))  # (age 0 synchronises with the end of the last step)
        timed_hist[age] = False
    drain()  # every gather of the timed steps completes inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    samples_per_step_per_gpu = nsteps * HOP * nstreams
    value = samples_per_step_per_gpu * steps * world / dt / 1e6  # MS/s, whole job
    path = h.last_path()[0]
    if args.diag and rank == 0 and path == 1:
        for c in range(nch):
            print(f"diag ch{c}: {h.tp_debug(c)[1].tolist()}", file=sys.stderr)
    h.close()
    res = {"value": value, "ms_per_step": dt / steps * 1e3, "nch": nch, "n_iq": n_iq, "nstreams": nstreams, "fft_log": fft_log, "nbat": nbat,
           "samples_per_step_per_gpu": samples_per_step_per_gpu, "kms": kms, "path": path, "gathering": gathering, "dev": dev,
           "centre": centre, "chans": chans}
    return res


def kernel_table(res):
    """Per-kernel lines from the HIP-event sums.  ms / bytes are PER LAUNCH (what rocprofv3's average shows); a long call is
    processed in chunks, so a kernel runs launches_per_step times per step, each launch over 1/launches_per_step of the step."""
    nch, n_iq = res["nch"], res["n_iq"]
    per_sample = {  # algorithmic HBM bytes per complex input sample, per kernel (DESIGN.md "Kernels")
        "k_channelize": 2.0 + (4.0 * nch + 8.0 * n_iq) / HOP,  # u8 I+Q read once + |bin| (+ raw I/Q) written   
        "k_demod": (8.0 * nch + 16.0 * n_iq) / HOP,             # |bin| (+ raw I/Q) read + audio (+ raw I/Q) written
        "k_tp_full": 5.0 * nch / HOP,                           # |bin| read + 16 B of block aggregates per 16 steps
        "k_tp_core": 5.0 * nch / HOP,                           # aggregates + raw samples read
        "k_tp_seg": 8.0 * nch / HOP,                            # |bin| read + audio written
    }
    kernels = {}
    for name, (tot, launches, reads) in res["kms"].items():
        per_step = launches / reads
        ms = tot / launches
        nbytes = res["samples_per_step_per_gpu"] * per_sample.get(name.split("#")[0], 0.0) / per_step
        kernels[name] = {"ms": ms, "launches_per_step": per_step, "ms_per_step": tot / reads, "steps_timed": reads,
                         "algorithmic_bytes": nbytes, "GBps": (nbytes / (ms * 1e-3) / 1e9) if ms > 0 and nbytes > 0 else None}
    return kernels


def traffic_of(traffic, name):
    """PMC bytes per launch of a bench kernel name; stage 1 runs as l64_entry (the plan-compiled lane-resident kernel),
    k_channelize9p or k_channelize<...> depending on the plan and the options."""
    b = name.split("#")[0]
    alias = {"k_channelize": ("k_channelize", "l64_entry", "k_channelize9p"), "k_tp_core": ("k_tp_core", "k_tp_core2"),  # (k_tp_core2: the split chain)
             "k_demod": ("k_demod_pw", "k_demod_uni", "k_demod_packed", "k_demod")}                                   # (k_demod_pw: four waves per channel)
    for k in alias.get(b, (b,)):
        if k in traffic:
            return traffic[k]
    return None


def roofline_block(res, kernels, traffic, traffic_src):
    nch, n_iq, n = res["nch"], res["n_iq"], 1 << res["fft_log"]
    path_bytes_per_sample = 2.0 + (4.0 * nch + 8.0 * n_iq) / HOP  # 2.2 B/sample @ 8 ch, 2.8 @ 32 ch
    # algorithmic flops per sample: a full N-point FFT per output + the per-channel loop (40-150 flop per
    # channel and output sample by channel type; 60 for plain AM, 110 as the mixed-plan mean)
    stage2 = 60.0 if all(c.modulation == 0 and c.bandwidth == 0 for c in res["chans"]) else 110.0
    flop_per_sample = 5.0 * n * res["fft_log"] / HOP + stage2 * nch / HOP
    step_s = res["ms_per_step"] * 1e-3
    sps = res["samples_per_step_per_gpu"]
    if not kernels:
        return {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    k = kernels[dom]
    launch_samples = sps / k["launches_per_step"]
    achieved = launch_samples * path_bytes_per_sample / (k["ms"] * 1e-3) / 1e9
    path_gbs = sps * path_bytes_per_sample / step_s / 1e9
    tflops = sps * flop_per_sample / step_s / 1e12
    limited_by = {"k_tp_core": "latency: the serial squelch core chain, one workgroup per channel (dependent VALU issue, not bytes)",
                  "k_channelize": "fp32 VALU / LDS issue of stage 1 (65 flop/B at fft 512: far right of the HBM ridge)",
                  "k_demod": "dependent VALU issue of the waves of one channel (the serial per-channel loop: its longest chain, pre_filter_.full_, where helper waves run)",
                  "k_tp_seg": "lane latency of the segment pass", "k_tp_full": "lane latency of the aggregate pass"}.get(dom.split("#")[0], "launch latency of the tail passes")
    sum_ms = sum(v["ms_per_step"] for v in kernels.values())
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_of(traffic, dom), "traffic_source": traffic_src,
            # `bound` names the roofline the contract prices against; what actually limits the step:
            "limited_by": limited_by,
            "critical_path_ms": k["ms_per_step"],            # the dominant kernel's time per step: the step cannot be shorter
            "sum_of_kernel_ms_per_step": sum_ms,             # > ms_per_step because consecutive steps overlap on several streams
            "overlap_factor": sum_ms / res["ms_per_step"],
            "definition": "achieved = SURVEY 8(d) algorithmic bytes per sample (path_bytes_per_sample) x samples per launch / the dominant "
                          "kernel's mean launch duration (HIP events on its launch stream)",
            "path_bytes_per_sample": path_bytes_per_sample,
            "achieved_path": path_gbs, "frac_path": path_gbs / HBM_PEAK_GBS,  # the same bytes over the step's wall time
            "kernel_own_GBps": k["GBps"], "frac_kernel_own": (k["GBps"] / HBM_PEAK_GBS) if k["GBps"] else None,
            "alu": {"flop_per_sample": flop_per_sample, "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tflops / FP32_PEAK_TFLOPS,
                    "definition": "5 N log2 N / hop + stage-2 flops per sample, over the step's wall time, against the fp32 vector peak"},
            "note": "the path is not HBM-bound at fft 512 (65 flop/B): stage 1 is fp32-VALU / LDS bound, stage 2 is a per-channel "
                    "recurrence; both fractions are reported as SURVEY 8(d) asks"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=0.0, help="capture length per stream per step (HBM-resident); 0 = the workload's default")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="length of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--noise-only", action="store_true", help="diagnostic: no carriers in the synthetic capture (squelch never opens)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of audio to rank 0 (N>1)")
    ap.add_argument("--open-only", action="store_true", help="gather only the batches whose squelch flag is not NO_SIGNAL")
    ap.add_argument("--no-config5", action="store_true", help="N>1: do not add the config5 measurements to the line")
    ap.add_argument("--diag", action="store_true", help="print the time-parallel path's per-channel counters to stderr")
    ap.add_argument("--no-overlap", action="store_true", help="do not let consecutive steps overlap (MI_OPT_EARLY_INPUT off)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2",
                    help="config2 = BASELINE configs[1], the bench line (default); the others are extra measurements, not the driver's line")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU for the many-stream workloads (default 64)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    pkg = load_package()
    torch.cuda.set_device(local_rank)
    force_gather = os.environ.get("BENCH_FORCE_GATHER") == "1"
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    gather_mode = None if (args.no_gather or not dist.is_initialized()) else ("open" if args.open_only else "full")
    res = run_workload(pkg, args, args.workload, args.steps, args.warmup, gather_mode, rank, world, local_rank)

    extra = {}
    if (world > 1 or os.environ.get("BENCH_FORCE_CONFIG5") == "1") and not args.no_config5 and args.workload == "config2":
        # BASELINE configs[4] on the same ranks: short runs, outside the timed region of the line's own metric
        saved = (args.streams, args.seconds)
        args.streams, args.seconds = 0, 0.0
        k5, w5 = max(3, min(args.steps, 10)), 2
        for label, mode in (("gather_full", "full"), ("gather_open_batches_only", "open"), ("no_gather", None)):
            r5 = run_workload(pkg, args, "config5", k5, w5, mode, rank, world, local_rank)
            extra[label] = {"value": r5["value"], "unit": "MS/s", "ms_per_step": r5["ms_per_step"], "steps": k5, "warmup": w5,
                            "x_realtime_per_stream": r5["value"] / world / r5["nstreams"] / (SAMPLE_RATE / 1e6),
                            "audio_bytes_to_rank0_per_step": (world - 1) * r5["nstreams"] * r5["nch"] * r5["nbat"] * WAVE_BATCH * 4 if mode == "full" else None}
        extra["workload"] = WORKLOADS["config5"]
        extra["streams_total"] = world * 64
        args.streams, args.seconds = saved

    if rank == 0:
        kernels = kernel_table(res)
        # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside the bench):
        # only quoted when the launch geometry is the one that was profiled (default --seconds, default chunking).
        traffic, traffic_src = {}, None
        pmc_name = PMC_PROFILES.get(args.workload, "")
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        default_geometry = not args.seconds and not args.streams and not args.noise_only and "MI_AIRBAND_TP_CHUNKS" not in os.environ
        if pmc_name and os.path.exists(pmc) and default_geometry:
            import csv
            for row in csv.DictReader(open(pmc)):
                traffic[row["kernel"]] = int(row["hbm_bytes_per_launch"])
            sha = hashlib.sha1(open(pmc, "rb").read()).hexdigest()[:12]
            traffic_src = (f"profiles/{pmc_name} (sha1 {sha}): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command "
                           f"(tools/make_profiles.sh), FETCH_SIZE x2 (gfx950); not re-measured inside this run")
        for name, k in kernels.items():
            k["traffic"] = traffic_of(traffic, name)
        nch, nstreams = res["nch"], res["nstreams"]
        out = {
            "metric": f"IQ MS/s processed (x real-time) @ {nch}ch fft_size={1 << res['fft_log']}",
            "value": res["value"],
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload],
                       "streams_per_gpu": nstreams, "channels": nch, "fft_size": 1 << res["fft_log"], "capture_seconds_per_step": res["nbat"] / 8.0,
                       "audio_gather_to_rank0": bool(res["gathering"]), "gather": gather_mode if res["gathering"] else None,
                       "gather_overlaps_next_step": bool(res["gathering"]), "partition": "shard.stream_range (stream-major)",
                       "stage2_path": ("time-parallel (plain AM rows) + serial kernel (the others), side by side" if "k_demod" in kernels else "time-parallel")
                       if res["path"] == 1 else "serial"},
            "x_realtime_per_stream": res["value"] / world / nstreams / (SAMPLE_RATE / 1e6),
            "roofline": roofline_block(res, kernels, traffic, traffic_src),
            "kernels": kernels,
        }
        if extra:
            out["config5"] = extra
        if world == 1 and args.cpu_seconds > 0 and args.workload == "config2":
            out["cpu_baseline"] = cpu_baseline(pkg, res["dev"], res["centre"], res["chans"], args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
