import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from bench import load_package
pkg = load_package()
centre, chans = pkg.config2_channels()
dev = pkg.device_cfg(centerfreq=centre)
nbat = 512; nsteps = nbat*2000; hop = 320
nbytes = ((nsteps+100)*hop + 1024 + 255)//256*256
s = torch.cuda.current_stream()
cfg = pkg.iqgen_cfg(sample_rate=2560000, gate_samples=2560000, carriers=pkg.carriers_for(centre, chans))
d_iq = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
pkg.iqgen_device(cfg, 0, 1, nbytes, 0, nbytes//2, d_iq.data_ptr(), s.cuda_stream)
d_wo = torch.empty((1,8,nsteps), dtype=torch.float32, device="cuda"); d_wo2 = torch.empty_like(d_wo); d_axc = torch.empty((1,8,nbat), dtype=torch.uint8, device='cuda')
h = pkg.Demod(dev, chans, max_batches=nbat)
h.set_option(pkg.OPT_EARLY_INPUT, 1)
h.process_device(d_iq.data_ptr(), nbytes, nbat, d_wo.data_ptr(), d_axc.data_ptr(), hip_stream=s.cuda_stream)
base = d_iq.data_ptr() + 100*hop
torch.cuda.synchronize()
for trial in range(2):
    t0 = time.perf_counter(); ts = []
    for k in range(10):
        a = time.perf_counter()
        h.process_device(base, nbytes-100*hop, nbat, (d_wo if k % 2 else d_wo2).data_ptr(), d_axc.data_ptr(), hip_stream=s.cuda_stream)
        ts.append(time.perf_counter()-a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue ms per call:", [round(x*1e3,2) for x in ts], "enqueue total", round((t1-t0)*1e3,2), "wall per call", round((t2-t0)*1e2,3), "ms")
