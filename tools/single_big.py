"""One large device-resident transform per call (batch = 1): per-call time and, under rocprofv3 --kernel-trace --stats, the kernels
behind it.  usage: single_big.py [log2n=20] [c32|c64] [reps=200]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "c64") else np.float32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n = 1 << L
f = kofft_amd.HipFftImpl(dt)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
x = torch.empty((n, 2), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
with torch.cuda.stream(stream):
    for _ in range(20):
        f.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, 1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(stream)
    for _ in range(reps):
        f.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, 1)
    e.record(stream); torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
es = 16 if dt == np.float64 else 8
print(f"2^{L} {np.dtype(dt).name} batch 1: {us:.1f} us per transform = {2*es*n/us/1e6:.2f} TB/s of algorithmic bytes ({2*es*n/us/1e6/8:.3f} of the roofline)")
