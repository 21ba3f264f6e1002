"""A launch-bound loop of small device calls, direct against replayed from a hipGraph (the calls are capturable once the context
has seen the size: tests/test_gpu_streams.py)."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import time, numpy as np, torch, kofft_amd
dev = torch.device("cuda", 0)
f = kofft_amd.HipFftImpl(np.float32)
s = torch.cuda.Stream(device=dev); f.set_stream(s.cuda_stream)
n, batch, calls = 1024, 4, 200
with torch.cuda.stream(s):
    x = torch.empty((calls, batch, n, 2), dtype=torch.float32, device=dev).uniform_(-1, 1)
    y = torch.empty_like(x)
    f.fft_dev_oop(x[0].data_ptr(), y[0].data_ptr(), n, batch)
torch.cuda.synchronize()
def direct():
    for c in range(calls):
        f.fft_dev_oop(x[c].data_ptr(), y[c].data_ptr(), n, batch)
ptrs = [(x[c].data_ptr(), y[c].data_ptr()) for c in range(calls)]
def direct_ptrs():
    for a, b in ptrs:
        f.fft_dev_oop(a, b, n, batch)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    direct_ptrs()
torch.cuda.synchronize()
for name, fn in (("direct calls", direct_ptrs), ("graph replay", g.replay)):
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 / calls * 1e6
    print(f"{name}: {us:.2f} us per call of {batch} x {n}-pt c32 transforms ({calls} calls per loop)")
