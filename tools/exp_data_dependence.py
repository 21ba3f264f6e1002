"""Does the streaming rate depend on the DATA?  The headline kernel (65536 x 4096 c32, out of place and in place) on uniform random values, on
values scaled to 1e-18, and on buffers full of NaN / zeros (what repeated in-place forward transforms of O(1) data turn into).
usage (GPU box): python3 tools/exp_data_dependence.py"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
n, batch = 4096, 65536
src = torch.empty((batch, n, 2), dtype=torch.float32, device="cuda")
dst = torch.empty_like(src)
a = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
for _ in range(300):
    a.mul_(1.0)
torch.cuda.synchronize(); del a


def t(fn, reps=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(reps):
            fn()
        e.record(stream); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, fill in (("uniform(-1, 1)", lambda x: x.uniform_(-1, 1)), ("uniform * 1e-18", lambda x: x.uniform_(-1, 1).mul_(1e-18)),
                   ("NaN", lambda x: x.fill_(float("nan"))), ("zeros", lambda x: x.zero_()), ("uniform(-1, 1) again", lambda x: x.uniform_(-1, 1))):
    with torch.cuda.stream(stream):
        fill(src); fill(dst)
    oop = t(lambda: f.fft_dev_oop(src.data_ptr(), dst.data_ptr(), n, batch))
    with torch.cuda.stream(stream):
        fill(dst)
        if "uniform(-1" in name:
            dst.mul_(1e-18)  # in place: keep 13 forward transforms finite
    inp = t(lambda: f.fft_dev(dst.data_ptr(), n, batch, False))
    print(f"{name:22s}: out of place {oop:.3f} ms ({2*8*n*batch/oop/1e6/8000:.3f})   in place {inp:.3f} ms ({2*8*n*batch/inp/1e6/8000:.3f})", flush=True)
