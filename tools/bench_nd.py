"""fft2d / fft3d (ndfft.rs:74-155) on device memory, in place.  usage: bench_nd.py [f32|f64] [DxRxC ...]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
dt = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
tdt = torch.float64 if dt == np.float64 else torch.float32
es = 16 if dt == np.float64 else 8
f = kofft_amd.HipFftImpl(dt)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or \
    [(1, 4096, 4096), (1, 2048, 8192), (1, 8192, 2048), (1, 1024, 1024), (1, 16384, 1024), (256, 256, 256), (64, 512, 512), (512, 512, 64),
     (16, 1024, 1024), (1, 1000, 1000), (100, 100, 100), (1, 4096, 520)]
for d, r, c in shapes:
    x = torch.empty((d, r, c, 2), dtype=tdt, device="cuda").uniform_(-1e-3, 1e-3)
    ts = []
    with torch.cuda.stream(stream):
        for it in range(8):
            x.uniform_(-1e-3, 1e-3)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            f.fftnd_dev(x.data_ptr(), d, r, c)
            e.record(stream); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
    ms = float(np.median(ts[3:]))
    axes = (d > 1) + (r > 1) + (c > 1)
    print(f"{np.dtype(dt).name} {d:4d} x {r:5d} x {c:5d}: {ms:8.3f} ms  {d*r*c/ms/1e6:8.1f} GPoints/s  ({2*es*d*r*c/ms/1e6/8000:.3f} of the roofline on one read + one write; {axes} axes)", flush=True)
