"""Non-power-of-two lengths (Bluestein arm) on device memory.  usage: bench_bluestein.py [f32|f64] [n:batch ...]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
dt = np.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else np.float32
tdt = torch.float64 if dt == np.float64 else torch.float32
es = 16 if dt == np.float64 else 8
f = kofft_amd.HipFftImpl(dt)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or \
    [(12, 1 << 22), (30, 1 << 21), (60, 1 << 20), (100, 1 << 19), (250, 1 << 18), (500, 1 << 17), (1000, 65536), (1000, 1), (2000, 32768),
     (4095, 16384), (12345, 4096), (100003, 256), (1000003, 16)]
for n, batch in cases:
    x = torch.empty((batch, n, 2), dtype=tdt, device="cuda").uniform_(-1, 1)
    y = torch.empty_like(x)
    with torch.cuda.stream(stream):
        for _ in range(3):
            f.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, batch)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            f.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, batch)
        e.record(stream); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{np.dtype(dt).name} n={n:8d} batch={batch:8d}: {ms:8.3f} ms  {batch*n/ms/1e6:8.1f} GPoints/s  ({2*es*batch*n/ms/1e6/8000:.3f} of the roofline on the algorithmic bytes)")
