"""Real transforms beyond the fused sizes on device memory.  usage: bench_rfft_big.py [n:batch ...]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1 << 20, 128), (1 << 17, 1024), (1 << 20, 1)]
for n, batch in cases:
    x = torch.empty((batch, n), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    y = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float32, device="cuda")
    w = torch.from_numpy(kofft_amd.hann(n)).cuda()
    for label, win in (("plain", None), ("window", w.data_ptr())):
        with torch.cuda.stream(stream):
            for _ in range(3):
                f.rfft_dev(x.data_ptr(), y.data_ptr(), win, n, batch)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            for _ in range(10):
                f.rfft_dev(x.data_ptr(), y.data_ptr(), win, n, batch)
            e.record(stream); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"rfft f32 n={n:8d} batch={batch:6d} {label:7s}: {ms:8.3f} ms  {batch*n/ms/1e6:8.1f} GSamples/s  ({8.0*batch*n/ms/1e6/8000:.3f} of the roofline)")
