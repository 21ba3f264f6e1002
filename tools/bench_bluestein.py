"""Non-power-of-two lengths (Bluestein arm) on device memory."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
for n, batch in ((1000, 65536), (1000, 1), (4095, 16384), (12345, 4096), (100003, 256), (1000003, 16)):
    x = torch.empty((batch, n, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1)
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
    print(f"n={n:8d} batch={batch:6d}: {ms:8.3f} ms  {batch*n/ms/1e6:8.1f} GPoints/s  ({16*batch*n/ms/1e6/8000:.3f} of the roofline on the algorithmic bytes)")
