import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import time, numpy as np, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
x = (np.random.rand(8192, 4096).astype(np.float32) + 1j*np.random.rand(8192, 4096).astype(np.float32)).astype(np.complex64)
f.fft_batch(x)  # warm
ts = []
for _ in range(5):
    t0 = time.perf_counter(); f.fft_batch(x); ts.append(time.perf_counter() - t0)
t = min(ts); pts = x.size
print(f"host-pointer C ABI (PCIe-inclusive, pageable numpy memory): {pts/t/1e9:.2f} GPoints/s, {2*x.nbytes/t/1e9:.1f} GB/s over PCIe, {t*1e3:.1f} ms for 8192 x 4096")
