"""Per-call latency of the host-pointer trait path (one transform per call), the way `stft()` or a Rust caller's
`fft.fft(&mut frame)` loop would use it."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import time, numpy as np, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
for n in (64, 1024, 4096, 16384, 65536, 131072):
    x = (np.random.rand(n).astype(np.float32) + 1j * np.random.rand(n).astype(np.float32)).astype(np.complex64)
    for _ in range(20):
        f.fft(x.copy())
    reps = 300
    bufs = [x.copy() for _ in range(reps)]
    t0 = time.perf_counter()
    for b in bufs:
        f.fft(b)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:6d}: {dt*1e6:7.1f} us per fft() call on host memory")

# streaming helpers: one frame per call
sig = np.random.rand(48000).astype(np.float32)
win = kofft_amd.hann(1024)
st = kofft_amd.StftStream(sig, win, 256, f)
ist = kofft_amd.IstftStream(1024, 256, win, f)
frame = np.zeros(1024, np.complex64)
n_fr = 0
t0 = time.perf_counter()
while st.next_frame(frame):
    ist.push_frame(frame)
    n_fr += 1
dt = (time.perf_counter() - t0) / n_fr
print(f"StftStream.next_frame + IstftStream.push_frame (1024 / 256): {dt*1e6:7.1f} us per frame")
