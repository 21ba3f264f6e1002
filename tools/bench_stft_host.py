"""stft() of BASELINE config 4 from HOST memory through the C ABI (PCIe-inclusive; informational, never `value`)."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import ctypes as C, time, numpy as np, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
lib = kofft_amd.load_library()
total, win_len, hop = 28_800_000, 1024, 256
rng = np.random.default_rng(1)
sig = rng.uniform(-1, 1, total).astype(np.float32)
win = kofft_amd.hann(win_len)
frames = -(-total // hop)
out = np.ones((frames, win_len), np.complex64)  # touched once: page faults of a fresh buffer are not the library's
def call():
    rc = lib.kofft_hip_stft_f32(f._ctx, C.c_void_p(sig.ctypes.data), C.c_size_t(total), C.c_void_p(win.ctypes.data), C.c_size_t(win_len),
                                C.c_size_t(hop), C.c_void_p(out.ctypes.data), C.c_size_t(frames))
    assert rc == 0, rc
for _ in range(2):
    call()
t = []
for _ in range(5):
    t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
ms = min(t) * 1e3
print(f"stft host {frames} x {win_len}: {ms:.1f} ms best of 5, {frames*win_len/ms/1e6:.2f} GPoints/s, {(sig.nbytes+out.nbytes)/ms/1e6:.1f} GB/s over PCIe")
