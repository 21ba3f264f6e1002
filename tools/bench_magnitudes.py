"""stft_magnitudes (visual/spectrogram.rs:52-76) on device memory over window sizes: config 4's stream (28.8 M samples), hop = window / 4.
usage: bench_magnitudes.py [win ...]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
total = 28_800_000
sig = torch.empty(total, dtype=torch.float32, device="cuda").uniform_(-1, 1)
for win in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]:
    hop = win // 4
    frames = -(-total // hop)
    mags = torch.empty((frames, win // 2), dtype=torch.float32, device="cuda")
    mx = torch.zeros(1, dtype=torch.float32, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(3):
            f.stft_magnitudes_dev(sig.data_ptr(), total, win, hop, mags.data_ptr(), frames, mx.data_ptr())
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            f.stft_magnitudes_dev(sig.data_ptr(), total, win, hop, mags.data_ptr(), frames, mx.data_ptr())
        e.record(stream); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    alg = 4 * total + 4 * frames * (win // 2)
    print(f"win {win:6d} hop {hop:5d} frames {frames:8d}: {ms:8.3f} ms  {frames*(win//2)/ms/1e6:8.1f} GMagnitudes/s  ({alg/ms/1e6/8000:.3f} of the roofline on the algorithmic bytes)")
    del mags
