"""STFT (stft.rs:76-105) on device memory for window lengths that are not powers of two (the reference calls fft.fft(frame) for ANY win_len:
Bluestein arm), next to a power-of-two neighbour.  usage: bench_stft_any.py [win:hop ...]"""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream(); f.set_stream(stream.cuda_stream)
total = 28_800_000
sig = torch.empty(total, dtype=torch.float32, device="cuda").uniform_(-1, 1)
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(400, 160), (512, 160), (1000, 250), (1024, 256), (1102, 441), (2000, 500), (3000, 750)]
for win, hop in shapes:
    frames = -(-total // hop)
    out = torch.empty((frames, win, 2), dtype=torch.float32, device="cuda")
    window = torch.from_numpy(kofft_amd.hann(win)).cuda()
    ts = []
    with torch.cuda.stream(stream):
        for it in range(8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            f.stft_dev(sig.data_ptr(), total, window.data_ptr(), win, hop, out.data_ptr(), 0, frames)
            e.record(stream); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
    ms = float(np.median(ts[3:]))
    alg = 4 * total + 8 * frames * win
    print(f"win {win:5d} hop {hop:5d} frames {frames:7d}: {ms:8.3f} ms  {frames*win/ms/1e6:8.1f} GPoints/s  ({alg/ms/1e6/8000:.3f} of the roofline on the algorithmic bytes)", flush=True)
    del out
