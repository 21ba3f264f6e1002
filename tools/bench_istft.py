"""ISTFT (stft.rs:117-156) on device memory over window / hop shapes, fused kernel against the two-kernel route (KOFFT_HIP_ISTFT_FUSED=0 in a second
context): ~0.92 GB of spectra per call.  usage: bench_istft.py [win:hop ...]"""
import os, sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch, kofft_amd
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or \
    [(256, 64), (512, 128), (1024, 1024), (1024, 512), (1024, 256), (1024, 128), (2048, 512), (4096, 1024), (4096, 512)]
impls = {}
for name, val in (("fused", "1"), ("two kernels", "0")):
    os.environ["KOFFT_HIP_ISTFT_FUSED"] = val
    impls[name] = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream()
for f in impls.values():
    f.set_stream(stream.cuda_stream)
for win, hop in shapes:
    frames = int(os.environ.get("BENCH_ISTFT_FRAMES", 115_200_000 // win))
    out_len = (frames - 1) * hop + win
    spec0 = torch.empty((frames, win, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    spec = torch.empty_like(spec0)
    window = torch.from_numpy(kofft_amd.hann(win)).cuda()
    out = torch.zeros(out_len, dtype=torch.float32, device="cuda")
    scratch = torch.zeros(out_len, dtype=torch.float32, device="cuda")
    line = f"win {win:5d} hop {hop:5d} frames {frames:7d}:"
    for name, f in impls.items():
        ts = []
        with torch.cuda.stream(stream):
            for it in range(8):
                spec.copy_(spec0); out.zero_()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(stream)
                f.istft_dev(spec.data_ptr(), frames, window.data_ptr(), win, hop, out.data_ptr(), out_len, scratch.data_ptr())
                e.record(stream); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
        ms = float(np.median(ts[3:]))
        alg = 2 * frames * win * 8 + 3 * out_len * 4
        line += f"  {name} {ms:7.3f} ms ({alg/ms/1e6/8000:.3f})"
    print(line, flush=True)
