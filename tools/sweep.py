#!/usr/bin/env python3
"""Size sweep of the device paths: kernel time and fraction of the 8 TB/s HBM roofline for every power of two.

Usage (GPU box):  python tools/sweep.py [--mb 512] [--out gpurun_out/sweep.json]
Each line: transform, dtype, n, batch, ms per launch, algorithmic GB/s, fraction of 8 TB/s.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import kofft_amd  # noqa: E402


def timeit(stream, launch, steps=20, warm=5):
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record(stream)
    for _ in range(steps):
        launch()
    e.record(stream)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=512, help="input size per launch in MiB")
    ap.add_argument("--out", default="gpurun_out/sweep.json")
    ap.add_argument("--kinds", default="c32,c64,rfft32,stft")
    ap.add_argument("--rfft-no-window", action="store_true", help="rfft32 without the Hann row window (default: with, like config 3)")
    ap.add_argument("--only-n", type=int, default=0, help="restrict every kind to this n")
    ap.add_argument("--max-n", type=int, default=0, help="skip sizes above this n")
    ap.add_argument("--min-n", type=int, default=0, help="skip sizes below this n")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(dev)
    rows = []
    kinds = args.kinds.split(",")
    with torch.cuda.stream(stream):
        # clock ramp
        a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
        for _ in range(200):
            a.mul_(1.0)
        torch.cuda.synchronize()
        del a
        for kind in kinds:
            if kind in ("c32", "c64"):
                dt = torch.float32 if kind == "c32" else torch.float64
                fft = kofft_amd.HipFftImpl(np.float32 if kind == "c32" else np.float64, device=0)
                fft.set_stream(stream.cuda_stream)
                esz = 8 if kind == "c32" else 16
                for L in range(1, 25):
                    n = 1 << L
                    if (args.only_n and n != args.only_n) or (args.max_n and n > args.max_n) or n < args.min_n:
                        continue
                    batch = max(1, (args.mb << 20) // (esz * n))
                    src = torch.empty((batch, n, 2), dtype=dt, device=dev).uniform_(-1, 1)
                    dst = torch.empty_like(src)
                    ms = timeit(stream, lambda: fft.fft_dev_oop(src.data_ptr(), dst.data_ptr(), n, batch))
                    gbs = 2 * esz * n * batch / ms / 1e6
                    rows.append(dict(kind=kind, n=n, batch=batch, ms=ms, gbs=gbs, frac=gbs / 8000))
                    print(f"{kind:7s} n={n:9d} batch={batch:9d} {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del src, dst
            elif kind == "ceil":
                # context: what plain torch kernels reach on this box (write-only, read-only, copy)
                for mb in (256, 1024):
                    x = torch.empty(mb << 18, dtype=torch.float32, device=dev).uniform_(-1, 1)
                    y = torch.empty_like(x)
                    for name, fn, nbytes in (("fill", lambda: y.fill_(1.0), x.numel() * 4), ("sum", lambda: x.sum(), x.numel() * 4),
                                             ("copy", lambda: y.copy_(x), x.numel() * 8)):
                        ms = timeit(stream, fn)
                        gbs = nbytes / ms / 1e6
                        rows.append(dict(kind=name, n=mb, batch=1, ms=ms, gbs=gbs, frac=gbs / 8000))
                        print(f"{name:7s} {mb:5d} MiB {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del x, y
            elif kind in ("rfft64", "irfft64"):
                fft = kofft_amd.HipFftImpl(np.float64, device=0)
                fft.set_stream(stream.cuda_stream)
                for L in range(2, 21):
                    n = 1 << L
                    if (args.only_n and n != args.only_n) or (args.max_n and n > args.max_n) or n < args.min_n:
                        continue
                    batch = max(1, (args.mb << 20) // (8 * n))
                    re = torch.empty((batch, n), dtype=torch.float64, device=dev).uniform_(-1, 1)
                    cx = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float64, device=dev).uniform_(-1, 1)
                    if kind == "rfft64":
                        ms = timeit(stream, lambda: fft.rfft_dev(re.data_ptr(), cx.data_ptr(), None, n, batch))
                    else:
                        ms = timeit(stream, lambda: fft.irfft_dev(cx.data_ptr(), re.data_ptr(), n, batch))
                    gbs = batch * (8 * n + 16 * (n // 2 + 1)) / ms / 1e6
                    rows.append(dict(kind=kind, n=n, batch=batch, ms=ms, gbs=gbs, frac=gbs / 8000))
                    print(f"{kind:7s} n={n:9d} batch={batch:9d} {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del re, cx
            elif kind == "rfft32":
                fft = kofft_amd.HipFftImpl(np.float32, device=0)
                fft.set_stream(stream.cuda_stream)
                for L in range(2, 22):
                    n = 1 << L
                    if (args.only_n and n != args.only_n) or (args.max_n and n > args.max_n) or n < args.min_n:
                        continue
                    batch = max(1, (args.mb << 20) // (4 * n))
                    src = torch.empty((batch, n), dtype=torch.float32, device=dev).uniform_(-1, 1)
                    dst = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float32, device=dev)
                    win = torch.from_numpy(kofft_amd.hann(n)).to(dev)
                    wptr = None if args.rfft_no_window else win.data_ptr()
                    ms = timeit(stream, lambda: fft.rfft_dev(src.data_ptr(), dst.data_ptr(), wptr, n, batch))
                    gbs = batch * (4 * n + 8 * (n // 2 + 1)) / ms / 1e6
                    rows.append(dict(kind=kind, n=n, batch=batch, ms=ms, gbs=gbs, frac=gbs / 8000))
                    print(f"{kind:7s} n={n:9d} batch={batch:9d} {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del src, dst
            elif kind == "irfft32":
                fft = kofft_amd.HipFftImpl(np.float32, device=0)
                fft.set_stream(stream.cuda_stream)
                for L in range(2, 22):
                    n = 1 << L
                    if (args.only_n and n != args.only_n) or (args.max_n and n > args.max_n) or n < args.min_n:
                        continue
                    batch = max(1, (args.mb << 20) // (4 * n))
                    src = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float32, device=dev).uniform_(-1, 1)
                    dst = torch.empty((batch, n), dtype=torch.float32, device=dev)
                    ms = timeit(stream, lambda: fft.irfft_dev(src.data_ptr(), dst.data_ptr(), n, batch))
                    gbs = batch * (4 * n + 8 * (n // 2 + 1)) / ms / 1e6
                    rows.append(dict(kind=kind, n=n, batch=batch, ms=ms, gbs=gbs, frac=gbs / 8000))
                    print(f"{kind:7s} n={n:9d} batch={batch:9d} {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del src, dst
            elif kind == "stft":
                fft = kofft_amd.HipFftImpl(np.float32, device=0)
                fft.set_stream(stream.cuda_stream)
                total = 28_800_000
                sig = torch.empty(total, dtype=torch.float32, device=dev).uniform_(-1, 1)
                for L in range(5, 15):
                    n = 1 << L
                    if (args.only_n and n != args.only_n) or (args.max_n and n > args.max_n) or n < args.min_n:
                        continue
                    hop = n // 4
                    frames = -(-total // hop)
                    win = torch.from_numpy(kofft_amd.hann(n)).to(dev)
                    dst = torch.empty((frames, n, 2), dtype=torch.float32, device=dev)
                    ms = timeit(stream, lambda: fft.stft_dev(sig.data_ptr(), total, win.data_ptr(), n, hop, dst.data_ptr(), 0, frames))
                    gbs = (4 * total + 8 * frames * n) / ms / 1e6
                    rows.append(dict(kind=kind, n=n, batch=frames, ms=ms, gbs=gbs, frac=gbs / 8000))
                    print(f"{kind:7s} n={n:9d} frames={frames:8d} {ms:8.4f} ms {gbs:8.1f} GB/s frac {gbs/8000:.3f}", flush=True)
                    del dst
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
