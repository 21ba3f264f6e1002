"""Config 5's intermediate: does the placement probe (KOFFT_HIP_BIG_PROBE, round 6) pick a fast allocation?

Fresh contexts alternate probe ON (default K candidates) / OFF (KOFFT_HIP_BIG_PROBE=0: whatever hipMalloc hands out) in ONE
process on the SAME input / output buffers; every context transforms `batch` 2^20-point c64 transforms `steps` times (HIP events
per step) after a warm call that allocates (and probes).  Printed per context: the probe's candidates (first factor us, chunk us,
pick) and the measured ms per 1024 transforms; at the end both groups' medians.  The c32 two-factor path (2^20 c32) and the
Bluestein arm's large-m path ride the same buffer: --kind c32 / blue.

usage (GPU box): python3 tools/exp_c64_pick.py [--contexts 6] [--batch 128] [--kind c64]"""
import argparse
import os
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=6)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--kind", default="c64")
    ap.add_argument("--k", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch

    import kofft_amd as K

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    if args.kind == "c64":
        n, dt, tdt = 1 << 20, np.float64, torch.float64
    elif args.kind == "c32":
        n, dt, tdt = 1 << 20, np.float32, torch.float32
    else:  # Bluestein, m = 2^17 > one workgroup
        n, dt, tdt = 40000, np.float32, torch.float32
    batch = args.batch * (2 if args.kind == "c32" else 1) * (16 if args.kind == "blue" else 1)
    x = torch.empty((batch, n, 2), dtype=tdt, device=dev).uniform_(-1.0, 1.0)
    y = torch.empty_like(x)
    a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    for _ in range(300):
        a.mul_(1.0)
    torch.cuda.synchronize(dev)
    lib = K.load_library()
    fn = getattr(lib, "kofft_hip_fft_c64_dev_oop" if dt == np.float64 else "kofft_hip_fft_c32_dev_oop")
    import ctypes as C

    res = {"on": [], "off": []}
    for c in range(args.contexts):
        mode = "on" if c % 2 == 0 else "off"
        os.environ["KOFFT_HIP_BIG_PROBE"] = str(args.k) if mode == "on" else "0"
        f = K.HipFftImpl(dt)
        f.set_stream(stream.cuda_stream)
        call = lambda: fn(f._ctx, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, batch, 0)  # noqa: E731
        assert call() == 0
        torch.cuda.synchronize(dev)
        info = f.big_probe_info()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for s, e in ev:
            s.record(stream)
            assert call() == 0
            e.record(stream)
        torch.cuda.synchronize(dev)
        ms = statistics.median(s.elapsed_time(e) for s, e in ev)
        per1024 = ms * 1024 / args.batch
        res[mode].append(per1024)
        print(f"ctx {c} probe {mode:3s} {per1024:8.3f} ms per 1024 x 2^20 (median of {args.steps})  {info}", flush=True)
        f.close()
    for mode in ("on", "off"):
        v = res[mode]
        print(f"probe {mode:3s}: median {statistics.median(v):.3f}  min {min(v):.3f}  max {max(v):.3f} ms per 1024 transforms ({len(v)} contexts)")


if __name__ == "__main__":
    main()
