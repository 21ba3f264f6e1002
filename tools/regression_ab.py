"""One-box A/B of the judged libraries of earlier rounds against HEAD: is the five-round drift of configs
#2 / #3 / #4 in the driver's records boxes, or code?

Every library (name=path; built from the round's judge commit by the recipe in tools/README.md into kofft_amd/lib_rNN/) gets
one context in THIS process; all of them transform the SAME device buffers on the same stream.  A cell = (library, workload):
untimed restore of the in-place buffer where there is one, `warm` launches, then `steps` launches with a HIP-event pair each.
Cells are interleaved library by library inside a round in a fresh random order per (round, workload), `rounds` rounds (the
first `skip-rounds` left out of the summary), so box drift lands on every library alike.  Output:
per (workload, library) the median / mean / min over rounds of the per-round mean step time, and the ratio to HEAD.

usage (GPU box): python3 tools/regression_ab.py --libs r02=kofft_amd/lib_r02/libkofft_hip.so ... head=kofft_amd/lib/libkofft_hip.so
"""
import argparse
import ctypes as C
import json
import statistics
from collections import defaultdict
from pathlib import Path

SZ = C.c_size_t
VP = C.c_void_p


class Lib:
    def __init__(self, path):
        self.lib = lib = C.CDLL(str(path))
        lib.kofft_hip_create.argtypes = [C.c_int, C.POINTER(VP)]
        lib.kofft_hip_set_stream.argtypes = [VP, VP]
        lib.kofft_hip_fft_c32_dev_oop.argtypes = [VP, VP, VP, SZ, SZ, C.c_int]
        lib.kofft_hip_fft_c32_dev.argtypes = [VP, VP, SZ, SZ, C.c_int]
        lib.kofft_hip_rfft_f32_dev.argtypes = [VP, VP, VP, VP, SZ, SZ]
        lib.kofft_hip_stft_f32_dev.argtypes = [VP, VP, SZ, VP, SZ, SZ, VP, SZ, SZ]
        self.ctx = VP()
        assert lib.kofft_hip_create(0, C.byref(self.ctx)) == 0

    def set_stream(self, s):
        assert self.lib.kofft_hip_set_stream(self.ctx, VP(s)) == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--skip-rounds", type=int, default=1, help="leading rounds left out of the summary (box warm-up)")
    ap.add_argument("--workloads", default="c2_oop,c2_inplace,c3,c4")
    ap.add_argument("--out", default="gpurun_out/regression_ab/cells.json")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    g = torch.Generator(device=dev)
    g.manual_seed(0x6B6F666674 + 2)
    wl = args.workloads.split(",")
    bufs = {}
    if "c2_oop" in wl or "c2_inplace" in wl:
        n, b = 4096, 65536
        pristine = torch.empty((b, n, 2), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g).mul_(1e-18)
        bufs["c2"] = (pristine, torch.empty_like(pristine), torch.empty_like(pristine))
    if "c3" in wl:
        n3, b3 = 2048, 1 << 20
        x3 = torch.empty((b3, n3), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g)
        y3 = torch.empty((b3, n3 // 2 + 1, 2), dtype=torch.float32, device=dev)
        w3 = (0.5 - 0.5 * torch.cos(2.0 * torch.pi * torch.arange(n3, device=dev, dtype=torch.float32) / n3)).contiguous()
        bufs["c3"] = (x3, y3, w3)
    if "c4" in wl:
        ln, win, hop = 28_800_000, 1024, 256
        frames = -(-ln // hop)
        sig = torch.empty(ln, dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g)
        w4 = (0.5 - 0.5 * torch.cos(2.0 * torch.pi * torch.arange(win, device=dev, dtype=torch.float32) / win)).contiguous()
        y4 = torch.empty((frames, win, 2), dtype=torch.float32, device=dev)
        bufs["c4"] = (sig, w4, y4, ln, win, hop, frames)
    a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    for _ in range(300):  # clock ramp
        a.mul_(1.0)
    torch.cuda.synchronize(dev)
    del a
    libs = []
    for spec in args.libs:
        name, path = spec.split("=", 1)
        lib = Lib(path)
        lib.set_stream(stream.cuda_stream)
        libs.append((name, lib))

    def make_call(lib, w):
        L, ctx = lib.lib, lib.ctx
        if w == "c2_oop":
            p, s, d = bufs["c2"]
            return (lambda: s.copy_(p)), (lambda: L.kofft_hip_fft_c32_dev_oop(ctx, VP(s.data_ptr()), VP(d.data_ptr()), 4096, 65536, 0)), 16 * 4096 * 65536
        if w == "c2_inplace":
            p, s, d = bufs["c2"]
            return (lambda: d.copy_(p)), (lambda: L.kofft_hip_fft_c32_dev(ctx, VP(d.data_ptr()), 4096, 65536, 0)), 16 * 4096 * 65536
        if w == "c3":
            x, y, wdw = bufs["c3"]
            return (lambda: None), (lambda: L.kofft_hip_rfft_f32_dev(ctx, VP(x.data_ptr()), VP(y.data_ptr()), VP(wdw.data_ptr()), 2048, 1 << 20)), (8192 + 8200) * (1 << 20)
        sig, w4, y4, ln, win, hop, frames = bufs["c4"]
        return (lambda: None), (lambda: L.kofft_hip_stft_f32_dev(ctx, VP(sig.data_ptr()), ln, VP(w4.data_ptr()), win, hop, VP(y4.data_ptr()), 0, frames)), 4 * ln + 8 * frames * win

    import random

    rng = random.Random(args.seed)
    cells, summary, nbytes = [], defaultdict(list), {}
    for rnd in range(args.rounds):
        for w in wl:
            order = list(libs)
            rng.shuffle(order)  # a fresh order per (round, workload): no library always follows the same neighbour
            for name, lib in order:
                restore, call, by = make_call(lib, w)
                nbytes[w] = by
                restore()
                for _ in range(args.warm):
                    assert call() == 0
                torch.cuda.synchronize(dev)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for s, e in ev:
                    s.record(stream)
                    assert call() == 0
                    e.record(stream)
                torch.cuda.synchronize(dev)
                ms = [s.elapsed_time(e) for s, e in ev]
                cells.append({"workload": w, "lib": name, "round": rnd, "ms_mean": statistics.mean(ms), "ms_median": statistics.median(ms), "ms_min": min(ms)})
                if rnd >= args.skip_rounds:
                    summary[(w, name)].append(statistics.mean(ms))
                print(f"round {rnd} {w:11s} {name:6s} mean {statistics.mean(ms):.4f} ms  median {statistics.median(ms):.4f}  min {min(ms):.4f}", flush=True)
    print("---- per (workload, library): median / mean / best of the per-round mean step times; frac of 8 TB/s at the median; vs head ----")
    for w in wl:
        head = statistics.median(summary[(w, libs[-1][0])])
        for name, _ in libs:
            v = summary[(w, name)]
            med = statistics.median(v)
            print(f"{w:11s} {name:6s} {med:.4f} {statistics.mean(v):.4f} {min(v):.4f} ms   frac {nbytes[w] / (med * 1e-3) / 8e12:.3f}   {100 * (med / head - 1):+.2f} % vs {libs[-1][0]}")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
