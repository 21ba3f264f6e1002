"""Config 2 (65536 x 4096 c32) IN PLACE against OUT OF PLACE, one process, identical buffers, interleaved cells.

Every cell: restore `1e-18 x pristine` into the buffer it transforms (untimed), `warm` launches, then `steps` launches with a
HIP-event pair per launch on the launch stream.  Cells cycle  oop(src->dst), inplace(dst), inplace(src)  for every library build
given with --libs (name=path) over --rounds rounds, so box drift lands on every form alike.  Under
`rocprofv3 --kernel-trace [--pmc ...] -- python3 tools/exp_inplace_ab.py ...` the trace / counter files are folded back per cell
with --parse (the cells file records how many library dispatches each cell made).

usage (GPU box): python3 tools/exp_inplace_ab.py [--libs head=kofft_amd/lib/libkofft_hip.so ld0=kofft_amd/lib_ld0/libkofft_hip.so]
                 python3 tools/exp_inplace_ab.py --parse <kernel_trace.csv> [--counters <counter_collection.csv>] --cells <cells.json>"""
import argparse
import csv
import ctypes as C
import json
import statistics
import sys
from collections import defaultdict
from pathlib import Path

N, BATCH = 4096, 65536


class Lib:
    def __init__(self, path):
        self.lib = C.CDLL(str(path))
        self.lib.kofft_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        self.lib.kofft_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kofft_hip_fft_c32_dev_oop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        self.lib.kofft_hip_fft_c32_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        self.ctx = C.c_void_p()
        assert self.lib.kofft_hip_create(0, C.byref(self.ctx)) == 0

    def set_stream(self, s):
        assert self.lib.kofft_hip_set_stream(self.ctx, C.c_void_p(s)) == 0

    def oop(self, src, dst, batch=BATCH):
        assert self.lib.kofft_hip_fft_c32_dev_oop(self.ctx, C.c_void_p(src), C.c_void_p(dst), N, batch, 0) == 0

    def inplace(self, buf, batch=BATCH):
        assert self.lib.kofft_hip_fft_c32_dev(self.ctx, C.c_void_p(buf), N, batch, 0) == 0


def parse(trace, cells_path, counters):
    cells = json.loads(Path(cells_path).read_text())
    rows = [r for r in csv.DictReader(open(trace)) if "fft_persist_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cnt = defaultdict(dict)
    if counters:
        for r in csv.DictReader(open(counters)):
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    names = sorted({k for d in cnt.values() for k in d})
    agg = defaultdict(lambda: {"us": [], **{n: [] for n in names}})
    i = 0
    for c in cells:
        part = rows[i:i + c["dispatches"]][c["warm_dispatches"]:]
        i += c["dispatches"]
        a = agg[c["cell"]]
        a["us"] += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in part]
        for n in names:
            a[n] += [cnt[r["Dispatch_Id"]][n] for r in part if n in cnt.get(r["Dispatch_Id"], {})]
    print("cell".ljust(28), " kernel us (mean / median / min)   ", " ".join(n[-24:].rjust(24) for n in names))
    for cell, a in agg.items():
        us = a["us"]
        print(f"{cell:28s} {statistics.mean(us):8.1f} {statistics.median(us):8.1f} {min(us):8.1f}  n={len(us):4d} ",
              " ".join((f"{statistics.mean(a[n]):24.5g}" if a[n] else " " * 24) for n in names))
    print("dispatches used", i, "of", len(rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=["head=kofft_amd/lib/libkofft_hip.so"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1e-18)
    ap.add_argument("--forms", default="oop,inplace_dst,inplace_src")
    ap.add_argument("--out", default="gpurun_out/inplace_ab/cells.json")
    ap.add_argument("--parse", default=None)
    ap.add_argument("--counters", default=None)
    ap.add_argument("--cells", default="gpurun_out/inplace_ab/cells.json")
    args = ap.parse_args()
    if args.parse:
        return parse(args.parse, args.cells, args.counters)
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    g = torch.Generator(device=dev)
    g.manual_seed(0x6B6F666674 + 2)
    pristine = torch.empty((BATCH, N, 2), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=g).mul_(args.scale)
    src = torch.empty_like(pristine)
    dst = torch.empty_like(pristine)
    print(f"pristine {pristine.data_ptr():#x} src {src.data_ptr():#x} dst {dst.data_ptr():#x}", flush=True)
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
    cells, summary = [], defaultdict(list)
    forms = args.forms.split(",")
    for rnd in range(args.rounds):
        for name, lib in libs:
            for form in forms:
                src.copy_(pristine)
                dst.copy_(pristine)
                if form == "oop":
                    call = lambda: lib.oop(src.data_ptr(), dst.data_ptr())  # noqa: E731
                elif form == "inplace_dst":
                    call = lambda: lib.inplace(dst.data_ptr())  # noqa: E731
                else:
                    call = lambda: lib.inplace(src.data_ptr())  # noqa: E731
                for _ in range(args.warm):
                    call()
                torch.cuda.synchronize(dev)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for s, e in ev:
                    s.record(stream)
                    call()
                    e.record(stream)
                torch.cuda.synchronize(dev)
                ms = [s.elapsed_time(e) for s, e in ev]
                ok = bool(torch.isfinite((dst if form != "inplace_src" else src)[:64]).all().item())
                cell = f"{name}:{form}"
                cells.append({"cell": cell, "round": rnd, "ms_mean": statistics.mean(ms), "ms_min": min(ms), "finite": ok,
                              "dispatches": args.warm + args.steps, "warm_dispatches": args.warm})
                summary[cell].append(statistics.mean(ms))
                print(f"round {rnd} {cell:28s} mean {statistics.mean(ms):.4f} ms  min {min(ms):.4f}  finite={ok}", flush=True)
    print("---- per cell over rounds (mean of block means / median / best block) ----")
    for cell, v in summary.items():
        frac = 16 * N * BATCH / (statistics.median(v) * 1e-3) / 8e12
        print(f"{cell:28s} {statistics.mean(v):.4f} {statistics.median(v):.4f} {min(v):.4f} ms   frac(median) {frac:.3f}")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
