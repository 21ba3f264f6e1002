"""Config 5's last factor: fast / slow mode per ALLOCATION of the intermediate (exp_c64_shift.py: the mode follows the context,
i.e. the hipMalloc of the 512 MiB scratch, not the byte shift).  Creates --contexts contexts (each with its own scratch, all kept
alive so that every one gets different physical memory), times one 32-transform chunk per call through each, then runs a second
round of calls per context (the one a --pmc pass is read for).  Run as `rocprofv3 --kernel-trace [--pmc ...] -- python3 tools/exp_c64_ctx.py ...` (python3 itself after `--`);
`--parse <kernel_trace.csv> [--counters <counter_collection.csv>]` prints per-context kernel times and counter means.

usage (GPU box): python3 tools/exp_c64_ctx.py [--contexts 12] [--out gpurun_out/exp3/cells.json]"""
import argparse
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from exp_c64_place import N, Lib  # noqa: E402


def parse(trace, cells_path, counters):
    cells = json.loads(Path(cells_path).read_text())
    rows = [r for r in csv.DictReader(open(trace)) if "kofft" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cnt = defaultdict(dict)  # dispatch id -> counter -> value
    if counters:
        for r in csv.DictReader(open(counters)):
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    i = 0
    names = sorted({k for d in cnt.values() for k in d})
    print("cell".ljust(26), "first us  last us ", " ".join(n[-22:].rjust(22) for n in names))
    for c in cells:
        part = rows[i:i + c["dispatches"]][c["warm_dispatches"]:]
        i += c["dispatches"]
        for kind in ("fft_tile_persist", "fft_rows_persist"):
            sel = [r for r in part if kind in r["Kernel_Name"]]
            if not sel:
                continue
            us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / len(sel) / 1e3
            vals = []
            for n in names:
                xs = [cnt[r["Dispatch_Id"]].get(n) for r in sel if r["Dispatch_Id"] in cnt]
                xs = [x for x in xs if x is not None]
                vals.append(f"{sum(xs) / len(xs):22.4g}" if xs else " " * 22)
            print(f"{c['cell']:26s} {kind[4:8]:5s} {us:8.1f} ", " ".join(vals))
    print("dispatches used", i, "of", len(rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="kofft_amd/lib/libkofft_hip.so")
    ap.add_argument("--contexts", type=int, default=12)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/exp3/cells.json")
    ap.add_argument("--parse", default=None)
    ap.add_argument("--counters", default=None)
    ap.add_argument("--cells", default="gpurun_out/exp3/cells.json")
    args = ap.parse_args()
    if args.parse:
        return parse(args.parse, args.cells, args.counters)
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    CH = 32
    chunk_bytes = CH * N * 16
    src = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    v = src.view(torch.float64)
    g = torch.Generator(device=dev)
    g.manual_seed(0x6B6F666674 + 5)
    for i in range(0, v.numel(), 1 << 27):
        v[i:i + (1 << 27)].uniform_(-1.0, 1.0, generator=g)
    print(f"src {src.data_ptr():#x} dst {dst.data_ptr():#x}", flush=True)
    # clock ramp
    a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    for _ in range(300):
        a.mul_(1.0)
    torch.cuda.synchronize(dev)
    cells = []
    libs = []

    def run(lib, name, warm, reps):
        for i in range(warm):
            lib.fft(src.data_ptr() + (i % args.chunks) * chunk_bytes, dst.data_ptr() + (i % args.chunks) * chunk_bytes, CH)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            c = (warm + i) % args.chunks
            lib.fft(src.data_ptr() + c * chunk_bytes, dst.data_ptr() + c * chunk_bytes, CH)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        cells.append({"cell": name, "ms": ms, "dispatches": 2 * (warm + reps), "warm_dispatches": 2 * warm})
        print(f"{name:26s} {ms:8.3f} ms", flush=True)

    for k in range(args.contexts):
        lib = Lib(args.lib)
        lib.set_stream(stream.cuda_stream)
        libs.append(lib)
        run(lib, f"ctx{k} first round", 2, 4)
    for k, lib in enumerate(libs):
        run(lib, f"ctx{k} second round", 1, 4)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
