"""Config 5 (1024 x 2^20 c64): what is the LAST factor kernel sensitive to?  (192 <-> 221 us per 512 MiB chunk
between boxes / runs with the first factor constant at 183 us.)  One process, several builds of the library side by side
(ctypes, one context each), cells:
  A  fresh allocations of the 16 GiB input / output pair in different orders (placement sensitivity), default build
  B  the same buffers through every build given with --libs (name=path; e.g. round 2's build, store-policy / mapping variants)
  C  one 512 MiB chunk (32 transforms) per call at different positions of the output buffer
  D  one chunk per call with the output shifted by a few hundred bytes .. 64 KiB
Run it as `rocprofv3 --kernel-trace ... -- python3 tools/exp_c64_place.py ...` (python3 itself after `--`) and feed the trace to --parse to get the two factor kernels apart: every cell prints the
number of library kernel dispatches it made, the parser walks the trace in dispatch order.

usage (GPU box): python3 tools/exp_c64_place.py --libs head=kofft_amd/lib/libkofft_hip.so r02=kofft_amd/lib_r02/libkofft_hip.so ...
                 python3 tools/exp_c64_place.py --parse <kernel_trace.csv> --cells gpurun_out/exp_c64_place.json"""
import argparse
import csv
import ctypes as C
import json
import sys
from pathlib import Path

N = 1 << 20


class Lib:
    def __init__(self, path):
        self.lib = C.CDLL(str(path))
        self.lib.kofft_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        self.lib.kofft_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kofft_hip_fft_c64_dev_oop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        self.lib.kofft_hip_destroy.argtypes = [C.c_void_p]
        self.ctx = C.c_void_p()
        rc = self.lib.kofft_hip_create(0, C.byref(self.ctx))
        assert rc == 0, rc

    def set_stream(self, s):
        assert self.lib.kofft_hip_set_stream(self.ctx, C.c_void_p(s)) == 0

    def fft(self, src, dst, batch):
        rc = self.lib.kofft_hip_fft_c64_dev_oop(self.ctx, C.c_void_p(src), C.c_void_p(dst), N, batch, 0)
        assert rc == 0, rc

    def close(self):
        self.lib.kofft_hip_destroy(self.ctx)


def parse(trace, cells_path):
    cells = json.loads(Path(cells_path).read_text())
    rows = [r for r in csv.DictReader(open(trace)) if "kofft" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for c in cells:
        k = c["dispatches"]
        part = rows[i:i + k]
        i += k
        # skip the warm-up calls' share
        part = part[c["warm_dispatches"]:]
        first = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part if "fft_tile_persist" in r["Kernel_Name"]]
        last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part if "fft_rows_persist" in r["Kernel_Name"]]
        other = [r["Kernel_Name"][:60] for r in part if "fft_tile_persist" not in r["Kernel_Name"] and "fft_rows_persist" not in r["Kernel_Name"]]
        f = sum(first) / max(1, len(first)) / 1e3
        la = sum(last) / max(1, len(last)) / 1e3
        print(f"{c['cell']:34s} ev {c['ms']:7.3f} ms | first {f:6.1f} us (min {min(first, default=0) / 1e3:6.1f})  last {la:6.1f} us "
              f"(min {min(last, default=0) / 1e3:6.1f} max {max(last, default=0) / 1e3:6.1f}) n={len(last)} {set(other) or ''}")
    print("dispatches used", i, "of", len(rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=["head=kofft_amd/lib/libkofft_hip.so"])
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--out", default="gpurun_out/exp_c64_place.json")
    ap.add_argument("--parse", default=None)
    ap.add_argument("--cells", default="gpurun_out/exp_c64_place.json")
    args = ap.parse_args()
    if args.parse:
        return parse(args.parse, args.cells)
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    libs = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        libs[name] = path
    B = args.batch
    CH = 32  # transforms per 512 MiB chunk
    cells = []

    def timed(lib, src, dst, batch, name, warm=1, reps=3):
        for _ in range(warm):
            lib.fft(src, dst, batch)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            lib.fft(src, dst, batch)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        per = 2 * ((batch + CH - 1) // CH)
        cells.append({"cell": name, "ms": ms, "dispatches": per * (warm + reps), "warm_dispatches": per * warm})
        print(f"{name:34s} {ms:8.3f} ms per call (batch {batch})", flush=True)

    def alloc(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def fill(t):
        v = t.view(torch.float64)
        step = 1 << 27
        g = torch.Generator(device=dev)
        g.manual_seed(0x6B6F666674 + 5)
        for i in range(0, v.numel(), step):
            v[i:i + step].uniform_(-1.0, 1.0, generator=g)

    head = Lib(libs["head"])
    head.set_stream(stream.cuda_stream)
    size = B * N * 16
    slack = 1 << 21
    # ---- A: placement
    orders = ["src,dst", "pad3G,src,dst", "dst,src", "pad1G+,dst,pad,src"]
    src = dst = None
    for o in orders:
        src = dst = None
        torch.cuda.empty_cache()
        keep = []
        if o == "src,dst":
            src, dst = alloc(size), alloc(size + slack)
        elif o == "pad3G,src,dst":
            keep.append(alloc(3 << 30))
            src, dst = alloc(size), alloc(size + slack)
        elif o == "dst,src":
            dst, src = alloc(size + slack), alloc(size)
        else:
            keep.append(alloc((1 << 30) + (6 << 20)))
            dst = alloc(size + slack)
            keep.append(alloc(300 << 20))
            src = alloc(size)
        fill(src)
        timed(head, src.data_ptr(), dst.data_ptr(), B, f"A {o} s={src.data_ptr():#x} d={dst.data_ptr():#x}")
        del keep
    # ---- B: builds side by side on the last pair of buffers
    for name, path in libs.items():
        lib = head if name == "head" else Lib(path)
        lib.set_stream(stream.cuda_stream)
        timed(lib, src.data_ptr(), dst.data_ptr(), B, f"B lib={name}")
        if lib is not head:
            torch.cuda.synchronize(dev)
            lib.close()
    # ---- C: one chunk per call, by position
    nch = B // CH
    for c in sorted(set([0, nch // 6, nch // 3, nch // 2, (2 * nch) // 3, nch - 1])):
        off = c * CH * N * 16
        timed(head, src.data_ptr() + off, dst.data_ptr() + off, CH, f"C chunk {c}", warm=2, reps=6)
    # ---- D: output shifted
    for sh in (0, 128, 256, 1024, 4096, 8192, 16384 + 128, 65536 + 256):
        timed(head, src.data_ptr(), dst.data_ptr() + sh, CH, f"D shift {sh}", warm=2, reps=6)
    # ---- D2: input and scratch relation: source shifted
    for sh in (128, 4096):
        timed(head, src.data_ptr() + sh, dst.data_ptr(), CH, f"D src shift {sh}", warm=2, reps=6)
    Path(args.out).parent.mkdir(exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
