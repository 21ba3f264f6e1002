"""Config 5's last factor: variants of the kernel on IDENTICAL buffers (the mode depends on the physical placement of the
intermediate and the output, so variants can only be compared on one (src, mid, dst) triple).  Libraries built with
-DKOFFT_EXP_API take the intermediate from this script (kofft_hip_exp_set_big_tmp).  The base library is first run on
--tries freshly allocated (mid, dst) pairs to find a slow and a fast placement; then every variant runs on both.
Run as `rocprofv3 --kernel-trace ... -- python3 tools/exp_c64_ab.py ...` (python3 itself after `--`: no env / shebang hop), parse with `exp_c64_ctx.py --parse`.

usage (GPU box): python3 tools/exp_c64_ab.py base=kofft_amd/lib/libkofft_hip.so st16=kofft_amd/lib_st16/libkofft_hip.so ..."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from exp_c64_place import N, Lib  # noqa: E402


def _chk(rc):
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--tries", type=int, default=10)
    ap.add_argument("--f32", action="store_true", help="Complex32 transforms")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/exp6/cells.json")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    import exp_c64_place as P
    P.N = 1 << args.log2n
    NN = P.N
    CH = args.batch
    chunk_bytes = CH * NN * (8 if args.f32 else 16)
    src = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    v = src.view(torch.float32 if args.f32 else torch.float64)
    g = torch.Generator(device=dev)
    g.manual_seed(0x6B6F666674 + 5)
    for i in range(0, v.numel(), 1 << 27):
        v[i:i + (1 << 27)].uniform_(-1.0, 1.0, generator=g)
    a = torch.empty(1 << 26, dtype=torch.float32, device=dev)
    for _ in range(300):
        a.mul_(1.0)
    torch.cuda.synchronize(dev)
    del a
    cells = []
    libs = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        lib = Lib(path)
        if args.f32:
            lib.lib.kofft_hip_fft_c32_dev_oop.argtypes = lib.lib.kofft_hip_fft_c64_dev_oop.argtypes
            lib.fft = (lambda L: (lambda s_, d_, b_: _chk(L.lib.kofft_hip_fft_c32_dev_oop(L.ctx, C.c_void_p(s_), C.c_void_p(d_), NN, b_, 0))))(lib)
        lib.set_stream(stream.cuda_stream)
        lib.lib.kofft_hip_exp_set_big_tmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        libs[name] = lib

    def run(lib, name, mid, dstbuf, warm=2, reps=6):
        assert lib.lib.kofft_hip_exp_set_big_tmp(lib.ctx, C.c_void_p(mid.data_ptr()), mid.numel()) == 0
        for i in range(warm):
            lib.fft(src.data_ptr() + (i % args.chunks) * chunk_bytes, dstbuf.data_ptr() + (i % args.chunks) * chunk_bytes, CH)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            c = (warm + i) % args.chunks
            lib.fft(src.data_ptr() + c * chunk_bytes, dstbuf.data_ptr() + c * chunk_bytes, CH)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        cells.append({"cell": name, "ms": ms, "dispatches": 2 * (warm + reps), "warm_dispatches": 2 * warm})
        print(f"{name:40s} {ms:8.3f} ms  mid {mid.data_ptr():#x} dst {dstbuf.data_ptr():#x}", flush=True)
        return ms

    base = libs[next(iter(libs))]
    pairs = []
    for t in range(args.tries):
        mid = torch.empty(chunk_bytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
        ms = run(base, f"probe {t}", mid, dst, reps=4)
        pairs.append((ms, mid, dst))
    pairs.sort(key=lambda p: p[0])
    fast, slow = pairs[0], pairs[-1]
    print(f"fastest probe {fast[0]:.3f} ms, slowest {slow[0]:.3f} ms", flush=True)
    for tag, (_, mid, dst) in (("fast", fast), ("slow", slow)):
        for rnd in range(2):
            for name, lib in libs.items():
                run(lib, f"{tag} r{rnd} {name}", mid, dst)
    # mixed pairs: is it the intermediate or the output?
    run(base, "mix fast-mid slow-dst", fast[1], slow[2])
    run(base, "mix slow-mid fast-dst", slow[1], fast[2])
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
