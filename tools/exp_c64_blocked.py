"""The block-interleaved c64 intermediate (KOFFT_HIP_BIG_BLOCKED) against the natural layout on IDENTICAL buffers: two contexts of one
-DKOFFT_EXP_API build (one created with KOFFT_HIP_BIG_BLOCKED=0), the same input, output and intermediate handed to both, over
--pairs freshly allocated (intermediate, output) pairs -- the placement of the intermediate decides a +-8 % mode, so
runs in two processes or two contexts with their own scratch cannot tell the layouts apart.
Run as `rocprofv3 --kernel-trace ... -- python3 tools/exp_c64_blocked.py ...` (python3 itself after `--`: no env / shebang hop), parse with `exp_c64_ctx.py --parse`.
usage (GPU box): python3 tools/exp_c64_blocked.py --lib kofft_amd/lib_exp/libkofft_hip.so [--log2n 20] [--batch 32]"""
import argparse
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import exp_c64_place as P  # noqa: E402


def _chk(rc):
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="kofft_amd/lib_exp/libkofft_hip.so")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--f32", action="store_true", help="Complex32 transforms (a -DKOFFT_BLOCKED_C32 build)")
    ap.add_argument("--out", default="gpurun_out/exp11/cells.json")
    args = ap.parse_args()
    import torch

    P.N = 1 << args.log2n
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    CH = args.batch
    ES = 8 if args.f32 else 16
    chunk_bytes = CH * P.N * ES
    chunks = 4
    src = torch.empty(chunks * chunk_bytes, dtype=torch.uint8, device=dev)
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
    libs = {}
    for name, env in (("blocked", "1"), ("natural", "0")):
        os.environ["KOFFT_HIP_BIG_BLOCKED"] = env
        lib = P.Lib(args.lib)
        if args.f32:
            lib.lib.kofft_hip_fft_c32_dev_oop.argtypes = lib.lib.kofft_hip_fft_c64_dev_oop.argtypes
            fn = lib.lib.kofft_hip_fft_c32_dev_oop
            lib.fft = (lambda L, f: (lambda s_, d_, b_: _chk(f(L.ctx, C.c_void_p(s_), C.c_void_p(d_), P.N, b_, 0))))(lib, fn)
        lib.set_stream(stream.cuda_stream)
        lib.lib.kofft_hip_exp_set_big_tmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        libs[name] = lib
    os.environ.pop("KOFFT_HIP_BIG_BLOCKED", None)
    cells = []

    def run(lib, name, mid, dst, warm=2, reps=5):
        assert lib.lib.kofft_hip_exp_set_big_tmp(lib.ctx, C.c_void_p(mid.data_ptr()), mid.numel()) == 0
        for i in range(warm):
            lib.fft(src.data_ptr() + (i % chunks) * chunk_bytes, dst.data_ptr() + (i % chunks) * chunk_bytes, CH)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            c = (warm + i) % chunks
            lib.fft(src.data_ptr() + c * chunk_bytes, dst.data_ptr() + c * chunk_bytes, CH)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        cells.append({"cell": name, "ms": ms, "dispatches": 2 * (warm + reps), "warm_dispatches": 2 * warm})
        print(f"{name:30s} {ms:8.4f} ms", flush=True)
        return ms

    keep = []
    for p in range(args.pairs):
        mid = torch.empty(chunk_bytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(chunks * chunk_bytes, dtype=torch.uint8, device=dev)
        keep.append((mid, dst))
        for rnd in range(2):
            for name, lib in libs.items():
                run(lib, f"pair{p} r{rnd} {name}", mid, dst)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
