"""Config 5's fast / slow modes, round 2 of the hunt (see exp_c64_ctx.py): what decides the mode?
  realloc   ONE context, its scratch released and re-allocated --reps times with dummy allocations in between (so the
            intermediate moves while the twiddle table stays): does the mode follow the intermediate?
  env:K=V,K=V   --contexts fresh contexts created under these environment knobs (e.g. KOFFT_HIP_BIG_CHUNK_MB=128,
            a -DKOFFT_EXP_API build with big_mid_nt forced to 0; the environment knob was removed in round 4), each timed on `batch` transforms per call
  dst       one context, the OUTPUT buffer re-allocated --reps times
Run as `rocprofv3 --kernel-trace ... -- python3 tools/exp_c64_modes.py ...` (python3 itself after `--`: no env / shebang hop), parse with `exp_c64_ctx.py --parse`.

usage (GPU box): python3 tools/exp_c64_modes.py realloc env: env:KOFFT_HIP_BIG_CHUNK_MB=128 dst [--batch 32]"""
import argparse
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from exp_c64_place import N, Lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="+")
    ap.add_argument("--lib", default="kofft_amd/lib/libkofft_hip.so")
    ap.add_argument("--contexts", type=int, default=6)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/exp4/cells.json")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    CH = args.batch
    chunk_bytes = CH * N * 16
    src = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    v = src.view(torch.float64)
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
    kernels_per_call = {}

    def run(lib, name, dstbuf, warm=2, reps=4, per_call=None):
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
        cells.append({"cell": name, "ms": ms, "dispatches": per_call * (warm + reps), "warm_dispatches": per_call * warm})
        print(f"{name:40s} {ms:8.3f} ms  dst {dstbuf.data_ptr():#x}", flush=True)

    for what in args.what:
        if what == "realloc":
            lib = Lib(args.lib)
            lib.set_stream(stream.cuda_stream)
            lib.lib.kofft_hip_release_scratch.argtypes = [C.c_void_p]
            keep = []
            for r in range(args.reps):
                run(lib, f"realloc {r}", dst, per_call=2)
                torch.cuda.synchronize(dev)
                assert lib.lib.kofft_hip_release_scratch(lib.ctx) == 0
                keep.append(torch.empty((300 + 37 * r) << 20, dtype=torch.uint8, device=dev))  # the next scratch lands elsewhere
            del keep
            torch.cuda.empty_cache()
        elif what == "dst":
            lib = Lib(args.lib)
            lib.set_stream(stream.cuda_stream)
            keep = []
            for r in range(args.reps):
                d2 = torch.empty(args.chunks * chunk_bytes, dtype=torch.uint8, device=dev)
                run(lib, f"dst {r}", d2, per_call=2)
                keep.append(d2)
            del keep
            torch.cuda.empty_cache()
        elif what.startswith("env:"):
            kv = [x for x in what[4:].split(",") if x]
            for x in kv:
                k, val = x.split("=")
                os.environ[k] = val
            chunk_mb = int(os.environ.get("KOFFT_HIP_BIG_CHUNK_MB", "512"))
            per_call = 2 * max(1, -(-(CH * N * 16) // (chunk_mb << 20)))
            libs = []
            for k in range(args.contexts):
                lib = Lib(args.lib)
                lib.set_stream(stream.cuda_stream)
                libs.append(lib)
                run(lib, f"{what[4:] or 'default'} ctx{k}", dst, per_call=per_call)
            torch.cuda.synchronize(dev)
            for lib in libs:
                lib.close()
            for x in kv:
                os.environ.pop(x.split("=")[0], None)
        else:
            raise SystemExit(f"unknown cell kind {what}")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
