"""Config 5: the last factor's fast / slow mode follows the INTERMEDIATE's placement (exp_c64_ab.py: fast mid + slow dst = fast).
How does one get a fast intermediate?  Times the two factor kernels (32 transforms per call) with the intermediate taken from
  torch      separate 512 MiB torch allocations (hipMalloc underneath)
  contig     hipExtMallocWithFlags(hipDeviceMallocContiguous) of 512 MiB
  window     512 MiB windows (2 MiB-aligned offsets) of ONE large allocation (plain and contiguous)
Library built with -DKOFFT_EXP_API.  Run as `rocprofv3 --kernel-trace ... -- python3 tools/exp_c64_alloc.py ...` (python3 itself after `--`: no env / shebang hop), parse with `exp_c64_ctx.py --parse`."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from exp_c64_place import N, Lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="kofft_amd/lib/libkofft_hip.so")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--big-gib", type=int, default=6)
    ap.add_argument("--out", default="gpurun_out/exp7/cells.json")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    CH = 32
    chunk_bytes = CH * N * 16
    chunks = 4
    src = torch.empty(chunks * chunk_bytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(chunks * chunk_bytes, dtype=torch.uint8, device=dev)
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
    lib = Lib(args.lib)
    lib.set_stream(stream.cuda_stream)
    L = lib.lib
    L.kofft_hip_exp_set_big_tmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.kofft_hip_exp_malloc.argtypes = [C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]
    L.kofft_hip_exp_free.argtypes = [C.c_void_p]
    cells = []

    def run(name, mid_ptr, warm=2, reps=4):
        assert L.kofft_hip_exp_set_big_tmp(lib.ctx, C.c_void_p(mid_ptr), chunk_bytes) == 0
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
        print(f"{name:40s} {ms:8.3f} ms  mid {mid_ptr:#x}", flush=True)

    def ext_malloc(nbytes, flags):
        p = C.c_void_p()
        rc = L.kofft_hip_exp_malloc(nbytes, flags, C.byref(p))
        return p.value if rc == 0 else None

    keep = []
    for t in range(args.n):
        m = torch.empty(chunk_bytes, dtype=torch.uint8, device=dev)
        keep.append(m)
        run(f"torch {t}", m.data_ptr())
    ptrs = []
    for t in range(args.n):
        p = ext_malloc(chunk_bytes, 0x4)
        if p is None:
            print("contiguous allocation refused", flush=True)
            break
        ptrs.append(p)
        run(f"contig {t}", p)
    big_bytes = args.big_gib << 30
    big = torch.empty(big_bytes, dtype=torch.uint8, device=dev)
    for t in range(args.n):
        off = t * ((big_bytes - chunk_bytes) // max(1, args.n - 1)) & ~((2 << 20) - 1)
        run(f"window torch +{off >> 20}M", big.data_ptr() + off)
    pbig = ext_malloc(big_bytes, 0x4)
    if pbig is not None:
        for t in range(args.n):
            off = t * ((big_bytes - chunk_bytes) // max(1, args.n - 1)) & ~((2 << 20) - 1)
            run(f"window contig +{off >> 20}M", pbig + off)
    else:
        print("big contiguous allocation refused", flush=True)
    # the same window again, shifted by 2 MiB steps (is the mode a property of the physical region?)
    for sh in (2, 4, 8, 64, 256):
        run(f"window torch +{sh}M", big.data_ptr() + (sh << 20))
    torch.cuda.synchronize(dev)
    L.kofft_hip_exp_set_big_tmp(lib.ctx, None, 0)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(cells, indent=1) + "\n")


if __name__ == "__main__":
    main()
