"""2-D / 3-D transforms on device memory: ms per call and fraction of the roofline per axis pass."""
import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import ctypes as C
import numpy as np, torch, kofft_amd
from kofft_amd import _lib
lib = _lib.load()
f = kofft_amd.HipFftImpl(np.float32)
stream = torch.cuda.Stream()
f.set_stream(stream.cuda_stream)
def run(depth, rows, cols):
    x = torch.empty((depth, rows, cols, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    call = lambda: lib.kofft_hip_fftnd_c32_dev(f._ctx, C.c_void_p(x.data_ptr()), depth, rows, cols, 0)
    with torch.cuda.stream(stream):
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(10):
            call()
        e.record(stream)
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    axes = 2 if depth == 1 else 3
    gbs = axes * 2 * x.numel() * 4 / ms / 1e6
    print(f"{depth} x {rows} x {cols}: {ms:8.3f} ms, {gbs:7.0f} GB/s over {axes} axis passes = {gbs/8000:.3f} of the roofline per pass")
import os
shapes = ((1, 4096, 4096), (1, 1024, 1024), (1, 8192, 2048), (1, 512, 16384), (256, 256, 256), (64, 512, 512), (1, 1024, 16384), (1, 2048, 8192), (1024, 128, 128))
for shape in shapes:
    run(*shape)
