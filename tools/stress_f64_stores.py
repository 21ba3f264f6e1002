"""Round 6: the f64 16-byte-store hazard (fft_device.hip.h: b128_store_guard).  Every f64 route whose stores follow arithmetic on the
stored registers, large batches, REPS times each, against the oracle; prints the number of differing elements per repetition.

usage (GPU box): [KOFFT_HIP_LIB=kofft_amd/lib_<variant>/libkofft_hip.so] python3 tools/stress_f64_stores.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("KOFFT_HIP_HOST_PIPELINE", "0")
import kofft_amd  # noqa: E402
from conftest import rand_c, seeded  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

oracle.build()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
f64 = kofft_amd.HipFftImpl(np.float64)
f32 = kofft_amd.HipFftImpl(np.float32)
total = 0


def ndiff(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.dtype.itemsize % 8 == 0 else int((a.view(np.uint32) != b.view(np.uint32)).sum())


def case(tag, run, want):
    global total
    counts = []
    for _ in range(REPS):
        counts.append(ndiff(run(), want))
    total += sum(counts)
    print(f"{tag:58s} differing 8-byte words per repetition: {counts}", flush=True)


def cfwd(impl, x):
    y = x.copy()
    impl.fft_batch(y)
    return y


def cinv(impl, x):
    y = x.copy()
    impl.fft_batch(y, inverse=True)
    return y


for n, batch in ((8192, 1024), (4096, 2561), (2048, 4100), (1024, 8200), (16384, 520), (65536, 300), (1 << 20, 20)):
    x = rand_c(seeded(n), (batch, n), np.complex128)
    want = oracle.fft_inplace_mt(x.copy())
    case(f"c64 forward n={n} x {batch}", lambda: cfwd(f64, x), want)
    winv = oracle.fft_inplace_mt(want.copy(), inverse=True)
    case(f"c64 inverse n={n} x {batch}", lambda: cinv(f64, want), winv)
for n, batch in ((32768, 520), (8192, 2100), (2048, 8200), (1 << 17, 130)):
    rng = seeded(n + 1)
    xr = rng.uniform(-1, 1, (batch, n))
    win = rng.uniform(0.1, 1, n)
    w1 = oracle.rfft_mt(xr, win)
    case(f"rfft64 windowed n={n} x {batch}", lambda: f64.rfft_batch(xr, win), w1)
    w2 = oracle.rfft_mt(xr)
    case(f"rfft64 n={n} x {batch}", lambda: f64.rfft_batch(xr), w2)
    w3 = oracle.irfft(w2, n)
    case(f"irfft64 n={n} x {batch}", lambda: f64.irfft_batch(w2, n), w3)
for n, batch in ((1000, 5000), (2000, 1200), (5000, 300), (12, 200000)):
    x = rand_c(seeded(n + 2), (batch, n), np.complex128)
    want = oracle.fft(x)
    case(f"bluestein c64 forward n={n} x {batch}", lambda: cfwd(f64, x), want)
    winv = oracle.ifft(want)
    case(f"bluestein c64 inverse n={n} x {batch}", lambda: cinv(f64, want), winv)
# control: the same shapes in f32 (8-byte stores; row pairs: packed 16-byte stores)
for n, batch in ((8192, 2048), (1 << 17, 130)):
    x = rand_c(seeded(n + 3), (batch, n))
    want = oracle.fft_inplace_mt(x.copy())
    case(f"c32 forward n={n} x {batch}", lambda: cfwd(f32, x), want)
    winv = oracle.fft_inplace_mt(want.copy(), inverse=True)
    case(f"c32 inverse n={n} x {batch}", lambda: cinv(f32, want), winv)
print("library:", os.environ.get("KOFFT_HIP_LIB", "default"), " total differing words:", total)
