"""Round 6 debugging aid: where do the large-batch f64 routes differ from the small-batch ones?  (tools/kernel_coverage.sh found that the
parity tests had stopped reaching them.)  usage (GPU box): python3 tools/dbg_f64.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["KOFFT_HIP_HOST_PIPELINE"] = "0"
import kofft_amd  # noqa: E402
from conftest import rand_c, seeded  # noqa: E402


def ctx(dtype, **env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        return kofft_amd.HipFftImpl(dtype)
    finally:
        for k in env:
            del os.environ[k]


def report(tag, a, b):
    a2 = a.reshape(a.shape[0], -1)
    b2 = b.reshape(b.shape[0], -1)
    bad = a2.view(np.uint8).reshape(a2.shape[0], -1) != b2.view(np.uint8).reshape(b2.shape[0], -1)
    esz = a2.dtype.itemsize
    bad_el = bad.reshape(a2.shape[0], a2.shape[1], esz).any(axis=2)
    rows = np.nonzero(bad_el.any(axis=1))[0]
    print(f"{tag}: {bad_el.sum()} differing elements in {rows.size} of {a2.shape[0]} rows", flush=True)
    for r in rows[:6]:
        cols = np.nonzero(bad_el[r])[0]
        print(f"   row {r}: {cols.size} elements, first {cols[:12].tolist()} last {cols[-4:].tolist()}")
        c = cols[0]
        print(f"      got {a2[r, c]!r} want {b2[r, c]!r}")
    if rows.size:
        print("   rows:", rows[:40].tolist(), "..." if rows.size > 40 else "")


for n, batch in ((8192, 1024), (4096, 2561), (4096, 2048)):
    x = rand_c(seeded(1), (batch, n), np.complex128)
    outs = []
    for p in ("1", "0"):
        f = ctx(np.float64, KOFFT_HIP_PERSIST64=p)
        y = x.copy()
        f.fft_batch(y)
        z = y.copy()
        f.fft_batch(z, inverse=True)
        outs.append((y, z))
        f.close()
    report(f"c64 n={n} batch={batch} forward persist vs generic", outs[0][0], outs[1][0])
    report(f"c64 n={n} batch={batch} inverse persist vs generic", outs[0][1], outs[1][1])

n, batch = 16384, 520
x = rand_c(seeded(2), (batch, n), np.complex128)
outs = []
for p in ("1", "0"):
    f = ctx(np.float64, KOFFT_HIP_REGFILE=p)
    y = x.copy()
    f.fft_batch(y)
    z = y.copy()
    f.fft_batch(z, inverse=True)
    outs.append((y, z))
    f.close()
report("c64 2^14 forward regfile vs factors", outs[0][0], outs[1][0])
report("c64 2^14 inverse regfile vs factors", outs[0][1], outs[1][1])

n, batch = 32768, 520
rng = seeded(3)
xr = rng.uniform(-1, 1, (batch, n))
win = rng.uniform(0.1, 1, n)
outs = []
for p in ("1", "0"):
    f = ctx(np.float64, KOFFT_HIP_REGFILE=p)
    outs.append((f.rfft_batch(xr, win), f.rfft_batch(xr)))
    f.close()
report("rfft64 32768 windowed regfile vs composed", outs[0][0], outs[1][0])
report("rfft64 32768 plain regfile vs composed", outs[0][1], outs[1][1])
