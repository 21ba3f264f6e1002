#!/usr/bin/env python3
"""Where does the large-n device path differ from the oracle?  (development tool)  usage: debug_big.py log2n batch [c32|c64]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import kofft_amd  # noqa: E402
from oracle import pyoracle as ko  # noqa: E402

log2n, batch = int(sys.argv[1]), int(sys.argv[2])
dt = np.complex128 if (len(sys.argv) > 3 and sys.argv[3] == "c64") else np.complex64
n = 1 << log2n
rng = np.random.default_rng(5)
real = np.float64 if dt == np.complex128 else np.float32
x = (rng.uniform(-1, 1, (batch, n)).astype(real) + 1j * rng.uniform(-1, 1, (batch, n)).astype(real)).astype(dt)
fft = kofft_amd.HipFftImpl(real)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for rep in range(reps):
    # fresh data every repetition: stale cache lines from the previous repetition would show
    x = (rng.uniform(-1, 1, (batch, n)).astype(real) + 1j * rng.uniform(-1, 1, (batch, n)).astype(real)).astype(dt)
    want = ko.fft(x[: min(batch, 4)])
    y = x.copy()
    fft.fft_batch(y)
    bad_total = 0
    for b in range(min(batch, 4)):
        bad = np.nonzero(y[b].view(np.uint8).reshape(n, -1) != want[b].view(np.uint8).reshape(n, -1))[0]
        bad = np.unique(bad)
        bad_total += bad.size
        if bad.size:
            print(f"rep {rep} transform {b}: {bad.size} bad elements; first {bad[:24].tolist()}")
            print("   bits of first bad index:", [format(int(i), f'0{log2n}b') for i in bad[:6]])
            print("   got", y[b][bad[:3]], "want", want[b][bad[:3]], "input there", x[b][bad[:3]])
    print(f"rep {rep}: {bad_total} bad elements in the first {min(batch, 4)} transforms")
