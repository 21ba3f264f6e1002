#!/usr/bin/env python3
"""Bit-equality of two dispatch routes of the same call, on one box: the environment knobs named on the command line are
set for context B only (contexts read them when they are created).

usage: python3 tools/ab_route_equal.py KOFFT_HIP_SPLIT=0 [--kinds fft,rfft,rfftw,irfft,stft,stftmag,fft64,ifft64] [--n 16384] [--batch 1300]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kofft_amd import api  # noqa: E402


def run(kind, n, batch, seed):
    rng = np.random.default_rng(seed)
    f = api.HipFftImpl(np.float64 if kind.endswith("64") else np.float32)
    try:
        if kind in ("fft64", "ifft64"):
            x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
            f.fft_batch(x, inverse=kind == "ifft64")
            return x
        if kind == "rfft":
            x = rng.standard_normal((batch, n), dtype=np.float32)
            return f.rfft_batch(x)
        if kind == "rfftw":
            x = rng.standard_normal((batch, n), dtype=np.float32)
            return f.rfft_batch(x, rng.standard_normal(n, dtype=np.float32))
        if kind == "irfft":
            x = (rng.standard_normal((batch, n // 2 + 1)) + 1j * rng.standard_normal((batch, n // 2 + 1))).astype(np.complex64)
            return f.irfft_batch(x, n)
        if kind == "fft":
            x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
            f.fft_batch(x)
            return x
        hop = max(1, n // 4)
        sig = rng.standard_normal(hop * (batch - 1) + n - 3, dtype=np.float32)  # last frame runs past the end
        if kind == "stft":
            return f.stft_into(sig, api.hann(n), hop, batch, check_frames=False)
        if kind == "stftmag":
            mags, peak = f.stft_magnitudes(sig, n, hop)
            return np.concatenate([np.asarray(mags).ravel(), np.asarray([peak], np.float32)])
        raise SystemExit(f"unknown kind {kind}")
    finally:
        f.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knobs", nargs="+")
    ap.add_argument("--kinds", default="rfft,rfftw,stft,stftmag")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--batch", type=int, default=600011)
    a = ap.parse_args()
    bad = 0
    for kind in a.kinds.split(","):
        n = a.n // 2 if kind.startswith("stft") else a.n  # --n is the REAL length: rfft n and STFT window n/2 share a kernel size
        for k in a.knobs:
            os.environ.pop(k.split("=")[0], None)
        ref = run(kind, n, a.batch, 7)
        for k in a.knobs:
            key, val = k.split("=")
            os.environ[key] = val
        got = run(kind, n, a.batch, 7)
        same = ref.shape == got.shape and np.array_equal(ref.view(np.uint8), got.view(np.uint8))
        print(f"{kind:8s} n={n} batch={a.batch}: {'bit-equal' if same else 'DIFFERENT'}")
        bad += not same
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
