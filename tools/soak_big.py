#!/usr/bin/env python3
"""Randomised differential soak of the kernels that only run from ~1000 transforms up at n >= 4096 (the wave-split kernels at
8192 / 16384, c64 4096 / 8192, rfft / irfft 16384, rfft 32768) and of the group-wide rfft stores (n = 128 .. 512): random batch sizes around
and above the dispatch thresholds, random STFT hops with frames running past the end, against the oracle, bit for bit.
usage (GPU box, repo root): python3 tools/soak_big.py [rounds=6] [seed=1]"""
import os
import sys
from pathlib import Path

# one launch per call on the batch the case names (the host pipeline would cut it into eight: tests/conftest.py)
os.environ.setdefault("KOFFT_HIP_HOST_PIPELINE", "0")

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import kofft_amd  # noqa: E402
from conftest import bits_equal, rand_c  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
    bad = n_cases = 0

    def check(ok, what):
        nonlocal bad, n_cases
        n_cases += 1
        if not ok:
            bad += 1
            print("FAIL", what, flush=True)

    for r in range(rounds):
        for n, lo in ((8192, 1024), (16384, 1024)):
            batch = int(rng.integers(lo, lo + 700))
            x = rand_c(rng, (batch, n))
            y = x.copy()
            inv = bool(rng.random() < 0.5)
            f32.fft_batch(y, inverse=inv)
            check(bits_equal(y, oracle.ifft(x) if inv else oracle.fft(x)), ("c32", n, batch, inv))
            hop = int(rng.integers(n // 8, n // 2))
            frames = int(rng.integers(lo, lo + 300))
            length = frames * hop - int(rng.integers(0, hop))
            sig = rng.uniform(-1, 1, length).astype(np.float32)
            win = rng.uniform(0, 1, n).astype(np.float32)
            fr = -(-length // hop)
            check(bits_equal(f32.stft_into(sig, win, hop, fr), oracle.stft(sig, win, hop, fr)), ("stft", n, hop, fr))
            hop2 = n // 4
            mags, mx = f32.stft_magnitudes(sig, n, hop2)
            wm, wmx = oracle.stft_magnitudes(sig, n, hop2)
            check(bits_equal(mags, wm) and mx == wmx, ("stft_magnitudes", n, hop2))
        for n, lo in ((4096, 2048), (8192, 1024)):
            batch = int(rng.integers(lo, lo + 500))
            x = rand_c(rng, (batch, n), np.complex128)
            y = x.copy()
            inv = bool(rng.random() < 0.5)
            f64.fft_batch(y, inverse=inv)
            check(bits_equal(y, oracle.ifft(x) if inv else oracle.fft(x)), ("c64", n, batch, inv))
        batch = int(rng.integers(1024, 1500))
        x = rng.uniform(-1, 1, (batch, 16384)).astype(np.float32)
        win = rng.uniform(0, 1, 16384).astype(np.float32) if rng.random() < 0.5 else None
        check(bits_equal(f32.rfft_batch(x, win), oracle.rfft(x, win)), ("rfft", 16384, batch, win is not None))
        spec = rand_c(rng, (batch, 8193))
        check(bits_equal(f32.irfft_batch(spec, 16384), oracle.irfft(spec, 16384)), ("irfft", 16384, batch))
        batch = int(rng.integers(1024, 1200))
        x = rng.uniform(-1, 1, (batch, 32768)).astype(np.float32)
        win = rng.uniform(0, 1, 32768).astype(np.float32) if rng.random() < 0.5 else None
        check(bits_equal(f32.rfft_batch(x, win), oracle.rfft(x, win)), ("rfft", 32768, batch, win is not None))
        spec = rand_c(rng, (batch, 16385))
        check(bits_equal(f32.irfft_batch(spec, 32768), oracle.irfft(spec, 32768)), ("irfft", 32768, batch))
        for n in (128, 256, 512):
            batch = int(rng.integers(256 * 2048 // n, 256 * 2048 // n + 3000))  # above the persistent kernel's threshold
            x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
            win = rng.uniform(0, 1, n).astype(np.float32) if rng.random() < 0.5 else None
            check(bits_equal(f32.rfft_batch(x, win), oracle.rfft(x, win)), ("rfft", n, batch, win is not None))
        # round 4: the register-file-resident kernels (batches from two transforms per CU; the threshold itself, 512, and around it:
        # the two-factor route below it), rfft / irfft of 65536 reals through them, the block-interleaved c64 intermediate and the
        # two-pass ndfft axes
        for dt, n, cdt in (("c32", 32768, np.complex64), ("c64", 16384, np.complex128)):
            impl = f32 if dt == "c32" else f64
            for batch in (int(rng.integers(512, 900)), int(rng.choice([510, 511, 512, 513]))):
                x = rand_c(rng, (batch, n), cdt)
                y = x.copy()
                inv = bool(rng.random() < 0.5)
                impl.fft_batch(y, inverse=inv)
                check(bits_equal(y, oracle.ifft(x) if inv else oracle.fft(x)), (dt, n, batch, inv))
        batch = int(rng.integers(512, 700))
        x = rng.uniform(-1, 1, (batch, 65536)).astype(np.float32)
        win = rng.uniform(0, 1, 65536).astype(np.float32) if rng.random() < 0.6 else None  # (a window: RowWindowIO on the register-file kernel)
        got = f32.rfft_batch(x, win)
        check(bits_equal(got, oracle.rfft(x, win)), ("rfft", 65536, batch, win is not None))
        check(bits_equal(f32.irfft_batch(got, 65536), oracle.irfft(got, 65536)), ("irfft", 65536, batch))
        batch = int(rng.integers(512, 600))
        x = rng.uniform(-1, 1, (batch, 32768))
        win = rng.uniform(0, 1, 32768) if rng.random() < 0.6 else None
        got = f64.rfft_batch(x, win)
        check(bits_equal(got, oracle.rfft(x, win)), ("rfft64", 32768, batch, win is not None))
        check(bits_equal(f64.irfft_batch(got, 32768), oracle.irfft(got, 32768)), ("irfft64", 32768, batch))
        for log2n in (14, 15, 16, 18):
            n = 1 << log2n
            batch = int(rng.integers((8192 >> (log2n // 2)) + 1, (8192 >> (log2n // 2)) + 40))  # just above the persistent factors' threshold
            x = rand_c(rng, (batch, n), np.complex128)
            y = x.copy()
            f64.fft_batch(y)
            check(bits_equal(y, oracle.fft(x)), ("c64 blocked", n, batch))
        for rows, cols in ((4096, int(rng.choice([512, 1024]))), (8192, 256)):
            x = rand_c(rng, (rows, cols))
            want = oracle.fft(np.ascontiguousarray(oracle.fft(x).T)).T  # rows, then columns
            data = x.reshape(-1).copy()
            f32.fftnd(data, 1, rows, cols)
            check(bits_equal(data.reshape(rows, cols), np.ascontiguousarray(want)), ("fft2d two-pass", rows, cols))
        # round 4, later: c32 two-factor sizes with the last factor on row pairs, the persistent Bluestein kernel (every m, batches from its
        # threshold up) and the fused ISTFT (random frame counts / output lengths around what the frames reach, both hop ratios)
        for log2n in (16, 17, 19, 21):
            n = 1 << log2n
            batch = int(rng.integers((8192 >> (log2n // 2)) + 1, (8192 >> (log2n // 2)) + 24))
            x = rand_c(rng, (batch, n))
            y = x.copy()
            inv = bool(rng.random() < 0.5)
            f32.fft_batch(y, inverse=inv)
            pick = sorted({0, batch // 2, batch - 1})
            check(bits_equal(y[pick], oracle.ifft(x[pick]) if inv else oracle.fft(x[pick])), ("c32 row pairs", n, batch, inv))
        for m_log, base in ((5, 300_000), (6, 150_000), (7, 90_000), (8, 60_000), (9, 20_000), (10, 14_000), (11, 5_000), (12, 2_500), (13, 1_100)):
            lo, hi = (1 << (m_log - 2)) + 1, (1 << (m_log - 1))  # lengths whose m = (2n - 1).next_power_of_two() is 2^m_log
            n = int(rng.integers(lo, hi + 1))
            if n & (n - 1) == 0:
                n -= 1
            if n < 3:
                n = 3
            batch = base + int(rng.integers(0, 50))
            for impl, cdt, ok in ((f32, np.complex64, True), (f64, np.complex128, m_log <= 12)):
                if not ok:
                    continue
                x = rand_c(rng, (batch, n), cdt)
                y = x.copy()
                inv = bool(rng.random() < 0.5)
                impl.fft_batch(y, inverse=inv)
                pick = sorted({0, 1, batch // 3, batch - 2, batch - 1})
                check(bits_equal(y[pick], oracle.ifft(x[pick]) if inv else oracle.fft(x[pick])), ("bluestein persistent", cdt.__name__, n, batch, inv))
        for win_len, frames_lo in ((256, 33000), (512, 9000), (1024, 9000), (2048, 4100), (4096, 2100)):
            hop = win_len // int(rng.choice([1, 2, 4, 8]))
            nframes = frames_lo + int(rng.integers(0, 40))
            reach = (nframes - 1) * hop + win_len
            out_len = reach + int(rng.choice([0, 1, -1, 2000, -2000, hop, -hop]))
            spec = rand_c(rng, (nframes, win_len))
            window = rng.uniform(0.05, 1, win_len).astype(np.float32)
            want = oracle.istft(spec, window, hop, out_len)
            out = np.zeros(out_len, np.float32)
            scratch = np.zeros(out_len, np.float32)
            kofft_amd.istft(spec, window, hop, out, scratch, f32)
            check(bits_equal(out, want), ("istft fused", win_len, hop, nframes, out_len - reach))
        print(f"round {r}: {n_cases} cases, {bad} failures", flush=True)
    print("soak done, failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
