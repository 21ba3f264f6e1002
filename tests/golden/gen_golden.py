#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/), NOT from reference code.

The reference (okian/kofft, Rust) cannot run in the build image and stores no golden vectors, so
these fixtures record the C restatement's outputs on the reference tests' own deterministic
inputs (generators cited per case) plus seeded uniform data.  Each case also stores the f64 DFT
of the same input and the measured relative L2 error, so that (a) the oracle and the HIP path are
both checked against committed bytes, and (b) anyone with a Rust toolchain can confirm the
restatement against kofft itself bit for bit.

Run from the repo root:  python tests/golden/gen_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as ko  # noqa: E402

OUT = Path(__file__).resolve().parent


def rel_l2(a, b):
    a = np.asarray(a, np.complex128)
    b = np.asarray(b, np.complex128)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d else float(np.linalg.norm(a - b))


def gen_inputs_c(n, kind, dtype):
    real = np.float32 if dtype == np.complex64 else np.float64
    i = np.arange(n, dtype=real)
    if kind == "ramp_quarter":      # tests/stockham_parity.rs:7-9   (i, -0.25 i)
        return (i - 1j * (i * real(0.25))).astype(dtype)
    if kind == "ramp_half":         # tests/pow2.rs:23-25            (i, -0.5 i)
        return (i - 1j * (i * real(0.5))).astype(dtype)
    if kind == "sin_cos":           # tests/stockham_large.rs:7-9    (sin i, cos i)
        return (np.sin(i) + 1j * np.cos(i)).astype(dtype)
    if kind == "ramp_double":       # tests/parallel_stockham.rs:8-10 (i, 2i)
        return (i + 1j * (2 * i)).astype(dtype)
    if kind == "ramp_real":         # tests/split64.rs:6             (i, 0)
        return i.astype(dtype)
    if kind == "basic_usage":       # examples/basic_usage.rs:232-234 sin(0.1 i)
        return np.sin(real(0.1) * i).astype(dtype)
    if kind == "uniform":
        rng = np.random.default_rng(0x6B6F6666 + n)
        return (rng.uniform(-1, 1, n).astype(real) + 1j * rng.uniform(-1, 1, n).astype(real)).astype(dtype)
    raise ValueError(kind)


def main():
    cases = {}
    # ---- complex FFT, f32
    c32 = [(2, "ramp_half"), (4, "ramp_half"), (8, "sin_cos"), (16, "sin_cos"), (32, "ramp_quarter"),
           (64, "ramp_quarter"), (256, "ramp_quarter"), (512, "sin_cos"), (1024, "sin_cos"), (1024, "basic_usage"),
           (2048, "uniform"), (4096, "ramp_double"), (4096, "uniform"), (8192, "uniform")]
    for n, kind in c32:
        x = gen_inputs_c(n, kind, np.complex64)
        y = ko.fft(x)
        ref = np.fft.fft(x.astype(np.complex128))
        cases[f"c32_{n}_{kind}"] = dict(x=x, y=y, y_inv=ko.ifft(x), dft64=ref, rel_err=rel_l2(y, ref))
    # ---- complex FFT, f64
    for n, kind in [(32, "ramp_real"), (64, "ramp_quarter"), (1024, "uniform"), (4096, "uniform")]:
        x = gen_inputs_c(n, kind, np.complex128)
        y = ko.fft(x)
        ref = np.fft.fft(x)
        cases[f"c64_{n}_{kind}"] = dict(x=x, y=y, y_inv=ko.ifft(x), dft64=ref, rel_err=rel_l2(y, ref))
    # ---- real FFT
    for n, kind in [(8, "count"), (32, "sin"), (2048, "uniform"), (2048, "uniform_hann")]:
        if kind == "count":      # lib.rs:433 [1..8]
            x = np.arange(1, n + 1, dtype=np.float32)
        elif kind == "sin":      # tests/rfft_arch_parity.rs:13  sin(i)
            x = np.sin(np.arange(n, dtype=np.float32)).astype(np.float32)
        else:
            x = np.random.default_rng(0x72666674 + n).uniform(-1, 1, n).astype(np.float32)
        win = ko.hann(n) if kind.endswith("hann") else None
        y = ko.rfft(x, win)
        xin = x if win is None else (x * win).astype(np.float32)
        ref = np.fft.rfft(xin.astype(np.float64))
        d = dict(x=x, y=y, dft64=ref, rel_err=rel_l2(y, ref), x_back=ko.irfft(y, n))
        if win is not None:
            d["window"] = win
        cases[f"rfft32_{n}_{kind}"] = d
    x = np.random.default_rng(0x72666674).uniform(-1, 1, 256)
    y = ko.rfft(x)
    cases["rfft64_256_uniform"] = dict(x=x, y=y, dft64=np.fft.rfft(x), rel_err=rel_l2(y, np.fft.rfft(x)),
                                       x_back=ko.irfft(y, 256))
    # ---- STFT: 4096-sample ramp+sine, hann(1024), hop 256 -> 16 required frames (+1 extra, all zero-padded tail)
    t = np.arange(4096, dtype=np.float32)
    sig = (t / np.float32(4096) + np.float32(0.5) * np.sin(np.float32(0.05) * t)).astype(np.float32)
    win = ko.hann(1024)
    frames = ko.stft(sig, win, 256, 17)
    framed = np.zeros((17, 1024), np.float64)
    for f in range(17):
        seg = sig[f * 256: f * 256 + 1024]
        framed[f, : seg.size] = (seg * win[: seg.size]).astype(np.float32)
    ref = np.fft.fft(framed, axis=1)
    cases["stft32_4096_w1024_h256"] = dict(signal=sig, window=win, hop=np.int64(256), frames=frames, dft64=ref,
                                           rel_err=rel_l2(frames, ref))
    # ---- tables
    for n in (8, 1024, 4096):
        cases[f"twiddles32_{n}"] = dict(table=ko.get_twiddles(n, np.float32),
                                        exact=np.exp(-2j * np.pi * np.arange(n // 2) / n))
    cases["twiddles64_1024"] = dict(table=ko.get_twiddles(1024, np.float64),
                                    exact=np.exp(-2j * np.pi * np.arange(512) / 1024))
    for m in (8, 1024):
        cases[f"rffttab32_{m}"] = dict(table=ko.rfft_table(m, np.float32), exact=np.exp(-1j * np.pi * np.arange(m) / m))
    cases["hann32_1024"] = dict(table=ko.hann(1024))

    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = np.asarray(v)
    np.savez_compressed(OUT / "hotpath_golden.npz", **flat)
    with open(OUT / "MANIFEST.txt", "w") as fh:
        fh.write("# case  rel_L2_error_vs_f64_DFT   (generated by tests/golden/gen_golden.py from oracle/)\n")
        for name, d in cases.items():
            if "rel_err" in d:
                fh.write(f"{name}  {d['rel_err']:.3e}\n")
    print(f"wrote {len(cases)} cases, {(OUT / 'hotpath_golden.npz').stat().st_size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
