#!/usr/bin/env python3
"""Exports the golden cases as flat little-endian .bin files + a manifest, for a harness in ANOTHER language.

Consumer: integration/rust/kofft-hip/tests/golden_pin.rs, which runs the real kofft crate (ScalarFftImpl, RfftPlanner,
stft, istft, hann, stft_magnitudes) on these inputs and compares BYTES.  `cargo test --test golden_pin` on any box with a
Rust toolchain turns "parity unpinned" into a reference pin (or shows exactly which case the restatement gets wrong).

Inputs and expected outputs come from oracle/ (the C restatement), NOT from reference code; the complex / real / STFT
cases are the arrays of hotpath_golden.npz, plus cases the .npz does not hold: Bluestein lengths (including n = 15 in f64,
the smallest length whose chirp table depends on the sincos lowering -- see tests/test_oracle_second_opinion.py), istft
and stft_magnitudes.

Layout: tests/golden/bin/<case>.<field>.bin (raw little-endian f32 / f64, complex interleaved re, im) and manifest.tsv:
    kind <TAB> case <TAB> key=value ...
Run from the repo root:  python tests/golden/export_bin.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as ko  # noqa: E402

HERE = Path(__file__).resolve().parent
OUT = HERE / "bin"

LCG_A, LCG_C = np.uint64(6364136223846793005), np.uint64(1442695040888963407)


def lcg_values(seed: int, count: int) -> np.ndarray:
    """count f32 values in [-1, 1), each exact: state <- state * A + C (mod 2^64), value = ((state >> 40) & 0xFFFFFF) / 2^23 - 1.
    The same three lines exist in golden_pin.rs (`lcg_values`): large route cases store a seed instead of their input."""
    with np.errstate(over="ignore"):
        apow = np.multiply.accumulate(np.full(count, LCG_A, np.uint64))          # A^1 .. A^count
        geo = np.add.accumulate(np.concatenate(([np.uint64(1)], apow[:-1])))      # 1 + A + .. + A^(k-1)
        states = apow * np.uint64(seed) + LCG_C * geo
    return (((states >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.float32) / np.float32(8388608.0) - np.float32(1.0)).astype(np.float32)


def fnv1a64(data: bytes) -> int:
    """FNV-1a over 8-byte little-endian words (the output arrays are multiples of 8 bytes)."""
    words = np.frombuffer(data, dtype="<u8")
    h = 0xCBF29CE484222325
    for w in words.tolist():
        h = ((h ^ w) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def put(case, field, arr):
    a = np.ascontiguousarray(arr)
    assert a.dtype.byteorder in ("<", "=", "|")
    (OUT / f"{case}.{field}.bin").write_bytes(a.tobytes())


def main():
    OUT.mkdir(exist_ok=True)
    for f in OUT.glob("*.bin"):
        f.unlink()
    g = np.load(HERE / "hotpath_golden.npz")
    lines = []
    stems = sorted({k.split("/")[0] for k in g.files})
    for s in stems:
        if s.startswith(("c32_", "c64_")):
            put(s, "x", g[f"{s}/x"]); put(s, "y", g[f"{s}/y"]); put(s, "y_inv", g[f"{s}/y_inv"])
            lines.append(f"fft\t{s}\tdtype={s[:3]}\tn={g[f'{s}/x'].size}")
        elif s.startswith("rffttab"):
            put(s, "table", g[f"{s}/table"])
            lines.append(f"rffttab\t{s}\tdtype=f{s[7:9]}\tm={g[f'{s}/table'].size}")
        elif s.startswith("rfft"):
            has_win = f"{s}/window" in g.files
            put(s, "x", g[f"{s}/x"]); put(s, "y", g[f"{s}/y"]); put(s, "x_back", g[f"{s}/x_back"])
            if has_win:
                put(s, "window", g[f"{s}/window"])
            lines.append(f"rfft\t{s}\tdtype=f{s[4:6]}\tn={g[f'{s}/x'].size}\twindow={int(has_win)}")
        elif s.startswith("stft32"):
            put(s, "signal", g[f"{s}/signal"]); put(s, "window", g[f"{s}/window"]); put(s, "frames", g[f"{s}/frames"])
            lines.append(f"stft\t{s}\tlen={g[f'{s}/signal'].size}\twin={g[f'{s}/window'].size}\thop={int(g[f'{s}/hop'])}"
                         f"\tframes={g[f'{s}/frames'].shape[0]}")
        elif s.startswith("twiddles"):
            put(s, "table", g[f"{s}/table"])
            lines.append(f"twiddles\t{s}\tdtype=f{s[8:10]}\tn={2 * g[f'{s}/table'].size}")
        elif s.startswith("hann"):
            put(s, "table", g[f"{s}/table"])
            lines.append(f"hann\t{s}\tlen={g[f'{s}/table'].size}")
    # ---- cases beyond the .npz --------------------------------------------------------------------------------
    rng = np.random.default_rng(0x6B6F6666 + 99)
    for dt, cdt, tag in ((np.float32, np.complex64, "c32"), (np.float64, np.complex128, "c64")):
        for n in (3, 12, 15, 30, 1000):
            x = (rng.uniform(-1, 1, n).astype(dt) + 1j * rng.uniform(-1, 1, n).astype(dt)).astype(cdt)
            case = f"{tag}_{n}_bluestein"
            put(case, "x", x); put(case, "y", ko.fft(x)); put(case, "y_inv", ko.ifft(x))
            lines.append(f"fft\t{case}\tdtype={tag}\tn={n}")
    sig = (0.5 * np.sin(2 * np.pi * 440.0 * np.arange(6000) / 48000.0) + 0.25 * rng.uniform(-1, 1, 6000)).astype(np.float32)
    for win_len, hop in ((1024, 256), (256, 64), (64, 48)):
        win = ko.hann(win_len)
        frames = -(-sig.size // hop)
        spec = ko.stft(sig, win, hop, frames)
        case = f"stft32_6000_w{win_len}_h{hop}"
        put(case, "signal", sig); put(case, "window", win); put(case, "frames", spec)
        lines.append(f"stft\t{case}\tlen={sig.size}\twin={win_len}\thop={hop}\tframes={frames}")
        out_len = sig.size
        fr = spec.copy()
        out = np.zeros(out_len, np.float32)
        scratch = np.zeros(out_len, np.float32)
        import ctypes as C
        rc = ko.lib().ko_istft_f32(C.c_void_p(fr.ctypes.data), C.c_size_t(frames), C.c_void_p(win.ctypes.data), C.c_size_t(win_len),
                                   C.c_size_t(hop), C.c_void_p(out.ctypes.data), C.c_size_t(out_len), C.c_void_p(scratch.ctypes.data),
                                   C.c_size_t(out_len))
        assert rc == 0
        case = f"istft32_6000_w{win_len}_h{hop}"
        put(case, "frames", spec); put(case, "window", win); put(case, "output", out); put(case, "scratch", scratch)
        lines.append(f"istft\t{case}\twin={win_len}\thop={hop}\tframes={frames}\tout_len={out_len}")
        mags, mx = ko.stft_magnitudes(sig, win_len, hop)
        case = f"mags32_6000_w{win_len}_h{hop}"
        put(case, "samples", sig); put(case, "mags", mags); put(case, "max", np.array([mx], np.float32))
        lines.append(f"mags\t{case}\tlen={sig.size}\twin={win_len}\thop={hop}\tframes={mags.shape[0]}")
    # ---- one case per DISPATCH ROUTE of the device beyond the fused single-workgroup sizes: inputs are
    # LCG streams (a seed in the manifest), expected spectra stored in full where they are small, as a hash where not.
    #   c32 8192 -> wave-split kernel (already above: c32_8192_uniform); c32 16384 -> largest single-workgroup transform;
    #   c32 2^15 -> two factors; c64 2^16 -> two factors, f64; c32 2^22 -> three factors (hash only: 32 MiB of spectrum).
    for tag, log2n, seed, full in (("c32", 14, 1401, True), ("c32", 15, 1501, True), ("c64", 16, 1601, True), ("c32", 22, 2201, False)):
        n = 1 << log2n
        v = lcg_values(seed, 2 * n)
        x = (v[0::2] + 1j * v[1::2]).astype(np.complex64 if tag == "c32" else np.complex128)
        y = ko.fft(x)
        case = f"{tag}_2p{log2n}_lcg"
        if full:
            put(case, "y", y)
            lines.append(f"fft_lcg\t{case}\tdtype={tag}\tn={n}\tseed={seed}")
        else:
            lines.append(f"fft_lcg_hash\t{case}\tdtype={tag}\tn={n}\tseed={seed}\tfnv1a64={fnv1a64(y.tobytes()):016x}")
    # FftStrategy::Radix4, the reference's own arm (fft.rs:1455-1548; kofft_hip_fft_radix4_* reproduces it on request)
    for tag, n, seed in (("c32", 1024, 3101), ("c32", 16, 3102), ("c64", 256, 3103)):
        v = lcg_values(seed, 2 * n)
        x = (v[0::2] + 1j * v[1::2]).astype(np.complex64 if tag == "c32" else np.complex128)
        case = f"{tag}_{n}_radix4"
        put(case, "x", x); put(case, "y", ko.fft_radix4(x))
        lines.append(f"radix4\t{case}\tdtype={tag}\tn={n}")
    (OUT / "manifest.tsv").write_text("\n".join(lines) + "\n")
    total = sum(f.stat().st_size for f in OUT.glob("*.bin"))
    print(f"{len(lines)} cases, {total / 1024:.0f} KiB in {OUT}")


if __name__ == "__main__":
    main()
