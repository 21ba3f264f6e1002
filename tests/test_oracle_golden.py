"""Oracle vs the committed golden vectors (tests/golden/, produced by tests/golden/gen_golden.py), and
the golden vectors vs an independent f64 DFT within the reference's measured twiddle-drift budget
(an exact FFT would sit at ~1e-7; kofft's recurrence tables put it at 1e-5..1e-4
for n >= 512).  CPU-only."""
import numpy as np
import pytest

from conftest import GOLDEN, bits_equal


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLDEN / "hotpath_golden.npz")
    cases = {}
    for key in z.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = z[key]
    return cases


def drift_budget(n: int, f64: bool) -> float:
    if f64:
        return 1e-12
    if n <= 256:
        return 1e-6
    if n <= 2048:
        return 5e-5
    return 2e-4


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.complex128) - b) / max(np.linalg.norm(b), 1e-300))


def test_complex_fft_bitwise_and_drift(oracle, gold):
    names = [k for k in gold if k.startswith(("c32_", "c64_"))]
    assert len(names) >= 18
    for name in names:
        g = gold[name]
        n = g["x"].size
        assert bits_equal(oracle.fft(g["x"]), g["y"]), name
        assert bits_equal(oracle.ifft(g["x"]), g["y_inv"]), name
        err = rel_l2(g["y"], g["dft64"])
        assert err <= drift_budget(n, name.startswith("c64")), (name, err)
        assert abs(err - float(g["rel_err"])) < 1e-12  # the stored figure is the measured one


def test_rfft_bitwise_and_drift(oracle, gold):
    for name in [k for k in gold if k.startswith(("rfft32_", "rfft64_"))]:
        g = gold[name]
        win = g.get("window")
        assert bits_equal(oracle.rfft(g["x"], win), g["y"]), name
        assert bits_equal(oracle.irfft(g["y"], g["x"].size), g["x_back"]), name
        assert rel_l2(g["y"], g["dft64"]) <= drift_budget(g["x"].size // 2, name.startswith("rfft64")), name
        if win is None:  # irfft(rfft(x)) == x to the reference's own tolerance (rfft.rs:904-906 / 933-935)
            tol = 1e-10 if name.startswith("rfft64") else 2e-4
            assert np.max(np.abs(g["x_back"] - g["x"])) < tol, name


def test_stft_bitwise_and_drift(oracle, gold):
    g = gold["stft32_4096_w1024_h256"]
    frames = oracle.stft(g["signal"], g["window"], int(g["hop"]), g["frames"].shape[0])
    assert bits_equal(frames, g["frames"])
    assert rel_l2(g["frames"], g["dft64"]) <= drift_budget(1024, False)
    # sharded evaluation (ko_stft_range) reproduces the same frames
    part = oracle.stft_range(g["signal"], g["window"], int(g["hop"]), 5, 7)
    assert bits_equal(part, g["frames"][5:12])
    # the 17th frame starts at 4096 == len: entirely zero-padded -> exactly zero spectrum
    assert np.all(g["frames"][16] == 0)


def test_tables_bitwise_and_drift(oracle, gold):
    assert bits_equal(oracle.get_twiddles(8), gold["twiddles32_8"]["table"])
    assert bits_equal(oracle.get_twiddles(1024), gold["twiddles32_1024"]["table"])
    assert bits_equal(oracle.get_twiddles(4096), gold["twiddles32_4096"]["table"])
    assert bits_equal(oracle.get_twiddles(1024, np.float64), gold["twiddles64_1024"]["table"])
    assert bits_equal(oracle.rfft_table(8), gold["rffttab32_8"]["table"])
    assert bits_equal(oracle.rfft_table(1024), gold["rffttab32_1024"]["table"])
    assert bits_equal(oracle.hann(1024), gold["hann32_1024"]["table"])
    # measured drift of the recurrence (5.9e-6 @1024, 3.2e-5 @4096, 3.5e-6 rfft m=1024)
    d1024 = np.max(np.abs(gold["twiddles32_1024"]["table"] - gold["twiddles32_1024"]["exact"]))
    d4096 = np.max(np.abs(gold["twiddles32_4096"]["table"] - gold["twiddles32_4096"]["exact"]))
    assert 1e-6 < d1024 < 2e-5 and 1e-5 < d4096 < 1e-4
    assert np.max(np.abs(gold["twiddles64_1024"]["table"] - gold["twiddles64_1024"]["exact"])) < 1e-13
    # the first entry is exactly (1, 0); the quarter-turn entry is NOT exactly (0, -1): no symmetry shortcuts
    t = gold["twiddles32_4096"]["table"]
    assert t[0] == 1 + 0j and t[1024] != 0 - 1j


def test_bin_export_matches_the_oracle(oracle):
    """tests/golden/bin (consumed by integration/rust/kofft-hip/tests/golden_pin.rs) is the oracle's output on the stored
    inputs: every case of the manifest is recomputed here, so the fixtures cannot go stale unnoticed."""
    bindir = GOLDEN / "bin"
    lines = [ln.split("\t") for ln in (bindir / "manifest.tsv").read_text().splitlines() if ln.strip()]
    assert len(lines) >= 45

    def rd(case, field, dtype):
        return np.frombuffer((bindir / f"{case}.{field}.bin").read_bytes(), dtype=dtype)

    kinds = set()
    for cols in lines:
        kind, case = cols[0], cols[1]
        p = dict(c.split("=") for c in cols[2:])
        kinds.add(kind)
        if kind == "fft":
            cdt = np.complex64 if p["dtype"] == "c32" else np.complex128
            x = rd(case, "x", cdt)
            assert x.size == int(p["n"])
            assert bits_equal(oracle.fft(x), rd(case, "y", cdt)), case
            assert bits_equal(oracle.ifft(x), rd(case, "y_inv", cdt)), case
        elif kind == "rfft":
            rdt, cdt = (np.float32, np.complex64) if p["dtype"] == "f32" else (np.float64, np.complex128)
            x = rd(case, "x", rdt)
            win = rd(case, "window", rdt) if p["window"] == "1" else None
            y = rd(case, "y", cdt)
            assert bits_equal(oracle.rfft(x, win), y), case
            assert bits_equal(oracle.irfft(y, x.size), rd(case, "x_back", rdt)), case
        elif kind == "stft":
            got = oracle.stft(rd(case, "signal", np.float32), rd(case, "window", np.float32), int(p["hop"]), int(p["frames"]))
            assert bits_equal(got.ravel(), rd(case, "frames", np.complex64)), case
        elif kind == "istft":
            fr = rd(case, "frames", np.complex64).reshape(int(p["frames"]), int(p["win"]))
            got = oracle.istft(fr, rd(case, "window", np.float32), int(p["hop"]), int(p["out_len"]))
            assert bits_equal(got, rd(case, "output", np.float32)), case
        elif kind == "mags":
            mags, mx = oracle.stft_magnitudes(rd(case, "samples", np.float32), int(p["win"]), int(p["hop"]))
            assert bits_equal(mags.ravel(), rd(case, "mags", np.float32)) and np.float32(mx) == rd(case, "max", np.float32)[0], case
        elif kind == "twiddles":
            dt = np.float32 if p["dtype"] == "f32" else np.float64
            assert bits_equal(oracle.get_twiddles(int(p["n"]), dt), rd(case, "table", np.complex64 if dt == np.float32 else np.complex128)), case
        elif kind == "rffttab":
            dt = np.float32 if p["dtype"] == "f32" else np.float64
            assert bits_equal(oracle.rfft_table(int(p["m"]), dt), rd(case, "table", np.complex64 if dt == np.float32 else np.complex128)), case
        elif kind == "hann":
            assert bits_equal(oracle.hann(int(p["len"])), rd(case, "table", np.float32)), case
        elif kind in ("fft_lcg", "fft_lcg_hash"):
            import importlib.util

            spec = importlib.util.spec_from_file_location("export_bin", GOLDEN / "export_bin.py")
            eb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(eb)
            cdt = np.complex64 if p["dtype"] == "c32" else np.complex128
            v = eb.lcg_values(int(p["seed"]), 2 * int(p["n"]))
            assert v.dtype == np.float32 and np.all(v >= -1) and np.all(v < 1)
            x = (v[0::2] + 1j * v[1::2]).astype(cdt)
            y = oracle.fft(x)
            if kind == "fft_lcg":
                assert bits_equal(y, rd(case, "y", cdt)), case
            else:
                assert f"{eb.fnv1a64(y.tobytes()):016x}" == p["fnv1a64"], case
        elif kind == "radix4":
            cdt = np.complex64 if p["dtype"] == "c32" else np.complex128
            assert bits_equal(oracle.fft_radix4(rd(case, "x", cdt)), rd(case, "y", cdt)), case
        else:
            raise AssertionError(kind)
    assert kinds == {"fft", "rfft", "stft", "istft", "mags", "twiddles", "rffttab", "hann", "fft_lcg", "fft_lcg_hash", "radix4"}


def test_lcg_stream_is_what_the_rust_harness_generates():
    """The three-line generator of export_bin.py / golden_pin.rs, restated with plain Python integers."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("export_bin", GOLDEN / "export_bin.py")
    eb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eb)
    state, want = 1401, []
    for _ in range(64):
        state = (state * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        want.append(np.float32(((state >> 40) & 0xFFFFFF)) / np.float32(8388608.0) - np.float32(1.0))
    assert bits_equal(eb.lcg_values(1401, 64), np.array(want, np.float32))
    assert eb.fnv1a64(bytes(range(16))) == 0x2B4A9E4A1B0F9B3D or True  # (value pinned below)
    h = 0xCBF29CE484222325
    for w in (int.from_bytes(bytes(range(8)), "little"), int.from_bytes(bytes(range(8, 16)), "little")):
        h = ((h ^ w) * 0x100000001B3) % (1 << 64)
    assert eb.fnv1a64(bytes(range(16))) == h
