"""Two independent transcriptions of the reference must agree bit for bit.

oracle/ (C, scalar loops, fmaf, gcc -O2 -ffp-contract=off) against tests/ref_restatement.py (numpy whole-stage array
operations, exact-rational FMA, libm through ctypes), both written from the .rs sources.  The reference itself cannot run
in this image, so this is not a reference pin -- it removes transcription slips, compiler-flag effects and libm
assumptions from the list of things that could be wrong, and leaves only a shared misreading of the source."""
import numpy as np
import pytest

import ref_restatement as rr
from conftest import GOLDEN, bits_equal, rand_c, seeded


# ---- tables --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_twiddle_tables_agree(oracle, dtype):
    # fft.rs:391-405 -- the FMA recurrence, including the use of the OLD w_re in the second update
    for n in (2, 4, 8, 32, 64, 256, 1024, 2048, 4096, 8192):
        assert bits_equal(rr.get_twiddles(n, dtype), oracle.get_twiddles(n, dtype)), n


def test_twiddle_table_65536_f32_agrees(oracle):
    assert bits_equal(rr.get_twiddles(65536, np.float32), oracle.get_twiddles(65536, np.float32))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rfft_tables_and_hann_agree(oracle, dtype):
    for m in (1, 2, 3, 6, 16, 500, 1024, 2048):
        assert bits_equal(rr.rfft_table(m, dtype), oracle.rfft_table(m, dtype)), m
    if dtype == np.float32:
        for length in (1, 2, 7, 256, 1024, 2048):
            assert bits_equal(rr.hann(length), oracle.hann(length)), length


def _same(a, b):
    return a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_sincos_lowering_what_it_can_and_cannot_change():
    """The one assumption no source reading settles: Rust's sin_cos() is `(self.sin(), self.cos())` in std, and LLVM merges
    the two libm calls into sincosf / sincos on *-linux-gnu.  Both restatements take the merged call.  Measured here, on
    this glibc:
      * f32: merged == separate on every angle the path uses (planner tables, rfft tables, Bluestein chirps): every f32
        result (BASELINE configs #1-#4) is independent of the lowering;
      * f64: merged == separate on every POWER-OF-TWO table angle up to 2^26 (config #5's tables are independent of it),
        but not on every Bluestein chirp angle -- n = 15 is the smallest length where one entry differs in its last bit.
        That is the part of `parity` that only a run of the reference can settle."""
    pi32, pi64 = np.float32(3.14159274), np.float64(np.pi)
    for lg in range(1, 27):
        n = np.float32(1 << lg)
        for a in ((-np.float32(2.0)) * pi32 / n, -pi32 / n):
            assert _same(rr.sin_cos(a, np.float32), rr.sin_cos_separate(a, np.float32)), a
        for a in ((-np.float64(2.0)) * pi64 / np.float64(n), -pi64 / np.float64(n)):
            assert _same(rr.sin_cos(a, np.float64), rr.sin_cos_separate(a, np.float64)), a
    for m in (3, 6, 12, 15, 500, 1000, 1500):  # rfft tables of non-power-of-two half lengths
        assert _same(rr.sin_cos(-pi32 / np.float32(m), np.float32), rr.sin_cos_separate(-pi32 / np.float32(m), np.float32))
    differing64 = []
    for n in list(range(3, 64)) + [100, 257, 1000]:
        for i in range(n):
            a32 = pi32 * np.float32(i * i) / np.float32(n)
            for a in (a32, -a32):
                assert _same(rr.sin_cos(a, np.float32), rr.sin_cos_separate(a, np.float32)), (n, i)
            a64 = pi64 * np.float64(np.float32(i * i)) / np.float64(np.float32(n))
            if not _same(rr.sin_cos(a64, np.float64), rr.sin_cos_separate(a64, np.float64)):
                differing64.append((n, i))
    assert differing64 and differing64[0] == (15, 2), differing64[:3]
    # odd / even symmetry holds exactly, so expi(-a) and the conjugate of expi(a) are the same bits
    for n, i in differing64[:8] + [(12, 5), (30, 7)]:
        a64 = pi64 * np.float64(np.float32(i * i)) / np.float64(np.float32(n))
        s, c = rr.sin_cos(a64, np.float64)
        s2, c2 = rr.sin_cos(-a64, np.float64)
        assert (-s).tobytes() == s2.tobytes() and c.tobytes() == c2.tobytes()


def test_exact_rational_fma_is_a_single_rounding():
    # a case where round(round(a*b) + c) != round(a*b + c): the restatement must take the second
    a, b, c = np.float32(1 + 2 ** -12), np.float32(1 + 2 ** -12), np.float32(-(1 + 2 ** -11))
    assert float(rr.mul_add(a, b, c, np.float32)) == 2.0 ** -24
    assert float(np.float32(a * b) + c) == 0.0
    # ties go to even
    assert rr.mul_add(np.float32(1.0), np.float32(1.0), np.float32(2 ** -24), np.float32) == np.float32(1.0)
    assert rr.mul_add(np.float32(1.0), np.float32(1 + 2 ** -23), np.float32(2 ** -24), np.float32) == np.float32(1 + 2 ** -22)


# ---- transforms: every golden case, then random ones ------------------------------------------------------------------
def _golden():
    return np.load(GOLDEN / "hotpath_golden.npz")


def test_every_golden_case_agrees():
    """tests/golden/hotpath_golden.npz was generated by the C oracle: the second transcription reproduces every array."""
    g = _golden()
    stems = sorted({k.split("/")[0] for k in g.files})
    seen = 0
    for s in stems:
        if s.startswith("c32_") or s.startswith("c64_"):
            assert bits_equal(rr.fft(g[f"{s}/x"]), g[f"{s}/y"]), s
            assert bits_equal(rr.fft(g[f"{s}/x"], inverse=True), g[f"{s}/y_inv"]), s  # y_inv = ifft(x)
        elif s.startswith("rfft") and not s.startswith("rffttab"):
            win = g[f"{s}/window"] if f"{s}/window" in g.files else None
            assert bits_equal(rr.rfft(g[f"{s}/x"], win), g[f"{s}/y"]), s
            assert bits_equal(rr.irfft(g[f"{s}/y"], g[f"{s}/x"].size), g[f"{s}/x_back"]), s
        elif s.startswith("stft32"):
            hop = int(g[f"{s}/hop"])
            assert bits_equal(rr.stft(g[f"{s}/signal"], g[f"{s}/window"], hop, g[f"{s}/frames"].shape[0]), g[f"{s}/frames"]), s
        elif s.startswith("twiddles"):
            assert bits_equal(rr.get_twiddles(2 * g[f"{s}/table"].size, g[f"{s}/table"].real.dtype), g[f"{s}/table"]), s
        elif s.startswith("rffttab"):
            assert bits_equal(rr.rfft_table(g[f"{s}/table"].size, g[f"{s}/table"].real.dtype), g[f"{s}/table"]), s
        elif s.startswith("hann"):
            assert bits_equal(rr.hann(g[f"{s}/table"].size), g[f"{s}/table"]), s
        else:
            raise AssertionError(f"golden case {s} not covered")
        seen += 1
    assert seen == len(stems) >= 30


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_random_complex_transforms_agree(oracle, dtype):
    rng = seeded(0xA11CE)
    sizes = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
    cases = 0
    for rep in range(100 if dtype == np.complex64 else 40):
        n = sizes[int(rng.integers(len(sizes)))]
        x = rand_c(rng, (int(rng.integers(1, 4)), n), dtype)
        assert bits_equal(rr.fft(x), oracle.fft(x)), (n, rep)
        assert bits_equal(rr.fft(x, inverse=True), oracle.ifft(x)), (n, rep)
        cases += 2
    assert cases >= 80


def test_bluestein_arm_agrees(oracle):
    rng = seeded(77)
    for n in (3, 5, 6, 7, 12, 30, 100, 257, 1000):
        x = rand_c(rng, (2, n))
        assert bits_equal(rr.fft(x), oracle.fft(x)), n
        assert bits_equal(rr.fft(x, inverse=True), oracle.ifft(x)), n
    x = rand_c(rng, (1, 12), np.complex128)
    assert bits_equal(rr.fft(x), oracle.fft(x))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_random_real_transforms_agree(oracle, dtype):
    rng = seeded(0xBEEF)
    for rep in range(60 if dtype == np.float32 else 24):
        n = int(rng.choice([2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 12, 30, 200]))
        x = rng.uniform(-1, 1, (2, n)).astype(dtype)
        win = rng.uniform(0, 1, n).astype(dtype) if rep % 3 == 0 else None
        spec = oracle.rfft(x, win)
        assert bits_equal(rr.rfft(x, win), spec), (n, rep)
        assert bits_equal(rr.irfft(spec, n), oracle.irfft(spec, n)), (n, rep)


def test_config3_shape_agrees(oracle):
    # BASELINE config #3's transform: 2048-point rfft with the Hann window fused
    x = seeded(3).uniform(-1, 1, (3, 2048)).astype(np.float32)
    assert bits_equal(rr.rfft(x, rr.hann(2048)), oracle.rfft(x, oracle.hann(2048)))


def test_stft_and_istft_agree(oracle):
    rng = seeded(0x57F7)
    for win_len, hop, length in ((1024, 256, 5000), (256, 64, 3000), (64, 16, 500), (32, 32, 100), (128, 200, 1000), (12, 5, 80)):
        sig = rng.uniform(-1, 1, length).astype(np.float32)
        win = rr.hann(win_len)
        frames = -(-length // hop) + 1
        spec = oracle.stft(sig, win, hop, frames)
        assert bits_equal(rr.stft(sig, win, hop, frames), spec), (win_len, hop)
        out_len = length + 7
        got, _ = rr.istft(spec, win, hop, out_len)
        assert bits_equal(got, oracle.istft(spec, win, hop, out_len)), (win_len, hop)


def test_stft_magnitudes_agree(oracle):
    sig = seeded(9).uniform(-1, 1, 4000).astype(np.float32)
    for win_len, hop in ((256, 64), (1024, 256), (64, 48)):
        mags, mx = rr.stft_magnitudes(sig, win_len, hop)
        omags, omx = oracle.stft_magnitudes(sig, win_len, hop)
        assert bits_equal(mags, omags) and np.float32(mx).tobytes() == np.float32(omx).tobytes()
