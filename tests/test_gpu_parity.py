"""Parity of the HIP path with the CPU oracle, through the C ABI (libkofft_hip.so), on a real MI355X.

The bar (task section 3 / BASELINE north_star): <= 1e-5 relative f32 vs ScalarFftImpl's arithmetic.
Because every butterfly performs the reference's un-fused operations on the reference's tables, the
tests assert the stronger property: BIT-EXACT equality with the oracle (and with the committed
golden vectors).  The tolerance constant is kept, written out, as the contractual fallback bar.
"""
import numpy as np
import pytest

from conftest import GOLDEN, bits_equal, rand_c, seeded

pytestmark = pytest.mark.gpu

REL_TOL_F32 = 1e-5   # north_star: "within 1e-5 relative f32"
REL_TOL_F64 = 1e-12


def rel_err(a, b):
    a = np.asarray(a).astype(np.complex128)
    b = np.asarray(b).astype(np.complex128)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def assert_parity(got, want, what, tol):
    if bits_equal(got, want):
        return
    err = rel_err(got, want)
    nbad = int(np.sum(np.asarray(got).view(np.uint8) != np.asarray(want).view(np.uint8)))
    raise AssertionError(f"{what}: not bit-exact ({nbad} differing bytes), rel L2 err {err:.3e} (contract bar {tol:g})")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLDEN / "hotpath_golden.npz")
    cases = {}
    for key in z.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = z[key]
    return cases


# ---- complex FFT ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fft_c32_matches_oracle(fft32, oracle, n):
    rng = seeded(100 + n)
    batch = 7 if n <= 4096 else 3  # not a multiple of the transforms-per-workgroup count
    x = rand_c(rng, (batch, n))
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"fft c32 n={n}", REL_TOL_F32)
    z = x.copy()
    fft32.fft_batch(z, inverse=True)
    assert_parity(z, oracle.ifft(x), f"ifft c32 n={n}", REL_TOL_F32)


@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 256, 1024, 4096, 8192])
def test_fft_c64_matches_oracle(fft64, oracle, n):
    rng = seeded(200 + n)
    x = rand_c(rng, (5, n), np.complex128)
    y = x.copy()
    fft64.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"fft c64 n={n}", REL_TOL_F64)
    z = x.copy()
    fft64.fft_batch(z, inverse=True)
    assert_parity(z, oracle.ifft(x), f"ifft c64 n={n}", REL_TOL_F64)


@pytest.mark.parametrize("batch", [1024, 1500, 2051])
def test_fft_c32_4096_streaming_path(fft32, oracle, batch):
    """Large batches take the persistent, prefetching kernel (fft_persist.hip.h); batch sizes that are not a
    multiple of the resident grid exercise its tail.  Same bits as the oracle, forward and inverse."""
    rng = seeded(700 + batch)
    x = rand_c(rng, (batch, 4096))
    y = x.copy()
    fft32.fft_batch(y)
    want = oracle.fft(x)
    assert_parity(y, want, f"streaming fft c32 batch={batch}", REL_TOL_F32)
    fft32.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(want), f"streaming ifft c32 batch={batch}", REL_TOL_F32)


def test_golden_complex_vectors(fft32, fft64, gold):
    for name, g in gold.items():
        if not name.startswith(("c32_", "c64_")):
            continue
        impl = fft32 if name.startswith("c32_") else fft64
        y = g["x"].copy()
        impl.fft(y)
        assert_parity(y, g["y"], name, REL_TOL_F32)
        z = g["x"].copy()
        impl.ifft(z)
        assert_parity(z, g["y_inv"], name + " (ifft)", REL_TOL_F32)


def test_reference_style_single_transforms(fft32, oracle):
    """The reference's own test inputs, through the trait-shaped API (one slice at a time)."""
    # tests/stockham_parity.rs:6-9
    for n in (32, 64, 128, 256):
        i = np.arange(n, dtype=np.float32)
        data = (i - 1j * (i * np.float32(0.25))).astype(np.complex64)
        expected = oracle.fft(data)
        fft32.stockham_fft(data)
        assert np.all(np.abs(data.real - expected.real) < 1e-3) and np.all(np.abs(data.imag - expected.imag) < 1e-3)
        assert bits_equal(data, expected)
    # tests/parallel_stockham.rs:7-10 (n = 4096)
    i = np.arange(4096, dtype=np.float32)
    data = (i + 1j * (2 * i)).astype(np.complex64)
    expected = oracle.fft(data)
    fft32.fft(data)
    assert bits_equal(data, expected)
    # lib.rs:178-199 impulse -> ones -> impulse
    d = np.array([1, 0, 0, 0], np.complex64)
    fft32.fft(d)
    assert np.all(np.abs(d - 1) < 1e-6)
    fft32.ifft(d)
    assert abs(d[0] - 1) < 1e-6 and np.all(np.abs(d[1:]) < 1e-6)
    # examples/basic_usage.rs:232-241 (BASELINE config #1): 1024-pt FFT of sin(0.1 i)
    sig = np.sin(np.float32(0.1) * np.arange(1024, dtype=np.float32)).astype(np.complex64)
    want = oracle.fft(sig)
    fft32.fft(sig)
    assert bits_equal(sig, want)


def test_special_values_follow_ieee(fft32, oracle):
    """Signed zeros, denormals and huge values take the same path as on the CPU (no flush, no shortcuts)."""
    x = np.zeros((4, 64), np.complex64)
    x[0, :] = -0.0
    x[1, 3] = np.float32(1e-42) + 1j * np.float32(-3e-43)   # subnormal
    x[2, :] = np.float32(3e37)                              # overflows to inf in the sum
    x[3, 5] = np.float32(-0.0) + 1j * np.float32(2.5)
    y = x.copy()
    fft32.fft_batch(y)
    want = oracle.fft(x)
    # inf - inf produces the platform's default NaN (x86 sets the sign bit, gfx950 does not): NaNs must sit in the
    # same places; every non-NaN value -- signed zeros and subnormals included -- must match bit for bit.
    yv, wv = y.view(np.float32), want.view(np.float32)
    assert np.array_equal(np.isnan(yv), np.isnan(wv))
    assert np.isnan(wv[2]).any() and not np.isnan(wv[[0, 1, 3]]).any()
    ok = ~np.isnan(wv)
    assert yv[ok].tobytes() == wv[ok].tobytes()


# ---- real FFT -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 256, 2048, 8192, 32768])
def test_rfft_f32_matches_oracle(fft32, oracle, n):
    rng = seeded(300 + n)
    x = rng.uniform(-1, 1, (6, n)).astype(np.float32)
    got = fft32.rfft_batch(x)
    want = oracle.rfft(x)
    assert_parity(got, want, f"rfft f32 n={n}", REL_TOL_F32)
    win = oracle.hann(n)
    assert_parity(fft32.rfft_batch(x, win), oracle.rfft(x, win), f"rfft+hann f32 n={n}", REL_TOL_F32)
    assert_parity(fft32.irfft_batch(want, n), oracle.irfft(want, n), f"irfft f32 n={n}", REL_TOL_F32)


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 16384])
def test_rfft_f64_matches_oracle(fft64, oracle, n):
    rng = seeded(400 + n)
    x = rng.uniform(-1, 1, (3, n))
    want = oracle.rfft(x)
    assert_parity(fft64.rfft_batch(x), want, f"rfft f64 n={n}", REL_TOL_F64)
    assert_parity(fft64.irfft_batch(want, n), oracle.irfft(want, n), f"irfft f64 n={n}", REL_TOL_F64)


def test_golden_rfft_vectors(fft32, fft64, gold):
    for name, g in gold.items():
        if not name.startswith(("rfft32_", "rfft64_")):
            continue
        impl = fft32 if name.startswith("rfft32_") else fft64
        x = g["x"][None, :]
        got = impl.rfft_batch(np.ascontiguousarray(x), g.get("window"))
        assert_parity(got[0], g["y"], name, REL_TOL_F32)
        back = impl.irfft_batch(np.ascontiguousarray(g["y"][None, :]), g["x"].size)
        assert_parity(back[0], g["x_back"], name + " (irfft)", REL_TOL_F32)


def test_realfftimpl_trait_surface(fft32, fft64, oracle):
    """rfft.rs:892-936 and tests/rfft_arch_parity.rs:11-27 through rfft_with_scratch / irfft_with_scratch."""
    x = np.arange(1, 9, dtype=np.float32)
    freq = np.zeros(5, np.complex64)
    scratch = np.zeros(4, np.complex64)
    fft32.rfft_with_scratch(x.copy(), freq, scratch)
    assert bits_equal(freq, oracle.rfft(x))
    out = np.zeros(8, np.float32)
    fft32.irfft_with_scratch(freq, out, scratch)
    assert np.all(np.abs(out - x) < 1e-5)
    assert abs(freq[0].imag) < 1e-6 and abs(freq[-1].imag) < 1e-6  # lib.rs:451-467
    # rfft_arch_parity: sin(i), size 32
    s = np.sin(np.arange(32, dtype=np.float32)).astype(np.float32)
    f32 = np.zeros(17, np.complex64)
    fft32.rfft(s.copy(), f32)
    want = oracle.rfft(s)
    assert np.all(np.abs(f32.real - want.real) < 1e-5) and np.all(np.abs(f32.imag - want.imag) < 1e-5)
    assert bits_equal(f32, want)
    # f64 round trip (rfft.rs:921-936)
    x64 = np.arange(1, 9, dtype=np.float64)
    f64 = np.zeros(5, np.complex128)
    fft64.rfft(x64.copy(), f64)
    o64 = np.zeros(8)
    fft64.irfft(f64, o64)
    assert np.all(np.abs(o64 - x64) < 1e-10)


# ---- STFT ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("win_len,hop,length", [(2, 1, 4), (4, 2, 8), (16, 4, 50), (64, 16, 1000), (256, 64, 3000),
                                                (1024, 256, 10000), (1024, 256, 4096), (4096, 1024, 20000)])
def test_stft_matches_oracle(fft32, oracle, win_len, hop, length):
    rng = seeded(500 + win_len)
    signal = rng.uniform(-1, 1, length).astype(np.float32)
    window = oracle.hann(win_len)
    frames = -(-length // hop) + 2  # two extra, fully/partly zero-padded frames (every provided frame is computed)
    got = fft32.stft_into(signal, window, hop, frames)
    want = oracle.stft(signal, window, hop, frames)
    assert_parity(got, want, f"stft win={win_len} hop={hop} len={length}", REL_TOL_F32)


def test_golden_stft_vectors(fft32, gold):
    g = gold["stft32_4096_w1024_h256"]
    got = fft32.stft_into(g["signal"], g["window"], int(g["hop"]), g["frames"].shape[0])
    assert_parity(got, g["frames"], "golden stft", REL_TOL_F32)


def test_stft_api_variants(fft32, oracle):
    import kofft_amd as K

    signal = np.arange(1, 9, dtype=np.float32)
    window = np.ones(4, np.float32)
    frames = [np.zeros(0, np.complex64) for _ in range(4)]  # vec![vec![]; 4]
    K.stft(signal, window, 2, frames, fft32)
    want = oracle.stft(signal, window, 2, 4)
    assert all(bits_equal(f, w) for f, w in zip(frames, want))
    # parallel(): same frames, and it accepts fewer frames than stft() demands (stft.rs:232-263)
    few = [np.zeros(0, np.complex64) for _ in range(2)]
    K.parallel(signal, window, 2, few, fft32)
    assert all(bits_equal(f, w) for f, w in zip(few, want[:2]))
    # frame() / StftStream (stft.rs:355-372, 160-206)
    buf = np.zeros(4, np.complex64)
    K.frame(signal, window, 6, buf, fft32)
    assert bits_equal(buf, want[3])
    stream = K.StftStream(signal, window, 2, fft32)
    seen = []
    while stream.next_frame(buf):
        seen.append(buf.copy())
    assert len(seen) == 4 and all(bits_equal(a, b) for a, b in zip(seen, want))
    # all-zero window -> exact zeros (stft.rs:700-720)
    z = fft32.stft_into(np.arange(1, 5, dtype=np.float32), np.zeros(2, np.float32), 1, 4)
    assert np.all(z == 0)


# ---- error behaviour through the mirrored API -------------------------------------------------------------
def test_error_variants_match_reference(fft32, oracle):
    import kofft_amd as K
    from kofft_amd import FftError

    def raises(variant, fn, *a):
        with pytest.raises(FftError) as e:
            fn(*a)
        assert e.value.code == variant, (e.value, variant)

    raises(FftError.EmptyInput, fft32.fft, np.zeros(0, np.complex64))                       # lib.rs:322-326
    raises(FftError.EmptyInput, fft32.ifft, np.zeros(0, np.complex64))
    raises(FftError.MismatchedLengths, fft32.fft_out_of_place, np.zeros(2, np.complex64), np.zeros(3, np.complex64))
    raises(FftError.InvalidStride, fft32.fft_strided, np.zeros(8, np.complex64), 0, np.zeros(4, np.complex64))
    raises(FftError.MismatchedLengths, fft32.fft_strided, np.zeros(6, np.complex64), 2, np.zeros(4, np.complex64))
    raises(FftError.MismatchedLengths, fft32.rfft, np.zeros(4, np.float32), np.zeros(4, np.complex64))  # lib.rs:470
    raises(FftError.EmptyInput, fft32.rfft, np.zeros(0, np.float32), np.zeros(1, np.complex64))
    raises(FftError.InvalidValue, fft32.rfft, np.zeros(3, np.float32), np.zeros(2, np.complex64))
    raises(FftError.MismatchedLengths, fft32.fft_split, np.zeros(4, np.float32), np.zeros(3, np.float32))
    raises(FftError.InvalidHopSize, K.stft, np.ones(4, np.float32), np.ones(2, np.float32), 0, [None] * 4, fft32)
    raises(FftError.MismatchedLengths, K.stft, np.zeros(10, np.float32), K.hann(4), 4, [None] * 2, fft32)  # tests/stft.rs
    with pytest.raises(FftError):
        K.StftStream(np.ones(4, np.float32), np.ones(2, np.float32), 0, fft32)
    s = K.StftStream(np.ones(4, np.float32), np.ones(2, np.float32), 1, fft32)
    raises(FftError.MismatchedLengths, s.next_frame, np.zeros(3, np.complex64))               # stft.rs:861-870
    # single element / strategy plumbing
    one = np.array([1 + 0j], np.complex64)
    fft32.fft(one)
    assert one[0] == 1 + 0j                                                                   # lib.rs:352-358
    d = np.arange(8).astype(np.complex64)
    e = d.copy()
    fft32.fft_with_strategy(d, K.FftStrategy.SplitRadix)
    fft32.fft(e)
    assert bits_equal(d, e)
    # real FFT lengths whose half is not a power of two take the composed path (rfft.rs:447 calls fft.fft for any m)
    x12 = np.arange(12, dtype=np.float32)
    out12 = np.zeros(7, np.complex64)
    fft32.rfft(x12.copy(), out12)
    assert bits_equal(out12, oracle.rfft(x12[None, :])[0])


def test_strided_split_and_batch_helpers(fft32, fft64, oracle):
    import kofft_amd as K

    rng = seeded(9)
    # fft_strided (fft.rs:1175-1199): every 3rd element, n = 16
    buf = rand_c(rng, 48)
    want = buf.copy()
    want[::3] = oracle.fft(buf[::3].copy())
    fft32.fft_strided(buf, 3, np.zeros(16, np.complex64))
    assert bits_equal(buf, want)
    # out-of-place strided
    src = rand_c(rng, 64)
    dst = np.zeros(32, np.complex64)
    fft32.fft_out_of_place_strided(src, 4, dst, 2)
    assert bits_equal(dst[::2], oracle.fft(src[::4].copy()))
    # fft_split / ifft_split (tests/split64.rs)
    n = 32
    re = np.arange(n, dtype=np.float64)
    im = np.zeros(n)
    fft64.fft_split(re, im)
    w = oracle.fft(np.arange(n).astype(np.complex128))
    assert np.all(np.abs(re - w.real) < 1e-10) and np.all(np.abs(im - w.imag) < 1e-10)
    fft64.ifft_split(re, im)
    assert np.all(np.abs(re - np.arange(n)) < 1e-8)
    # batch()/multi_channel() over a list of slices (fft.rs:2156-2191), ragged lengths included
    vs = [rand_c(rng, 64), rand_c(rng, 64), rand_c(rng, 256)]
    want = [oracle.fft(v) for v in vs]
    K.batch(fft32, vs)
    assert all(bits_equal(a, b) for a, b in zip(vs, want))
    K.batch_inverse(fft32, vs)
    assert all(bits_equal(a, oracle.ifft(b)) for a, b in zip(vs, want))


# ---- streaming (persistent, prefetching) kernels: large batches ----------------------------------------------
@pytest.mark.parametrize("batch", [8192, 9001])
def test_fft_c32_1024_streaming_path(fft32, oracle, batch):
    """n = 1024 with >= 8192 transforms runs the wave-synchronous persistent kernel (one wavefront per transform)."""
    rng = seeded(800 + batch)
    x = rand_c(rng, (batch, 1024))
    y = x.copy()
    fft32.fft_batch(y)
    want = oracle.fft(x)
    assert_parity(y, want, f"streaming fft c32 n=1024 batch={batch}", REL_TOL_F32)
    fft32.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(want), f"streaming ifft c32 n=1024 batch={batch}", REL_TOL_F32)


@pytest.mark.parametrize("batch", [4096, 4099])
def test_n2048_streaming_paths(fft32, oracle, batch):
    """n = 2048 (two wavefronts per transform, block-synchronised exchange): complex, rfft 4096 and STFT 2048 variants."""
    rng = seeded(950 + batch)
    x = rand_c(rng, (batch, 2048))
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"streaming fft c32 n=2048 batch={batch}", REL_TOL_F32)
    r = rng.uniform(-1, 1, (batch, 4096)).astype(np.float32)
    win = oracle.hann(4096)
    assert_parity(fft32.rfft_batch(r, win), oracle.rfft(r, win), f"streaming rfft n=4096 batch={batch}", REL_TOL_F32)
    sig = rng.uniform(-1, 1, 512 * batch + 100).astype(np.float32)
    w2 = oracle.hann(2048)
    frames = -(-sig.size // 512)
    assert_parity(fft32.stft_into(sig, w2, 512, frames), oracle.stft(sig, w2, 512, frames), "streaming stft win=2048", REL_TOL_F32)


@pytest.mark.parametrize("batch,windowed", [(8192, True), (8200, False), (10001, True)])
def test_rfft_2048_streaming_path(fft32, oracle, batch, windowed):
    """BASELINE config #3 shape (2048-pt rfft + Hann) at a batch that takes the persistent kernel with the window and
    the post-pass table staged in LDS; 10001 rows exercise the tail (rows are 8200 B: only 8-byte aligned)."""
    rng = seeded(900 + batch)
    x = rng.uniform(-1, 1, (batch, 2048)).astype(np.float32)
    win = oracle.hann(2048) if windowed else None
    got = fft32.rfft_batch(x, win)
    assert_parity(got, oracle.rfft(x, win), f"streaming rfft n=2048 batch={batch} windowed={windowed}", REL_TOL_F32)


@pytest.mark.parametrize("length,hop", [(2_200_000, 256), (1_100_003, 128)])
def test_stft_1024_streaming_path(fft32, oracle, length, hop):
    """BASELINE config #4 shape (1024-pt Hann, hop 256) with > 8192 frames: persistent kernel, frames that run off the
    end of the signal are zero-filled by the buffer bounds check; two extra frames lie entirely past the end."""
    rng = seeded(1000 + hop)
    signal = rng.uniform(-1, 1, length).astype(np.float32)
    window = oracle.hann(1024)
    frames = -(-length // hop) + 2
    got = fft32.stft_into(signal, window, hop, frames)
    want = oracle.stft(signal, window, hop, frames)
    assert_parity(got, want, f"streaming stft len={length} hop={hop}", REL_TOL_F32)
    assert np.all(got[-1] == 0)


# ---- large n: two-factor path (fft_big.hip.h) ------------------------------------------------------------------
@pytest.mark.parametrize("log2n,batch", [(15, 3), (16, 2), (17, 2), (20, 2), (21, 2), (22, 1), (23, 3), (24, 1)])
def test_fft_c32_large_n(fft32, oracle, log2n, batch):
    """n > 16384 runs as two (from 2^21: three) factors over HBM with the ONE reference table T_n; still the reference's
    butterflies.  21..24 cover the three-factor splits 7+7+7, 8+7+7, 8+8+7 and 8+8+8."""
    n = 1 << log2n
    rng = seeded(1100 + log2n)
    x = rand_c(rng, (batch, n))
    y = x.copy()
    fft32.fft_batch(y)
    want = oracle.fft(x)
    assert_parity(y, want, f"large fft c32 n=2^{log2n}", REL_TOL_F32)
    fft32.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(want), f"large ifft c32 n=2^{log2n}", REL_TOL_F32)


@pytest.mark.parametrize("log2n,batch", [(14, 3), (15, 2), (18, 2), (20, 3), (21, 1), (22, 2)])
def test_fft_c64_large_n(fft64, oracle, log2n, batch):
    """BASELINE config #5's transform (2^20-point Complex64) and smaller two-factor sizes, bit for bit."""
    n = 1 << log2n
    rng = seeded(1200 + log2n)
    x = rand_c(rng, (batch, n), np.complex128)
    y = x.copy()
    fft64.fft_batch(y)
    want = oracle.fft(x)
    assert_parity(y, want, f"large fft c64 n=2^{log2n}", REL_TOL_F64)
    fft64.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(want), f"large ifft c64 n=2^{log2n}", REL_TOL_F64)
    if log2n == 20:  # tests/split64.rs-style truth check at cfg5's size: f64 drift budget 2.3e-11
        ref = np.fft.fft(x[0])
        assert rel_err(want[0], ref) < 1e-9


@pytest.mark.parametrize("dtype,log2n,batch", [("c64", 20, 18), ("c64", 16, 300), ("c64", 22, 3), ("c32", 17, 200), ("c32", 20, 40),
                                               ("c32", 22, 6), ("c64", 19, 40), ("c32", 21, 11), ("c64", 21, 5), ("c64", 21, 9), ("c32", 15, 64), ("c32", 19, 16)])
def test_large_n_persistent_factor_kernels(fft32, fft64, oracle, dtype, log2n, batch):
    """Batches large enough for the persistent factor kernels (fft_tile_persist_kernel / fft_rows_persist_kernel: every
    resident workgroup walks several tiles, prefetching the next; the last factor keeps its table entries resident per row
    tile): forward and inverse, the first / middle / last transforms bit for bit against the oracle, Parseval on all."""
    n = 1 << log2n
    cdt = np.complex128 if dtype == "c64" else np.complex64
    fft = fft64 if dtype == "c64" else fft32
    rng = seeded(3400 + log2n)
    x = rand_c(rng, (batch, n), cdt)
    y = x.copy()
    fft.fft_batch(y)
    pick = sorted({0, 1, batch // 2, batch - 1})
    want = oracle.fft(x[pick])
    tol = REL_TOL_F64 if dtype == "c64" else REL_TOL_F32
    assert_parity(y[pick], want, f"persistent factors {dtype} 2^{log2n}", tol)
    ex = (np.abs(x) ** 2).sum(axis=1, dtype=np.float64)
    ey = (np.abs(y) ** 2).sum(axis=1, dtype=np.float64) / n
    assert np.max(np.abs(ey - ex) / ex) < (1e-8 if dtype == "c64" else 3e-3)  # the recurrence tables drift (2^22 f64: 1.1e-9)
    fft.fft_batch(y, inverse=True)
    assert_parity(y[pick], oracle.ifft(want), f"persistent factors inverse {dtype} 2^{log2n}", tol)


@pytest.mark.parametrize("dtype,log2n,batch", [("c32", 15, 515), ("c64", 14, 520), ("c32", 15, 1030), ("c64", 14, 777)])
def test_regfile_resident_kernels_every_transform(oracle, dtype, log2n, batch, monkeypatch):
    """Round 4: c32 n = 2^15 and c64 n = 2^14 (256 KiB per transform) run ONE pass over HBM with the transform held in the CU's
    register file (fft_regfile.hip.h: 1024 threads x 32 / 16 points, exchanges through LDS in a real and an imaginary round) once
    the batch gives every CU two transforms; KOFFT_HIP_REGFILE=0 keeps the two-factor path.  EVERY transform against the oracle,
    forward and inverse, and both routes byte for byte."""
    import kofft_amd

    n = 1 << log2n
    cdt = np.complex128 if dtype == "c64" else np.complex64
    rdt = np.float64 if dtype == "c64" else np.float32
    tol = REL_TOL_F64 if dtype == "c64" else REL_TOL_F32
    x = rand_c(seeded(4300 + log2n + batch), (batch, n), cdt)
    want = oracle.fft(x)
    outs = []
    for regfile in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_REGFILE", regfile)  # read when the context is created
        f = kofft_amd.HipFftImpl(rdt)
        y = x.copy()
        f.fft_batch(y)
        assert_parity(y, want, f"regfile={regfile} {dtype} 2^{log2n} x {batch}", tol)
        z = y.copy()
        f.fft_batch(z, inverse=True)
        outs.append((y, z))
    assert bits_equal(outs[0][1], outs[1][1])
    assert_parity(outs[0][1], oracle.ifft(want), f"regfile inverse {dtype} 2^{log2n} x {batch}", tol)


@pytest.mark.parametrize("dtype,n,batch", [("f32", 65536, 515), ("f32", 65536, 1027), ("f64", 32768, 520)])
def test_windowed_rfft_through_the_regfile_kernel(oracle, dtype, n, batch, monkeypatch):
    """rfft of 65536 (f32) / 32768 (f64) samples WITH a row window: the m-point transform of the packed row runs in the register-file
    kernel with the window product on its loads (RowWindowIO; without it the window rides on the factor path's first load: two passes).
    EVERY row against the oracle, and byte for byte against KOFFT_HIP_REGFILE=0; without a window (plain loads) alike; and irfft of the
    result, whose pre-pass (bins e and m - e -> element e) rides on the same kernel's loads."""
    import kofft_amd

    rdt = np.float32 if dtype == "f32" else np.float64
    rng = seeded(4400 + n + batch)
    x = rng.uniform(-1, 1, (batch, n)).astype(rdt)
    win = rng.uniform(0.1, 1, n).astype(rdt)
    want = oracle.rfft(x, win)
    want_plain = oracle.rfft(x)
    outs = []
    for regfile in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_REGFILE", regfile)  # read when the context is created
        f = kofft_amd.HipFftImpl(rdt)
        got = f.rfft_batch(x, win)
        assert_parity(got, want, f"windowed rfft regfile={regfile} {dtype} n={n} x {batch}", REL_TOL_F32 if dtype == "f32" else REL_TOL_F64)
        assert bits_equal(f.rfft_batch(x), want_plain)
        # ... and back: irfft with the pre-pass on the same kernel's loads (IrfftRowIO), every row
        back = f.irfft_batch(got, n)
        assert_parity(back, oracle.irfft(got, n), f"irfft regfile={regfile} {dtype} n={n} x {batch}", REL_TOL_F32 if dtype == "f32" else REL_TOL_F64)
        outs.append((got, back))
    assert bits_equal(outs[0][0], outs[1][0]) and bits_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("log2n,batch", [(15, 40), (16, 40), (17, 33), (18, 20), (19, 18), (20, 10), (21, 9)])
def test_c32_last_factor_on_row_pairs(oracle, log2n, batch, monkeypatch):
    """Round 4: the c32 last factor of the two-factor path runs on PAIRS of adjacent rows -- two Complex<f32> values as one 16-byte value
    (f32x2 "scalars": every operation elementwise, each row with its own table entry) through the c64 kernel's structure, 16-row tiles,
    128-byte runs on both sides; the first factor interleaves the rows of a pair in the intermediate (blk_r = 1).
    KOFFT_HIP_BIG_ROW_PAIRS=0 keeps one row per thread slot.  EVERY transform of the batch byte for byte between the two routes, forward
    and inverse; first / middle / last against the oracle."""
    import kofft_amd

    n = 1 << log2n
    x = rand_c(seeded(4500 + log2n), (batch, n))
    outs = []
    for pairs in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_BIG_ROW_PAIRS", pairs)  # read when the context is created
        monkeypatch.setenv("KOFFT_HIP_REGFILE", "0")          # (2^15: keep it on the factor path whatever the batch)
        f = kofft_amd.HipFftImpl(np.float32)
        y = x.copy()
        f.fft_batch(y)
        z = y.copy()
        f.fft_batch(z, inverse=True)
        outs.append((y, z))
    assert bits_equal(outs[0][0], outs[1][0]) and bits_equal(outs[0][1], outs[1][1])
    pick = [0, batch // 2, batch - 1]
    want = oracle.fft(x[pick])
    assert_parity(outs[0][0][pick], want, f"row pairs c32 2^{log2n}", REL_TOL_F32)
    assert_parity(outs[0][1][pick], oracle.ifft(want), f"row pairs inverse c32 2^{log2n}", REL_TOL_F32)


@pytest.mark.parametrize("log2n,batch", [(14, 70), (15, 33), (17, 40), (19, 20), (20, 10)])
def test_c64_factor_intermediate_layouts_agree(oracle, log2n, batch, monkeypatch):
    """Round 4: between the two persistent factor kernels the c64 intermediate is block-interleaved (BigColsIO::out_lane: one
    contiguous 1 KiB run per wavefront store, the last factor's row tile one contiguous stream); KOFFT_HIP_BIG_BLOCKED=0 keeps
    the natural matrix layout.  Same butterflies either way: EVERY transform of the batch byte for byte, forward and inverse,
    and the first / last against the oracle."""
    import kofft_amd

    n = 1 << log2n
    x = rand_c(seeded(4100 + log2n), (batch, n), np.complex128)
    outs = []
    for blocked in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_BIG_BLOCKED", blocked)  # read when the context is created
        f = kofft_amd.HipFftImpl(np.float64)
        y = x.copy()
        f.fft_batch(y)
        z = y.copy()
        f.fft_batch(z, inverse=True)
        outs.append((y, z))
    assert bits_equal(outs[0][0], outs[1][0]) and bits_equal(outs[0][1], outs[1][1])
    pick = [0, batch - 1]
    want = oracle.fft(x[pick])
    assert_parity(outs[0][0][pick], want, f"blocked intermediate c64 2^{log2n}", REL_TOL_F64)
    assert_parity(outs[0][1][pick], oracle.ifft(want), f"blocked intermediate inverse c64 2^{log2n}", REL_TOL_F64)


# ---- real / STFT lengths beyond the fused kernels: composed from fft_dev ------------------------------
@pytest.mark.parametrize("n,batch", [(2, 5), (6, 4), (12, 7), (30, 3), (1000, 9), (65536, 3), (1 << 20, 2), (40000, 2)])
def test_rfft_irfft_any_length_f32(fft32, oracle, n, batch):
    """rfft_direct / irfft_direct call fft.fft / fft.ifft on the half length whatever it is (rfft.rs:447, 504): Bluestein for
    non-powers of two (fft.rs:1083-1132), the factor path beyond 2^14.  Window product, transform and post-pass are separate
    kernels here; every element still sees the reference's operations in the reference's order."""
    rng = seeded(3000 + n % 977)
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    win = rng.uniform(0, 1, n).astype(np.float32)
    for w in (None, win):
        got = fft32.rfft_batch(x, w)
        want = oracle.rfft(x, w)
        assert_parity(got, want, f"composed rfft n={n} window={w is not None}", REL_TOL_F32)
    spec = oracle.rfft(x)
    assert_parity(fft32.irfft_batch(spec, n), oracle.irfft(spec, n), f"composed irfft n={n}", REL_TOL_F32)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_pointwise_factors_folded_into_the_factor_kernels(oracle, fused):
    """Round 3: for inner lengths beyond one workgroup's transform the Bluestein products ride on the factor
    kernels' first load / last stores (BigColsIO PRE_CHIRP, BigRowsIO POST_BLUE_MID / POST_BLUE_OUT) and the row window of the
    real transform on the first load (PRE_WINDOW).  Batches large enough for the PERSISTENT factor kernels (the small-batch
    tests above take the one-tile kernels' load() / store() forms), forward and inverse, f32 and f64, against the oracle; the
    separate pointwise kernels (KOFFT_HIP_BLUESTEIN_FUSED=0) give the same bits."""
    import os

    import kofft_amd

    os.environ["KOFFT_HIP_BLUESTEIN_FUSED"] = fused
    try:
        f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
    finally:
        del os.environ["KOFFT_HIP_BLUESTEIN_FUSED"]
    rng = seeded(6100)
    x = rand_c(rng, (160, 20000))            # m = 65536: two factors, 160 * 256 columns
    pick = [0, 1, 77, 159]
    y = x.copy()
    f32.fft_batch(y)
    assert bits_equal(y[pick], oracle.fft(x[pick]))
    y = x.copy()
    f32.fft_batch(y, inverse=True)
    assert bits_equal(y[pick], oracle.ifft(x[pick]))
    xd = rand_c(rng, (96, 9001), np.complex128)   # m = 32768 in f64
    yd = xd.copy()
    f64.fft_batch(yd)
    assert bits_equal(yd[[0, 50, 95]], oracle.fft(xd[[0, 50, 95]]))
    yd = xd.copy()
    f64.fft_batch(yd, inverse=True)
    assert bits_equal(yd[[0, 50, 95]], oracle.ifft(xd[[0, 50, 95]]))
    r = rng.uniform(-1, 1, (80, 65536)).astype(np.float32)   # rfft: m = 32768 complex, window folded into the first factor
    win = rng.uniform(0, 1, 65536).astype(np.float32)
    got = f32.rfft_batch(r, win)
    assert bits_equal(got[[0, 41, 79]], oracle.rfft(r[[0, 41, 79]], win))
    rd = rng.uniform(-1, 1, (70, 32768))                        # f64: m = 16384
    wd = rng.uniform(0, 1, 32768)
    gd = f64.rfft_batch(rd, wd)
    assert bits_equal(gd[[0, 69]], oracle.rfft(rd[[0, 69]], wd))


@pytest.mark.parametrize("n,batch", [(12, 5), (1000, 4), (65536, 2), (34, 3)])
def test_rfft_irfft_any_length_f64(fft64, oracle, n, batch):
    rng = seeded(3100 + n % 977)
    x = rng.uniform(-1, 1, (batch, n))
    got = fft64.rfft_batch(x)
    want = oracle.rfft(x)
    assert_parity(got, want, f"composed rfft f64 n={n}", REL_TOL_F64)
    assert_parity(fft64.irfft_batch(want, n), oracle.irfft(want, n), f"composed irfft f64 n={n}", REL_TOL_F64)


@pytest.mark.parametrize("win_len,hop,length", [(12, 5, 200), (1000, 250, 9000), (32768, 8192, 100_000), (3, 1, 20)])
def test_stft_istft_any_window_length(fft32, oracle, win_len, hop, length):
    """stft calls fft.fft(frame) for any win_len (stft.rs:91-103); istft, inverse_parallel and stft_magnitudes follow."""
    import kofft_amd as K

    rng = seeded(3200 + win_len % 977)
    signal = rng.uniform(-1, 1, length).astype(np.float32)
    window = oracle.hann(win_len)
    frames = -(-length // hop) + 1
    spec = oracle.stft(signal, window, hop, frames)
    got = fft32.stft_into(signal, window, hop, frames)
    assert_parity(got, spec, f"composed stft win={win_len}", REL_TOL_F32)
    want = oracle.istft(spec, window, hop, length)
    out = np.zeros(length, np.float32)
    K.istft(spec.copy(), window, hop, out, np.zeros(length, np.float32), fft32)
    assert_parity(out, want, f"istft win={win_len}", REL_TOL_F32)
    mags, mx = fft32.stft_magnitudes(signal, win_len, hop)
    wm, wmx = oracle.stft_magnitudes(signal, win_len, hop)
    assert bits_equal(mags, wm) and mx == wmx


@pytest.mark.parametrize("depth,rows,cols", [(1, 12, 10), (1, 100, 6), (6, 5, 7), (1, 3, 32768), (2, 1000, 16),
                                             (1, 3, 1 << 22), (1, 1 << 22, 2)])  # panels with one side beyond 65535 tiles
def test_ndfft_any_axis_length(fft32, oracle, depth, rows, cols):
    """ndfft runs FftImpl::fft on rows and fft_strided down the other axes for any length (ndfft.rs:89-98, 131-151): axes the
    strided kernel does not cover go through transpose -> batched fft_dev (Bluestein / factor path) -> transpose."""
    rng = seeded(3300 + rows * 7 + cols)
    x = rand_c(rng, (depth, rows, cols))
    want = x.copy()
    if depth > 1:
        want = _oracle_axis(oracle, want, 0)
        want = _oracle_axis(oracle, want, 1)
        want = _oracle_axis(oracle, want, 2)
    else:
        want = _oracle_axis(oracle, want, 2)
        want = _oracle_axis(oracle, want, 1)
    y = x.copy().reshape(-1)
    fft32.fftnd(y, depth, rows, cols)
    assert_parity(y.reshape(depth, rows, cols), want, f"ndfft {depth}x{rows}x{cols}", REL_TOL_F32)


# ---- ISTFT -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("win_len,hop,length", [(4, 2, 8), (16, 4, 100), (64, 64, 1000), (256, 32, 5000), (1024, 256, 40000),
                                                (1024, 300, 9000)])
def test_istft_matches_oracle(fft32, oracle, win_len, hop, length):
    """stft::istft (stft.rs:117-156): per-sample overlap-add in frame order + normalisation, bit for bit; round trip
    STFT -> ISTFT recovers the signal wherever the window-square sum is not tiny."""
    import kofft_amd as K

    rng = seeded(1300 + win_len + hop)
    signal = rng.uniform(-1, 1, length).astype(np.float32)
    window = oracle.hann(win_len)
    nframes = -(-length // hop)
    spec = oracle.stft(signal, window, hop, nframes)
    want = oracle.istft(spec, window, hop, length)
    frames = spec.copy()
    out = np.zeros(length, np.float32)
    scratch = np.full(length, 7.0, np.float32)  # must be overwritten
    K.istft(frames, window, hop, out, scratch, fft32)
    assert_parity(out, want, f"istft win={win_len} hop={hop}", REL_TOL_F32)
    assert_parity(frames, oracle.ifft(spec), "istft leaves the inverse-transformed frames behind", REL_TOL_F32)
    if hop <= win_len // 2:
        ok = scratch > 1e-3
        assert np.max(np.abs(out[ok] - signal[ok])) < 1e-3


@pytest.mark.parametrize("win_len,hop,frames", [(12, 5, 300_000), (30, 10, 150_000), (60, 20, 90_000), (100, 40, 60_000), (200, 80, 40_001), (400, 160, 20_001),
                                                (500, 125, 14_003), (1000, 250, 5_001), (1102, 441, 2_600), (2500, 625, 1_100)])
def test_stft_window_not_a_power_of_two_large_frame_counts(oracle, monkeypatch, win_len, hop, frames):
    """stft.rs:91-103 calls fft.fft(frame) for ANY window length: lengths that are not powers of two take the Bluestein arm.  With enough
    frames the framing product rides on the persistent Bluestein kernel's loads (BlueStftSrc: one pass over HBM; the composed route writes
    the frames out, then transforms them in place).  Frames that run off the end of the signal included; head, middle and tail against the
    oracle, EVERY frame against the composed route (KOFFT_HIP_BLUESTEIN_PERSIST=0)."""
    import kofft_amd

    rng = seeded(2600 + win_len)
    total = (frames - 3) * hop + win_len // 3  # the last frames are partly / wholly past the end
    sig = rng.uniform(-1, 1, total).astype(np.float32)
    win = rng.uniform(0.1, 1, win_len).astype(np.float32)
    f = kofft_amd.HipFftImpl(np.float32)
    got = f.stft_into(sig, win, hop, frames, check_frames=False)
    for first, count in ((0, 3), (frames // 2, 2), (frames - 5, 5)):
        assert bits_equal(got[first:first + count], oracle.stft_range(sig, win, hop, first, count)), (first, count)
    monkeypatch.setenv("KOFFT_HIP_BLUESTEIN_PERSIST", "0")
    g = kofft_amd.HipFftImpl(np.float32)
    assert bits_equal(got, g.stft_into(sig, win, hop, frames, check_frames=False))


@pytest.mark.parametrize("win_len,hop,nframes,out_delta", [
    (1024, 256, 9001, 0), (1024, 256, 9003, 1500), (1024, 256, 9000, -777), (1024, 512, 9002, 300),
    (512, 256, 9001, 0), (2048, 512, 4101, 5), (2048, 1024, 4100, -3000), (4096, 1024, 2101, 0), (4096, 2048, 2100, 4097),
    (256, 128, 33001, 3), (256, 64, 33000, -70), (256, 32, 33003, 0), (512, 512, 9001, 0), (512, 128, 9002, 1), (512, 64, 9003, -9),
    (1024, 1024, 9000, 17), (1024, 128, 9001, 0), (2048, 2048, 4100, 0), (2048, 256, 4101, -1), (4096, 4096, 2100, 0), (4096, 512, 2101, 100),
    # (round 6, tools/kernel_coverage.sh: runs of at least 2 x (win / hop) frames per workgroup -- the counts above left <11, 3>, <12, 2>, <12, 3> on the two-kernel route)
    (2048, 256, 8301, 0), (4096, 1024, 4101, 5), (4096, 512, 8201, -9),
])
def test_istft_fused_kernel_large_frame_counts(oracle, monkeypatch, win_len, hop, nframes, out_delta):
    """Frame counts that give every workgroup of the chip-sized grid a run of frames take istft_fused_kernel (inverse transform + ordered
    overlap-add in one kernel, the seams between the workgroups' runs and the tail on the overlap-add kernel): output, scratch and the
    frames left behind are the oracle's bytes -- output lengths equal to, beyond and short of what the frames reach; and the same call
    with KOFFT_HIP_ISTFT_FUSED=0 (two kernels) accumulating into a NON-ZERO output gives the same bytes as the fused one."""
    import kofft_amd as K

    rng = seeded(1400 + win_len + hop + nframes)
    spec = rand_c(rng, (nframes, win_len))
    window = (oracle.hann(win_len) + np.float32(0.01)).astype(np.float32)
    out_len = (nframes - 1) * hop + win_len + out_delta
    want = oracle.istft(spec, window, hop, out_len)
    f = K.HipFftImpl(np.float32)
    frames = spec.copy()
    out = np.zeros(out_len, np.float32)
    scratch = np.full(out_len, 7.0, np.float32)
    K.istft(frames, window, hop, out, scratch, f)
    assert_parity(out, want, f"fused istft win={win_len} hop={hop} frames={nframes}", REL_TOL_F32)
    pick = sorted({0, 1, 2, 3, 4, nframes // 2, nframes - 3, nframes - 2, nframes - 1})
    assert_parity(frames[pick], oracle.ifft(spec[pick]), "fused istft leaves the inverse-transformed frames behind", REL_TOL_F32)
    # against the two-kernel route, accumulating into a non-zero output (stft.rs:144: output[..] +=)
    base = rng.uniform(-1, 1, out_len).astype(np.float32)
    out_f, scr_f, fr_f = base.copy(), np.zeros(out_len, np.float32), spec.copy()
    K.istft(fr_f, window, hop, out_f, scr_f, f)
    monkeypatch.setenv("KOFFT_HIP_ISTFT_FUSED", "0")
    g = K.HipFftImpl(np.float32)
    out_g, scr_g, fr_g = base.copy(), np.zeros(out_len, np.float32), spec.copy()
    K.istft(fr_g, window, hop, out_g, scr_g, g)
    assert bits_equal(out_f, out_g) and bits_equal(scr_f, scr_g) and bits_equal(fr_f, fr_g)
    assert bits_equal(scr_f, scratch)
    # inverse_parallel (tiny sums become 0, frames untouched): same route underneath
    keep = spec.copy()
    out_p = np.zeros(out_len, np.float32)
    K.inverse_parallel(keep, window, hop, out_p, f)
    assert bits_equal(keep, spec)
    want_p = want.copy()
    want_p[scratch <= 1e-8] = 0.0
    assert bits_equal(out_p, want_p)


def test_istft_reference_tests(fft32, oracle):
    import ctypes as C

    import kofft_amd as K
    from kofft_amd import FftError

    # stft.rs:560-580 test_stft_istft_batch_roundtrip (list-of-frames form)
    signal = np.arange(1, 9, dtype=np.float32)
    window = np.ones(4, np.float32)
    frames = [np.zeros(0, np.complex64) for _ in range(4)]
    K.stft(signal, window, 2, frames, fft32)
    output = np.zeros(8, np.float32)
    scratch = np.zeros(8, np.float32)
    K.istft(frames, window, 2, output, scratch, fft32)
    assert np.all(np.abs(output - signal) < 1e-4)
    # stft.rs:654-663 output shorter than the frames cover: Ok, just not filled
    fr = [np.zeros(4, np.complex64)]
    K.istft(fr, window, 2, np.zeros(2, np.float32), np.zeros(2, np.float32), fft32)
    # stft.rs:666-675 frame size mismatch; 838-846 zero hop; scratch length mismatch (stft.rs:128)
    with pytest.raises(FftError) as e:
        K.istft([np.zeros(3, np.complex64)], window, 2, np.zeros(8, np.float32), np.zeros(8, np.float32), fft32)
    assert e.value.code == FftError.MismatchedLengths
    with pytest.raises(FftError) as e:
        K.istft([np.zeros(4, np.complex64)], window, 0, np.zeros(8, np.float32), np.zeros(8, np.float32), fft32)
    assert e.value.code == FftError.InvalidHopSize
    with pytest.raises(FftError) as e:
        K.istft([np.zeros(4, np.complex64)], window, 2, np.zeros(8, np.float32), np.zeros(7, np.float32), fft32)
    assert e.value.code == FftError.MismatchedLengths
    # istft accumulates into the caller's output (stft.rs:144: "+=")
    spec = oracle.stft(signal, window, 2, 4)
    pre = np.full(8, 10.0, np.float32)
    want = pre.copy()
    fr2 = spec.copy()
    scr = np.zeros(8, np.float32)
    L = oracle.lib()
    L.ko_istft_f32(C.c_void_p(fr2.ctypes.data), C.c_size_t(4), C.c_void_p(window.ctypes.data), C.c_size_t(4), C.c_size_t(2),
                   C.c_void_p(want.ctypes.data), C.c_size_t(8), C.c_void_p(scr.ctypes.data), C.c_size_t(8))
    got = pre.copy()
    K.istft(spec.copy(), window, 2, got, np.zeros(8, np.float32), fft32)
    assert bits_equal(got, want)


# ---- STFT magnitudes -----------------------------------------------------------------------------------
@pytest.mark.parametrize("win_len,hop,length", [(4, 2, 10), (64, 16, 1000), (256, 64, 5000), (1024, 256, 20000), (1024, 256, 2_200_000)])
def test_stft_magnitudes_match_oracle(fft32, oracle, win_len, hop, length):
    """visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): Hann STFT, first win_len/2 bins as
    sqrt(re*re + im*im) in f32, and the maximum.  The last size takes the streaming kernel with the fused store."""
    rng = seeded(1400 + win_len)
    samples = rng.uniform(-1, 1, length).astype(np.float32)
    want, want_max = oracle.stft_magnitudes(samples, win_len, hop)  # oracle/kofft_oracle.c: spectrogram.rs:52-76 restated
    mags, mx = fft32.stft_magnitudes(samples, win_len, hop)
    assert_parity(mags, want, f"stft_magnitudes win={win_len}", REL_TOL_F32)
    assert bits_equal(mags, want)
    assert mx == want_max


# ---- 2-D / 3-D FFT ---------------------------------------------------------------------------------------
def _oracle_axis(oracle, x, axis, inverse=False):
    moved = np.ascontiguousarray(np.moveaxis(x, axis, -1))
    return np.ascontiguousarray(np.moveaxis(oracle.fft(moved, inverse=inverse), -1, axis))


@pytest.mark.parametrize("rows,cols", [(4, 4), (8, 32), (64, 16), (256, 128), (1024, 64), (32, 4096), (2, 1), (1, 8)])
def test_fft2d_matches_oracle(fft32, oracle, rows, cols):
    """ndfft::fft2d_inplace (ndfft.rs:74-101): rows, then strided columns -- every line an independent reference FFT."""
    import kofft_amd as K

    rng = seeded(1500 + rows + cols)
    x = rand_c(rng, (rows, cols))
    want = _oracle_axis(oracle, _oracle_axis(oracle, x, 1), 0)
    data = x.reshape(-1).copy()
    K.fft2d_inplace(data, rows, cols, fft32, np.zeros(rows, np.complex64))
    assert_parity(data.reshape(rows, cols), want, f"fft2d {rows}x{cols}", REL_TOL_F32)
    fft32.fftnd(data, 1, rows, cols, inverse=True)  # round trip along the same axes (ndfft.rs tests, 165-176)
    back = _oracle_axis(oracle, _oracle_axis(oracle, want, 1, True), 0, True)
    assert_parity(data.reshape(rows, cols), back, f"ifft2d {rows}x{cols}", REL_TOL_F32)


@pytest.mark.parametrize("n,batch,windowed", [(1 << 17, 40, True), (1 << 17, 37, False), (1 << 18, 19, True), (1 << 19, 20, False), (1 << 20, 9, True),
                                              (1 << 21, 5, True), (1 << 22, 3, True), (1 << 17, 1100, True)])
def test_rfft_of_2p17_to_2p22_reals_every_row(oracle, n, batch, windowed):
    """rfft of 2^17 .. 2^22 reals (inner length m = 2^16 .. 2^21: first factor with the row window on its loads, last factor, post-pass
    kernel -- the composed route of real_impl.hip.h), EVERY output row against the oracle bit for bit, with and without a window,
    ragged batches, and 1100 transforms (two chunks of the intermediate).  (Written in round 5 for the two-pass variant with the post-pass
    in the last factor's epilogue, which passed it and was removed as slower; kept as the every-row check of these sizes.)"""
    import kofft_amd

    rng = seeded(7700 + n % 977 + batch)
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    x[0, :] = 0.0
    x[0, 1] = 1.0  # (an impulse row: every bin has unit modulus before the window)
    win = oracle.hann(n) if windowed else None
    want = oracle.rfft_mt(x, win)
    f = kofft_amd.HipFftImpl(np.float32)
    got = f.rfft_batch(x, win)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"rfft n={n} batch={batch}"


@pytest.mark.parametrize("rows,cols", [(1024, 4096), (2048, 4096), (4096, 4096), (1024, 1024), (2048, 1024), (4096, 1024), (1024, 2048),
                                       (2048, 2048), (4096, 2048), (512, 4096), (8192, 2048), (8192, 1024), (8192, 4096)])
def test_fft2d_fused_two_passes(fft32, oracle, rows, cols):
    """Round 5 (fft_nd_fused.hip.h): c32 images with rows of 1024 / 2048 / 4096 points and 1024 .. 4096 rows -- the row transforms and
    the columns' first two Stockham stages in one pass (four rows per workgroup step), the remaining column stages as one column-tile
    pass with the frequency prefix K = 2 bits (512 rows: below one group per CU, the three-pass route; 8192 rows, round 6: the column pass
    on 2^11-point tiles of 8 columns).  Forward and inverse, EVERY
    value against the oracle's rows-then-columns (ndfft.rs:89-98) bit for bit; an impulse image with its exact answer (ndfft.rs tests)."""
    rng = seeded(5200 + rows + 3 * cols)
    x = rand_c(rng, (rows, cols))
    x[3, 5] = 0  # (a few exact zeros and a large value: sign-of-zero and cancellation paths)
    x[rows - 1, cols - 1] = 1e6

    def axis(a, ax, inverse=False):
        moved = np.moveaxis(a, ax, -1).copy(order="C")  # (a copy even when the axis is already last: the oracle works in place)
        oracle.fft_inplace_mt(moved, inverse=inverse)
        return np.ascontiguousarray(np.moveaxis(moved, -1, ax))
    want = axis(axis(x, 1), 0)
    data = x.reshape(-1).copy()
    fft32.fftnd(data, 1, rows, cols)
    assert bits_equal(data.reshape(rows, cols), want), f"fft2d {rows}x{cols}"
    fft32.fftnd(data, 1, rows, cols, inverse=True)
    back = axis(axis(want, 1, True), 0, True)
    assert bits_equal(data.reshape(rows, cols), back), f"ifft2d {rows}x{cols}"
    imp = np.zeros((rows, cols), np.complex64)
    imp[0, 0] = 2.5 - 1.0j
    d2 = imp.reshape(-1).copy()
    fft32.fftnd(d2, 1, rows, cols)
    assert np.all(d2 == np.complex64(2.5 - 1.0j))


@pytest.mark.parametrize("depth,rows,cols", [(2, 2, 2), (4, 8, 16), (16, 32, 64), (64, 4, 256)])
def test_fft3d_matches_oracle(fft32, fft64, oracle, depth, rows, cols):
    """ndfft::fft3d_inplace (ndfft.rs:114-155): z, then y, then x."""
    import kofft_amd as K

    rng = seeded(1600 + depth)
    for impl, dt in ((fft32, np.complex64), (fft64, np.complex128)):
        x = rand_c(rng, (depth, rows, cols), dt)
        want = _oracle_axis(oracle, _oracle_axis(oracle, _oracle_axis(oracle, x, 0), 1), 2)
        data = x.reshape(-1).copy()
        scratch = (np.zeros(depth, dt), np.zeros(rows, dt), np.zeros(cols, dt))
        K.fft3d_inplace(data, depth, rows, cols, impl, scratch)
        assert_parity(data.reshape(depth, rows, cols), want, f"fft3d {depth}x{rows}x{cols} {np.dtype(dt).name}", REL_TOL_F32)


def test_ndfft_error_variants(fft32):
    import kofft_amd as K
    from kofft_amd import FftError

    with pytest.raises(FftError) as e:
        K.fft2d_inplace(np.zeros(7, np.complex64), 2, 4, fft32, np.zeros(2, np.complex64))
    assert e.value.code == FftError.MismatchedLengths
    with pytest.raises(FftError) as e:
        K.fft2d_inplace(np.zeros(8, np.complex64), 2, 4, fft32, np.zeros(3, np.complex64))
    assert e.value.code == FftError.MismatchedLengths
    K.fft2d_inplace(np.zeros(0, np.complex64), 0, 4, fft32, np.zeros(0, np.complex64))  # Ok: nothing to do
    with pytest.raises(FftError):
        K.flatten_2d([[1, 2], [3]])


# ---- Bluestein arm: non-power-of-two lengths ---------------------------------------------------------------
@pytest.mark.parametrize("n", [3, 5, 6, 7, 12, 15, 100, 1000, 4095, 5000, 100_003])
def test_fft_non_power_of_two_matches_oracle(fft32, oracle, n):
    """fft.rs:1088-1132: a = x*chirp, fft_m, *fft(b), conj, fft_m, conj, /m, *chirp with m = next_pow2(2n-1).  The
    chirps come from the reference's f32 recipe on the host, the m-point transforms are the ordinary kernels."""
    rng = seeded(1700 + n)
    batch = 3 if n < 50_000 else 1
    x = rand_c(rng, (batch, n))
    y = x.copy()
    fft32.fft_batch(y)
    want = oracle.fft(x)
    assert_parity(y, want, f"bluestein fft c32 n={n}", REL_TOL_F32)
    fft32.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(want), f"bluestein ifft c32 n={n}", REL_TOL_F32)


@pytest.mark.parametrize("n", [3, 12, 15, 1000])
def test_fft_non_power_of_two_f64(fft64, oracle, n):
    rng = seeded(1800 + n)
    x = rand_c(rng, (2, n), np.complex128)
    y = x.copy()
    fft64.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"bluestein fft c64 n={n}", REL_TOL_F64)


@pytest.mark.parametrize("one_kernel", ["1", "0"])
def test_bluestein_one_launch_and_two_launch_routes(oracle, one_kernel, monkeypatch):
    """m = 32 ... 4096 (c32) / 1024 (c64) run the whole arm in one launch (bluestein_wg_kernel), larger m and
    KOFFT_HIP_BLUESTEIN_ONE=0 the two fused launches through a scratch: both must be the oracle's bytes, forward and
    inverse, at the m boundaries of either route (n = 17 -> m = 64, 2048 -> 4096, 2049 -> 8192, ...)."""
    import kofft_amd

    monkeypatch.setenv("KOFFT_HIP_BLUESTEIN_ONE", one_kernel)  # read when the context is created
    f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
    for n in (9, 17, 33, 100, 255, 500, 513, 1025, 2047, 2049, 4097):
        x = rand_c(seeded(2100 + n), (5, n))
        y = x.copy()
        f32.fft_batch(y)
        want = oracle.fft(x)
        assert_parity(y, want, f"bluestein route {one_kernel} fft c32 n={n}", REL_TOL_F32)
        f32.fft_batch(y, inverse=True)
        assert_parity(y, oracle.ifft(want), f"bluestein route {one_kernel} ifft c32 n={n}", REL_TOL_F32)
    for n in (9, 33, 100, 257, 511, 513, 1030):
        x = rand_c(seeded(2200 + n), (3, n), np.complex128)
        y = x.copy()
        f64.fft_batch(y)
        want = oracle.fft(x)
        assert_parity(y, want, f"bluestein route {one_kernel} fft c64 n={n}", REL_TOL_F64)
        f64.fft_batch(y, inverse=True)
        assert_parity(y, oracle.ifft(want), f"bluestein route {one_kernel} ifft c64 n={n}", REL_TOL_F64)


@pytest.mark.parametrize("dtype,n,batch", [
    # batch >= CUs x workgroups per CU x transforms per workgroup (complex_impl.hip.h: blue_persist_pays), several rounds of the grid
    ("c32", 12, 300_000 + 37), ("c32", 30, 150_000 + 5), ("c32", 60, 90_000 + 21), ("c32", 100, 60_000 + 9), ("c32", 250, 20_000 + 3),
    ("c32", 500, 14_000 + 7), ("c32", 1000, 5_000 + 3), ("c32", 1024 + 1, 2_500 + 2), ("c32", 2000, 2_500 + 1),
    ("c32", 4095, 1_100 + 1),
    ("c64", 12, 196_608 + 37), ("c64", 30, 98_304 + 5), ("c64", 60, 65_536 + 21), ("c64", 100, 20_000 + 5), ("c64", 250, 16_384 + 3),
    ("c64", 500, 5_000 + 1), ("c64", 1000, 2_500 + 3), ("c64", 2000, 1_100 + 1),
])
def test_bluestein_persistent_kernel_large_batches(oracle, dtype, n, batch):
    """Batches that give every workgroup of the chip-sized grid several transforms run bluestein_persist_kernel (twiddles and fft(b) in
    registers, next input prefetched): m = 32 ... 8192 (c32), m = 32 ... 4096 (c64), ragged last round, forward and inverse,
    first / last / middle rows against the oracle and every row against the one-workgroup-per-transform kernel (small batches)."""
    import kofft_amd

    cdt = np.complex64 if dtype == "c32" else np.complex128
    f = kofft_amd.HipFftImpl(np.float32 if dtype == "c32" else np.float64)
    x = rand_c(seeded(2300 + n), (batch, n), cdt)
    y = x.copy()
    f.fft_batch(y)
    pick = sorted({0, 1, 2, batch // 3, batch // 2, batch - 3, batch - 2, batch - 1})
    want = oracle.fft(x[pick])
    assert_parity(y[pick], want, f"bluestein persistent fft {dtype} n={n}", REL_TOL_F32 if dtype == "c32" else REL_TOL_F64)
    # every row: the same rows in small batches (the non-persistent kernel, itself checked against the oracle above and elsewhere)
    step = 64
    rows = np.arange(0, batch, max(1, batch // 2000))
    ref = np.concatenate([_small_batches(f, x[rows[i:i + step]]) for i in range(0, rows.size, step)])
    assert bits_equal(y[rows], ref), f"bluestein persistent vs per-transform kernel {dtype} n={n}"
    f.fft_batch(y, inverse=True)
    assert_parity(y[pick], oracle.ifft(want), f"bluestein persistent ifft {dtype} n={n}", REL_TOL_F32 if dtype == "c32" else REL_TOL_F64)


def _small_batches(f, rows):
    out = rows.copy()
    f.fft_batch(out)
    return out


def test_bluestein_reference_tests(fft32, fft64, oracle):
    # tests/bluestein.rs:32-66: n = 15 against the naive f32 DFT within 1e-3
    n = 15
    i = np.arange(n, dtype=np.float32)
    x = (i + 1j * (i * np.float32(0.5))).astype(np.complex64)
    y = x.copy()
    fft32.fft(y)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.all(np.abs(y - ref) < 1e-3)
    # lib.rs:267-282 test_fft_ifft_nonpow2_f32: n = 3 round trip within 1e-5
    d = np.array([1, 2, 3], np.complex64)
    fft32.fft(d)
    fft32.ifft(d)
    assert np.all(np.abs(d - np.array([1, 2, 3])) < 1e-5)
    # tests/split64.rs:20-34: n = 12 f64, SoA entry == AoS entry
    re = np.arange(12, dtype=np.float64)
    im = np.zeros(12)
    aos = np.arange(12).astype(np.complex128)
    fft64.fft(aos)
    fft64.fft_split(re, im)
    assert np.all(np.abs(aos.real - re) < 1e-10) and np.all(np.abs(aos.imag - im) < 1e-10)


def test_inverse_parallel_and_inverse_frame(fft32, oracle):
    """stft.rs:289-343 and 384-399 (+ the frame round trip of stft.rs:526-557)."""
    import kofft_amd as K

    rng = seeded(1900)
    signal = rng.uniform(-1, 1, 3000).astype(np.float32)
    window = oracle.hann(256)
    hop = 64
    nframes = -(-signal.size // hop)
    spec = oracle.stft(signal, window, hop, nframes)
    want = oracle.istft(spec, window, hop, signal.size)  # same sums; differs only where the norm is <= 1e-8 (sample 0 of a Hann window)
    keep = spec.copy()
    out = np.zeros(signal.size, np.float32)
    K.inverse_parallel(spec, window, hop, out, fft32)
    assert bits_equal(spec, keep)                        # frames are cloned, not transformed (stft.rs:310)
    assert out[0] == 0.0 and bits_equal(out[1:], want[1:])
    # frame / inverse_frame streaming round trip with a rectangular window (stft.rs:526-557)
    sig = np.arange(1, 9, dtype=np.float32)
    win = np.ones(4, np.float32)
    output = np.zeros(8, np.float32)
    norm = np.zeros(8, np.float32)
    buf = np.zeros(4, np.complex64)
    for pos in range(0, 8, 2):
        K.frame(sig, win, pos, buf, fft32)
        K.inverse_frame(buf, win, pos, output, fft32)
        for i in range(4):
            if pos + i < 8:
                norm[pos + i] += win[i] * win[i]
    output[norm > 1e-8] /= norm[norm > 1e-8]
    assert np.all(np.abs(output - sig) < 1e-4)


@pytest.mark.parametrize("n,batch,offset", [(2048, 8300, 1), (2048, 8192, 5), (4096, 4200, 3), (2048, 8192, 16)])
def test_rfft_streaming_unaligned_output_base(fft32, oracle, n, batch, offset):
    """The streaming rfft epilogue aligns its stores to 128-byte lines from the ABSOLUTE address of every output row:
    the result must not depend on where the caller's output buffer starts (here `offset` complex values into a
    device allocation), and nothing before or after the rows may be touched."""
    torch = pytest.importorskip("torch")
    rng = seeded(2100 + n + offset)
    x = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
    win = oracle.hann(n)
    rows = n // 2 + 1
    guard = 64
    d_in = torch.from_numpy(x).cuda()
    d_win = torch.from_numpy(win).cuda()
    d_out = torch.full((guard + offset + batch * rows + guard, 2), 7.5, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    base = d_out.data_ptr() + 8 * (guard + offset)
    fft32.rfft_dev(d_in.data_ptr(), base, d_win.data_ptr(), n, batch)
    fft32.synchronize()
    torch.cuda.synchronize()
    h = d_out.cpu().numpy()
    got = h[guard + offset:guard + offset + batch * rows].copy().view(np.complex64).reshape(batch, rows)
    assert_parity(got, oracle.rfft(x, win), f"unaligned rfft n={n} offset={offset}", REL_TOL_F32)
    assert np.all(h[:guard + offset] == 7.5) and np.all(h[guard + offset + batch * rows:] == 7.5)


@pytest.mark.parametrize("batch", [1024, 1031])
def test_n8192_streaming_paths(fft32, oracle, batch):
    """n = 8192 (512 threads per transform, four register passes, one workgroup per CU): complex forward / inverse and
    STFT with an 8192-sample window on the persistent kernel."""
    rng = seeded(990 + batch)
    x = rand_c(rng, (batch, 8192))
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"streaming fft c32 n=8192 batch={batch}", REL_TOL_F32)
    z = x.copy()
    fft32.fft_batch(z, inverse=True)
    assert_parity(z, oracle.ifft(x), f"streaming ifft c32 n=8192 batch={batch}", REL_TOL_F32)
    sig = rng.uniform(-1, 1, 2048 * batch + 77).astype(np.float32)
    w = oracle.hann(8192)
    frames = -(-sig.size // 2048)
    assert_parity(fft32.stft_into(sig, w, 2048, frames), oracle.stft(sig, w, 2048, frames), "streaming stft win=8192", REL_TOL_F32)
    # the magnitude policy rides the same kernel (the running maximum is carried across the delayed stores)
    mags, mx = fft32.stft_magnitudes(sig, 8192, 2048)
    wm, wmx = oracle.stft_magnitudes(sig, 8192, 2048)
    assert bits_equal(mags, wm) and mx == wmx


@pytest.mark.parametrize("batch", [1024, 1025, 1279, 1281, 2048 + 3])
def test_n8192_wave_split_kernel_around_the_grid(fft32, oracle, batch):
    """fft_split_persist_kernel (n = 8192: one workgroup per CU, every wavefront owning 1024 points, one s_barrier per
    transform, results stored one step late): batch sizes around multiples of the grid (256 workgroups), so that workgroups
    with 4 and 5 transforms, odd and even counts (the two LDS buffers alternate) and the empty-descriptor prefetch all occur;
    both settings of KOFFT_HIP_SPLIT agree bit for bit with the oracle."""
    import os

    import kofft_amd

    rng = seeded(7100 + batch)
    x = rand_c(rng, (batch, 8192))
    want = oracle.fft(x)
    for split in ("1", "0"):
        os.environ["KOFFT_HIP_SPLIT"] = split
        try:
            f = kofft_amd.HipFftImpl(np.float32)
        finally:
            del os.environ["KOFFT_HIP_SPLIT"]
        y = x.copy()
        f.fft_batch(y)
        assert bits_equal(y, want), f"KOFFT_HIP_SPLIT={split} batch={batch}"
        f.fft_batch(y, inverse=True)
        assert bits_equal(y, oracle.ifft(want)), f"inverse KOFFT_HIP_SPLIT={split} batch={batch}"


@pytest.mark.parametrize("n,batch", [(8192, 1024), (8192, 1025), (8192, 1281), (4096, 2048), (4096, 2561)])
def test_c64_persistent_kernels_around_the_grid(oracle, n, batch):
    """c64 n = 8192 from num_cus * 4 transforms up (n = 4096: num_cus * 8): fft_persist_kernel<double, 13 / 12> (one 512-thread /
    two 256-thread workgroups per CU, the next transform's loads in flight, table entries from global memory in every pass) --
    bit for bit the generic kernel's and the oracle's results, forward and inverse, with workgroups of 4 and 5 transforms and
    the empty-descriptor prefetch."""
    import os

    import kofft_amd

    rng = seeded(7300 + batch)
    x = rand_c(rng, (batch, n), np.complex128)
    want = oracle.fft(x)
    for persist in ("1", "0"):
        os.environ["KOFFT_HIP_PERSIST64"] = persist
        try:
            f = kofft_amd.HipFftImpl(np.float64)
        finally:
            del os.environ["KOFFT_HIP_PERSIST64"]
        y = x.copy()
        f.fft_batch(y)
        assert bits_equal(y, want), f"KOFFT_HIP_PERSIST64={persist} batch={batch}"
        f.fft_batch(y, inverse=True)
        assert bits_equal(y, oracle.ifft(want)), f"inverse KOFFT_HIP_PERSIST64={persist} batch={batch}"


def test_istft_stream_reconstructs_and_flushes(fft32, oracle):
    """tests/istft_stream.rs:4-52 and stft.rs:679-690, against the mirror's IstftStream."""
    import kofft_amd as K

    signal = np.arange(1, 9, dtype=np.float32)
    win_len, hop = 4, 2
    window = np.ones(win_len, np.float32)
    sstream = K.StftStream(signal, window, hop, fft32)
    istream = K.IstftStream(win_len, hop, window, fft32)
    frame = np.zeros(win_len, np.complex64)
    frames, out_stream = [], []
    while sstream.next_frame(frame):
        frames.append(frame.copy())
        out_stream.append(istream.push_frame(frame))
    tail = istream.flush()
    out_stream = np.concatenate(out_stream + [tail])
    offline = np.zeros(signal.size + win_len - hop, np.float32)
    scratch = np.zeros_like(offline)
    K.istft([f.copy() for f in frames], window, hop, offline, scratch, fft32)
    assert bits_equal(out_stream[:signal.size], offline[:signal.size])
    assert tail.size == win_len - hop and bits_equal(tail, offline[signal.size:])
    assert istream.flush().size == 0                      # subsequent calls return an empty slice
    assert K.IstftStream(win_len, hop, window, fft32).flush().size == 0   # no frames processed
    with pytest.raises(K.FftError) as e:
        istream.push_frame(np.zeros(win_len - 1, np.complex64))          # stft.rs:679-690
    assert e.value.code == K.FftError.MismatchedLengths
    with pytest.raises(K.FftError) as e:
        K.IstftStream(win_len, 0, window, fft32)
    assert e.value.code == K.FftError.InvalidHopSize
    # a longer stream with a Hann window: equals the batch istft bit for bit where both are defined
    rng = seeded(3100)
    sig = rng.uniform(-1, 1, 2000).astype(np.float32)
    w = oracle.hann(256)
    nfr = -(-sig.size // 64)
    spec = oracle.stft(sig, w, 64, nfr)
    st2 = K.IstftStream(256, 64, w, fft32)
    got = np.concatenate([st2.push_frame(spec[i]) for i in range(nfr)] + [st2.flush()])
    want = oracle.istft(spec.copy(), w, 64, got.size)
    assert bits_equal(got, want)


@pytest.mark.parametrize("length,win_len,hop", [(12_000_001, 64, 5_000_000), (70_000, 256, 300), (3000, 1024, 1), (100_000, 16, 7),
                                                (9_000_000, 2048, 4_500_000)])
def test_stft_unusual_hops(fft32, oracle, length, win_len, hop):
    """Hops larger than the window (gaps), hop = 1 (maximal overlap), hops that are not a multiple of anything, and hops
    beyond the reach of a workgroup's 32-bit buffer offsets (the kernels fall back to per-element addressing there)."""
    rng = seeded(4000 + win_len + hop % 1000)
    signal = rng.uniform(-1, 1, length).astype(np.float32)
    window = oracle.hann(win_len)
    frames = -(-length // hop)
    got = fft32.stft_into(signal, window, hop, frames)
    want = oracle.stft(signal, window, hop, frames)
    assert_parity(got, want, f"stft len={length} win={win_len} hop={hop}", REL_TOL_F32)


@pytest.mark.parametrize("batch", [16384, 16391])
def test_n512_streaming_paths(fft32, oracle, batch):
    """n = 512 on the persistent kernel (8 points per thread, one wavefront per transform, three passes of three
    stages): complex forward / inverse, rfft 1024 (+ window, aligned epilogue with 8 registers) and STFT 512."""
    rng = seeded(970 + batch)
    x = rand_c(rng, (batch, 512))
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"streaming fft c32 n=512 batch={batch}", REL_TOL_F32)
    z = x.copy()
    fft32.fft_batch(z, inverse=True)
    assert_parity(z, oracle.ifft(x), f"streaming ifft c32 n=512 batch={batch}", REL_TOL_F32)
    r = rng.uniform(-1, 1, (batch, 1024)).astype(np.float32)
    win = oracle.hann(1024)
    assert_parity(fft32.rfft_batch(r, win), oracle.rfft(r, win), f"streaming rfft n=1024 batch={batch}", REL_TOL_F32)
    assert_parity(fft32.rfft_batch(r), oracle.rfft(r, None), f"streaming rfft n=1024 (no window) batch={batch}", REL_TOL_F32)
    sig = rng.uniform(-1, 1, 128 * batch + 33).astype(np.float32)
    w2 = oracle.hann(512)
    frames = -(-sig.size // 128)
    assert_parity(fft32.stft_into(sig, w2, 128, frames), oracle.stft(sig, w2, 128, frames), "streaming stft win=512", REL_TOL_F32)


@pytest.mark.parametrize("n,batch", [(1024, 16390), (2048, 8200), (4096, 4101), (8192, 1029)])
def test_irfft_streaming_paths(fft32, oracle, n, batch):
    """irfft on the persistent kernel: input[k] prefetched through the row's descriptor, input[m-k] from the partner lane
    (m <= 1024: ds_bpermute) or from a natural-order LDS copy of the row (m = 2048, 4096: more than one wavefront per
    transform), the W table in LDS, k = 0 selected branch-free; rows are (m+1)*8 bytes, 8-byte aligned."""
    rng = seeded(5000 + n)
    spec = rand_c(rng, (batch, n // 2 + 1))
    spec[:, 0].imag = 0
    spec[:, -1].imag = 0
    got = fft32.irfft_batch(spec, n)
    want = oracle.irfft(spec, n)
    assert_parity(got, want, f"streaming irfft n={n} batch={batch}", REL_TOL_F32)
    # Hermitian-consistent input round-trips through rfft within the table drift
    x = rng.uniform(-1, 1, (64, n)).astype(np.float32)
    back = fft32.irfft_batch(fft32.rfft_batch(x), n)
    assert np.max(np.abs(back - x)) < 5e-4


@pytest.mark.parametrize("n,batch", [(256, 32768), (256, 32771), (128, 65536), (128, 65539), (64, 131075)])
def test_small_n_streaming_paths(fft32, oracle, n, batch):
    """n = 256 / 128 / 64 on the persistent kernel: 2 / 4 / 4 transforms per wavefront behind one group descriptor (batches that
    are not a multiple of the group leave a partly filled last group): complex, rfft (+ window, aligned epilogue with
    per-lane row offsets), irfft (reversed second load) and STFT."""
    rng = seeded(6000 + n + batch % 7)
    x = rand_c(rng, (batch, n))
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), f"streaming fft c32 n={n} batch={batch}", REL_TOL_F32)
    z = x.copy()
    fft32.fft_batch(z, inverse=True)
    assert_parity(z, oracle.ifft(x), f"streaming ifft c32 n={n} batch={batch}", REL_TOL_F32)
    r = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
    win = oracle.hann(2 * n)
    assert_parity(fft32.rfft_batch(r, win), oracle.rfft(r, win), f"streaming rfft n={2 * n} batch={batch}", REL_TOL_F32)
    spec = rand_c(rng, (batch, n + 1))
    assert_parity(fft32.irfft_batch(spec, 2 * n), oracle.irfft(spec, 2 * n), f"streaming irfft n={2 * n} batch={batch}", REL_TOL_F32)
    hop = n // 4
    sig = rng.uniform(-1, 1, hop * batch + 5).astype(np.float32)
    w2 = oracle.hann(n)
    frames = -(-sig.size // hop) + 1
    assert_parity(fft32.stft_into(sig, w2, hop, frames, check_frames=False), oracle.stft_range(sig, w2, hop, 0, frames),
                  f"streaming stft win={n}", REL_TOL_F32)
    part = sig[:hop * 33000 + 3]
    nfr = -(-part.size // hop)
    mags, mx = fft32.stft_magnitudes(part, n, hop)
    full = oracle.stft_range(part, w2, hop, 0, nfr)[:, :n // 2]
    want = np.sqrt(full.real * full.real + full.imag * full.imag, dtype=np.float32)
    assert bits_equal(mags, want) and mx == want.max()


@pytest.mark.parametrize("batch,windowed", [(1024, True), (1027, False)])
def test_rfft_8192_streaming_path(fft32, oracle, batch, windowed):
    """rfft n = 8192 (inner 4096-point transform) on the persistent kernel: window pairs in registers, post-pass table in
    LDS, line-aligned epilogue with 256 threads per row."""
    rng = seeded(7000 + batch)
    x = rng.uniform(-1, 1, (batch, 8192)).astype(np.float32)
    win = oracle.hann(8192) if windowed else None
    assert_parity(fft32.rfft_batch(x, win), oracle.rfft(x, win), f"streaming rfft n=8192 batch={batch}", REL_TOL_F32)


def test_fft_plan(fft32, fft64, oracle):
    """fft.rs:2361-2387 (plan fft/ifft, out of place), 2581-2610 (length mismatches), over the mirror's FftPlan."""
    import kofft_amd as K

    plan = K.FftPlan(4, K.FftStrategy.SplitRadix, fft32)
    data = np.array([1, 2, 3, 4], np.complex64)
    orig = data.copy()
    plan.fft(data)
    assert bits_equal(data, oracle.fft(orig))
    plan.ifft(data)
    assert np.all(np.abs(data.real - orig.real) < 1e-4)
    out = np.zeros(4, np.complex64)
    plan.fft_out_of_place(np.ones(4, np.complex64), out)
    assert bits_equal(out, oracle.fft(np.ones(4, np.complex64)))
    out2 = np.zeros(4, np.complex64)
    plan.ifft_out_of_place(out, out2)
    assert bits_equal(out2, oracle.ifft(out))
    p2 = K.FftPlan(4, K.FftStrategy.Radix2, fft32)
    for call in (lambda: p2.fft(np.zeros(3, np.complex64)), lambda: p2.ifft(np.zeros(3, np.complex64)),
                 lambda: p2.fft_out_of_place(np.zeros(4, np.complex64), np.zeros(3, np.complex64)),
                 lambda: p2.ifft_out_of_place(np.zeros(3, np.complex64), np.zeros(4, np.complex64))):
        with pytest.raises(K.FftError) as e:
            call()
        assert e.value.code == K.FftError.MismatchedLengths
    # an f32 plan is the Stockham transform for EVERY strategy: Radix2 / Radix4 take the *_with_twiddles shortcut
    # (fft.rs:2016-2035 -> stockham_fft, fft.rs:1645-1660), SplitRadix / Auto reach it through fft_with_strategy
    rng = seeded(8100)
    x = rand_c(rng, (64,))
    want = oracle.fft(x)
    for strat in (K.FftStrategy.Radix2, K.FftStrategy.Radix4, K.FftStrategy.SplitRadix, K.FftStrategy.Auto):
        y = x.copy()
        K.FftPlan(64, strat, fft32).fft(y)
        assert bits_equal(y, want)
        K.FftPlan(64, strat, fft32).ifft(y)
        assert bits_equal(y, oracle.ifft(want))
    # an f64 plan has no shortcut: Radix4 -> fft_with_strategy -> fft_radix4 (fft.rs:2037, 1356), the reference's bytes --
    # which from n = 16 are not the DFT
    xd = rand_c(rng, (256,), np.complex128)
    for strat in (K.FftStrategy.Radix2, K.FftStrategy.SplitRadix, K.FftStrategy.Auto):
        yd = xd.copy()
        K.FftPlan(256, strat, fft64).fft(yd)
        assert bits_equal(yd, oracle.fft(xd))
    yd = xd.copy()
    K.FftPlan(256, K.FftStrategy.Radix4, fft64).fft(yd)
    assert bits_equal(yd, oracle.fft_radix4(xd[None])[0])
    assert np.abs(yd - np.fft.fft(xd)).max() > 1e-3
    # ... and with the opt-out every strategy is the true transform
    plain = K.HipFftImpl(np.float64, radix4_compat=False)
    yd = xd.copy()
    K.FftPlan(256, K.FftStrategy.Radix4, plain).fft(yd)
    assert bits_equal(yd, oracle.fft(xd))
    assert np.abs(yd - np.fft.fft(xd)).max() < 1e-10


def test_host_pointer_pipeline_large_batches(fft32, fft64, oracle):
    """Host-pointer batches of >= 128 MiB go through the device in eight overlapped chunks (upload / kernel / download
    on three streams, downloads from a helper thread): results must be those of one pass.  Ragged last chunk included."""
    rng = seeded(9500)
    x = rand_c(rng, (16387, 1024))                       # 134 MB each way, 8 chunks of 2049 rows (last: 2044)
    y = x.copy()
    fft32.fft_batch(y)
    assert_parity(y, oracle.fft(x), "pipelined host fft c32", REL_TOL_F32)
    fft32.fft_batch(y, inverse=True)
    assert_parity(y, oracle.ifft(oracle.fft(x)), "pipelined host ifft c32", REL_TOL_F32)
    r = rng.uniform(-1, 1, (12301, 4096)).astype(np.float32)   # 201 MB in, 202 MB out
    win = oracle.hann(4096)
    spec = fft32.rfft_batch(r, win)
    assert_parity(spec, oracle.rfft(r, win), "pipelined host rfft", REL_TOL_F32)
    assert_parity(fft32.irfft_batch(spec, 4096), oracle.irfft(spec, 4096), "pipelined host irfft", REL_TOL_F32)
    xd = rand_c(rng, (4099, 2048), np.complex128)        # 134 MB each way in f64
    yd = xd.copy()
    fft64.fft_batch(yd)
    assert_parity(yd, oracle.fft(xd), "pipelined host fft c64", REL_TOL_F64)


@pytest.mark.parametrize("log2n,batch", [(15, 1), (16, 3), (17, 1), (18, 2), (19, 1), (20, 1)])
def test_single_large_transforms_narrow_tiles(fft32, fft64, oracle, log2n, batch, monkeypatch):
    """One large transform (or a handful) runs its factors on narrower tiles so that every CU gets a workgroup
    (launch_sub_one_tile: 8 -> 4 -> 2 columns); KOFFT_HIP_BIG_NARROW=0 keeps the full-width tiles.  Same bytes either way."""
    import kofft_amd

    n = 1 << log2n
    x32 = rand_c(seeded(9700 + log2n), (batch, n))
    x64 = rand_c(seeded(9800 + log2n), (batch, n), np.complex128)
    w32, w64 = oracle.fft(x32), oracle.fft(x64)
    for narrow in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_BIG_NARROW", narrow)  # read when the context is created
        f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
        y = x32.copy()
        f32.fft_batch(y)
        assert_parity(y, w32, f"single large c32 2^{log2n} x {batch} narrow={narrow}", REL_TOL_F32)
        f32.fft_batch(y, inverse=True)
        assert_parity(y, oracle.ifft(w32), f"single large c32 inverse 2^{log2n} narrow={narrow}", REL_TOL_F32)
        y = x64.copy()
        f64.fft_batch(y)
        assert_parity(y, w64, f"single large c64 2^{log2n} x {batch} narrow={narrow}", REL_TOL_F64)


def test_host_pointer_stft_large(fft32, oracle):
    """stft() from host memory with >= 128 MiB of spectra (one upload, one launch, one download: chunking the download
    behind the kernels measured slower) -- frames past the end of the signal included."""
    import kofft_amd as K

    rng = seeded(9600)
    hop, win_len = 256, 1024
    sig = rng.uniform(-1, 1, 16_500 * hop + 77).astype(np.float32)
    win = oracle.hann(win_len)
    frames = -(-sig.size // hop) + 5  # five frames wholly past the end: zero spectra (stft.rs:95-99)
    got = fft32.stft_into(sig, win, hop, frames)
    assert got.nbytes >= 128 << 20
    assert_parity(got, oracle.stft(sig, win, hop, frames), "large host stft", REL_TOL_F32)


@pytest.mark.parametrize("depth,rows,cols", [(1, 2048, 16), (1, 1024, 64), (1, 4096, 128), (1024, 2, 8), (4, 1024, 32), (2, 2048, 4),
                                             (1, 4096, 512), (2, 4096, 256), (4096, 2, 256), (1, 8192, 256), (1, 16384, 64), (3, 4096, 192),
                                             (1, 2048, 1024), (2048, 4, 256), (2, 2048, 512)])
def test_ndfft_long_strided_axes(fft32, fft64, oracle, depth, rows, cols):
    """Long axes that are not contiguous: the strided kernel up to 2048 points; from 4096 points (and 16 MiB of data) two
    column-tile passes with the axis's own table when the number of adjacent lines is a power of two (round 4: AxisLastIO; 2^12 =
    2^5 x 2^7, 2^13 = 2^6 x 2^7, 2^14 = 2^7 x 2^7, one case with two outer blocks; c32 axes of 2^11 points = 2^6 x 2^5 too), otherwise transpose -> batched transform ->
    transpose back (the 192-column case, every f64 axis of 16384 points).  Every line is still the reference's 1-D transform."""
    rng = seeded(1700 + depth + rows + cols)
    for impl, dt, tol in ((fft32, np.complex64, REL_TOL_F32), (fft64, np.complex128, REL_TOL_F64)):
        x = rand_c(rng, (depth, rows, cols), dt)
        if depth > 1:   # z, y, x (ndfft.rs:131-151)
            want = _oracle_axis(oracle, _oracle_axis(oracle, _oracle_axis(oracle, x, 0), 1), 2)
        else:           # rows, then columns (ndfft.rs:89-98)
            want = _oracle_axis(oracle, _oracle_axis(oracle, x, 2), 1)
        data = x.reshape(-1).copy()
        impl.fftnd(data, depth, rows, cols)
        assert_parity(data.reshape(depth, rows, cols), want, f"fftnd {depth}x{rows}x{cols} {np.dtype(dt).name}", tol)
        impl.fftnd(data, depth, rows, cols, inverse=True)
        # (the f64 kernels for n <= 16 carry the reference's f32-literal constants: 1e-7, not 1e-13)
        # (16384-point f32 axes: the reference's recurrence table has drifted 2e-4 by then)
        assert np.max(np.abs(data.reshape(depth, rows, cols) - x)) < ((2e-3 if max(depth, rows, cols) < 16384 else 8e-3) if dt == np.complex64 else 1e-6)


@pytest.mark.parametrize("n,batch", [(4096, 4095), (4096, 4097), (8192, 1023), (8192, 1025), (2048, 8191), (2048, 8193)])
def test_irfft_dispatch_boundaries(fft32, oracle, n, batch):
    """Either side of the batch sizes where irfft moves from the generic kernel (row staged once through LDS) to the persistent
    ones (partner lane for m <= 1024, natural-order LDS copy for m = 2048 / 4096): the same bytes on both sides."""
    rng = seeded(9900 + n + batch)
    spec = rand_c(rng, (batch, n // 2 + 1))
    spec[:, 0].imag = 0
    spec[:, -1].imag = 0
    pick = sorted({0, 1, batch // 3, batch - 2, batch - 1})
    got = fft32.irfft_batch(spec, n)
    assert_parity(got[pick], oracle.irfft(spec[pick], n), f"irfft n={n} batch={batch}", REL_TOL_F32)


@pytest.mark.parametrize("dtype,log2n,batches", [("c32", 17, (1, 2, 3, 5, 8, 9)), ("c32", 15, (1, 4, 15, 17, 33)), ("c64", 16, (1, 2, 7, 9, 31, 33))])
def test_large_n_batches_around_the_tile_width_steps(fft32, fft64, oracle, dtype, log2n, batches):
    """Batch counts on either side of the points where the large-n factors switch tile width (every CU must get a workgroup)
    and from the one-tile-per-workgroup kernels to the persistent ones."""
    n = 1 << log2n
    cdt = np.complex128 if dtype == "c64" else np.complex64
    fft = fft64 if dtype == "c64" else fft32
    tol = REL_TOL_F64 if dtype == "c64" else REL_TOL_F32
    rng = seeded(9950 + log2n)
    x = rand_c(rng, (max(batches), n), cdt)
    want = oracle.fft(x)
    for b in batches:
        y = x[:b].copy()
        fft.fft_batch(y)
        assert_parity(y, want[:b], f"{dtype} 2^{log2n} batch {b}", tol)


@pytest.mark.parametrize("batch", [1024, 1283])
def test_n16384_wave_split_kernels(fft32, oracle, batch):
    """n = 16384.  Default: fft_split_wide_persist_kernel (512 threads, 32 points each, a wavefront owns 2048 points = 128-byte
    runs both ways; one LDS buffer, two barriers, pass B1's 31 table entries resident in registers).  KOFFT_HIP_SPLIT=0: the
    generic kernel.  Complex forward / inverse, STFT and magnitudes with a 16384-sample window, both routes against the oracle
    bit for bit; workgroups with 4, 5 and 6 transforms."""
    import os

    import kofft_amd

    rng = seeded(7300 + batch)
    x = rand_c(rng, (batch, 16384))
    want = oracle.fft(x)
    sig = rng.uniform(-1, 1, 4096 * batch + 77).astype(np.float32)
    w = oracle.hann(16384)
    frames = -(-sig.size // 4096)
    want_stft = oracle.stft(sig, w, 4096, frames)
    wm, wmx = oracle.stft_magnitudes(sig, 16384, 4096)
    for knob, val in (("KOFFT_HIP_SPLIT", "1"), ("KOFFT_HIP_SPLIT", "0")):  # the default route, then the generic kernel
        os.environ[knob] = val
        try:
            f = kofft_amd.HipFftImpl(np.float32)
        finally:
            del os.environ[knob]
        y = x.copy()
        f.fft_batch(y)
        assert bits_equal(y, want), f"{knob}={val} batch={batch}"
        f.fft_batch(y, inverse=True)
        assert bits_equal(y, oracle.ifft(want)), f"inverse {knob}={val} batch={batch}"
        assert bits_equal(f.stft_into(sig, w, 4096, frames), want_stft), f"stft {knob}={val}"
        mags, mx = f.stft_magnitudes(sig, 16384, 4096)
        assert bits_equal(mags, wm) and mx == wmx, f"stft_magnitudes {knob}={val}"


@pytest.mark.parametrize("batch", [1024, 1285])
def test_rfft_irfft_n16384_persistent_kernels(oracle, batch):
    """rfft / irfft of 16384 reals from num_cus * 4 rows up: fft_persist_kernel<float, 13> with the rfft epilogue (post-pass table in
    LDS beside the 8192-point exchange buffer) and with irfft's paired input through the exchange buffer -- against the oracle
    and the generic kernels (KOFFT_HIP_RFFT13_PERSIST=0), plain and with a row window."""
    import os

    import kofft_amd

    rng = seeded(7500 + batch)
    x = rng.uniform(-1, 1, (batch, 16384)).astype(np.float32)
    win = rng.uniform(0.1, 1, 16384).astype(np.float32)
    want = oracle.rfft(x)
    want_w = oracle.rfft(x, win)
    spec = rand_c(rng, (batch, 8193))
    want_inv = oracle.irfft(spec, 16384)
    for persist in ("1", "0"):
        os.environ["KOFFT_HIP_RFFT13_PERSIST"] = persist
        try:
            f = kofft_amd.HipFftImpl(np.float32)
        finally:
            del os.environ["KOFFT_HIP_RFFT13_PERSIST"]
        assert bits_equal(f.rfft_batch(x), want), f"rfft KOFFT_HIP_RFFT13_PERSIST={persist} batch={batch}"
        assert bits_equal(f.rfft_batch(x, win), want_w), f"windowed rfft KOFFT_HIP_RFFT13_PERSIST={persist} batch={batch}"
        assert bits_equal(f.irfft_batch(spec, 16384), want_inv), f"irfft KOFFT_HIP_RFFT13_PERSIST={persist} batch={batch}"


@pytest.mark.parametrize("batch", [1024, 1100])
def test_rfft_irfft_n32768_inside_the_wave_split_kernel(oracle, batch):
    """rfft of 32768 reals from num_cus * 4 rows up: fft_split_wide_persist_kernel with the post-pass (rfft.rs:450-463) as its
    epilogue -- results back into the thread's own row cells, one more barrier, X[k] from Y[k], Y[m-k] (XOR-addressed; the K = 0
    threads pair differently) and W[k], stores rotated onto whole lines (all 16 row alignments occur); with a row window the
    window pairs are re-read per transform.  irfft of 32768 reals on the same kernel: the input row staged through the exchange
    buffer so that every thread reads its partners input[m-k] back (pre-pass, rfft.rs:487-506).  Against the oracle and the
    generic kernels (KOFFT_HIP_RFFT14_WIDE=0)."""
    import os

    import kofft_amd

    rng = seeded(7900 + batch)
    x = rng.uniform(-1, 1, (batch, 32768)).astype(np.float32)
    want = oracle.rfft(x)
    win = rng.uniform(0.1, 1, 32768).astype(np.float32)
    want_w = oracle.rfft(x, win)
    spec = rand_c(rng, (batch, 16385))
    want_inv = oracle.irfft(spec, 32768)
    for wide in ("1", "0"):
        os.environ["KOFFT_HIP_RFFT14_WIDE"] = wide
        try:
            f = kofft_amd.HipFftImpl(np.float32)
        finally:
            del os.environ["KOFFT_HIP_RFFT14_WIDE"]
        assert bits_equal(f.rfft_batch(x), want), f"KOFFT_HIP_RFFT14_WIDE={wide} batch={batch}"
        assert bits_equal(f.rfft_batch(x, win), want_w), f"windowed, KOFFT_HIP_RFFT14_WIDE={wide} batch={batch}"
        assert bits_equal(f.irfft_batch(spec, 32768), want_inv), f"irfft, KOFFT_HIP_RFFT14_WIDE={wide} batch={batch}"


@pytest.mark.parametrize("win_len,frames", [(32, 300), (64, 3000), (256, 40000), (1024, 9000), (4096, 2100), (8192, 1100), (16384, 1030)])
def test_stft_keeps_special_values(fft32, oracle, win_len, frames):
    """The reference multiplies every input (x * w, +0) by the first stage's T[0] = (1, 0) like any other entry, so an Inf or NaN
    sample puts a NaN into the IMAGINARY part (Inf * 0) and a -0.0 sample keeps its sign.  Signals salted with +-Inf, NaN, +-0,
    denormals and huge values, one kernel family per size (one thread per transform, persistent, wave-split): NaN positions and
    every other bit equal to the oracle's.  (A real-input first stage -- 3 plain instructions instead of 5 packed ones, exact
    under this test -- was measured in round 3: +-1 % on STFT 1024 .. 16384, not kept.)"""
    rng = seeded(7700 + win_len)
    hop = max(1, win_len // 4)
    length = hop * frames - min(3, hop - 1)  # stft.rs:81-84: frames == ceil(len / hop); the last frames run past the end
    sig = rng.uniform(-1, 1, length).astype(np.float32)
    specials = np.array([np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-42, -1e-42, 3e38, -3e38], np.float32)
    where = rng.integers(0, length, max(16, length // 997))
    sig[where] = specials[rng.integers(0, specials.size, where.size)]
    sig[: win_len // 2] = 0.0  # a stretch of exact zeros: sums that cancel to +-0
    sig[win_len // 2: win_len] = -0.0
    window = rng.uniform(0, 1, win_len).astype(np.float32)
    window[rng.integers(0, win_len, 3)] = 0.0
    got = fft32.stft_into(sig, window, hop, frames)
    want = oracle.stft(sig, window, hop, frames)
    g, w = got.view(np.float32), np.asarray(want).view(np.float32)
    ng, nw = np.isnan(g), np.isnan(w)
    assert nw.any() and np.array_equal(ng, nw)
    assert bits_equal(np.where(ng, np.float32(0), g), np.where(nw, np.float32(0), w))


# ---- FftStrategy::Radix4: the reference's bytes BY DEFAULT (strict drop-in) ---------------------------
def _plan_ifft_reference(oracle, x, forward):
    """FftPlan::ifft (fft.rs:2040-2055) around ``forward``: conj, fft, conj * 1/(n as f32 -> T) -- IEEE elementwise."""
    real = np.float32 if x.dtype == np.complex64 else np.float64
    y = forward(np.conj(x))
    scale = real(1) / real(np.float32(x.shape[-1]))
    out = np.empty_like(y)
    out.real = y.real * scale
    out.imag = (-y.imag) * scale
    return out


@pytest.mark.parametrize("n", [1, 4, 16, 64, 256, 1024, 4096, 65536, 1 << 20, 8, 32, 12])
def test_radix4_compat_reproduces_the_reference_arm(oracle, n, monkeypatch):
    """A DEFAULT-CONSTRUCTED implementation (no flag, no environment) sends fft_with_strategy(.., Radix4) to
    ScalarFftImpl::fft_radix4 (fft.rs:1356, 1455-1548) and returns its bytes, f32 and f64: powers of four run the reference's
    swap loop, butterfly4 and running-product twiddles; other lengths fall back to fft() (fft.rs:1457-1460).  FftPlan
    follows fft.rs:2012-2055: the f32 Radix4 plan is the Stockham transform (the *_with_twiddles shortcut), the f64 one is
    fft_radix4, and plan.ifft is conj / that fft / conj * scale.  radix4_compat=False / KOFFT_HIP_RADIX4_COMPAT=0 opts out."""
    import kofft_amd as K

    monkeypatch.delenv("KOFFT_HIP_RADIX4_COMPAT", raising=False)
    pow4 = n >= 16 and (n & (n - 1)) == 0 and (n.bit_length() - 1) % 2 == 0
    for dt, cdt in ((np.float32, np.complex64), (np.float64, np.complex128)):
        rng = seeded(8800 + n % 9973)
        x = rand_c(rng, (3, n), cdt)
        want = oracle.fft_radix4(x)
        dflt = K.HipFftImpl(dt)
        assert dflt.radix4_compat is True
        y = x.copy()
        dflt.fft_radix4_batch(y)
        assert bits_equal(y, want), f"fft_radix4 {cdt.__name__} n={n}"
        z = x[0].copy()
        dflt.fft_with_strategy(z, K.FftStrategy.Radix4)
        assert bits_equal(z, want[0])
        for strat in (K.FftStrategy.Radix2, K.FftStrategy.SplitRadix, K.FftStrategy.Auto):
            z = x[0].copy()
            dflt.fft_with_strategy(z, strat)
            assert bits_equal(z, oracle.fft(x[:1])[0])
        # FftPlan::fft / ifft
        plan = K.FftPlan(n, K.FftStrategy.Radix4, dflt)
        z = x[1].copy()
        plan.fft(z)
        plan_fwd = (lambda a: oracle.fft(a)) if dt == np.float32 else (lambda a: oracle.fft_radix4(a))
        assert bits_equal(z, plan_fwd(x[1:2])[0]), f"plan.fft {cdt.__name__} n={n}"
        z = x[2].copy()
        plan.ifft(z)
        assert bits_equal(z, _plan_ifft_reference(oracle, x[2:3], plan_fwd)[0]), f"plan.ifft {cdt.__name__} n={n}"
        if pow4:
            assert not bits_equal(want[0], oracle.fft(x[:1])[0])  # the reference's arm is not the transform
        # the opt-out: flag, then environment
        plain = K.HipFftImpl(dt, radix4_compat=False)
        v = x[0].copy()
        plain.fft_with_strategy(v, K.FftStrategy.Radix4)
        assert bits_equal(v, oracle.fft(x[:1])[0])
        v = x[0].copy()
        K.FftPlan(n, K.FftStrategy.Radix4, plain).ifft(v)
        assert bits_equal(v, oracle.ifft(x[:1])[0])
    monkeypatch.setenv("KOFFT_HIP_RADIX4_COMPAT", "0")
    assert K.HipFftImpl(np.float32).radix4_compat is False
    monkeypatch.setenv("KOFFT_HIP_RADIX4_COMPAT", "1")
    assert K.HipFftImpl(np.float32).radix4_compat is True


def test_radix4_compat_env_and_device_pointers(oracle, monkeypatch):
    import torch

    import kofft_amd as K

    monkeypatch.delenv("KOFFT_HIP_RADIX4_COMPAT", raising=False)
    f = K.HipFftImpl(np.float32)
    assert f.radix4_compat is True
    x = rand_c(seeded(8899), (5, 1024))
    want = oracle.fft_radix4(x)
    d = torch.from_numpy(x.view(np.float32).reshape(5, 1024, 2)).to("cuda")
    o = torch.empty_like(d)
    lib = K.load_library()
    assert lib.kofft_hip_fft_radix4_c32_dev(f._ctx, C_void(d.data_ptr()), C_void(o.data_ptr()), 1024, 5) == 0
    assert lib.kofft_hip_fft_radix4_c32_dev(f._ctx, C_void(d.data_ptr()), C_void(d.data_ptr()), 1024, 5) == 0  # in place
    f.synchronize() if hasattr(f, "synchronize") else lib.kofft_hip_synchronize(f._ctx)
    assert bits_equal(o.cpu().numpy().view(np.complex64).reshape(5, 1024), want)
    assert bits_equal(d.cpu().numpy().view(np.complex64).reshape(5, 1024), want)
    # the plan's inverse loop, device pointers, out of place and in place
    d = torch.from_numpy(x.view(np.float32).reshape(5, 1024, 2)).to("cuda")
    wanti = _plan_ifft_reference(oracle, x, oracle.fft_radix4)
    assert lib.kofft_hip_ifft_radix4_c32_dev(f._ctx, C_void(d.data_ptr()), C_void(o.data_ptr()), 1024, 5) == 0
    assert lib.kofft_hip_ifft_radix4_c32_dev(f._ctx, C_void(d.data_ptr()), C_void(d.data_ptr()), 1024, 5) == 0
    f.synchronize()
    assert bits_equal(o.cpu().numpy().view(np.complex64).reshape(5, 1024), wanti)
    assert bits_equal(d.cpu().numpy().view(np.complex64).reshape(5, 1024), wanti)
    assert lib.kofft_hip_fft_radix4_c32_dev(f._ctx, C_void(d.data_ptr()), C_void(o.data_ptr()), 1 << 28, 1) == -2  # UNSUPPORTED, like fft


def test_radix4_arm_beyond_2p20(oracle):
    """Powers of four past 2^20 (round 5 returned UNSUPPORTED there): 4^11 = 2^22 f32 and f64, one transform each, byte for
    byte against the oracle's restatement of fft_radix4; tables are O(n) host work once per (context, n)."""
    import kofft_amd as K

    n = 1 << 22
    for dt, cdt in ((np.float32, np.complex64), (np.float64, np.complex128)):
        x = rand_c(seeded(8877), (1, n), cdt)
        f = K.HipFftImpl(dt)
        y = x[0].copy()
        f.fft_with_strategy(y, K.FftStrategy.Radix4)
        assert bits_equal(y, oracle.fft_radix4(x)[0]), cdt.__name__


def C_void(p):
    import ctypes

    return ctypes.c_void_p(p)
