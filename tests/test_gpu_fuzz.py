"""Randomised differential test: the C ABI against the oracle over (operation, length, batch) drawn at random, with
batches placed around the thresholds where the dispatch switches kernels (generic <-> persistent, small <-> workgroup,
one workgroup <-> two factors).  Fixed seeds: every run checks the same cases.  Bit-exact or it fails."""
import numpy as np
import pytest

from conftest import bits_equal, rand_c, seeded

pytestmark = pytest.mark.gpu

# batches at which the f32 dispatch changes kernel (256 CUs x 4, 16, 32, 64), +-1, plus small ones
EDGE_BATCHES = [1, 2, 3, 7, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385]
MAX_POINTS = 1 << 23


def pick_batch(rng, n):
    cap = max(1, MAX_POINTS // n)
    if rng.random() < 0.6:
        c = [b for b in EDGE_BATCHES if b <= cap]
        return int(rng.choice(c))
    return int(rng.integers(1, cap + 1))


def same(got, want):
    g, w = np.asarray(got), np.asarray(want)
    nan_g, nan_w = np.isnan(g.view(g.real.dtype)), np.isnan(w.view(w.real.dtype))
    return g.shape == w.shape and np.array_equal(nan_g, nan_w) and bits_equal(np.where(nan_g, 0, g.view(g.real.dtype)),
                                                                               np.where(nan_w, 0, w.view(w.real.dtype)))


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_complex(fft32, fft64, oracle, seed):
    rng = seeded(9000 + seed)
    for _ in range(14):
        log2n = int(rng.integers(0, 17))
        n = 1 << log2n
        if rng.random() < 0.15:
            n = int(rng.integers(3, 200))  # Bluestein arm
        double = rng.random() < 0.35
        batch = pick_batch(rng, n * (2 if double else 1))
        inverse = bool(rng.random() < 0.4)
        x = rand_c(rng, (batch, n), np.complex128 if double else np.complex64)
        y = x.copy()
        (fft64 if double else fft32).fft_batch(y, inverse=inverse)
        want = oracle.ifft(x) if inverse else oracle.fft(x)
        assert same(y, want), (n, batch, double, inverse)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_real(fft32, fft64, oracle, seed):
    rng = seeded(9100 + seed)
    for _ in range(12):
        double = rng.random() < 0.25
        log2n = int(rng.integers(1, 15 if double else 16))  # inner transform n/2 <= 2^13 (f64) / 2^14 (f32): one workgroup
        n = 1 << log2n
        batch = pick_batch(rng, n)
        dt = np.float64 if double else np.float32
        f = fft64 if double else fft32
        if rng.random() < 0.5:
            x = rng.uniform(-1, 1, (batch, n)).astype(dt)
            win = rng.uniform(0, 1, n).astype(dt) if rng.random() < 0.6 else None
            assert same(f.rfft_batch(x, win), oracle.rfft(x, win)), ("rfft", n, batch, double, win is not None)
        else:
            spec = rand_c(rng, (batch, n // 2 + 1), np.complex128 if double else np.complex64)
            assert same(f.irfft_batch(spec, n), oracle.irfft(spec, n)), ("irfft", n, batch, double)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_stft(fft32, oracle, seed):
    rng = seeded(9200 + seed)
    for _ in range(10):
        win_len = 1 << int(rng.integers(0, 14))
        hop = int(rng.integers(1, 2 * win_len + 2))
        frames_target = pick_batch(rng, win_len)
        length = max(1, min(frames_target * hop - int(rng.integers(0, hop)), 6_000_000))
        frames = -(-length // hop) + int(rng.integers(0, 3))  # up to two frames entirely past the end
        if frames * win_len > MAX_POINTS:
            frames = max(1, MAX_POINTS // win_len)
        signal = rng.uniform(-1, 1, length).astype(np.float32)
        window = rng.uniform(0, 1, win_len).astype(np.float32)
        got = fft32.stft_into(signal, window, hop, frames, check_frames=False)
        want = oracle.stft_range(signal, window, hop, 0, frames)
        assert same(got, want), (win_len, hop, length, frames)
