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


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_multi_gpu_forms_on_one_card(oracle, seed):
    """kofft_hip_multi_* with G logical devices on card 0 (worker threads, shards, halos, empty tail shards), host and device
    forms, at random sizes: the union of the shards is the single-device result, bit for bit.  (The RCCL exchange needs
    distinct devices: tests/test_gpu_multi.py::test_multi_all_cards_of_the_box_with_rccl.)"""
    import torch

    import kofft_amd

    rng = seeded(9300 + seed)
    dev = torch.device("cuda", 0)
    for _ in range(5):
        g = int(rng.integers(1, 7))
        m = kofft_amd.HipMulti(g, devices=[0] * g)
        # STFT, host form: frames may be fewer than devices, frames past the end, any hop
        win_len = 1 << int(rng.integers(2, 12))
        hop = int(rng.integers(1, 2 * win_len))
        length = int(rng.integers(1, 60_000))
        frames = -(-length // hop) + int(rng.integers(0, 3))
        if frames * win_len > (1 << 22):
            hop = max(hop, length * win_len // (1 << 22) + 1)
            frames = -(-length // hop)
        sig = rng.uniform(-1, 1, length).astype(np.float32)
        win = rng.uniform(0, 1, win_len).astype(np.float32)
        want = oracle.stft(sig, win, hop, frames)
        assert same(m.stft(sig, win, hop, frames), want), ("stft host", g, win_len, hop, length, frames)
        # STFT, device form from per-device slices, outputs in the handle's buffers
        dw = torch.from_numpy(win).to(dev)
        slices = []
        for r in range(g):
            first, count = m.stft_slice(length, win_len, hop, frames, r)
            slices.append(torch.from_numpy(sig[first:first + count].copy()).to(dev))
        torch.cuda.synchronize(dev)
        outs = [torch.zeros((max(m.shard(frames, r)[1], 1), win_len, 2), dtype=torch.float32, device=dev) for r in range(g)]
        torch.cuda.synchronize(dev)
        m.stft_dev([t.data_ptr() for t in slices], length, [dw.data_ptr()] * g, win_len, hop, frames, d_out=[t.data_ptr() for t in outs])
        m.synchronize()
        for r in range(g):
            first, count = m.shard(frames, r)
            if count:
                got = outs[r].cpu().numpy().view(np.complex64).reshape(-1, win_len)[:count]
                assert same(got, want[first:first + count]), ("stft dev", g, r)
        # batched complex and real rows, host forms
        n = 1 << int(rng.integers(1, 13))
        batch = int(rng.integers(1, 40))
        double = rng.random() < 0.4
        x = rand_c(rng, (batch, n), np.complex128 if double else np.complex64)
        y = x.copy()
        inverse = bool(rng.random() < 0.5)
        m.fft_batch(y, inverse=inverse)
        assert same(y, oracle.ifft(x) if inverse else oracle.fft(x)), ("fft", g, n, batch, double, inverse)
        rows = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
        w2 = rng.uniform(0, 1, 2 * n).astype(np.float32) if rng.random() < 0.5 else None
        assert same(m.rfft_batch(rows, w2), oracle.rfft(rows, w2)), ("rfft", g, n, batch)
        m.close()
