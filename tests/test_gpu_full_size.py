"""BASELINE.json's full-size configurations on the device-pointer path (torch = device memory only).
Round 5: EVERY transform of every BASELINE config is compared with the oracle bit for bit (the C restatement, batch-parallel over
the host's cores through oracle.pyoracle's *_mt entries: 0.27 G points of config #2 take ~0.1 s on 16 cores, config #5's 1024
transforms of 2^20 c64 points well under a minute); the size-independent properties (impulses, Parseval, round trips, Hermitian
symmetry) stay beside them."""
import numpy as np
import pytest

from conftest import bits_equal, seeded


def same_bits(a: np.ndarray, b: np.ndarray) -> bool:
    """bits_equal without the two byte-string copies (multi-GiB arrays)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    u = np.uint64 if a.dtype.itemsize % 8 == 0 else np.uint32
    return a.shape == b.shape and a.dtype == b.dtype and bool(np.array_equal(a.reshape(-1).view(u), b.reshape(-1).view(u)))

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev_fft():
    import kofft_amd

    assert torch.cuda.is_available()
    f = kofft_amd.HipFftImpl(np.float32, device=0)
    s = torch.cuda.Stream()
    f.set_stream(s.cuda_stream)
    return f, s


def test_cfg2_65536x4096_c32(dev_fft, oracle):
    """config #2: 65 536 x 4096 c32 forward, in place on the device."""
    fft, stream = dev_fft
    n, batch = 4096, 65536
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(1234)
        x = torch.empty((batch, n, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
        # plant known-answer rows: an impulse at position p transforms to exp(-2 pi i k p / n) (|.| == 1 exactly at p = 0)
        x[7] = 0
        x[7, 0, 0] = 1.0
        x[batch - 1] = 0
        x[batch - 1, 0, 1] = -2.0
        y = x.clone()
        fft.fft_dev(y.data_ptr(), n, batch, False)
        stream.synchronize()
        # (1) the oracle on EVERY one of the 65 536 transforms, bit for bit (in place on the host copy, batch-parallel)
        want = x.cpu().numpy().view(np.complex64).reshape(batch, n)
        oracle.fft_inplace_mt(want)
        got = y.cpu().numpy().view(np.complex64).reshape(batch, n)
        assert same_bits(got, want)
        del got
        # (2) impulse rows: exactly constant spectra (lib.rs:178-199 at full scale)
        assert torch.all(y[7, :, 0] == 1.0) and torch.all(y[7, :, 1] == 0.0)
        assert torch.all(y[batch - 1, :, 0] == 0.0) and torch.all(y[batch - 1, :, 1] == -2.0)
        # (3) Parseval on every transform: sum |X|^2 = n * sum |x|^2 within the drift budget (8.5e-5 relative)
        ex = (x.double() ** 2).sum(dim=(1, 2))
        ey = (y.double() ** 2).sum(dim=(1, 2))
        rel = ((ey / n - ex).abs() / ex.clamp_min(1e-30)).max().item()
        assert rel < 5e-4, rel
        # (4) round trip: ifft(fft(x)) == x within f32 round-off + table drift, every element
        fft.fft_dev(y.data_ptr(), n, batch, True)
        stream.synchronize()
        err = (y - x).abs().max().item()
        assert err < 2e-3, err
        # (5) ... and the inverse of every transform against the oracle's ifft of the oracle's spectra, bit for bit
        oracle.fft_inplace_mt(want, inverse=True)
        assert same_bits(y.cpu().numpy().view(np.complex64).reshape(batch, n), want)


def test_cfg3_rfft_2048_hann_device(dev_fft, oracle):
    """config #3 at full size: 2^20 rows x 2048 f32 + Hann (8 GiB in, 8 GiB out)."""
    import kofft_amd

    fft, stream = dev_fft
    n, batch = 2048, 1 << 20
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(77)
        x = torch.empty((batch, n), dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
        win_h = kofft_amd.hann(n)
        win = torch.from_numpy(win_h).cuda()
        out = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float32, device="cuda")
        fft.rfft_dev(x.data_ptr(), out.data_ptr(), win.data_ptr(), n, batch)
        stream.synchronize()
        # EVERY one of the 2^20 rows against the oracle, bit for bit, in slices of 2^17 rows (1 GiB in, 1 GiB out on the host)
        for r0 in range(0, batch, 1 << 17):
            want = oracle.rfft_mt(x[r0:r0 + (1 << 17)].cpu().numpy(), win_h)
            got = out[r0:r0 + (1 << 17)].cpu().numpy().view(np.complex64).reshape(1 << 17, n // 2 + 1)
            assert same_bits(got, want), r0
        del want, got
        # DC and Nyquist bins are exactly real for every row (rfft.rs:451-452)
        assert torch.all(out[:, 0, 1] == 0) and torch.all(out[:, n // 2, 1] == 0)
        # DC bin = windowed sum, every row, within f32 accumulation error
        worst = 0.0
        for r0 in range(0, batch, 1 << 18):  # in slices: the f64 temporaries of one slice are 4 GiB
            dc = (x[r0:r0 + (1 << 18)].double() * win.double()).sum(dim=1)
            worst = max(worst, (out[r0:r0 + (1 << 18), 0, 0].double() - dc).abs().max().item())
        assert worst < 5e-3, worst


def test_cfg4_stft_10min_48k(dev_fft, oracle):
    """config #4: 28.8 M samples, 1024-pt Hann, hop 256 -> 112 500 frames (single GPU here; shards in test_dist)."""
    import kofft_amd

    fft, stream = dev_fft
    total, win_len, hop = 28_800_000, 1024, 256
    frames = -(-total // hop)
    assert frames == 112_500
    with torch.cuda.stream(stream):
        t = torch.arange(total, dtype=torch.float32, device="cuda")
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        sig = 0.5 * torch.sin(2 * np.pi * 440.0 * t / 48000.0) + 0.25 * torch.empty_like(t).uniform_(-1, 1, generator=g)
        del t
        win_h = kofft_amd.hann(win_len)
        win = torch.from_numpy(win_h).cuda()
        out = torch.empty((frames, win_len, 2), dtype=torch.float32, device="cuda")
        fft.stft_dev(sig.data_ptr(), total, win.data_ptr(), win_len, hop, out.data_ptr(), 0, frames)
        stream.synchronize()
        # EVERY one of the 112 500 frames against the oracle, bit for bit (the last 3 are partly zero-padded)
        sig_h = sig.cpu().numpy()
        want = oracle.stft_mt(sig_h, win_h, hop, frames)
        assert same_bits(out.cpu().numpy().view(np.complex64).reshape(frames, win_len), want)
        # (the serial entry and the threaded one agree on the ragged end)
        assert bits_equal(want[frames - 4:], oracle.stft_range(sig_h, win_h, hop, frames - 4, 4))
        del want
        # real input: Hermitian symmetry X[k] = conj(X[n-k]) holds to round-off on every frame
        a = out[:, 1:win_len // 2, :]
        b = out[:, win_len // 2 + 1:, :].flip(1)
        assert (a[..., 0] - b[..., 0]).abs().max().item() < 1e-2
        assert (a[..., 1] + b[..., 1]).abs().max().item() < 1e-2
        # sharded evaluation == whole evaluation, bit for bit (frames [f0, f1) of rank 3 of 8)
        per = -(-frames // 8)
        f0, f1 = 3 * per, min(4 * per, frames)
        part = torch.empty((f1 - f0, win_len, 2), dtype=torch.float32, device="cuda")
        fft.stft_dev(sig.data_ptr(), total, win.data_ptr(), win_len, hop, part.data_ptr(), f0, f1 - f0)
        stream.synchronize()
        assert torch.equal(part, out[f0:f1])


def test_cfg4_istft_of_the_spectra_full_size(dev_fft, oracle, monkeypatch):
    """SURVEY 8(f) row 1 at the bench's size: ISTFT of config #4's 112 500 spectra (fused inverse transform + overlap-add, seams and tail on
    the ordered overlap-add kernel).  The WHOLE output and scratch against the oracle bit for bit (the C restatement takes ~2 s for
    115 M points), the frames left behind on a sample, the round trip against the signal, and every byte against the two-kernel route."""
    import kofft_amd

    fft, stream = dev_fft
    total, win_len, hop = 28_800_000, 1024, 256
    frames = -(-total // hop)
    with torch.cuda.stream(stream):
        t = torch.arange(total, dtype=torch.float32, device="cuda")
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        sig = 0.5 * torch.sin(2 * np.pi * 440.0 * t / 48000.0) + 0.25 * torch.empty_like(t).uniform_(-1, 1, generator=g)
        del t
        win_h = kofft_amd.hann(win_len)
        win = torch.from_numpy(win_h).cuda()
        spec = torch.empty((frames, win_len, 2), dtype=torch.float32, device="cuda")
        fft.stft_dev(sig.data_ptr(), total, win.data_ptr(), win_len, hop, spec.data_ptr(), 0, frames)
        stream.synchronize()
        spec_h = spec.cpu().numpy().view(np.complex64).reshape(frames, win_len)
        want = oracle.istft(spec_h, win_h, hop, total)
        work = spec.clone()
        out = torch.zeros(total, dtype=torch.float32, device="cuda")
        scratch = torch.full((total,), 7.0, dtype=torch.float32, device="cuda")
        fft.istft_dev(work.data_ptr(), frames, win.data_ptr(), win_len, hop, out.data_ptr(), total, scratch.data_ptr())
        stream.synchronize()
        assert bits_equal(out.cpu().numpy(), want)
        pick = [0, 1, 2, 3, 219, 220, 221, 56_250, frames - 2, frames - 1]  # (220 frames per workgroup run at 512 workgroups: a seam)
        assert bits_equal(work[pick].cpu().numpy().view(np.complex64).reshape(len(pick), win_len), oracle.ifft(spec_h[pick]))
        ok = scratch > 1e-3
        assert (out[ok] - sig[ok]).abs().max().item() < 1e-3
        # the two-kernel route: same bytes everywhere
        monkeypatch.setenv("KOFFT_HIP_ISTFT_FUSED", "0")
        two = kofft_amd.HipFftImpl(np.float32, device=0)
        two.set_stream(stream.cuda_stream)
        work2 = spec.clone()
        out2 = torch.zeros(total, dtype=torch.float32, device="cuda")
        scratch2 = torch.zeros(total, dtype=torch.float32, device="cuda")
        two.istft_dev(work2.data_ptr(), frames, win.data_ptr(), win_len, hop, out2.data_ptr(), total, scratch2.data_ptr())
        stream.synchronize()
        assert torch.equal(out.view(torch.int32), out2.view(torch.int32)) and torch.equal(scratch.view(torch.int32), scratch2.view(torch.int32))
        assert torch.equal(work.view(torch.int32), work2.view(torch.int32))


def test_bluestein_65536x1000_full_size(dev_fft, oracle, monkeypatch):
    """SURVEY 8(f) row 4 at the bench's size: 65 536 x 1000-pt c32 through the Bluestein arm (persistent kernel).  A sample of rows against
    the oracle bit for bit, EVERY row against the one-workgroup-per-transform kernel (KOFFT_HIP_BLUESTEIN_PERSIST=0), and the round trip."""
    import kofft_amd

    fft, stream = dev_fft
    n, batch = 1000, 65536
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(77)
        x = torch.empty((batch, n, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
        y = torch.empty_like(x)
        fft.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, batch)
        stream.synchronize()
        idx = torch.tensor([0, 1, 2, 1023, 1024, 2047, 2048, 32768, 50_000, 65534, 65535], device="cuda")
        got = y[idx].cpu().numpy().view(np.complex64).reshape(len(idx), n)
        assert bits_equal(got, oracle.fft(x[idx].cpu().numpy().view(np.complex64).reshape(len(idx), n)))
        monkeypatch.setenv("KOFFT_HIP_BLUESTEIN_PERSIST", "0")
        old = kofft_amd.HipFftImpl(np.float32, device=0)
        old.set_stream(stream.cuda_stream)
        y2 = torch.empty_like(x)
        old.fft_dev_oop(x.data_ptr(), y2.data_ptr(), n, batch)
        stream.synchronize()
        assert torch.equal(y.view(torch.int32), y2.view(torch.int32))
        fft.fft_dev(y.data_ptr(), n, batch, True)
        stream.synchronize()
        assert (y - x).abs().max().item() < 2e-3


def test_cfg5_1024x2p20_c64(oracle):
    """config #5 at full size: 1024 x 2^20-point Complex64 forward (16 GiB in, 16 GiB out), two-factor device path."""
    import kofft_amd

    n, batch = 1 << 20, 1024
    fft = kofft_amd.HipFftImpl(np.float64, device=0)
    stream = torch.cuda.Stream()
    fft.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(2020)
        x = torch.empty((batch, n, 2), dtype=torch.float64, device="cuda").uniform_(-1, 1, generator=g)
        x[5] = 0
        x[5, 0, 0] = 3.0  # impulse at 0 -> constant spectrum, exactly (lib.rs:178-199)
        y = torch.empty_like(x)
        fft.fft_dev_oop(x.data_ptr(), y.data_ptr(), n, batch, False)
        stream.synchronize()
        # (1) the oracle on EVERY one of the 1024 transforms, bit for bit, 32 at a time (one 512 MiB chunk of the device's two-factor
        # schedule per slice; the oracle works in place on the host copy of the inputs)
        for b0 in range(0, batch, 32):
            want = x[b0:b0 + 32].cpu().numpy().view(np.complex128).reshape(32, n)
            oracle.fft_inplace_mt(want)
            got = y[b0:b0 + 32].cpu().numpy().view(np.complex128).reshape(32, n)
            assert same_bits(got, want), b0
        del want, got
        assert torch.all(y[5, :, 0] == 3.0) and torch.all(y[5, :, 1] == 0.0)
        # (2) Parseval on every transform (f64 tables drift ~1e-13 relative over 2^19 recurrence steps)
        worst = 0.0
        for b0 in range(0, batch, 64):
            ex = (x[b0:b0 + 64] ** 2).sum(dim=(1, 2))
            ey = (y[b0:b0 + 64] ** 2).sum(dim=(1, 2))
            worst = max(worst, ((ey / n - ex).abs() / ex).max().item())
        assert worst < 1e-9, worst
        # (3) inverse round trip, every element
        fft.fft_dev(y.data_ptr(), n, batch, True)
        stream.synchronize()
        err = 0.0
        for b0 in range(0, batch, 64):
            err = max(err, (y[b0:b0 + 64] - x[b0:b0 + 64]).abs().max().item())
        assert err < 1e-9, err
    del x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,batch", [(8192, 8192 + 7), (16384, 4096 + 3)])
def test_wave_split_kernels_at_sweep_size(dev_fft, oracle, n, batch):
    """The sizes tools/sweep.py times the wave-split kernels at (512 MiB per launch, 32 / 16 transforms per workgroup, a ragged
    tail): EVERY transform against the oracle bit for bit (the oracle does 67 M points in well under a second), forward in
    place on the device, then the inverse, plus an STFT whose frames overlap fourfold at the same window length."""
    fft, stream = dev_fft
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda")
        g.manual_seed(4321 + n)
        x = torch.empty((batch, n, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
        xh = x.cpu().numpy().view(np.complex64).reshape(batch, n)
        y = x.clone()
        fft.fft_dev(y.data_ptr(), n, batch, False)
        stream.synchronize()
        want = oracle.fft(xh)
        assert bits_equal(y.cpu().numpy().view(np.complex64).reshape(batch, n), want)
        fft.fft_dev(y.data_ptr(), n, batch, True)
        stream.synchronize()
        assert bits_equal(y.cpu().numpy().view(np.complex64).reshape(batch, n), oracle.ifft(want))
        del x, y
        hop = n // 4
        frames = 5000 if n == 8192 else 2500
        sig = torch.empty(hop * frames + 33, dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
        import kofft_amd

        win = torch.from_numpy(kofft_amd.hann(n)).cuda()
        nframes = -(-sig.numel() // hop)
        out = torch.empty((nframes, n, 2), dtype=torch.float32, device="cuda")
        fft.stft_dev(sig.data_ptr(), sig.numel(), win.data_ptr(), n, hop, out.data_ptr(), 0, nframes)
        stream.synchronize()
        ws = oracle.stft(sig.cpu().numpy(), win.cpu().numpy(), hop, nframes)
        assert bits_equal(out.cpu().numpy().view(np.complex64).reshape(nframes, n), ws)
