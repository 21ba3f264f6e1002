"""kofft_hip_stft_f32_multi / kofft_hip_multi_* on a real device.

The builder's GPU box has one card, so most tests run G = 1 (the same code path as G = 8: per-device context, stream,
worker thread, slice upload, slot layout, ncclCommInitAll + grouped ncclAllGather, with one rank) or G logical devices
mapped onto card 0 (everything but the RCCL exchange, which refuses duplicate devices).  The tests at the end are
parametrised on the box: with >= 2 cards they run the real thing, RCCL included, without code changes.  The partition
arithmetic for G > 1 is covered on CPU too (tests/test_dist_gloo.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal, seeded

pytestmark = pytest.mark.gpu


def _signal(n, seed=3):
    rng = seeded(seed)
    t = np.arange(n, dtype=np.float32)
    return (0.5 * np.sin(2 * np.pi * 440.0 * t / 48000.0) + 0.25 * rng.uniform(-1, 1, n)).astype(np.float32)


@pytest.mark.parametrize("allgather", [False, True])
def test_multi_stft_matches_oracle(oracle, allgather):
    import kofft_amd

    sig = _signal(50_000)
    win = kofft_amd.hann(1024)
    hop = 256
    frames = -(-sig.size // hop)
    want = oracle.stft(sig, win, hop, frames)
    m = kofft_amd.HipMulti(1)
    got, ptrs = m.stft(sig, win, hop, frames, allgather=allgather, want_device_ptrs=True)
    assert bits_equal(got, want)
    assert len(ptrs) == 1 and ptrs[0] != 0
    comp, gath = m.last_timing()
    assert comp > 0.0 and (gath > 0.0 if allgather else gath == 0.0)
    t = m.last_timing_ex()  # phases apart: kernel_ms has no copy inside, the round-2 call reports exactly that
    assert t["kernel_ms"] == comp and t["gather_ms"] == gath
    assert t["upload_ms"] > 0.0 and t["download_ms"] > 0.0
    assert t["wall_ms"] >= max(t["upload_ms"], t["kernel_ms"], t["download_ms"])
    # the handle is reusable, buffers and communicator kept
    got2 = m.stft(sig, win, hop, frames + 3, allgather=allgather)  # extra frames: zero-padded past the end (stft.rs:95-99)
    assert bits_equal(got2, oracle.stft(sig, win, hop, frames + 3))
    m.close()


def test_one_shot_multi_entry(oracle):
    import kofft_amd

    sig = _signal(9_000, seed=5)
    win = kofft_amd.hann(256)
    frames = -(-sig.size // 64)
    want = oracle.stft(sig, win, 64, frames)
    assert bits_equal(kofft_amd.stft_multi(1, sig, win, 64, frames), want)
    assert bits_equal(kofft_amd.stft_multi(1, sig, win, 64, frames, allgather=True), want)


def test_multi_rejects_more_devices_than_present():
    import kofft_amd

    cnt = C.c_int()
    assert kofft_amd.load_library().kofft_hip_device_count(C.byref(cnt)) == 0
    with pytest.raises(kofft_amd.FftError) as e:
        kofft_amd.HipMulti(cnt.value + 1)
    assert e.value.variant == "InvalidValue"


def test_multi_shard_matches_python_partition():
    import kofft_amd
    from kofft_amd.dist import shard_range

    m = kofft_amd.HipMulti(1)
    for total in (0, 1, 7, 112_500):
        assert m.shard(total, 0) == (0, total)
    m.close()
    # the C partition is the one dist.py and bench.py use (ceil(total / world) each): restated for G = 8 from the header's rule
    for total in (0, 5, 8, 112_500):
        per = -(-total // 8)
        for r in range(8):
            lo, hi = shard_range(total, r, 8)
            assert (lo, hi) == (min(r * per, total), min((r + 1) * per, total))


def test_multi_fft_batch(oracle):
    import kofft_amd
    from conftest import rand_c

    x = rand_c(seeded(11), (37, 512))
    want = oracle.fft(x)
    m = kofft_amd.HipMulti(1)
    y = x.copy()
    m.fft_batch(y)
    assert bits_equal(y, want)
    m.fft_batch(y, inverse=True)
    assert bits_equal(y, oracle.ifft(want))
    m.close()


@pytest.mark.parametrize("g", [2, 3, 8])
def test_multi_partition_on_one_card(oracle, g):
    """G logical devices mapped onto the one physical card (devices = [0] * G): G contexts, G streams, G slices with their
    halos, G shards written to their places -- everything of the G > 1 path except the RCCL exchange (RCCL refuses duplicate
    devices), on hardware."""
    import kofft_amd

    sig = _signal(37_003, seed=7 + g)
    win = kofft_amd.hann(1024)
    hop = 256
    frames = -(-sig.size // hop) + 2  # two frames wholly past the end: zero spectra (stft.rs:95-99)
    want = oracle.stft(sig, win, hop, frames)
    m = kofft_amd.HipMulti(g, devices=[0] * g)
    got, ptrs = m.stft(sig, win, hop, frames, allgather=False, want_device_ptrs=True)
    assert bits_equal(got, want)
    assert len(set(ptrs)) == g  # one buffer per logical device
    per = -(-frames // g)
    for r in range(g):
        assert m.shard(frames, r) == (min(r * per, frames), min((r + 1) * per, frames) - min(r * per, frames))
    # batched complex transforms shard the same way, no exchange at all
    from conftest import rand_c
    x = rand_c(seeded(31 + g), (2 * g + 1, 256))
    y = x.copy()
    m.fft_batch(y)
    assert bits_equal(y, oracle.fft(x))
    m.close()


@pytest.mark.parametrize("g", [1, 2, 5])
def test_multi_c64_and_rfft_rows_host(oracle, g):
    """BASELINE configs #5 and #3 at the boundary: c64 batches and windowed real rows in G contiguous blocks, no exchange
    (fft.rs:2156-2175, rfft.rs:264-282), each block through its device's worker thread."""
    import kofft_amd
    from conftest import rand_c

    m = kofft_amd.HipMulti(g, devices=[0] * g)
    x = rand_c(seeded(41 + g), (2 * g + 3, 1024), np.complex128)
    y = x.copy()
    m.fft_batch(y)
    assert bits_equal(y, oracle.fft(x))
    m.fft_batch(y, inverse=True)
    assert bits_equal(y, oracle.ifft(oracle.fft(x)))
    t = m.last_timing_ex()
    assert t["kernel_ms"] > 0.0 and t["upload_ms"] > 0.0 and t["download_ms"] > 0.0 and t["gather_ms"] == 0.0
    rows = seeded(43 + g).uniform(-1, 1, (3 * g + 1, 2048)).astype(np.float32)
    win = kofft_amd.hann(2048)
    assert bits_equal(m.rfft_batch(rows, win), oracle.rfft(rows, win))
    assert bits_equal(m.rfft_batch(rows), oracle.rfft(rows, None))
    # fewer rows than devices: the tail devices own nothing
    few = rows[: max(1, g - 1)]
    assert bits_equal(m.rfft_batch(few, win), oracle.rfft(few, win))
    with pytest.raises(kofft_amd.FftError) as e:
        m.rfft_batch(rows[:, :2047])
    assert e.value.variant == "InvalidValue"  # odd n (rfft.rs:436)
    m.close()


def _device_of(r, ndev):
    return r % ndev


@pytest.mark.parametrize("g", [1, 3])
def test_multi_device_resident_forms(oracle, g):
    """The *_dev twins: one device pointer per device, asynchronous, results and inputs never leave the cards.  G logical
    devices on card 0 (G = 1 also runs the all-gather)."""
    import torch

    import kofft_amd
    from conftest import rand_c

    dev = torch.device("cuda", 0)
    before = torch.cuda.current_device()
    m = kofft_amd.HipMulti(g, devices=[0] * g)
    # complex batches, f32 and f64, in place
    for dtype, double in ((np.complex64, False), (np.complex128, True)):
        x = rand_c(seeded(51 + g), (4 * g + 1, 512), dtype)
        real = np.float64 if double else np.float32
        blocks = []
        for r in range(g):
            first, count = m.shard(x.shape[0], r)
            blocks.append(torch.from_numpy(x[first:first + count].view(real).copy()).to(dev))
        torch.cuda.synchronize(dev)
        m.fft_dev([b.data_ptr() for b in blocks], 512, x.shape[0], double=double)
        m.synchronize()
        got = np.concatenate([b.cpu().numpy().view(dtype) for b in blocks], axis=0)
        assert bits_equal(got.reshape(x.shape), oracle.fft(x))
        t = m.last_timing_ex()
        assert t["kernel_ms"] > 0.0 and t["upload_ms"] == 0.0 and t["download_ms"] == 0.0 and t["wall_ms"] == 0.0
    # windowed real rows
    rows = seeded(53 + g).uniform(-1, 1, (2 * g + 1, 1024)).astype(np.float32)
    win = kofft_amd.hann(1024)
    dwin = torch.from_numpy(win).to(dev)
    ins, outs = [], []
    for r in range(g):
        first, count = m.shard(rows.shape[0], r)
        ins.append(torch.from_numpy(rows[first:first + count].copy()).to(dev))
        outs.append(torch.zeros((max(count, 1), 513, 2), dtype=torch.float32, device=dev))
    torch.cuda.synchronize(dev)
    m.rfft_dev([t_.data_ptr() for t_ in ins], [t_.data_ptr() for t_ in outs], [dwin.data_ptr()] * g, 1024, rows.shape[0])
    m.synchronize()
    got = np.concatenate([outs[r].cpu().numpy().view(np.complex64).reshape(-1, 513)[: m.shard(rows.shape[0], r)[1]] for r in range(g)])
    assert bits_equal(got, oracle.rfft(rows, win))
    # STFT from per-device slices (block + halo); caller's output buffers for even ranks, the handle's for odd ones
    sig = _signal(41_111, seed=9 + g)
    w1k = kofft_amd.hann(1024)
    hop = 256
    frames = -(-sig.size // hop) + 1
    want = oracle.stft(sig, w1k, hop, frames)
    dw = torch.from_numpy(w1k).to(dev)
    slices, mine = [], []
    for r in range(g):
        first, count = m.stft_slice(sig.size, 1024, hop, frames, r)
        slices.append(torch.from_numpy(sig[first:first + count].copy()).to(dev))
        cnt = m.shard(frames, r)[1]
        mine.append(torch.zeros((max(cnt, 1), 1024, 2), dtype=torch.float32, device=dev) if r % 2 == 0 else None)
    torch.cuda.synchronize(dev)
    ptrs = m.stft_dev([t_.data_ptr() for t_ in slices], sig.size, [dw.data_ptr()] * g, 1024, hop, frames,
                      d_out=[t_.data_ptr() if t_ is not None else 0 for t_ in mine])
    m.synchronize()
    assert all(p != 0 for p in ptrs)
    for r in range(0, g, 2):
        first, count = m.shard(frames, r)
        assert ptrs[r] == mine[r].data_ptr()
        assert bits_equal(mine[r].cpu().numpy().view(np.complex64).reshape(-1, 1024)[:count], want[first:first + count])
    if g == 1:  # the exchange with one rank: gathered layout in the caller's buffer
        full = torch.zeros((frames, 1024, 2), dtype=torch.float32, device=dev)
        m.stft_dev([slices[0].data_ptr()], sig.size, [dw.data_ptr()], 1024, hop, frames, allgather=True, d_out=[full.data_ptr()])
        t = m.last_timing_ex()
        assert t["kernel_ms"] > 0.0 and t["gather_ms"] > 0.0
        assert bits_equal(full.cpu().numpy().view(np.complex64).reshape(frames, 1024), want)
    assert torch.cuda.current_device() == before  # the calling thread's device is left as found
    m.close()


@pytest.mark.parametrize("g", [1, 2, 3, 5, 8])
def test_multi_direct_gather_on_one_card(oracle, g):
    """Round 4: the exchange without RCCL -- every device pushes its slot to every peer (hipMemcpyPeerAsync on a stream per peer,
    ordered by events only).  G logical devices on card 0 (peer = self is an ordinary device copy): host form and device form,
    EVERY device's gathered buffer checked incl. the zero padding of the short last slot; at G = 1 both forms byte for byte; the
    handle reports which form ran."""
    import torch

    import kofft_amd

    dev = torch.device("cuda", 0)
    sig = _signal(61_007 + 13 * g, seed=21 + g)
    win = kofft_amd.hann(512)
    hop = 128
    frames = -(-sig.size // hop) + 1
    want = oracle.stft(sig, win, hop, frames)
    per = -(-frames // g)
    m = kofft_amd.HipMulti(g, devices=[0] * g)
    assert m.gather_mode() == {"configured": "rccl", "last": None}
    m.set_gather("direct")
    got, ptrs = m.stft(sig, win, hop, frames, allgather=True, want_device_ptrs=True)
    assert bits_equal(got, want)
    assert m.gather_mode() == {"configured": "direct", "last": "direct"}
    t = m.last_timing_ex()
    assert t["kernel_ms"] > 0.0 and (t["gather_ms"] > 0.0 or g == 1)
    # device form, caller's buffers pre-filled with a sentinel: every logical device ends up with the whole spectrogram
    dw = torch.from_numpy(win).to(dev)
    slices, outs = [], []
    for r in range(g):
        first, count = m.stft_slice(sig.size, 512, hop, frames, r)
        slices.append(torch.from_numpy(sig[first:first + count].copy()).to(dev))
        outs.append(torch.full((g * per, 512, 2), 7.0, dtype=torch.float32, device=dev))
    torch.cuda.synchronize(dev)
    m.stft_dev([t_.data_ptr() for t_ in slices], sig.size, [dw.data_ptr()] * g, 512, hop, frames, allgather=True,
               d_out=[t_.data_ptr() for t_ in outs])
    m.synchronize()
    pad = np.zeros((g * per - frames, 512), np.complex64)
    for r in range(g):
        have = outs[r].cpu().numpy().view(np.complex64).reshape(g * per, 512)
        assert bits_equal(have[:frames], want), f"logical device {r}: gathered spectrogram differs"
        assert bits_equal(have[frames:], pad), f"logical device {r}: slot padding not zero"
    # a call without an exchange says so
    m.stft(sig, win, hop, frames, allgather=False)
    assert m.gather_mode()["last"] is None
    if g == 1:  # both forms on the one rank RCCL accepts here
        m.set_gather("rccl")
        again = m.stft(sig, win, hop, frames, allgather=True)
        assert m.gather_mode()["last"] == "rccl" and bits_equal(again, got)
    with pytest.raises(kofft_amd.FftError):
        m.set_gather(3)
    m.close()


def test_direct_gather_waits_for_a_reader_on_the_destination_stream(oracle):
    """ADVICE r4 (kofft_multi.hip, gather_direct): the peer copy r -> p must order behind work ALREADY QUEUED on the destination's
    stream[p] -- the stream kofft_hip_multi_context hands to callers so that they can order consumers of the gathered buffer on it.
    Two asynchronous device-form calls into the SAME output buffers; between them a slow reader (a long spin, then a snapshot copy of
    the gathered buffer) queued on every stream[p].  The snapshots must hold the FIRST call's spectrogram on every logical device, the
    buffers the SECOND call's.  Without the wait on ev[1][p] the second call's copies overwrite slot r while the reader still waits."""
    import torch

    import kofft_amd

    g = 3
    dev = torch.device("cuda", 0)
    win = kofft_amd.hann(512)
    hop = 128
    sig_a = _signal(80_011, seed=5)
    sig_b = _signal(80_011, seed=6)
    frames = -(-sig_a.size // hop)
    want_a = oracle.stft(sig_a, win, hop, frames)
    want_b = oracle.stft(sig_b, win, hop, frames)
    per = -(-frames // g)
    m = kofft_amd.HipMulti(g, devices=[0] * g)
    m.set_gather("direct")
    dw = torch.from_numpy(win).to(dev)
    sl_a, sl_b, outs, snaps = [], [], [], []
    for r in range(g):
        first, count = m.stft_slice(sig_a.size, 512, hop, frames, r)
        sl_a.append(torch.from_numpy(sig_a[first:first + count].copy()).to(dev))
        sl_b.append(torch.from_numpy(sig_b[first:first + count].copy()).to(dev))
        outs.append(torch.full((g * per, 512, 2), 7.0, dtype=torch.float32, device=dev))
        snaps.append(torch.zeros((g * per, 512, 2), dtype=torch.float32, device=dev))
    torch.cuda.synchronize(dev)
    optr = [t_.data_ptr() for t_ in outs]
    m.stft_dev([t_.data_ptr() for t_ in sl_a], sig_a.size, [dw.data_ptr()] * g, 512, hop, frames, allgather=True, d_out=optr)
    for p in range(g):  # the consumer of call 1's gathered buffer, ordered on stream[p]; on device 0 behind ~30 ms of spinning
        _, st = m.context(p)
        with torch.cuda.stream(torch.cuda.ExternalStream(st, device=dev)):
            if p == 0:  # ONE slow destination: its peers' kernels of call 2 finish long before this reader does
                torch.cuda._sleep(60_000_000)
            snaps[p].copy_(outs[p])
    m.stft_dev([t_.data_ptr() for t_ in sl_b], sig_b.size, [dw.data_ptr()] * g, 512, hop, frames, allgather=True, d_out=optr)
    m.synchronize()
    torch.cuda.synchronize(dev)
    for p in range(g):
        snap = snaps[p].cpu().numpy().view(np.complex64).reshape(g * per, 512)
        assert bits_equal(snap[:frames], want_a), f"logical device {p}: the reader saw slots of the NEXT call"
        have = outs[p].cpu().numpy().view(np.complex64).reshape(g * per, 512)
        assert bits_equal(have[:frames], want_b), f"logical device {p}: second call's gathered spectrogram differs"
    m.close()


def test_multi_gather_mode_from_the_environment(monkeypatch):
    import kofft_amd

    monkeypatch.setenv("KOFFT_HIP_MULTI_GATHER", "direct")
    m = kofft_amd.HipMulti(1)
    assert m.gather_mode()["configured"] == "direct"
    m.close()


def _ndev():
    import kofft_amd

    cnt = C.c_int()
    kofft_amd.load_library().kofft_hip_device_count(C.byref(cnt))
    return cnt.value


def test_multi_all_cards_of_the_box_with_rccl(oracle):
    """G = every card of the box (skipped on a one-card box): the RCCL all-gather for real, checked on EVERY device's
    gathered buffer, host form and device form; the sharded batched forms beside it."""
    import torch

    import kofft_amd
    from conftest import rand_c

    g = _ndev()
    if g < 2:
        pytest.skip("one card: the RCCL exchange between devices needs at least two")
    g = min(g, 8)
    before = torch.cuda.current_device()
    sig = _signal(300_007, seed=77)
    win = kofft_amd.hann(1024)
    hop = 256
    frames = -(-sig.size // hop)
    want = oracle.stft(sig, win, hop, frames)
    per = -(-frames // g)
    m = kofft_amd.HipMulti(g)
    got, ptrs = m.stft(sig, win, hop, frames, allgather=True, want_device_ptrs=True)
    assert bits_equal(got, want)
    t = m.last_timing_ex()
    assert t["gather_ms"] > 0.0 and t["kernel_ms"] > 0.0
    assert all(p != 0 for p in ptrs)
    # device form, caller's buffers: every device must end up with the whole spectrogram
    slices, wins, outs = [], [], []
    for r in range(g):
        d = torch.device("cuda", r)
        first, count = m.stft_slice(sig.size, 1024, hop, frames, r)
        slices.append(torch.from_numpy(sig[first:first + count].copy()).to(d))
        wins.append(torch.from_numpy(win).to(d))
        outs.append(torch.full((g * per, 1024, 2), 7.0, dtype=torch.float32, device=d))
    for r in range(g):
        torch.cuda.synchronize(torch.device("cuda", r))
    m.stft_dev([t_.data_ptr() for t_ in slices], sig.size, [t_.data_ptr() for t_ in wins], 1024, hop, frames, allgather=True,
               d_out=[t_.data_ptr() for t_ in outs])
    m.synchronize()
    pad = np.zeros((g * per - frames, 1024), np.complex64)
    for r in range(g):
        have = outs[r].cpu().numpy().view(np.complex64).reshape(g * per, 1024)
        assert bits_equal(have[:frames], want), f"device {r}: gathered spectrogram differs"
        assert bits_equal(have[frames:], pad), f"device {r}: slot padding not zero"
    # the same exchange as peer copies (no RCCL): the A/B the first multi-card run should carry (direct ~0.75 ms vs ring ~5 ms)
    t_rccl = m.last_timing_ex()
    m.set_gather("direct")
    for o in outs:
        o.fill_(7.0)
    for r in range(g):
        torch.cuda.synchronize(torch.device("cuda", r))
    m.stft_dev([t_.data_ptr() for t_ in slices], sig.size, [t_.data_ptr() for t_ in wins], 1024, hop, frames, allgather=True,
               d_out=[t_.data_ptr() for t_ in outs])
    m.synchronize()
    t_direct = m.last_timing_ex()
    assert m.gather_mode()["last"] == "direct"
    for r in range(g):
        have = outs[r].cpu().numpy().view(np.complex64).reshape(g * per, 1024)
        assert bits_equal(have[:frames], want), f"device {r}: direct gather differs"
        assert bits_equal(have[frames:], pad), f"device {r}: slot padding not zero (direct)"
    print(f"all-gather over {g} cards: rccl {t_rccl['gather_ms']:.3f} ms, direct {t_direct['gather_ms']:.3f} ms")
    m.set_gather("rccl")
    # collective-free batched forms over the real devices
    x = rand_c(seeded(79), (8 * g + 3, 4096))
    y = x.copy()
    m.fft_batch(y)
    assert bits_equal(y, oracle.fft(x))
    rows = seeded(81).uniform(-1, 1, (4 * g + 1, 2048)).astype(np.float32)
    w2k = kofft_amd.hann(2048)
    assert bits_equal(m.rfft_batch(rows, w2k), oracle.rfft(rows, w2k))
    assert torch.cuda.current_device() == before
    m.close()
