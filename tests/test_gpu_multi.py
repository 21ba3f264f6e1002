"""kofft_hip_stft_f32_multi / kofft_hip_multi_* on a real device.

The GPU box has one card, so G = 1 here: the same code path as G = 8 (per-device context, stream, slice upload, slot
layout, ncclCommInitAll + grouped ncclAllGather), with one rank.  The partition arithmetic for G > 1 is covered on CPU
(tests/test_dist_gloo.py, test_multi_shard_matches_python_partition below needs a device only for the handle)."""
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
