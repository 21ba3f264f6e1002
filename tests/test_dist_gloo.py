"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise the sharding and all-gather logic of
kofft_amd/dist.py.  The oracle stands in for the device inside `compute_frames` (the product has no CPU path); on a
GPU the same code runs with HipFftImpl.stft_dev and backend "nccl" (RCCL)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def test_shard_range_covers_everything_once():
    from kofft_amd.dist import frames_required, shard_range, signal_span

    for total in (0, 1, 7, 8, 112_500, 65_536):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert all(0 <= lo <= hi <= total for lo, hi in blocks)
    # BASELINE config #4 on 8 GPUs: 14 063 frames per rank, 14 059 on the last
    assert frames_required(28_800_000, 256) == 112_500
    sizes = [hi - lo for lo, hi in (shard_range(112_500, r, 8) for r in range(8))]
    assert sizes == [14_063] * 7 + [14_059]
    lo, hi = signal_span(14_063, 14_063, 256, 1024, 28_800_000)
    assert lo == 14_063 * 256 and hi - lo == 14_062 * 256 + 1024  # slice + 768-sample halo


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, length: int, hop: int, win_len: int, q, extra: int = 1):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kofft_amd.dist import frames_required, stft_sharded
        from oracle import pyoracle as ko

        rng = np.random.default_rng(1234)  # same signal on every rank (replicated input)
        signal = rng.uniform(-1, 1, length).astype(np.float32)
        window = ko.hann(win_len)
        frames = frames_required(length, hop) + extra  # (default) one extra, fully zero-padded frame

        def compute(first, count):
            out = ko.stft_range(signal, window, hop, first, count)
            return torch.from_numpy(out.view(np.float32).reshape(count, win_len, 2).copy())

        full, (f0, f1) = stft_sharded(compute, frames, win_len, gather=True)
        want = ko.stft(signal, window, hop, frames).view(np.float32).reshape(frames, win_len, 2)
        ok = full.shape == (frames, win_len, 2) and full.numpy().tobytes() == want.tobytes()
        local, _ = stft_sharded(compute, frames, win_len, gather=False)
        ok = ok and local.numpy().tobytes() == want[f0:f1].tobytes()
        # barrier + max-over-ranks timing protocol of bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok), (f0, f1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,length,hop,win_len", [(2, 5000, 64, 256), (3, 4097, 100, 128)])
def test_stft_sharded_allgather_gloo(world, length, hop, win_len):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, length, hop, win_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in results) == list(range(world))
    assert all(ok for _, ok, _ in results), results


def test_stft_sharded_world8_config4_frame_count():
    """BASELINE config #4's partition for real: 28.8 M samples at hop 256 = 112 500 frames over 8 ranks -- 14 063 frames on ranks
    0..6, 14 059 on the last, whose all-gather slot is padded with four zero frames.  A short window keeps it cheap;
    every rank checks the whole gathered spectrogram and its own block against the oracle."""
    world, length, hop, win_len = 8, 28_800_000, 256, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, length, hop, win_len, q, 0)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    blocks = dict((r, b) for r, _, b in results)
    assert [blocks[r][1] - blocks[r][0] for r in range(world)] == [14_063] * 7 + [14_059]
    assert blocks[7] == (98_441, 112_500)
