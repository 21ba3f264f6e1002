"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Batches and STFT frames are independent units (fft.rs:2160-2162, stft.rs:91-103), so every rank transforms its own
contiguous block and NO data-path collective is needed; the only real exchange is the optional all-gather of STFT
spectra (BASELINE config #4).  The signal is replicated (or host-sliced with its win_len - hop halo) rather than
halo-exchanged: each rank reads samples [f0*hop, (f1-1)*hop + win_len) of it.

`compute_frames(first, count) -> tensor[count, win_len, 2]` is injected so that the partition/gather logic can be
exercised on CPU with gloo (tests/test_dist_gloo.py, where the oracle plays the device); on a GPU it is
`HipFftImpl.stft_dev` writing into a device tensor.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` units owned by `rank`: ceil(total/world) each, the tail short or empty."""
    per = -(-int(total) // int(world)) if world > 0 else int(total)
    lo = min(rank * per, total)
    hi = min((rank + 1) * per, total)
    return lo, hi


def frames_required(signal_len: int, hop: int) -> int:
    """stft.rs:86: ceil(len / hop)."""
    return -(-int(signal_len) // int(hop))


def signal_span(first: int, count: int, hop: int, win_len: int, signal_len: int) -> Tuple[int, int]:
    """Samples [lo, hi) that frames [first, first+count) can see (its slice plus the win_len - hop halo)."""
    if count <= 0:
        return 0, 0
    lo = min(first * hop, signal_len)
    hi = min((first + count - 1) * hop + win_len, signal_len)
    return lo, max(hi, lo)


def stft_sharded(compute_frames: Callable[[int, int], torch.Tensor], frames: int, win_len: int, *,
                 rank: Optional[int] = None, world: Optional[int] = None, gather: bool = True,
                 group=None) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Compute this rank's frames and (optionally) all-gather the spectra.

    Returns (tensor, (f0, f1)).  With gather=False the tensor is this rank's [f1-f0, win_len, 2] block.  With
    gather=True it is the full [frames, win_len, 2] spectrogram on every rank: one `all_gather_into_tensor` of equal
    ceil(frames/world)-frame slots (the short tail is zero-padded, then trimmed).
    """
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    f0, f1 = shard_range(frames, rank, world)
    local = compute_frames(f0, f1 - f0)
    if local.shape[0] != f1 - f0:
        raise ValueError("compute_frames returned the wrong number of frames")
    if not gather or world == 1:
        return local, (f0, f1)
    per = -(-frames // world)
    slot = torch.zeros((per, win_len, 2), dtype=local.dtype, device=local.device)
    slot[: f1 - f0] = local
    full = torch.empty((world * per, win_len, 2), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, slot, group=group)
    return full[:frames], (f0, f1)
