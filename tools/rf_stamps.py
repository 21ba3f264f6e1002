#!/usr/bin/env python3
"""Phase timeline of fft_regfile_persist_kernel from s_memtime stamps (library built with -DKOFFT_RF_STAMPS: tools/build_variant.sh
stamps -DKOFFT_RF_STAMPS).  Stamps of the first 4 workgroups x 8 transforms x 16 wavefronts:
  0 loop top | 1 inputs landed | 2 A0 done | 3 past the top barrier | 4 A-local exchange done | 5 A1 done | 6 block-wide exchange done |
  7 B0 done | 8 B-local exchange done | 9 B1 + stores + next loads issued
Prints, per phase, the mean over wavefronts / transforms (steady-state transforms 2 .. 6) in stamp ticks and in microseconds
(s_memtime counts at about 2 GHz on this part -- measured against HIP events over the whole kernel -- not at the 100 MHz of the constant
refclk: 0.5 ns per tick), and the spread between the first and the last wavefront at each stamp.

usage (GPU box): KOFFT_HIP_LIB=kofft_amd/lib_stamps/libkofft_hip.so python3 tools/rf_stamps.py [c32|c64]"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kofft_amd  # noqa: E402

NAMES = ["top", "inputs landed", "A0", "top barrier", "A-local exch", "A1", "block exch", "B0", "B-local exch", "B1+st+ld issue"]


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "c32"
    dev = torch.device("cuda:0")
    f32 = kind == "c32"
    n = 32768 if f32 else 16384
    batch = 2048
    dt = torch.float32 if f32 else torch.float64
    fft = kofft_amd.HipFftImpl(np.float32 if f32 else np.float64, device=0)
    src = torch.empty((batch, n, 2), dtype=dt, device=dev).uniform_(-1, 1)
    dst = torch.empty_like(src)
    for _ in range(5):
        fft.fft_dev_oop(src.data_ptr(), dst.data_ptr(), n, batch)
    torch.cuda.synchronize()
    lib = C.CDLL(os.environ.get("KOFFT_HIP_LIB", "kofft_amd/lib/libkofft_hip.so"))
    fn = getattr(lib, "kofft_hip_exp_rf_stamps_f32" if f32 else "kofft_hip_exp_rf_stamps_f64")
    fn.argtypes = [C.c_void_p, C.c_size_t]
    buf = np.zeros(4 * 8 * 16 * 16, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(4, 8, 16, 16).astype(np.int64)  # [wg][iter][wave][stamp]
    tick_us = 0.0005  # s_memtime: ~2 GHz here
    print(f"{kind} n={n}: ticks of 0.5 ns; transforms 2..6 of workgroups 0..3")
    sel = st[:, 2:7]
    per = sel[..., 1:10] - sel[..., 0:9]            # phase durations per wave
    nxt = st[:, 3:8, :, 0] - st[:, 2:7, :, 9]       # end of loop body -> next top (zero-ish)
    total = st[:, 3:8, :, 0] - st[:, 2:7, :, 0]
    print(f"  per transform (top -> next top): mean {total.mean() * tick_us:7.2f} us  min {total.min() * tick_us:7.2f}  max {total.max() * tick_us:7.2f}")
    for i in range(9):
        d = per[..., i]
        print(f"  {NAMES[i]:>16s} -> {NAMES[i + 1]:<16s} mean {d.mean() * tick_us:6.2f} us   first wave {d.min(axis=2).mean() * tick_us:6.2f}  last wave {d.max(axis=2).mean() * tick_us:6.2f}")
    print(f"  loop end -> next top: {nxt.mean() * tick_us:6.2f} us")
    # spread of arrival at each stamp across the 16 waves
    for i in range(10):
        sp = sel[..., i].max(axis=2) - sel[..., i].min(axis=2)
        print(f"  spread between wavefronts at '{NAMES[i]}': {sp.mean() * tick_us:6.2f} us")
    one = st[0, 3]
    t0 = one[:, 0].min()
    print("  workgroup 0, transform 3, microseconds from the first wavefront's top:")
    for w in range(16):
        print("   wave %2d: " % w + " ".join(f"{(one[w, i] - t0) * tick_us:6.2f}" for i in range(10)))


if __name__ == "__main__":
    main()
