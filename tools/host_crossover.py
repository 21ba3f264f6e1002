"""Host-pointer (true drop-in) calls against the CPU: from which batch does `fft.fft_batch(host array)` beat the reference's algorithm on the
host cores?  (INTEGRATION section 4 should name the crossover, not only say "PCIe-bound".)

For n in {1024, 4096} and batch = 1 .. 4096: wall time of the C-ABI host-pointer entry (upload, kernel, download, synchronise; pageable numpy
memory) against the oracle port of kofft's Stockham path on ONE host core and on all cores the process may use (batch split over threads).
usage (GPU box): python3 tools/host_crossover.py"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

import kofft_amd
from oracle import pyoracle as ko


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    f = kofft_amd.HipFftImpl(np.float32)
    cores = ko.host_threads()
    print(f"host cores usable: {cores}")
    for n in (1024, 4096):
        print(f"n = {n}: batch, device path from host memory us, CPU 1 core us, CPU {cores} cores us   (per call)")
        cross1 = crossN = None
        for lb in range(0, 13):
            b = 1 << lb
            rng = np.random.default_rng(lb)
            x = (rng.uniform(-1, 1, (b, n)) + 1j * rng.uniform(-1, 1, (b, n))).astype(np.complex64)
            y = x.copy()
            f.fft_batch(y)  # warm: tables, staging buffers
            reps = 30 if b <= 256 else 8
            t_dev = best(lambda: f.fft_batch(y), reps)
            z = x.copy()
            fn = ko.lib().ko_fft_batch_f32  # the serial C entry itself: no Python thread is started for the one-core figure
            zp = C.c_void_p(z.ctypes.data)
            t_1 = best(lambda: fn(zp, C.c_size_t(n), C.c_size_t(b), 0), reps)
            # all cores: Python starts a thread per block (~0.1 ms each), so only batches that outlast that are timed
            t_n = best(lambda: ko.fft_inplace_mt(z, threads=cores), reps) if b >= 512 else float("nan")
            if cross1 is None and t_dev < t_1:
                cross1 = b
            if crossN is None and t_n == t_n and t_dev < t_n:
                crossN = b
            print(f"  {b:6d} {t_dev * 1e6:10.1f} {t_1 * 1e6:10.1f} {t_n * 1e6:10.1f}", flush=True)
        print(f"n = {n}: the device path from host memory is faster than ONE core from batch {cross1}, than {cores} cores (timed from batch 512 only) from batch {crossN}")


if __name__ == "__main__":
    main()
