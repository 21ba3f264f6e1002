#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): batched 4096-point Complex32 FFT.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: every rank transforms its own 65 536 x 4096
c32 batch (BASELINE config #2, 2 GiB, resident in HBM), reading a pristine input buffer and writing
the spectra to a second buffer (same kernel and bytes as the in-place call, but repeated steps do
not overflow f32).  Batches shard across ranks with no data-path collective (weak scaling); the only
collectives are the barriers of the timing protocol and a MAX over ranks of the elapsed time.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for how each field is produced.
torch is plumbing only (device memory, streams, torch.distributed); the transform is libkofft_hip.so.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec per GPU

WORKLOADS = {
    # name: (description, n, batch, algorithmic bytes per unit, unit of `value`)
    "fft4096": "batched 65536 x 4096-pt Complex32 forward FFT (BASELINE config #2)",
    "rfft2048": "batched 2^20 x 2048-pt f32 rfft + Hann (BASELINE config #3)",
    "stft1024": "STFT 28.8M-sample f32 stream, 1024-pt Hann, hop 256 (BASELINE config #4, frames sharded)",
    "c64_2p20": "batched 1024 x 2^20-pt Complex64 forward FFT (BASELINE config #5)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="untimed pre-warm-up: keep the GPU busy this long so DVFS has left idle clocks "
                         "(measured: per-launch time falls 1.17 -> 0.81 ms over the first ~40 ms of load)")
    ap.add_argument("--workload", default="fft4096", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inplace", action="store_true", help="fft4096 only: transform the buffer in place (values overflow after ~10 steps; timing study only)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work for the cpu_baseline sample")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# cpu_baseline: the oracle (a port of kofft's CPU algorithm) on the host cores, bounded sample
# ---------------------------------------------------------------------------------------------------
def host_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota if there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline_fft4096(target_seconds: float):
    from oracle import pyoracle as ko

    n = 4096
    cores = host_cores()
    rng = np.random.default_rng(0x6B6F666674 + 2)

    def make(batch):
        return (rng.uniform(-1, 1, (batch, n)).astype(np.float32)
                + 1j * rng.uniform(-1, 1, (batch, n)).astype(np.float32)).astype(np.complex64)

    lib = ko.lib()
    import ctypes as C

    chunk = 256  # transforms per call: 8 MiB per thread, re-copied from a pristine source before every call

    def run(reps):
        srcs = [make(chunk) for _ in range(cores)]
        busy = [0.0] * cores

        def work(i):
            buf = np.empty_like(srcs[i])
            for _ in range(reps):
                np.copyto(buf, srcs[i])  # untimed: keeps repeated in-place transforms from overflowing
                t0 = time.perf_counter()
                rc = lib.ko_fft_batch_f32(C.c_void_p(buf.ctypes.data), C.c_size_t(n), C.c_size_t(chunk), 0)
                busy[i] += time.perf_counter() - t0
                assert rc == 0
        ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return max(busy)

    dt = max(run(1), 1e-4)
    reps = int(min(max(target_seconds / dt, 1), 4000))
    dt = run(reps)
    transforms = cores * reps * chunk
    return {
        "value": transforms * n / dt / 1e9,
        "unit": "GPoints/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{transforms} x 4096-pt c32 transforms ({cores} threads x {reps} calls x {chunk}, one planner per call), "
                  f"oracle/ C restatement of kofft's Stockham path (-O2, no FMA), busiest thread {dt:.1f} s",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import kofft_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only implementation of this path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    fft = kofft_amd.HipFftImpl(np.float32, device=local_rank)
    # A dedicated (non-default) torch stream: the library launches on it and the HIP events below are
    # recorded on the same stream, so each event pair brackets exactly one launch.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    fft.set_stream(stream.cuda_stream)

    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6B6F666674 + 2 + rank)

    if args.workload == "fft4096":
        n, batch = 4096, args.batch or 65536
        src = torch.empty((batch, n, 2), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
        dst = torch.empty_like(src)
        units_per_step = batch * n                      # complex points
        alg_bytes = 16 * units_per_step                 # 8 B read + 8 B written per point
        unit = "GPoints/s"
        metric = "batched 4096-pt c32 FFT throughput, GPoints/s (achieved HBM GB/s: roofline.achieved)"
        launch = lambda: fft.fft_dev_oop(src.data_ptr(), dst.data_ptr(), n, batch, False)  # noqa: E731
        if args.inplace:
            launch = lambda: fft.fft_dev(src.data_ptr(), n, batch, False)  # noqa: E731
        cfg = {"workload": WORKLOADS["fft4096"], "n": n, "batch_per_gpu": batch, "layout": "interleaved re/im, contiguous",
               "direction": "forward", "sharding": f"batch x{world}, no collective"}
    elif args.workload == "rfft2048":
        n, batch = 2048, args.batch or (1 << 20)
        src = torch.empty((batch, n), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0, generator=gen)
        dst = torch.empty((batch, n // 2 + 1, 2), dtype=torch.float32, device=dev)
        win = torch.from_numpy(kofft_amd.hann(n)).to(dev)
        units_per_step = batch * n                      # real samples
        alg_bytes = batch * (4 * n + 8 * (n // 2 + 1))
        unit = "GSamples/s"
        metric = "batched 2048-pt f32 rfft + Hann throughput"
        launch = lambda: fft.rfft_dev(src.data_ptr(), dst.data_ptr(), win.data_ptr(), n, batch)  # noqa: E731
        cfg = {"workload": WORKLOADS["rfft2048"], "n": n, "batch_per_gpu": batch, "sharding": f"batch x{world}, no collective"}
    elif args.workload == "c64_2p20":
        n, batch = 1 << 20, args.batch or 1024
        fft64 = kofft_amd.HipFftImpl(np.float64, device=local_rank)
        fft64.set_stream(stream.cuda_stream)
        src = torch.empty((batch, n, 2), dtype=torch.float64, device=dev).uniform_(-1.0, 1.0, generator=gen)
        dst = torch.empty_like(src)
        units_per_step = batch * n
        alg_bytes = 32 * units_per_step                 # 16 B read + 16 B written per point
        unit = "GPoints/s"
        metric = "batched 2^20-pt Complex64 forward FFT throughput"
        launch = lambda: fft64.fft_dev_oop(src.data_ptr(), dst.data_ptr(), n, batch, False)  # noqa: E731
        cfg = {"workload": WORKLOADS["c64_2p20"], "n": n, "batch_per_gpu": batch, "sharding": f"batch x{world}, no collective",
               "passes_over_hbm": 2}
    else:
        total_len, win_len, hop = 28_800_000, 1024, 256
        from kofft_amd.dist import frames_required, shard_range

        frames_total = frames_required(total_len, hop)
        f0, f1 = shard_range(frames_total, rank, world)
        t = torch.arange(total_len, dtype=torch.float32, device=dev)
        sig = 0.5 * torch.sin(2 * np.pi * 440.0 * t / 48000.0) + 0.25 * torch.empty_like(t).uniform_(-1, 1, generator=gen)
        del t
        win = torch.from_numpy(kofft_amd.hann(win_len)).to(dev)
        count = f1 - f0
        dst = torch.empty((count, win_len, 2), dtype=torch.float32, device=dev)
        units_per_step = count * win_len                # output complex points of this rank
        alg_bytes = 4 * min(total_len - f0 * hop, (count - 1) * hop + win_len) + 8 * units_per_step
        unit = "GPoints/s"
        metric = "STFT 1024-pt Hann hop-256 throughput (output points, compute only)"
        launch = lambda: fft.stft_dev(sig.data_ptr(), total_len, win.data_ptr(), win_len, hop, dst.data_ptr(), f0, count)  # noqa: E731
        cfg = {"workload": WORKLOADS["stft1024"], "signal_len": total_len, "win_len": win_len, "hop": hop,
               "frames_total": frames_total, "sharding": f"frames x{world}, strong", "collective": "none in the timed region"}

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    # clock ramp (untimed, not counted as warm-up steps), then the W warm-up steps of the contract
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(8):
            launch()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        launch()
    torch.cuda.synchronize(dev)

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)   # HIP events on the stream the kernel is launched on
        launch()
        ends[i].record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    kern_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]

    # BASELINE config #4's exchange step, timed apart from the compute (it dominates: SURVEY 8e)
    allgather_ms = None
    if args.workload == "stft1024" and world > 1:
        per = -(-frames_total // world)
        slot = torch.zeros((per, win_len, 2), dtype=torch.float32, device=dev)
        slot[:count] = dst
        full = torch.empty((world * per, win_len, 2), dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_gather_into_tensor(full, slot)
        torch.cuda.synchronize(dev)
        barrier()
        tg = time.perf_counter()
        for _ in range(10):
            dist.all_gather_into_tensor(full, slot)
        torch.cuda.synchronize(dev)
        barrier()
        allgather_ms = (time.perf_counter() - tg) / 10 * 1e3
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    units = torch.tensor([float(units_per_step)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_units = float(units.item()) * args.steps

    if rank == 0:
        if os.environ.get("KOFFT_BENCH_VERBOSE"):
            print("# per-launch ms: " + " ".join(f"{m:.3f}" for m in kern_ms), file=sys.stderr)
        avg_kernel_s = float(np.mean(kern_ms)) / 1e3
        achieved = alg_bytes / avg_kernel_s / 1e9
        traffic = None
        tfile = ROOT / "profiles" / f"traffic_{args.workload}.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("workload") == args.workload:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": metric,
            "value": total_units / elapsed / 1e9,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "stft1024" else "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.workload == "c64_2p20" else "f32",
            "data": "synthetic",
            "config": cfg,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "kernel_ms_min": float(np.min(kern_ms)),
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if allgather_ms is not None:
            out["allgather"] = {"ms_per_step": allgather_ms, "bytes_gathered_per_rank": int(full.numel() * 4),
                                "backend": "nccl (RCCL over xGMI), all_gather_into_tensor"}
        if args.workload == "fft4096" and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_fft4096(args.cpu_seconds)
        elif args.workload == "fft4096" and world > 1:
            out["cpu_baseline"] = None  # timed on rank 0 at N=1 only
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
