"""kofft_hip_set_stream: a caller that switches streams between *_dev calls (torch style) must not race on the context's
shared scratch (the large-n intermediate, the Bluestein work buffer and its cached fft(b) table): VERDICT r1 item 9,
ADVICE kofft_hip.hip:1266."""
import numpy as np
import pytest

from conftest import bits_equal, rand_c, seeded

pytestmark = pytest.mark.gpu


def test_switching_streams_between_calls_is_ordered(oracle):
    import torch

    import kofft_amd

    dev = torch.device("cuda", 0)
    fft = kofft_amd.HipFftImpl(np.float32)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n, batch = 1 << 16, 48  # two-factor path: both calls go through the one big_tmp intermediate
    xa, xb = rand_c(seeded(21), (batch, n)), rand_c(seeded(22), (batch, n))
    wa, wb = oracle.fft(xa), oracle.fft(xb)
    da = torch.from_numpy(xa.view(np.float32).reshape(batch, n, 2)).to(dev)
    db = torch.from_numpy(xb.view(np.float32).reshape(batch, n, 2)).to(dev)
    oa, ob = torch.empty_like(da), torch.empty_like(db)
    torch.cuda.synchronize(dev)
    for _ in range(4):
        fft.set_stream(s1.cuda_stream)
        fft.fft_dev_oop(da.data_ptr(), oa.data_ptr(), n, batch)
        fft.set_stream(s2.cuda_stream)
        fft.fft_dev_oop(db.data_ptr(), ob.data_ptr(), n, batch)
    torch.cuda.synchronize(dev)
    assert bits_equal(oa.cpu().numpy().view(np.complex64).reshape(batch, n), wa)
    assert bits_equal(ob.cpu().numpy().view(np.complex64).reshape(batch, n), wb)
    fft.set_stream(0)


def test_bluestein_table_built_on_one_stream_used_on_another(oracle):
    import torch

    import kofft_amd

    dev = torch.device("cuda", 0)
    fft = kofft_amd.HipFftImpl(np.float32)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n, batch = 1000, 512
    x = rand_c(seeded(23), (batch, n))
    want = oracle.fft(x)
    d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).to(dev)
    o1, o2 = torch.empty_like(d), torch.empty_like(d)
    torch.cuda.synchronize(dev)
    fft.set_stream(s1.cuda_stream)
    fft.fft_dev_oop(d.data_ptr(), o1.data_ptr(), n, batch)   # builds chirp / fft(b) for n = 1000
    fft.set_stream(s2.cuda_stream)
    fft.fft_dev_oop(d.data_ptr(), o2.data_ptr(), n, batch)   # reads the cached tables and the shared work buffer
    torch.cuda.synchronize(dev)
    assert bits_equal(o1.cpu().numpy().view(np.complex64).reshape(batch, n), want)
    assert bits_equal(o2.cpu().numpy().view(np.complex64).reshape(batch, n), want)
    fft.set_stream(0)


def test_release_scratch_and_reuse(oracle):
    """kofft_hip_release_scratch frees what the largest call left behind; the next call allocates again and still matches."""
    import kofft_amd

    fft = kofft_amd.HipFftImpl(np.float32)
    x = rand_c(seeded(41), (3, 1 << 16))
    want = oracle.fft(x)
    y = x.copy()
    fft.fft_batch(y)          # staging + the two-factor intermediate
    assert bits_equal(y, want)
    z = rand_c(seeded(42), (5, 1000))
    wz = oracle.fft(z)
    fft.fft_batch(z)          # Bluestein work buffer
    assert bits_equal(z, wz)
    fft.release_scratch()
    y = x.copy()
    fft.fft_batch(y)
    assert bits_equal(y, want)
    fft.release_scratch()
    fft.release_scratch()     # idempotent


def test_device_calls_can_be_captured_into_a_hip_graph(oracle):
    """Once a context has seen a (size, kind) -- tables uploaded, scratch allocated, kernel attributes set -- its *_dev calls
    are nothing but kernel launches on the context's stream, so a launch-bound loop of small calls (one STFT frame batch, one
    transform after another) can be captured into a hipGraph and replayed: same bytes as the direct calls."""
    import torch

    import kofft_amd

    dev = torch.device("cuda", 0)
    fft = kofft_amd.HipFftImpl(np.float32)
    s = torch.cuda.Stream(device=dev)
    fft.set_stream(s.cuda_stream)
    rng = seeded(31)
    n, batch, calls = 1024, 4, 12
    xs = [rand_c(rng, (batch, n)) for _ in range(calls)]
    sig = rng.uniform(-1, 1, 6000).astype(np.float32)
    win = kofft_amd.hann(256)
    frames = -(-sig.size // 64)
    with torch.cuda.stream(s):
        ds = [torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).to(dev) for x in xs]
        outs = [torch.empty_like(d) for d in ds]
        dsig, dwin = torch.from_numpy(sig).to(dev), torch.from_numpy(win).to(dev)
        spec = torch.empty((frames, 256, 2), dtype=torch.float32, device=dev)
        # warm-up outside the capture: planner tables and kernel attributes are set up on first use
        fft.fft_dev_oop(ds[0].data_ptr(), outs[0].data_ptr(), n, batch)
        fft.stft_dev(dsig.data_ptr(), sig.size, dwin.data_ptr(), 256, 64, spec.data_ptr(), 0, frames)
        for o in outs:
            o.zero_()
        spec.zero_()
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for d, o in zip(ds, outs):
            fft.fft_dev_oop(d.data_ptr(), o.data_ptr(), n, batch)
        fft.stft_dev(dsig.data_ptr(), sig.size, dwin.data_ptr(), 256, 64, spec.data_ptr(), 0, frames)
    torch.cuda.synchronize(dev)
    assert float(outs[-1].abs().sum()) == 0.0  # captured, not run
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize(dev)
    for x, o in zip(xs, outs):
        assert bits_equal(o.cpu().numpy().view(np.complex64).reshape(batch, n), oracle.fft(x))
    assert bits_equal(spec.cpu().numpy().view(np.complex64).reshape(frames, 256), oracle.stft(sig, win, 64, frames))
    fft.set_stream(0)


def test_two_contexts_in_two_threads(oracle):
    """ScalarFftImpl is Send + !Sync: one context per thread.  Two threads, each with its own contexts, run
    different transforms at the same time (ctypes drops the GIL around every call): the library's only shared state is
    read-only after the first use per device (kernel attributes, the RCCL binding), so every result is the oracle's."""
    import threading

    import kofft_amd

    errors = []

    def worker(seed, sizes):
        try:
            f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
            rng = seeded(seed)
            for rep in range(3):
                for n, batch in sizes:
                    x = rand_c(rng, (batch, n))
                    y = x.copy()
                    f32.fft_batch(y)
                    if not bits_equal(y, oracle.fft(x)):
                        errors.append(("c32", n, batch, rep))
                    xd = rand_c(rng, (max(batch // 4, 1), n), np.complex128)
                    yd = xd.copy()
                    f64.fft_batch(yd)
                    if not bits_equal(yd, oracle.fft(xd)):
                        errors.append(("c64", n, batch, rep))
                    r = rng.uniform(-1, 1, (batch, 2 * n)).astype(np.float32)
                    spec = f32.rfft_batch(r)
                    if not bits_equal(spec, oracle.rfft(r, None)) or not bits_equal(f32.irfft_batch(spec, 2 * n), oracle.irfft(spec, 2 * n)):
                        errors.append(("real", n, batch, rep))
        except Exception as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append(repr(e))

    a = threading.Thread(target=worker, args=(41, [(1024, 300), (1000, 40), (1 << 16, 6), (64, 5000)]))
    b = threading.Thread(target=worker, args=(42, [(4096, 130), (12, 700), (1 << 15, 9), (512, 900)]))
    a.start(); b.start(); a.join(); b.join()
    assert not errors, errors


def test_host_pipeline_on_every_entry_point(oracle, monkeypatch):
    """The host-pointer entry points move batches of 128 MiB and more in eight chunks -- upload of chunk c + 1, kernels of chunk c and
    download of chunk c - 1 in flight together (kofft_hip.hip: pipeline_chunks).  The session runs with the pipeline OFF (conftest.py:
    so that every test's batch reaches the kernel it names); here it is ON, on every entry point that has it -- complex in place
    (c32 / c64, forward and inverse), rfft with and without a row window, irfft -- with batches that do not divide by eight, EVERY row
    against the oracle, and byte for byte against the same calls with the pipeline off."""
    import kofft_amd

    rng = seeded(4242)
    xc = rand_c(rng, (4100 + 3, 4096))                              # 2 x 134 MB
    xz = rand_c(rng, (1025 + 6, 8192), np.complex128)               # 2 x 135 MB
    xr = rng.uniform(-1, 1, (8200 + 5, 4096)).astype(np.float32)    # 134 MB in, 134 MB out
    xd = rng.uniform(-1, 1, (4100 + 1, 4096)).astype(np.float64)
    win = rng.uniform(0.1, 1, 4096).astype(np.float32)
    outs = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("KOFFT_HIP_HOST_PIPELINE", pipe)  # read when the context is created
        f32, f64 = kofft_amd.HipFftImpl(np.float32), kofft_amd.HipFftImpl(np.float64)
        yc = xc.copy()
        f32.fft_batch(yc)
        ic = yc.copy()
        f32.fft_batch(ic, inverse=True)
        yz = xz.copy()
        f64.fft_batch(yz)
        iz = yz.copy()
        f64.fft_batch(iz, inverse=True)
        rw, rp = f32.rfft_batch(xr, win), f32.rfft_batch(xr)
        back = f32.irfft_batch(rp, 4096)
        rd = f64.rfft_batch(xd)
        outs.append((yc, ic, yz, iz, rw, rp, back, rd))
        f32.close()
        f64.close()
    on, off = outs
    assert all(bits_equal(a, b) for a, b in zip(on, off)), "pipeline on / off differ"
    yc, ic, yz, iz, rw, rp, back, rd = on
    assert bits_equal(yc, oracle.fft(xc)) and bits_equal(ic, oracle.ifft(yc))
    assert bits_equal(yz, oracle.fft(xz)) and bits_equal(iz, oracle.ifft(yz))
    assert bits_equal(rw, oracle.rfft(xr, win)) and bits_equal(rp, oracle.rfft(xr))
    assert bits_equal(back, oracle.irfft(rp, 4096)) and bits_equal(rd, oracle.rfft(xd))


@pytest.mark.parametrize("kind,dtype,n,batch", [
    ("c", "c32", 1000, 20001),        # Bluestein through the chunks (160 MB each way)
    ("c", "c32", 1 << 20, 17),        # rows of exactly 8 MiB (the largest the pipeline takes), one more than its smallest batch
    ("c", "c64", 4096, 2051),
    ("c", "c64", 1 << 16, 131),       # the factor path under every chunk
    ("r", "f32", 1000, 40001),        # an even length that is not a power of two: the composed real route
    ("r", "f64", 8192, 2100),
    ("r", "f32", 1 << 17, 300),
])
def test_host_pipeline_shapes(oracle, monkeypatch, kind, dtype, n, batch):
    """More shapes through the chunked host path (pipeline ON): chunk sizes that do not divide the batch, rows at the pipeline's size
    limit, lengths that take the Bluestein arm / the factor path / the composed real route inside every chunk; every row, both directions."""
    import kofft_amd

    monkeypatch.setenv("KOFFT_HIP_HOST_PIPELINE", "1")
    real = np.float64 if dtype in ("c64", "f64") else np.float32
    f = kofft_amd.HipFftImpl(real)
    rng = seeded(4300 + n + batch)
    if kind == "c":
        x = rand_c(rng, (batch, n), np.complex128 if dtype == "c64" else np.complex64)
        y = x.copy()
        f.fft_batch(y)
        want = oracle.fft_inplace_mt(x.copy())
        assert bits_equal(y, want), f"pipelined fft {dtype} n={n} x {batch}"
        f.fft_batch(y, inverse=True)
        assert bits_equal(y, oracle.fft_inplace_mt(want, inverse=True)), f"pipelined ifft {dtype} n={n} x {batch}"
    else:
        x = rng.uniform(-1, 1, (batch, n)).astype(real)
        got = f.rfft_batch(x)
        assert bits_equal(got, oracle.rfft_mt(x)), f"pipelined rfft {dtype} n={n} x {batch}"
        back = f.irfft_batch(got, n)
        rows = sorted({0, 1, batch // 8, batch // 8 + 1, batch // 2, batch - 2, batch - 1})  # around the chunk seams and the ends
        assert bits_equal(back[rows], oracle.irfft(got[rows], n)), f"pipelined irfft {dtype} n={n} x {batch}"
    f.close()


def test_bench_two_rank_rehearsal_carries_the_single_process_gather_ab():
    """bench.py's N > 1 line on a one-GPU box: `--gpus 2 --rehearse-one-card` runs the whole two-rank protocol with
    both ranks on cuda:0 (gloo process group: NOT a measurement) and, after the per-rank part, rank 0 alone drives config #4 through
    the single-process multi-device handle over two logical devices -- the line must carry `multi_single_process` with the direct
    gather's time, the check that every device holds the same gathered spectrogram, and every BASELINE config in roofline.configs."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--rehearse-one-card", "--steps", "3", "--warmup", "1",
                          "--min-seconds", "0", "--ramp-ms", "20", "--no-cpu-baseline", "--detail-file", "/tmp/kofft_bench_rehearsal_detail.json"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert len(lines[0]) < 6000  # the driver's tail buffer keeps the whole line
    assert line["n_gpus"] == 2 and "rehearsal" in line
    ab = line["multi_single_process"]
    assert ab["devices"] == 2 and ab["logical_devices_on_one_card"] is True, ab
    assert ab["gather_direct_ms"] > 0 and ab["kernel_ms"] > 0 and ab["gathered_identical_on_every_device"] is True, ab
    cfgs = line["roofline"]["configs"]
    assert {"#2_inplace", "#2_oop", "#3", "#4", "#5"} <= set(cfgs), sorted(cfgs)
    assert cfgs["#5"]["cap"] == 0.5 and "allgather_ms" in cfgs["#4"]
    assert line["config"]["form"].startswith("in place") and line["values_finite"] is True


def test_bench_single_rank_line_is_compact_and_complete():
    """The N = 1 line: one JSON line below the driver's tail buffer, the in-place form as the headline with its
    out-of-place twin beside it, and every BASELINE config and SURVEY 8(f) row summarised inside `roofline.configs`."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "3", "--warmup", "1", "--min-seconds", "0", "--ramp-ms", "20",
                          "--cpu-seconds", "1", "--detail-file", "/tmp/kofft_bench_n1_detail.json"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000, len(lines[0])
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "GPoints/s" and line["dtype"] == "f32" and line["steps"] == 3 and line["n_gpus"] == 1
    assert line["config"]["form"].startswith("in place") and line["values_finite"] is True
    rf = line["roofline"]
    # (both sides are rounded to 6 significant digits in the line: 5e-6 covers the worst case of the two roundings)
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 5e-6
    cfgs = rf["configs"]
    assert list(cfgs)[:2] == ["#2_inplace", "#2_oop"]
    assert {"#3", "#4", "#5", "f1", "f2", "f3", "f4"} <= set(cfgs), sorted(cfgs)
    for k, c in cfgs.items():
        assert "error" not in c and 0.0 < c["frac"] < 1.0 and c["ms"] > 0, (k, c)
        # one clock per entry: frac is the algorithmic bytes over the printed ms; the binding roofline is named
        assert c["bound"] in ("hbm", "valu", "lds") and c["kernel_ms"] > 0, (k, c)
        if k != "#2_oop":  # (the twin has no counters of its own: profiles/traffic_fft4096.json is the in-place form)
            assert 0.0 < c["issue_frac"] < 1.0 and 0.0 <= c["lds_frac"] < 1.0, (k, c)
    alg2 = 16 * 4096 * 65536
    assert abs(cfgs["#2_inplace"]["frac"] - alg2 / (cfgs["#2_inplace"]["ms"] * 1e-3) / 8e12) < 2e-4
    assert cfgs["#5"]["cap"] == 0.5
    pr = cfgs["#5"]["probe"]
    assert pr["n"] >= 2 and pr["pick_us"][1] <= pr["worst_us"][1]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    for v in line.values():  # the driver truncates long strings: keep every one short
        if isinstance(v, str):
            assert len(v) <= 120, v
