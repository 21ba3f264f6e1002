"""Every KOFFT_HIP_* route switch of the library (kofft_hip.hip: kofft_hip_create), each in its NON-DEFAULT setting, against the oracle
(a switch that no test flips is a wrong-answer path one `export` away).  The switches are read when a context is
created, so every case makes its own contexts.  The table is checked against the source on CPU (tests/test_abi_symbols.py): a new getenv without a row fails there."""
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import bits_equal, rand_c, seeded

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _complex(oracle, dtype, n, batch, seed, check=None, inverse=True):
    import kofft_amd

    cdt = np.complex64 if dtype == "c32" else np.complex128
    f = kofft_amd.HipFftImpl(np.float32 if dtype == "c32" else np.float64)
    x = rand_c(seeded(seed), (batch, n), cdt)
    y = x.copy()
    f.fft_batch(y)
    pick = sorted(set(check if check is not None else range(batch)))
    want = oracle.fft(x[pick])
    assert bits_equal(y[pick], want), f"{dtype} n={n} batch={batch}: forward differs"
    if inverse:
        f.fft_batch(y, inverse=True)
        assert bits_equal(y[pick], oracle.ifft(want)), f"{dtype} n={n} batch={batch}: inverse differs"


def _edges(batch):
    return [0, 1, batch // 2, batch - 2, batch - 1]


def case_no_persist(oracle):
    import kofft_amd

    _complex(oracle, "c32", 4096, 1300, 11, check=_edges(1300))
    f = kofft_amd.HipFftImpl(np.float32)
    rows = seeded(12).uniform(-1, 1, (2100, 2048)).astype(np.float32)
    win = kofft_amd.hann(2048)
    pick = _edges(2100)
    assert bits_equal(f.rfft_batch(rows, win)[pick], oracle.rfft(rows[pick], win))
    sig = seeded(13).uniform(-1, 1, 300_000).astype(np.float32)
    w1k = kofft_amd.hann(1024)
    frames = -(-sig.size // 256)
    assert bits_equal(f.stft_into(sig, w1k, 256, frames), oracle.stft(sig, w1k, 256, frames))


def case_grid_pct(oracle):
    _complex(oracle, "c32", 4096, 1300, 21, check=_edges(1300))
    _complex(oracle, "c64", 1 << 16, 300, 22, check=_edges(300), inverse=False)
    _complex(oracle, "c32", 1000, 4300, 23, check=_edges(4300))  # the persistent Bluestein kernel on half its grid
    case_istft_two_kernels(oracle)                               # the fused ISTFT on half its grid (longer runs, other seams)


def case_three_min_high(oracle):  # 2^22 as TWO factors
    import kofft_amd

    _complex(oracle, "c32", 1 << 22, 2, 31)
    # ... of 2^11 points each, also under Bluestein's pointwise steps and the row window (fft_wg_kernel<T, 11, .., BigColsIO<T, .., PRE_CHIRP /
    # PRE_WINDOW>>; c64: BigRowsIO<double, .., POST_BLUE_*> at 2^11 too)
    _complex(oracle, "c32", (1 << 21) - 1, 2, 34)
    _complex(oracle, "c64", (1 << 21) - 1, 1, 35)
    for dt, n in ((np.float32, 1 << 23), (np.float64, 1 << 23)):
        f = kofft_amd.HipFftImpl(dt)
        rng = seeded(36)
        x = rng.uniform(-1, 1, (1, n)).astype(dt)
        win = rng.uniform(0.1, 1, n).astype(dt)
        assert bits_equal(f.rfft_batch(x, win), oracle.rfft(x, win)), f"windowed rfft {dt.__name__} n=2^23"


def case_three_min_low(oracle):  # 2^21 as THREE factors
    _complex(oracle, "c32", 1 << 21, 3, 32)
    _complex(oracle, "c64", 1 << 21, 2, 33, inverse=False)


def case_small32(oracle):
    import kofft_amd

    _complex(oracle, "c32", 32, 5000, 41)
    # the real transforms and the STFT whose inner length is 32 (fft_wg_kernel<float, 5, .., RfftIO / IrfftIO / StftIO / StftMagIO>)
    f = kofft_amd.HipFftImpl(np.float32)
    rows = seeded(42).uniform(-1, 1, (3001, 64)).astype(np.float32)
    got = f.rfft_batch(rows)
    assert bits_equal(got, oracle.rfft(rows))
    assert bits_equal(f.irfft_batch(got, 64), oracle.irfft(got, 64))
    sig = seeded(43).uniform(-1, 1, 40_000).astype(np.float32)
    win = kofft_amd.hann(32)
    frames = -(-sig.size // 8)
    assert bits_equal(f.stft_into(sig, win, 8, frames), oracle.stft(sig, win, 8, frames))
    mags, mx = f.stft_magnitudes(sig, 32, 8)
    want, want_max = oracle.stft_magnitudes(sig, 32, 8)
    assert bits_equal(mags, want) and mx == want_max


def case_big_persist(oracle):
    _complex(oracle, "c64", 1 << 16, 300, 51, check=_edges(300))
    _complex(oracle, "c32", 1 << 17, 200, 52, check=_edges(200))


def case_big_persist_three(oracle):
    case_big_persist(oracle)
    # three factors, every one of them one tile per workgroup (fft_wg_kernel<T, 7 .. 9, .., BigMidIO<T>>: by default even ONE such transform
    # has enough units for the persistent kernels)
    _complex(oracle, "c32", 1 << 22, 3, 53)
    _complex(oracle, "c32", 1 << 24, 1, 54)
    _complex(oracle, "c64", 1 << 23, 2, 55)
    _complex(oracle, "c32", 1 << 26, 1, 56, inverse=False)
    _complex(oracle, "c64", 1 << 22, 1, 57)
    _complex(oracle, "c64", 1 << 26, 1, 58, inverse=False)
    _complex(oracle, "c32", (1 << 21) - 1, 1, 59)  # Bluestein, m = 2^22 in three factors: the pointwise steps on one-tile-per-workgroup factors
    _fft2d(oracle, "c32", 16384, 128)              # the column axis as 2^7 x 2^7, both passes one tile per workgroup


def _fft2d(oracle, dtype, rows, cols):
    import kofft_amd

    cdt = np.complex64 if dtype == "c32" else np.complex128
    f = kofft_amd.HipFftImpl(np.float32 if dtype == "c32" else np.float64)
    x = rand_c(seeded(rows + cols), (rows, cols), cdt)
    want = oracle.fft(np.ascontiguousarray(oracle.fft(x).T)).T
    data = x.reshape(-1).copy()
    kofft_amd.fft2d_inplace(data, rows, cols, f, np.zeros(rows, cdt))
    assert bits_equal(data.reshape(rows, cols), np.ascontiguousarray(want)), f"fft2d {dtype} {rows} x {cols}"
    f.fftnd(data, 1, rows, cols, inverse=True)
    w = np.ascontiguousarray(want)
    back = oracle.fft(np.ascontiguousarray(oracle.fft(w, inverse=True).T), inverse=True).T
    assert bits_equal(data.reshape(rows, cols), np.ascontiguousarray(back)), f"inverse fft2d {dtype} {rows} x {cols}"


def case_rfft_wide(oracle):  # rfft / irfft of 32768 and 16384 reals
    import kofft_amd

    f = kofft_amd.HipFftImpl(np.float32)
    for n, batch in ((32768, 1100), (16384, 1100)):
        rows = seeded(61 + n).uniform(-1, 1, (batch, n)).astype(np.float32)
        pick = _edges(batch)
        got = f.rfft_batch(rows)
        assert bits_equal(got[pick], oracle.rfft(rows[pick])), f"rfft {n}"
        assert bits_equal(f.irfft_batch(got, n)[pick], oracle.irfft(got[pick], n)), f"irfft {n}"


def case_persist64(oracle):
    _complex(oracle, "c64", 4096, 1100, 71, check=_edges(1100))
    _complex(oracle, "c64", 8192, 1100, 72, check=_edges(1100))


def case_persist_small(oracle):
    import kofft_amd

    f = kofft_amd.HipFftImpl(np.float32)
    sig = seeded(81).uniform(-1, 1, 700_000).astype(np.float32)
    for win_len in (64, 128, 256):
        win = kofft_amd.hann(win_len)
        hop = win_len // 4
        frames = -(-sig.size // hop)
        assert bits_equal(f.stft_into(sig, win, hop, frames), oracle.stft(sig, win, hop, frames)), f"stft {win_len}"
    spec = rand_c(seeded(82), (9000, 129))
    spec[:, 0].imag = 0
    spec[:, -1].imag = 0
    assert bits_equal(f.irfft_batch(spec, 256), oracle.irfft(spec, 256))


def case_split(oracle):
    import kofft_amd

    _complex(oracle, "c32", 8192, 1100, 91, check=_edges(1100))
    _complex(oracle, "c32", 16384, 1100, 92, check=_edges(1100))
    # STFT with an 8192-sample window: fft_persist_kernel<float, 13, .., StftIO> (the wave-split kernel otherwise)
    f = kofft_amd.HipFftImpl(np.float32)
    sig = seeded(93).uniform(-1, 1, 1100 * 2048 + 77).astype(np.float32)
    win = kofft_amd.hann(8192)
    frames = -(-sig.size // 2048)
    got = f.stft_into(sig, win, 2048, frames)
    for first, count in ((0, 2), (frames // 2, 2), (frames - 4, 4)):
        assert bits_equal(got[first:first + count], oracle.stft_range(sig, win, 2048, first, count)), (first, count)
    mags, mx = f.stft_magnitudes(sig, 8192, 2048)  # ... and <.., StftMagIO>
    want, want_max = oracle.stft_magnitudes(sig, 8192, 2048)
    assert bits_equal(mags, want) and mx == want_max


def case_regfile(oracle):
    _complex(oracle, "c32", 32768, 520, 101, check=_edges(520))
    _complex(oracle, "c64", 16384, 520, 102, check=_edges(520))


def case_host_pipeline(oracle):
    _complex(oracle, "c32", 4096, 5000, 111, check=_edges(5000), inverse=False)  # 160 MB: eight chunks through the pipeline


def case_zero_copy(oracle):
    import kofft_amd

    f = kofft_amd.HipFftImpl(np.float32)
    x = rand_c(seeded(121), (1024,))
    y = x.copy()
    f.fft(y)
    assert bits_equal(y, oracle.fft(x[None])[0])
    f.ifft(y)
    assert bits_equal(y, oracle.ifft(oracle.fft(x[None]))[0])


def _nd(oracle, dtype, depth, rows, cols, seed):
    import kofft_amd

    cdt = np.complex64 if dtype == "c32" else np.complex128
    f = kofft_amd.HipFftImpl(np.float32 if dtype == "c32" else np.float64)
    x = rand_c(seeded(seed), (depth, rows, cols), cdt)

    def axis(a, ax):
        moved = np.ascontiguousarray(np.moveaxis(a, ax, -1))
        out = oracle.fft(moved.reshape(-1, moved.shape[-1])).reshape(moved.shape)
        return np.ascontiguousarray(np.moveaxis(out, -1, ax))
    want = axis(axis(x, 2), 1) if depth == 1 else axis(axis(axis(x, 0), 1), 2)
    data = x.reshape(-1).copy()
    f.fftnd(data, depth, rows, cols)
    assert bits_equal(data.reshape(depth, rows, cols), want), f"fftnd {depth}x{rows}x{cols} {dtype}"


def case_nd_transpose(oracle):  # 4096-point axis over 520 (not a power of two) adjacent lines: transposes by default, strided kernel here
    _nd(oracle, "c32", 1, 4096, 520, 131)


def case_nd_two_pass(oracle):  # power-of-two line counts: two column-tile passes by default, transposes here
    _nd(oracle, "c32", 1, 4096, 512, 141)
    _nd(oracle, "c64", 2, 4096, 128, 142)
    _nd(oracle, "c32", 1, 2048, 1024, 143)


def case_nd_fused(oracle):  # 4096-point rows: rows + two column stages fused by default, rows then two column-tile passes here
    _nd(oracle, "c32", 1, 1024, 4096, 144)


def case_bluestein(oracle):
    _complex(oracle, "c32", 1000, 700, 151)      # one launch by default
    _complex(oracle, "c32", 12345, 40, 152)      # fused into the transforms' loads / stores by default
    _complex(oracle, "c64", 40000, 9, 153, inverse=False)  # folded into the factor kernels by default


def case_bluestein_persist(oracle):  # large batches: the persistent kernel by default, one workgroup per XPB transforms here
    _complex(oracle, "c32", 1000, 4200, 155, check=_edges(4200))
    _complex(oracle, "c64", 60, 66000, 156, check=_edges(66000))


def case_istft_two_kernels(oracle):  # a frame count the fused kernel takes by default
    import kofft_amd as K

    f = K.HipFftImpl(np.float32)
    spec = rand_c(seeded(157), (9001, 1024))
    window = oracle.hann(1024)
    out_len = 9000 * 256 + 1024
    out = np.zeros(out_len, np.float32)
    scratch = np.zeros(out_len, np.float32)
    K.istft(spec.copy(), window, 256, out, scratch, f)
    assert bits_equal(out, oracle.istft(spec, window, 256, out_len))


def case_big_narrow(oracle):
    _complex(oracle, "c32", 1 << 17, 1, 161)
    _complex(oracle, "c64", 1 << 18, 2, 162)


def case_first11(oracle):
    _complex(oracle, "c32", 1 << 21, 11, 171, check=[0, 5, 10])
    _complex(oracle, "c64", 1 << 21, 5, 172, check=[0, 4], inverse=False)


def case_blocked(oracle):
    _complex(oracle, "c64", 1 << 17, 40, 181, check=_edges(40))


def case_chunk(oracle):  # several chunks, the last one short
    _complex(oracle, "c64", 1 << 20, 10, 191, check=[0, 3, 4, 9], inverse=False)
    _complex(oracle, "c32", 1 << 19, 40, 192, check=[0, 15, 16, 39], inverse=False)


def case_rfft_regfile_two_passes(oracle):  # rfft of 65536 (f32) / 32768 (f64) reals: one pass by default (post-pass = the kernel's epilogue), two here
    import kofft_amd

    for dt, n in ((np.float32, 65536), (np.float64, 32768)):
        f = kofft_amd.HipFftImpl(dt)
        rows = seeded(64 + n).uniform(-1, 1, (530, n)).astype(dt)
        win = seeded(65 + n).uniform(0.1, 1, n).astype(dt)
        pick = _edges(530)
        assert bits_equal(f.rfft_batch(rows)[pick], oracle.rfft(rows[pick])), f"rfft {n}"
        assert bits_equal(f.rfft_batch(rows, win)[pick], oracle.rfft(rows[pick], win)), f"windowed rfft {n}"


def case_probe_off(oracle):  # 160 MiB chunks: K candidate placements of the intermediate timed by default, none here
    import kofft_amd

    f = kofft_amd.HipFftImpl(np.float64)
    x = rand_c(seeded(195), (10, 1 << 20), np.complex128)
    y = x.copy()
    f.fft_batch(y)
    assert f.big_probe_info()["n"] == 0
    assert bits_equal(y[[0, 9]], oracle.fft(x[[0, 9]]))


KNOBS = [
    ("KOFFT_HIP_NO_PERSIST", "1", case_no_persist),
    ("KOFFT_HIP_PERSIST_GRID_PCT", "50", case_grid_pct),
    ("KOFFT_HIP_BIG_THREE_MIN", "23", case_three_min_high),
    ("KOFFT_HIP_BIG_THREE_MIN", "21", case_three_min_low),
    ("KOFFT_HIP_SMALL32", "0", case_small32),
    ("KOFFT_HIP_BIG_PERSIST", "0", case_big_persist_three),
    ("KOFFT_HIP_RFFT14_WIDE", "0", case_rfft_wide),
    ("KOFFT_HIP_RFFT13_PERSIST", "0", case_rfft_wide),
    ("KOFFT_HIP_PERSIST64", "0", case_persist64),
    ("KOFFT_HIP_PERSIST_SMALL", "0", case_persist_small),
    ("KOFFT_HIP_SPLIT", "0", case_split),
    ("KOFFT_HIP_REGFILE", "0", case_regfile),
    ("KOFFT_HIP_RFFT_REGFILE_EPI", "0", case_rfft_regfile_two_passes),
    ("KOFFT_HIP_HOST_PIPELINE", "1", case_host_pipeline),  # (conftest.py runs the session with the pipeline off: ON is the setting to flip here)
    ("KOFFT_HIP_ZERO_COPY", "0", case_zero_copy),
    ("KOFFT_HIP_ND_TRANSPOSE", "0", case_nd_transpose),
    ("KOFFT_HIP_ND_TWO_PASS", "0", case_nd_two_pass),
    ("KOFFT_HIP_ND_FUSED", "0", case_nd_fused),
    ("KOFFT_HIP_BLUESTEIN_FUSED", "0", case_bluestein),
    ("KOFFT_HIP_BLUESTEIN_ONE", "0", case_bluestein),
    ("KOFFT_HIP_BLUESTEIN_PERSIST", "0", case_bluestein_persist),
    ("KOFFT_HIP_ISTFT_FUSED", "0", case_istft_two_kernels),
    ("KOFFT_HIP_BIG_NARROW", "0", case_big_narrow),
    ("KOFFT_HIP_BIG_FIRST11", "0", case_first11),
    ("KOFFT_HIP_BIG_BLOCKED", "0", case_blocked),
    ("KOFFT_HIP_BIG_ROW_PAIRS", "0", case_big_persist),
    ("KOFFT_HIP_BIG_CHUNK_MB", "64", case_chunk),
    ("KOFFT_HIP_BIG_PROBE", "0", case_probe_off),
]


@pytest.mark.parametrize("name,value,case", KNOBS, ids=[f"{k}={v}" for k, v, _ in KNOBS])
def test_route_switch_in_its_non_default_setting(oracle, monkeypatch, name, value, case):
    monkeypatch.setenv(name, value)
    case(oracle)


def test_placement_probe_of_the_intermediate_runs_once_and_touches_nothing(oracle, monkeypatch):
    """Round 6: the first large-n call of a context (full chunks of >= 128 MiB) times its factor kernels through K candidate
    allocations of the intermediate -- on scratch input / output, so an IN-PLACE call's data is not transformed twice -- and keeps
    the fastest.  Results are those of the oracle; the probe reports its candidates; a second call does not probe again; a smaller
    call (no full 128 MiB chunk) does not probe at all; release_scratch forgets the pick."""
    import torch

    import kofft_amd as K

    monkeypatch.delenv("KOFFT_HIP_BIG_PROBE", raising=False)
    for dt, cdt, n, batch in ((np.float64, np.complex128, 1 << 20, 9), (np.float32, np.complex64, 1 << 20, 17)):
        f = K.HipFftImpl(dt)
        small = rand_c(seeded(301), (2, n), cdt)
        y = small.copy()
        f.fft_batch(y)
        assert f.big_probe_info()["n"] == 0, "a 2-transform call is below the probe's threshold"
        assert bits_equal(y, oracle.fft(small))
        x = rand_c(seeded(302), (batch, n), cdt)
        d = torch.from_numpy(x.view(dt).reshape(batch, n, 2)).to("cuda")
        fn = f._fn(f"fft_{f._cplx}_dev")
        import ctypes as C

        assert fn(f._ctx, C.c_void_p(d.data_ptr()), n, batch, 0) == 0  # IN PLACE: the buffer grows, the probe runs
        f.synchronize()
        info = f.big_probe_info()
        assert info["n"] == 5 and 0 <= info["pick"] < 5, info
        assert info["total_us"][info["pick"]] == min(info["total_us"]) and min(info["first_us"]) > 0, info
        got = d.cpu().numpy().view(cdt).reshape(batch, n)
        pick = [0, batch // 2, batch - 1]
        assert bits_equal(got[pick], oracle.fft(x[pick])), f"{cdt.__name__}: in-place result after the probe"
        assert fn(f._ctx, C.c_void_p(d.data_ptr()), n, batch, 1) == 0  # no second probe: same figures
        f.synchronize()
        assert f.big_probe_info() == info
        f.release_scratch()
        assert f.big_probe_info()["n"] == 0
        f.close()
