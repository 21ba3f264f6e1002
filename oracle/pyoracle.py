"""numpy/ctypes face of the CPU oracle (oracle/libkofft_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/kofft_oracle.h.  Importable from tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg; never from kofft_amd/.
Function names follow the reference's (fft / ifft / rfft / irfft / stft / istft / hann /
get_twiddles); every call goes to the C restatement, which cites the reference lines.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "libkofft_oracle.so"
_lib = None


class OracleError(Exception):
    """Positive codes are kofft FftError discriminant+1 (same numbering as the C ABI)."""

    def __init__(self, code: int):
        self.code = int(code)
        super().__init__(f"oracle status {code}: {lib().ko_strerror(code).decode()}")


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    srcs = [_DIR / "kofft_oracle.c", _DIR / "kofft_oracle_impl.inc", _DIR / "kofft_oracle.h"]
    if force or not _SO.exists() or any(s.stat().st_mtime > _SO.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_DIR), "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        _lib.ko_strerror.restype = C.c_char_p
        for name in ("ko_planner_new_f32", "ko_planner_new_f64"):
            getattr(_lib, name).restype = C.c_void_p
    return _lib


_SZ = C.c_size_t


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _sfx(dtype) -> str:
    dt = np.dtype(dtype)
    if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
        return "f32"
    if dt in (np.dtype(np.float64), np.dtype(np.complex128)):
        return "f64"
    raise TypeError(dt)


def _cdt(sfx):
    return np.complex64 if sfx == "f32" else np.complex128


def _rdt(sfx):
    return np.float32 if sfx == "f32" else np.float64


def get_twiddles(n: int, dtype=np.float32) -> np.ndarray:
    """FftPlanner::get_twiddles (fft.rs:370-408)."""
    s = _sfx(dtype)
    out = np.empty(n // 2, _cdt(s))
    _chk(getattr(lib(), f"ko_twiddles_{s}")(_SZ(n), _p(out)))
    return out


def rfft_table(m: int, dtype=np.float32) -> np.ndarray:
    """build_twiddle_table (rfft.rs:172-183)."""
    s = _sfx(dtype)
    out = np.empty(m, _cdt(s))
    _chk(getattr(lib(), f"ko_rfft_table_{s}")(_SZ(m), _p(out)))
    return out


def hann(length: int) -> np.ndarray:
    out = np.empty(length, np.float32)
    _chk(lib().ko_hann_f32(_SZ(length), _p(out)))
    return out


def fft(x: np.ndarray, inverse: bool = False) -> np.ndarray:
    """ScalarFftImpl::fft / ifft over the last axis (all leading axes are the batch).  Returns a new array."""
    s = _sfx(x.dtype)
    a = np.ascontiguousarray(x, _cdt(s)).copy()
    n = a.shape[-1] if a.ndim else 0
    batch = a.size // n if n else (1 if a.ndim <= 1 else int(np.prod(a.shape[:-1])))
    _chk(getattr(lib(), f"ko_fft_batch_{s}")(_p(a), _SZ(n), _SZ(batch), int(bool(inverse))))
    return a


def ifft(x: np.ndarray) -> np.ndarray:
    return fft(x, inverse=True)


def fft_radix4(x: np.ndarray) -> np.ndarray:
    """ScalarFftImpl::fft_radix4 (fft.rs:1455-1548) over the last axis: what fft_with_strategy(.., Radix4) computes."""
    s = _sfx(x.dtype)
    a = np.ascontiguousarray(x, _cdt(s)).copy()
    n = a.shape[-1] if a.ndim else 0
    batch = a.size // n if n else (1 if a.ndim <= 1 else int(np.prod(a.shape[:-1])))
    _chk(getattr(lib(), f"ko_fft_radix4_batch_{s}")(_p(a), _SZ(n), _SZ(batch)))
    return a


def rfft(x: np.ndarray, window: np.ndarray | None = None) -> np.ndarray:
    """RfftPlanner::rfft_with_scratch over the last axis; optional row window multiplied in first."""
    s = _sfx(x.dtype)
    a = np.ascontiguousarray(x, _rdt(s))
    n = a.shape[-1]
    batch = a.size // n if n else 1
    out = np.empty(a.shape[:-1] + (n // 2 + 1,), _cdt(s))
    w = None if window is None else np.ascontiguousarray(window, _rdt(s))
    _chk(getattr(lib(), f"ko_rfft_batch_{s}")(_p(a), _p(out), _p(w), _SZ(n), _SZ(batch)))
    return out


def irfft(x: np.ndarray, n: int) -> np.ndarray:
    s = _sfx(x.dtype)
    a = np.ascontiguousarray(x, _cdt(s))
    batch = a.size // a.shape[-1] if a.shape[-1] else 1
    out = np.empty(a.shape[:-1] + (n,), _rdt(s))
    _chk(getattr(lib(), f"ko_irfft_batch_{s}")(_p(a), _p(out), _SZ(n), _SZ(batch)))
    return out


def stft(signal: np.ndarray, window: np.ndarray, hop: int, frames: int) -> np.ndarray:
    """stft::stft (stft.rs:76-105) into a [frames, win_len] array."""
    sig = np.ascontiguousarray(signal, np.float32)
    win = np.ascontiguousarray(window, np.float32)
    out = np.zeros((frames, win.size), np.complex64)
    _chk(lib().ko_stft_f32(_p(sig), _SZ(sig.size), _p(win), _SZ(win.size), _SZ(hop), _p(out), _SZ(frames)))
    return out


def stft_range(signal: np.ndarray, window: np.ndarray, hop: int, first: int, count: int) -> np.ndarray:
    sig = np.ascontiguousarray(signal, np.float32)
    win = np.ascontiguousarray(window, np.float32)
    out = np.zeros((count, win.size), np.complex64)
    _chk(lib().ko_stft_range_f32(_p(sig), _SZ(sig.size), _p(win), _SZ(win.size), _SZ(hop), _p(out), _SZ(first),
                                 _SZ(count)))
    return out


def istft(frames: np.ndarray, window: np.ndarray, hop: int, out_len: int) -> np.ndarray:
    """stft::istft (stft.rs:117-156) into a fresh zero output."""
    fr = np.ascontiguousarray(frames, np.complex64).copy()
    win = np.ascontiguousarray(window, np.float32)
    out = np.zeros(out_len, np.float32)
    scratch = np.zeros(out_len, np.float32)
    _chk(lib().ko_istft_f32(_p(fr), _SZ(fr.shape[0]), _p(win), _SZ(win.size), _SZ(hop), _p(out), _SZ(out_len),
                            _p(scratch), _SZ(out_len)))
    return out


def stft_magnitudes(samples: np.ndarray, win_len: int, hop: int):
    """visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): (mags[frames, win_len/2], max_mag)."""
    sig = np.ascontiguousarray(samples, np.float32)
    frames = -(-sig.size // int(hop)) if hop else 0
    mags = np.zeros((frames, int(win_len) // 2), np.float32)
    mx = C.c_float(0.0)
    _chk(lib().ko_stft_magnitudes_f32(_p(sig), _SZ(sig.size), _SZ(int(win_len)), _SZ(int(hop)), _p(mags), C.byref(mx)))
    return mags, float(mx.value)


# ---- threaded batch entries (full-size parity tests: EVERY transform of a BASELINE config against the oracle) ----------------
# Each worker thread calls the plain C batch entry on its own contiguous block of the batch (one planner per call, nothing shared:
# the C side has no global state); ctypes releases the GIL for the duration of the call.
def host_threads() -> int:
    """Cores this process may use: affinity mask, capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def _run_blocks(total: int, threads: int | None, work) -> None:
    """work(first, count) over `total` units split into contiguous blocks, one thread each; the first failure is re-raised."""
    if total <= 0:
        return  # an empty batch: nothing to do, like the serial entries
    k = max(1, min(threads or host_threads(), total))
    per = -(-total // k)
    errs: list[BaseException] = []

    def run(first):
        try:
            work(first, min(per, total - first))
        except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
            errs.append(e)
    ths = [threading.Thread(target=run, args=(f,)) for f in range(0, total, per)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]


def fft_inplace_mt(a: np.ndarray, inverse: bool = False, threads: int | None = None) -> np.ndarray:
    """ScalarFftImpl::fft / ifft on every row of the C-contiguous [batch, n] complex array `a`, IN PLACE (no copy), batch split over threads."""
    s = _sfx(a.dtype)
    assert a.ndim == 2 and a.flags.c_contiguous and a.dtype == _cdt(s)
    batch, n = a.shape
    fn = getattr(lib(), f"ko_fft_batch_{s}")
    row = n * a.itemsize

    def work(first, count):
        _chk(fn(C.c_void_p(a.ctypes.data + first * row), _SZ(n), _SZ(count), int(bool(inverse))))
    _run_blocks(batch, threads, work)
    return a


def rfft_mt(x: np.ndarray, window: np.ndarray | None = None, threads: int | None = None) -> np.ndarray:
    """RfftPlanner::rfft_with_scratch on every row of the [batch, n] real array (optional window product first): [batch, n/2+1] complex."""
    s = _sfx(x.dtype)
    a = np.ascontiguousarray(x, _rdt(s))
    assert a.ndim == 2
    batch, n = a.shape
    out = np.empty((batch, n // 2 + 1), _cdt(s))
    w = None if window is None else np.ascontiguousarray(window, _rdt(s))
    fn = getattr(lib(), f"ko_rfft_batch_{s}")

    def work(first, count):
        _chk(fn(C.c_void_p(a.ctypes.data + first * n * a.itemsize), C.c_void_p(out.ctypes.data + first * (n // 2 + 1) * out.itemsize),
                _p(w), _SZ(n), _SZ(count)))
    _run_blocks(batch, threads, work)
    return out


def stft_mt(signal: np.ndarray, window: np.ndarray, hop: int, frames: int, threads: int | None = None) -> np.ndarray:
    """stft::stft (stft.rs:76-105), frame ranges split over threads (ko_stft_range_f32: the same arithmetic per frame)."""
    sig = np.ascontiguousarray(signal, np.float32)
    win = np.ascontiguousarray(window, np.float32)
    out = np.zeros((frames, win.size), np.complex64)
    if hop == 0 or frames < -(-sig.size // hop) or frames == 0 or win.size == 0:
        return stft(sig, win, hop, frames)  # the error paths / empty cases: the serial entry decides

    def work(first, count):
        _chk(lib().ko_stft_range_f32(_p(sig), _SZ(sig.size), _p(win), _SZ(win.size), _SZ(hop),
                                     C.c_void_p(out.ctypes.data + first * win.size * 8), _SZ(first), _SZ(count)))
    _run_blocks(frames, threads, work)
    return out
