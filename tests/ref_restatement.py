"""A SECOND, independent restatement of the reference's hot-path arithmetic (test infrastructure only).

Why it exists: the reference (okian/kofft, Rust) cannot be built in this image and holds no golden vectors, so the C
oracle under oracle/ is pinned by nothing the reference produced.  The best substitute available here is a second
transcription made straight from the .rs sources with different machinery, and a test (test_oracle_second_opinion.py)
that both agree bit for bit.  A shared misreading of the source would still pass; a slip in either transcription, in a
compiler flag, or in a libm assumption does not.

Independent of oracle/ in:
  * language and structure -- numpy array operations over a whole Stockham stage at a time (every numpy ufunc call is
    one IEEE operation per element, rounded to the array's dtype; nothing is fused) instead of scalar C loops;
  * FMA -- `mul_add` is evaluated in exact rational arithmetic and rounded ONCE to the target type (ties to even),
    instead of calling fmaf()/fma();
  * trigonometry -- glibc libm through ctypes (Rust's f32::sin / cos / sin_cos lower to sinf / cosf / sincosf on
    x86_64-unknown-linux-gnu); `sin_cos_separate` lets a test measure the one assumption that cannot be read off the
    source: whether LLVM merges the two calls into sincos[f], and where that changes a bit.

Each function cites the reference lines it follows (okian/kofft v0.1.5).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
from fractions import Fraction

import numpy as np

_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _name, _t in (("sinf", C.c_float), ("cosf", C.c_float), ("sin", C.c_double), ("cos", C.c_double)):
    getattr(_libm, _name).restype = _t
    getattr(_libm, _name).argtypes = [_t]
_libm.sincosf.restype = None
_libm.sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
_libm.sincos.restype = None
_libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]


def _real(dtype):
    return np.float32 if np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64


def _cplx(dtype):
    return np.complex64 if _real(dtype) is np.float32 else np.complex128


def sin_cos_separate(x, dtype):
    """(sin, cos) from two libm calls: what Rust's std source says literally (`(self.sin(), self.cos())`)."""
    if _real(dtype) is np.float32:
        return np.float32(_libm.sinf(float(x))), np.float32(_libm.cosf(float(x)))
    return np.float64(_libm.sin(float(x))), np.float64(_libm.cos(float(x)))


def sin_cos(x, dtype):
    """Float::sin_cos (num.rs:56-58 / 93-95) as a linux-gnu build executes it: LLVM merges the sin and the cos of one
    operand into ONE sincosf / sincos libcall.  (f32: bit-identical to two calls; f64: not always -- see the tests.)"""
    if _real(dtype) is np.float32:
        s, c = C.c_float(), C.c_float()
        _libm.sincosf(float(x), C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)
    s, c = C.c_double(), C.c_double()
    _libm.sincos(float(x), C.byref(s), C.byref(c))
    return np.float64(s.value), np.float64(c.value)


def _round_once(q: Fraction, real, sign_hint: float):
    """Exact rational -> nearest `real`, ties to even.  sign_hint supplies the sign of an exact zero."""
    if q == 0:
        return real(np.copysign(0.0, sign_hint))
    if real is np.float64:
        return np.float64(q.numerator / q.denominator)  # int / int true division is correctly rounded
    c = np.float32(q.numerator / q.denominator)  # within one ulp of the answer (double rounding): test the neighbours
    cands = [np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))]
    best = None
    for v in cands:
        if not np.isfinite(v):
            continue
        err = abs(Fraction(float(v)) - q)
        even = (int(np.float32(v).view(np.uint32)) & 1) == 0
        key = (err, 0 if even else 1)
        if best is None or key < best[0]:
            best = (key, v)
    return np.float32(best[1])


def mul_add(a, b, c, real):
    """Float::mul_add (num.rs:62-65 / 99-102) = f32::mul_add / f64::mul_add: a*b + c with ONE rounding."""
    fa, fb, fc = float(a), float(b), float(c)
    return _round_once(Fraction(fa) * Fraction(fb) + Fraction(fc), real, fa * fb + fc)


# ---- planner tables ---------------------------------------------------------------------------------------------
def get_twiddles(n: int, dtype=np.float32) -> np.ndarray:
    """FftPlanner::get_twiddles (fft.rs:391-405)."""
    real = _real(dtype)
    half = n // 2
    # let angle = -T::from_f32(2.0) * T::pi() / T::from_f32(n as f32);
    pi = np.float32(3.14159274) if real is np.float32 else np.float64(np.pi)
    angle = (-real(np.float32(2.0))) * pi / real(np.float32(n))
    sin_step, cos_step = sin_cos(angle, real)
    out = np.empty(half, _cplx(dtype))
    w_re, w_im = real(1.0), real(0.0)
    for k in range(half):
        out[k] = complex(w_re, w_im)
        tmp = w_re
        w_re = mul_add(w_re, cos_step, -(w_im * sin_step), real)
        w_im = mul_add(w_im, cos_step, tmp * sin_step, real)
    return out


def rfft_table(m: int, dtype=np.float32) -> np.ndarray:
    """build_twiddle_table (rfft.rs:172-183): current = current.mul(w), the un-fused Complex::mul (num.rs:161-166)."""
    real = _real(dtype)
    pi = np.float32(3.14159274) if real is np.float32 else np.float64(np.pi)
    angle = -pi / real(np.float32(m))
    s, c = sin_cos(angle, real)
    out = np.empty(m, _cplx(dtype))
    cr, ci = real(1.0), real(0.0)
    for k in range(m):
        out[k] = complex(cr, ci)
        cr, ci = real(cr * c - ci * s), real(cr * s + ci * c)
    return out


def hann(length: int) -> np.ndarray:
    """window::hann (window.rs:24-28): 0.5 - 0.5 * (2.0 * PI * i as f32 / len as f32).cos(), all f32."""
    two_pi = np.float32(2.0) * np.float32(3.14159274)
    out = np.empty(length, np.float32)
    for i in range(length):
        x = np.float32(two_pi * np.float32(i)) / np.float32(length)
        out[i] = np.float32(0.5) - np.float32(0.5) * np.float32(_libm.cosf(float(x)))
    return out


# ---- complex helpers (num.rs:127-166, the non-FMA arm) on split arrays -------------------------------------------
def _mul(ar, ai, br, bi):
    return ar * br - ai * bi, ar * bi + ai * br


def _split(x, dtype):
    real = _real(dtype)
    x = np.asarray(x)
    return np.ascontiguousarray(x.real, real).copy(), np.ascontiguousarray(x.imag, real).copy()


def _join(re, im, dtype):
    out = np.empty(re.shape, _cplx(dtype))
    out.real, out.imag = re, im
    return out


# ---- small kernels (fft_kernels.rs) -------------------------------------------------------------------------------
def _small(re, im, real):
    """fft2 / fft4 / fft8 / fft16 (fft_kernels.rs:4-224) on python lists of numpy scalars."""
    n = len(re)
    x = [(real(re[i]), real(im[i])) for i in range(n)]
    add = lambda a, b: (a[0] + b[0], a[1] + b[1])  # noqa: E731
    sub = lambda a, b: (a[0] - b[0], a[1] - b[1])  # noqa: E731
    mul = lambda a, b: (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])  # noqa: E731
    w1 = (real(0.0), -real(1.0))
    s = real(np.float32(0.70710677))

    def quad(p0, p1, p2, p3):  # the fft4 body: inputs (x0, x2 | x1, x3) pattern of fft_kernels.rs:13-29
        a0, a1, a2, a3 = add(p0, p2), sub(p0, p2), add(p1, p3), sub(p1, p3)
        t = mul(a3, w1)
        return add(a0, a2), add(a1, t), sub(a0, a2), sub(a1, t)

    def oct_(ev, od):  # fft_kernels.rs:69-85: e + o*tw, e - o*tw with tw = 1, (s,-s), (0,-1), (-s,-s)
        t = [od[0], mul(od[1], (s, -s)), mul(od[2], w1), mul(od[3], (-s, -s))]
        return [add(ev[i], t[i]) for i in range(4)] + [sub(ev[i], t[i]) for i in range(4)]

    if n == 2:
        y = [add(x[0], x[1]), sub(x[0], x[1])]
    elif n == 4:
        # fft_kernels.rs:13-29: even0 = a0+a2, even1 = a0-a2, odd0 = a1+a3, odd1 = a1-a3
        q = quad(x[0], x[1], x[2], x[3])
        y = [q[0], q[1], q[2], q[3]]
    elif n == 8:
        ev = quad(x[0], x[2], x[4], x[6])   # fft_kernels.rs:46-56
        od = quad(x[1], x[3], x[5], x[7])   # 58-67
        y = oct_(ev, od)
    else:
        ea = quad(x[0], x[4], x[8], x[12])  # fft_kernels.rs:112-121
        eb = quad(x[2], x[6], x[10], x[14])  # 123-131
        e8 = oct_(ea, eb)                    # 133-144
        oa = quad(x[1], x[5], x[9], x[13])   # 147-155
        ob = quad(x[3], x[7], x[11], x[15])  # 157-165
        o8 = oct_(oa, ob)                    # 167-178
        c1, s1 = real(np.float32(0.9238795)), real(np.float32(-0.38268343))
        c2, s2 = real(np.float32(0.70710677)), real(np.float32(-0.70710677))
        c3, s3 = real(np.float32(0.38268343)), real(np.float32(-0.9238795))
        c4, s4 = real(0.0), real(np.float32(-1.0))
        tw = [None, (c1, s1), (c2, s2), (c3, s3), (c4, s4), (-c3, s3), (-c2, s2), (-c1, s1)]  # 181-197
        o = [o8[0]] + [mul(o8[i], tw[i]) for i in range(1, 8)]
        y = [add(e8[i], o[i]) for i in range(8)] + [sub(e8[i], o[i]) for i in range(8)]
    return [v[0] for v in y], [v[1] for v in y]


# ---- the transform -------------------------------------------------------------------------------------------------
def _stockham(re, im, tw):
    """fft_split_simd (fft.rs:834-898 f32, 959-1037 f64): radix-2 Stockham autosort, one numpy op per IEEE operation."""
    n = re.size
    twr, twi = np.ascontiguousarray(tw.real), np.ascontiguousarray(tw.imag)
    n1, n2 = 1, n
    while n1 < n:
        n2 >>= 1
        sr, si = re.reshape(n1, 2, n2), im.reshape(n1, 2, n2)
        wr, wi = twr[np.arange(n1) * n2][:, None], twi[np.arange(n1) * n2][:, None]
        er, ei, odr, odi = sr[:, 0, :], si[:, 0, :], sr[:, 1, :], si[:, 1, :]
        t_re = odr * wr - odi * wi
        t_im = odr * wi + odi * wr
        dr, di = np.empty((2, n1, n2), re.dtype), np.empty((2, n1, n2), re.dtype)
        dr[0], di[0] = er + t_re, ei + t_im   # dst[k*n2 + j]
        dr[1], di[1] = er - t_re, ei - t_im   # dst[(k + n1)*n2 + j]
        re, im = dr.reshape(n), di.reshape(n)
        n1 <<= 1
    return re, im


_tw_cache: dict = {}
_blue_cache: dict = {}


def _twiddles_cached(n, real):
    key = (n, real)
    if key not in _tw_cache:
        _tw_cache[key] = get_twiddles(n, real)
    return _tw_cache[key]


def _fft_1d(re, im, real):
    """ScalarFftImpl::fft (fft.rs:1054-1132) on one transform."""
    n = re.size
    if n == 0:
        raise ValueError("EmptyInput")
    if n == 1:
        return re, im
    if n <= 16 and n & (n - 1) == 0:
        r, i = _small(list(re), list(im), real)
        return np.array(r, real), np.array(i, real)
    if n & (n - 1) == 0:
        return _stockham(re, im, _twiddles_cached(n, real))
    # Bluestein (fft.rs:1088-1132) with get_bluestein's tables (fft.rs:411-433)
    key = (n, real)
    if key not in _blue_cache:
        m = 1 << (2 * n - 2).bit_length()   # (2n-1).next_power_of_two()
        if m < 2 * n - 1:
            m <<= 1
        pi = np.float32(3.14159274) if real is np.float32 else np.float64(np.pi)
        cr, ci = np.empty(n, real), np.empty(n, real)
        br, bi = np.zeros(m, real), np.zeros(m, real)
        for i in range(n):
            angle = pi * real(np.float32(i * i)) / real(np.float32(n))
            s, c = sin_cos(-angle, real)
            cr[i], ci[i] = c, s
            s, c = sin_cos(angle, real)
            br[i], bi[i] = c, s
        for i in range(1, n):
            br[m - i], bi[m - i] = br[i], bi[i]
        fr, fi = _fft_1d(br, bi, real)
        _blue_cache[key] = (cr, ci, fr, fi, m)
    cr, ci, fr, fi, m = _blue_cache[key]
    ar, ai = np.zeros(m, real), np.zeros(m, real)
    ar[:n], ai[:n] = _mul(re, im, cr, ci)
    ar, ai = _fft_1d(ar, ai, real)
    ar, ai = _mul(ar, ai, fr, fi)
    ai = -ai
    ar, ai = _fft_1d(ar, ai, real)
    ai = -ai
    scale = real(1.0) / real(np.float32(m))
    ar, ai = ar * scale, ai * scale
    return _mul(ar[:n], ai[:n], cr, ci)


def fft(x: np.ndarray, inverse: bool = False) -> np.ndarray:
    """fft (fft.rs:1054) / ifft (fft.rs:1134-1174: conj, fft, conj, * 1/(n as f32)) over the last axis."""
    x = np.asarray(x)
    real = _real(x.dtype)
    flat = x.reshape(-1, x.shape[-1])
    out = np.empty(flat.shape, _cplx(x.dtype))
    n = x.shape[-1]
    for b in range(flat.shape[0]):
        re, im = _split(flat[b], real)
        if inverse and n > 1:
            im = -im
        re, im = _fft_1d(re, im, real)
        if inverse and n > 1:
            scale = real(1.0) / real(np.float32(n))
            im = -im
            re, im = re * scale, im * scale
        out[b] = _join(re, im, real)
    return out.reshape(x.shape)


def rfft(x: np.ndarray, window: np.ndarray | None = None) -> np.ndarray:
    """rfft_direct (rfft.rs:425-465) over the last axis; the optional window is the framing product of stft.rs:96."""
    x = np.asarray(x)
    real = _real(x.dtype)
    flat = x.reshape(-1, x.shape[-1])
    n = x.shape[-1]
    m = n // 2
    tab = rfft_table(m, real)
    wr, wi = np.ascontiguousarray(tab.real), np.ascontiguousarray(tab.imag)
    out = np.empty((flat.shape[0], m + 1), _cplx(real))
    half = real(np.float32(0.5))
    for b in range(flat.shape[0]):
        row = flat[b] if window is None else flat[b] * np.asarray(window, real)
        yr, yi = _fft_1d(row[0::2].astype(real).copy(), row[1::2].astype(real).copy(), real)
        o_re, o_im = np.empty(m + 1, real), np.empty(m + 1, real)
        o_re[0], o_im[0] = yr[0] + yi[0], real(0.0)
        o_re[m], o_im[m] = yr[0] - yi[0], real(0.0)
        k = np.arange(1, m)
        a_r, a_i = yr[k], yi[k]
        b_r, b_i = yr[m - k], -yi[m - k]
        sum_r, sum_i, dif_r, dif_i = a_r + b_r, a_i + b_i, a_r - b_r, a_i - b_i
        t_r, t_i = _mul(wr[k], wi[k], dif_r, dif_i)
        o_re[1:m], o_im[1:m] = (sum_r + t_i) * half, (sum_i + (-t_r)) * half
        out[b] = _join(o_re, o_im, real)
    return out.reshape(x.shape[:-1] + (m + 1,))


def irfft(x: np.ndarray, n: int) -> np.ndarray:
    """irfft_direct (rfft.rs:468-508) over the last axis."""
    x = np.asarray(x)
    real = _real(x.dtype)
    m = n // 2
    flat = x.reshape(-1, m + 1)
    tab = rfft_table(m, real)
    wr, wi = np.ascontiguousarray(tab.real), -np.ascontiguousarray(tab.imag)
    out = np.empty((flat.shape[0], n), real)
    half = real(np.float32(0.5))
    for b in range(flat.shape[0]):
        xr, xi = _split(flat[b], real)
        s_r, s_i = np.empty(m, real), np.empty(m, real)
        s_r[0], s_i[0] = (xr[0] + xr[m]) * half, (xr[0] - xr[m]) * half
        k = np.arange(1, m)
        a_r, a_i, b_r, b_i = xr[k], xi[k], xr[m - k], -xi[m - k]
        sum_r, sum_i, dif_r, dif_i = a_r + b_r, a_i + b_i, a_r - b_r, a_i - b_i
        t_r, t_i = _mul(wr[k], wi[k], dif_r, dif_i)
        s_r[1:], s_i[1:] = (sum_r - t_i) * half, (sum_i - (-t_r)) * half
        if m > 1:  # fft.ifft: n == 1 returns before any conjugation (fft.rs:1139)
            s_i = -s_i
            s_r, s_i = _fft_1d(s_r, s_i, real)
            scale = real(1.0) / real(np.float32(m))
            s_i = -s_i
            s_r, s_i = s_r * scale, s_i * scale
        out[b, 0::2], out[b, 1::2] = s_r, s_i
    return out.reshape(x.shape[:-1] + (n,))


def stft(signal: np.ndarray, window: np.ndarray, hop: int, frames: int) -> np.ndarray:
    """stft::stft (stft.rs:91-103): every provided frame, zero past the end of the signal."""
    signal = np.asarray(signal, np.float32)
    window = np.asarray(window, np.float32)
    wl = window.size
    out = np.empty((frames, wl), np.complex64)
    for f in range(frames):
        start = f * hop
        seg = np.zeros(wl, np.float32)
        have = max(0, min(wl, signal.size - start))
        seg[:have] = signal[start:start + have] * window[:have]
        re, im = _fft_1d(seg, np.zeros(wl, np.float32), np.float32)
        out[f] = _join(re, im, np.float32)
    return out


def istft(frames: np.ndarray, window: np.ndarray, hop: int, out_len: int):
    """stft::istft (stft.rs:117-156) from a zeroed output: (output, window-square sums)."""
    window = np.asarray(window, np.float32)
    wl = window.size
    output, scratch = np.zeros(out_len, np.float32), np.zeros(out_len, np.float32)
    time = fft(np.asarray(frames, np.complex64), inverse=True)
    for f in range(time.shape[0]):
        start = f * hop
        have = max(0, min(wl, out_len - start))
        output[start:start + have] = output[start:start + have] + time[f, :have].real * window[:have]
        scratch[start:start + have] = scratch[start:start + have] + window[:have] * window[:have]
    big = scratch > np.float32(1e-8)
    output[big] = output[big] / scratch[big]
    return output, scratch


def stft_magnitudes(samples: np.ndarray, win_len: int, hop: int):
    """visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76)."""
    samples = np.asarray(samples, np.float32)
    frames = -(-samples.size // hop)
    spec = stft(samples, hann(win_len), hop, frames)[:, : win_len // 2]
    re, im = np.ascontiguousarray(spec.real), np.ascontiguousarray(spec.imag)
    mags = np.sqrt(re * re + im * im)
    mx = np.float32(0.0)
    for v in mags.ravel():  # `if mag > max_mag`: a NaN is never selected
        if v > mx:
            mx = v
    return mags.astype(np.float32), float(mx)
