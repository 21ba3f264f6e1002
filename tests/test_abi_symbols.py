"""The C-ABI library loads without a GPU and exports exactly what include/kofft_hip.h declares."""
import ctypes as C

import numpy as np

from conftest import bits_equal


def test_library_loads_and_exports_every_declared_symbol(hiplib):
    from kofft_amd import _lib

    declared = _lib.header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(hiplib, name), f"{name} declared in include/kofft_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "python prototypes out of step with the header"
    assert b"gfx950" in hiplib.kofft_hip_version()


def test_library_contains_gfx950_code_object():
    from kofft_amd import _lib

    blob = _lib.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"fft_wg_kernel" in blob


def test_strerror_names_follow_ffterror_order(hiplib):
    # fft.rs:447-454 declaration order
    names = [b"EmptyInput", b"NonPowerOfTwoNoStd", b"MismatchedLengths", b"InvalidStride", b"InvalidHopSize",
             b"InvalidValue"]
    for code, name in enumerate(names, start=1):
        assert name in hiplib.kofft_hip_strerror(code)
    assert hiplib.kofft_hip_strerror(0) == b"Ok"


def test_host_table_recipes_match_oracle_bit_for_bit(hiplib, oracle):
    """The product's planner recipes (kofft_amd/csrc/tables.cpp) against the oracle's restatement."""
    import kofft_amd

    for n in (2, 8, 32, 1024, 4096, 65536):
        assert bits_equal(kofft_amd.FftPlanner(np.float32).get_twiddles(n), oracle.get_twiddles(n, np.float32)), n
        assert bits_equal(kofft_amd.FftPlanner(np.float64).get_twiddles(n), oracle.get_twiddles(n, np.float64)), n
    for m in (1, 2, 16, 256, 1024, 2048):
        assert bits_equal(kofft_amd.RfftPlanner(np.float32).get_twiddles(m), oracle.rfft_table(m, np.float32)), m
        assert bits_equal(kofft_amd.RfftPlanner(np.float64).get_twiddles(m), oracle.rfft_table(m, np.float64)), m
    for length in (1, 8, 1024, 2048):
        assert bits_equal(kofft_amd.hann(length), oracle.hann(length))


def test_planner_strategy_and_cache():
    import kofft_amd

    p = kofft_amd.FftPlanner()
    assert p.plan_strategy(4096) is kofft_amd.FftStrategy.SplitRadix  # fft.rs:438-444
    assert p.plan_strategy(12) is kofft_amd.FftStrategy.Auto
    assert p.plan_strategy(1) is kofft_amd.FftStrategy.Auto
    assert p.get_twiddles(8) is p.get_twiddles(8)  # tests/rfft_twiddles.rs:12-15: cached table reused


def test_argument_validation_needs_no_device(hiplib):
    """The reference's error order, checked through the C ABI with a null context (no GPU touched).
    Every FftError variant the hot path can produce is covered; a valid request then reports the null context."""
    null = C.c_void_p(None)
    buf = np.zeros(64, np.float32)
    p = C.c_void_p(buf.ctypes.data)
    sz = C.c_size_t
    # fft.rs:1056 / 1136
    assert hiplib.kofft_hip_fft_c32(null, p, sz(0), sz(1), 0) == 1
    assert hiplib.kofft_hip_fft_c64(null, p, sz(0), sz(1), 1) == 1
    assert hiplib.kofft_hip_fft_c32(null, p, sz(0), sz(0), 0) == 0       # batch() over no slices
    assert hiplib.kofft_hip_fft_c32(null, p, sz(1), sz(1), 0) == 0       # n == 1: Ok, nothing to do
    assert hiplib.kofft_hip_fft_c32(null, p, sz(12), sz(1), 0) == -3     # Bluestein arm: a valid request
    assert hiplib.kofft_hip_fft_c32(null, p, sz(1 << 27), sz(1), 0) == -2   # beyond the two-factor path (2^26)
    assert hiplib.kofft_hip_fft_c32(null, p, sz(8), sz(1), 0) == -3      # valid request, null context
    # fft.rs:1181-1190
    assert hiplib.kofft_hip_fft_c32_strided(null, p, sz(8), sz(0), sz(4), 0) == 4
    assert hiplib.kofft_hip_fft_c32_strided(null, p, sz(8), sz(2), sz(0), 0) == 0
    assert hiplib.kofft_hip_fft_c32_strided(null, p, sz(6), sz(2), sz(4), 0) == 3
    # rfft.rs:433-443 / 476-486
    assert hiplib.kofft_hip_rfft_f32(null, p, p, None, sz(0), sz(1)) == 1
    assert hiplib.kofft_hip_rfft_f32(null, p, p, None, sz(7), sz(1)) == 6
    assert hiplib.kofft_hip_irfft_f32(null, p, p, sz(0), sz(1)) == 1
    assert hiplib.kofft_hip_irfft_f64(null, p, p, sz(9), sz(1)) == 6
    assert hiplib.kofft_hip_rfft_f32(null, p, p, None, sz(8), sz(1)) == -3
    # stft.rs:83-89
    assert hiplib.kofft_hip_stft_f32(null, p, sz(10), p, sz(4), sz(0), p, sz(3)) == 5
    assert hiplib.kofft_hip_stft_f32(null, p, sz(10), p, sz(4), sz(4), p, sz(2)) == 3   # tests/stft.rs:6-14
    assert hiplib.kofft_hip_stft_f32(null, p, sz(0), p, sz(4), sz(2), p, sz(0)) == 0
    assert hiplib.kofft_hip_stft_f32(null, p, sz(10), p, sz(0), sz(4), p, sz(3)) == 1   # empty window -> fft(&mut [])
    assert hiplib.kofft_hip_stft_parallel_f32(null, p, sz(10), p, sz(4), sz(0), p, sz(1)) == 5
    assert hiplib.kofft_hip_stft_parallel_f32(null, p, sz(10), p, sz(4), sz(4), p, sz(2)) == -3  # no frames check
    assert hiplib.kofft_hip_stft_f32_dev(null, p, sz(10), p, sz(4), sz(0), p, sz(0), sz(1)) == 5


def test_multi_gpu_entry_validates_without_a_device(hiplib):
    """kofft_hip_stft_f32_multi: stft::stft's checks in its order (stft.rs:83-87) come before any device call."""
    buf = np.zeros(64, np.float32)
    p = C.c_void_p(buf.ctypes.data)
    sz = C.c_size_t
    f = hiplib.kofft_hip_stft_f32_multi
    assert f(0, p, sz(10), p, sz(4), sz(2), p, sz(5), 0) == 6     # ngpu <= 0 -> InvalidValue
    assert f(-3, p, sz(10), p, sz(4), sz(2), p, sz(5), 1) == 6
    assert f(2, p, sz(10), p, sz(4), sz(0), p, sz(5), 0) == 5     # hop == 0 -> InvalidHopSize
    assert f(2, p, sz(10), p, sz(4), sz(4), p, sz(2), 0) == 3     # frames < ceil(len/hop) -> MismatchedLengths (tests/stft.rs:6-14)
    assert f(2, p, sz(0), p, sz(4), sz(2), p, sz(0), 1) == 0      # nothing to do
    assert f(2, p, sz(10), p, sz(0), sz(4), p, sz(3), 0) == 1     # empty window -> fft(&mut []) -> EmptyInput
    assert f(2, p, sz(10), None, sz(4), sz(4), p, sz(3), 0) == -3  # null window
    assert f(2, p, sz(10), p, sz(4), sz(4), None, sz(3), 0) == -3  # null output
    # handle form: null handle / bad arguments
    h = C.c_void_p()
    assert hiplib.kofft_hip_multi_create(0, None, C.byref(h)) == 6 and not h.value
    assert hiplib.kofft_hip_multi_create(1, None, None) == -3
    assert hiplib.kofft_hip_multi_destroy(None) == -3
    assert hiplib.kofft_hip_multi_ngpu(None) == 0
    assert hiplib.kofft_hip_multi_stft_f32(None, p, sz(10), p, sz(4), sz(0), p, sz(3), 0, None) == 5
    assert hiplib.kofft_hip_multi_stft_f32(None, p, sz(10), p, sz(4), sz(4), p, sz(2), 0, None) == 3
    assert hiplib.kofft_hip_multi_stft_f32(None, p, sz(10), p, sz(4), sz(4), p, sz(3), 0, None) == -3
    assert hiplib.kofft_hip_multi_fft_c32(None, p, sz(0), sz(1), 0) == 1
    assert hiplib.kofft_hip_multi_fft_c32(None, p, sz(8), sz(0), 0) == 0
    assert hiplib.kofft_hip_multi_fft_c32(None, p, sz(8), sz(1), 0) == -3
    assert b"RCCL" in hiplib.kofft_hip_strerror(-5)


def test_every_route_switch_of_the_library_has_a_gpu_test_row():
    """tests/test_gpu_knobs.py flips every KOFFT_HIP_* switch kofft_hip_create reads; the table there must list exactly those."""
    import re
    import sys
    from pathlib import Path

    ROOT = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_knobs import KNOBS

    src = (ROOT / "kofft_amd" / "csrc" / "kofft_hip.hip").read_text()
    in_source = set(re.findall(r'getenv\("(KOFFT_HIP_[A-Z0-9_]+)"\)', src))
    in_table = {k for k, _, _ in KNOBS}
    assert in_source == in_table, (sorted(in_source - in_table), sorted(in_table - in_source))
