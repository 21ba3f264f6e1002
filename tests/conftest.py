"""Shared fixtures.  `-m "not gpu"` runs here (no GPU): oracle, golden vectors, host logic, ABI symbols.
`-m gpu` runs on a real MI355X: parity of the HIP path against the oracle, through the C ABI."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Round 6 (tools/kernel_coverage.sh): host batches of 128 MiB and more go through the chunked upload / kernel / download pipeline
    # (kofft_hip.hip: use_host_pipeline) -- EIGHT kernel launches of batch / 8 transforms each.  The parity tests are written against
    # the kernels' own batch thresholds ("n = 1024 with >= 8192 transforms runs the persistent kernel"); through the pipeline their
    # batches fell below those thresholds and 345 of the library's 867 kernel instantiations -- every register-file kernel, most
    # persistent ones -- were never launched by this suite.  The session therefore runs with the pipeline OFF (one launch per call, on
    # the batch the test names); the pipeline itself is tested where a test switches it ON (test_gpu_streams.py:
    # test_host_pipeline_on_every_entry_point, test_gpu_knobs.py).  Read when a context is created; an explicit setting wins.
    os.environ.setdefault("KOFFT_HIP_HOST_PIPELINE", "0")


_TWICE_MAX_BYTES = 256 << 20  # (larger arrays: the second trip over PCIe costs more than it is likely to find)


def _nan_safe_equal(a, b) -> bool:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.fixture(autouse=True, scope="session")
def _every_batched_host_call_runs_twice():
    """Round 6: the f64 store hazard corrupted a few transforms per THOUSAND, differently on every run -- a test that compares three rows of
    a batch with the oracle does not see that, a second run of the same call does.  For the whole session the batched host-pointer methods of
    HipFftImpl run TWICE on the same input and the two results must be the same bytes (the oracle comparison of the test itself then sees the
    first).  KOFFT_TEST_TWICE=0 switches it off."""
    if os.environ.get("KOFFT_TEST_TWICE", "1") == "0":
        yield
        return
    import kofft_amd

    cls = kofft_amd.HipFftImpl
    saved = {name: getattr(cls, name) for name in ("fft_batch", "rfft_batch", "irfft_batch", "stft_into", "stft_magnitudes", "fftnd")}

    def in_place(name):
        real = saved[name]

        def wrapper(self, data, *args, **kw):
            snap = np.array(data, copy=True) if isinstance(data, np.ndarray) and data.nbytes <= _TWICE_MAX_BYTES else None
            real(self, data, *args, **kw)
            if snap is not None:
                real(self, snap, *args, **kw)
                assert _nan_safe_equal(data, snap), f"{name}: two runs of the same call differ (a sporadic device fault)"
        return wrapper

    def returning(name):
        real = saved[name]

        def wrapper(self, *args, **kw):
            first = real(self, *args, **kw)
            if sum(x.nbytes for x in (first if isinstance(first, tuple) else (first,)) if isinstance(x, np.ndarray)) > _TWICE_MAX_BYTES:
                return first
            second = real(self, *args, **kw)
            a, b = (first, second) if isinstance(first, tuple) else ((first,), (second,))
            for x, y in zip(a, b):
                same = _nan_safe_equal(x, y) if isinstance(x, np.ndarray) else (x == y or (x != x and y != y))
                assert same, f"{name}: two runs of the same call differ (a sporadic device fault)"
            return first
        return wrapper

    for name in ("fft_batch", "fftnd"):
        setattr(cls, name, in_place(name))
    for name in ("rfft_batch", "irfft_batch", "stft_into", "stft_magnitudes"):
        setattr(cls, name, returning(name))
    yield
    for name, fn in saved.items():
        setattr(cls, name, fn)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/pyoracle.py): test infrastructure only."""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hiplib():
    from kofft_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def fft32():
    """A HipFftImpl<f32> on device 0 (GPU tests only)."""
    from kofft_amd import HipFftImpl

    return HipFftImpl(np.float32)


@pytest.fixture(scope="session")
def fft64():
    from kofft_amd import HipFftImpl

    return HipFftImpl(np.float64)


GOLDEN = ROOT / "tests" / "golden"


def seeded(seed: int):
    return np.random.default_rng(seed)


def rand_c(rng, shape, dtype=np.complex64):
    real = np.float32 if dtype == np.complex64 else np.float64
    return (rng.uniform(-1, 1, shape).astype(real) + 1j * rng.uniform(-1, 1, shape).astype(real)).astype(dtype)


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """Bitwise equality (so -0.0 != 0.0 and NaN payloads count)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()
