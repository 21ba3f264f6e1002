"""ctypes binding of the C ABI declared in include/kofft_hip.h.

The shared library is the product; there is no CPU fallback.  If it has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C
kofft_amd/csrc``) every use raises :class:`LibraryMissing`.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libkofft_hip.so"
HEADER_PATH = _PKG.parent / "include" / "kofft_hip.h"


class LibraryMissing(RuntimeError):
    pass


_c_f = C.POINTER(C.c_float)
_c_d = C.POINTER(C.c_double)
_sz = C.c_size_t
_ctx = C.c_void_p

# name -> (restype, argtypes).  Kept in step with include/kofft_hip.h;
# tests/test_abi_symbols.py parses the header and checks both directions.
SIGNATURES = {
    "kofft_hip_strerror": (C.c_char_p, [C.c_int]),
    "kofft_hip_last_error": (C.c_char_p, [_ctx]),
    "kofft_hip_version": (C.c_char_p, []),
    "kofft_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kofft_hip_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "kofft_hip_destroy": (C.c_int, [_ctx]),
    "kofft_hip_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "kofft_hip_synchronize": (C.c_int, [_ctx]),
    "kofft_hip_release_scratch": (C.c_int, [_ctx]),
    "kofft_hip_big_probe_info": (C.c_int, [_ctx, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kofft_hip_twiddles_f32": (C.c_int, [_sz, C.c_void_p]),
    "kofft_hip_twiddles_f64": (C.c_int, [_sz, C.c_void_p]),
    "kofft_hip_rfft_table_f32": (C.c_int, [_sz, C.c_void_p]),
    "kofft_hip_rfft_table_f64": (C.c_int, [_sz, C.c_void_p]),
    "kofft_hip_hann_f32": (C.c_int, [_sz, C.c_void_p]),
    "kofft_hip_fft_c32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c64": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c32_dev": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c64_dev": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c32_dev_oop": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c64_dev_oop": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_fft_radix4_c32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz]),
    "kofft_hip_fft_radix4_c64": (C.c_int, [_ctx, C.c_void_p, _sz, _sz]),
    "kofft_hip_fft_radix4_c32_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_fft_radix4_c64_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_ifft_radix4_c32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz]),
    "kofft_hip_ifft_radix4_c64": (C.c_int, [_ctx, C.c_void_p, _sz, _sz]),
    "kofft_hip_ifft_radix4_c32_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_ifft_radix4_c64_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_fft_c32_strided": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_fft_c64_strided": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_rfft_f32": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_rfft_f32_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_irfft_f32": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_irfft_f32_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_rfft_f64": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_rfft_f64_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_irfft_f64": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_irfft_f64_dev": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz]),
    "kofft_hip_stft_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz]),
    "kofft_hip_stft_parallel_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz]),
    "kofft_hip_stft_frame_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p]),
    "kofft_hip_istft_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz, C.c_void_p, _sz]),
    "kofft_hip_istft_parallel_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz]),
    "kofft_hip_istft_frame_f32": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, _sz, _sz, C.c_void_p, _sz]),
    "kofft_hip_istft_f32_dev": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz, C.c_void_p, _sz]),
    "kofft_hip_stft_magnitudes_f32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_void_p, _sz, C.c_void_p]),
    "kofft_hip_stft_magnitudes_f32_dev": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_void_p, _sz, C.c_void_p]),
    "kofft_hip_fftnd_c32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_fftnd_c64": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_fftnd_c32_dev": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_fftnd_c64_dev": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, _sz, C.c_int]),
    "kofft_hip_stft_f32_dev": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz, _sz]),
    # multi-GPU (single process, one context per device; RCCL bound at run time)
    "kofft_hip_multi_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_ctx)]),
    "kofft_hip_multi_destroy": (C.c_int, [_ctx]),
    "kofft_hip_multi_last_error": (C.c_char_p, [_ctx]),
    "kofft_hip_multi_ngpu": (C.c_int, [_ctx]),
    "kofft_hip_multi_shard": (C.c_int, [_ctx, _sz, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "kofft_hip_multi_last_timing": (C.c_int, [_ctx, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "kofft_hip_multi_stft_f32": (C.c_int, [_ctx, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz, C.c_int, C.POINTER(C.c_void_p)]),
    "kofft_hip_stft_f32_multi": (C.c_int, [C.c_int, C.c_void_p, _sz, C.c_void_p, _sz, _sz, C.c_void_p, _sz, C.c_int]),
    "kofft_hip_multi_fft_c32": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_multi_fft_c64": (C.c_int, [_ctx, C.c_void_p, _sz, _sz, C.c_int]),
    "kofft_hip_multi_rfft_f32": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, _sz, _sz]),
    # device-resident twins: arrays of ngpu device pointers, asynchronous
    "kofft_hip_multi_fft_c32_dev": (C.c_int, [_ctx, C.POINTER(C.c_void_p), _sz, _sz, C.c_int]),
    "kofft_hip_multi_fft_c64_dev": (C.c_int, [_ctx, C.POINTER(C.c_void_p), _sz, _sz, C.c_int]),
    "kofft_hip_multi_rfft_f32_dev": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _sz, _sz]),
    "kofft_hip_multi_stft_f32_dev": (C.c_int, [_ctx, C.POINTER(C.c_void_p), _sz, C.POINTER(C.c_void_p), _sz, _sz, _sz, C.c_int,
                                               C.POINTER(C.c_void_p)]),
    "kofft_hip_multi_stft_slice": (C.c_int, [_ctx, _sz, _sz, _sz, _sz, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "kofft_hip_multi_context": (C.c_int, [_ctx, C.c_int, C.POINTER(_ctx), C.POINTER(C.c_void_p)]),
    "kofft_hip_multi_synchronize": (C.c_int, [_ctx]),
    "kofft_hip_multi_last_timing_ex": (C.c_int, [_ctx] + [C.POINTER(C.c_float)] * 5),
    "kofft_hip_multi_set_gather": (C.c_int, [_ctx, C.c_int]),
    "kofft_hip_multi_gather_mode": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def header_symbols() -> list[str]:
    """Function names declared in include/kofft_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kofft_hip_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load libkofft_hip.so (once) and attach the prototypes.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP runtime under the same soname (libamdhip64.so.7).  Whichever copy is
    # loaded first serves the whole process; torch cannot find the GPU through the system copy, while this library is
    # happy with either.  So when torch is installed it goes first.
    import sys
    if "torch" not in sys.modules and not os.environ.get("KOFFT_HIP_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = Path(os.environ.get("KOFFT_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise LibraryMissing(
            f"{path} not found: the HIP library is the only implementation of this path. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift; let it surface
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
