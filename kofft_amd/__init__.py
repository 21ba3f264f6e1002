"""kofft_amd -- MI355X (gfx950) implementation of kofft's FFT -> rFFT -> STFT hot path.

The product is ``kofft_amd/lib/libkofft_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/kofft_hip.h``); this package is the Python host mirror of the reference's operator
interface on top of it.  There is no CPU fallback: without the library every call raises.
"""
from .api import (DeviceError, fft2d_inplace, fft3d_inplace, flatten_2d, flatten_3d, FftError, FftPlan, FftPlanner, FftStrategy, HipFftImpl, HipMulti, IstftStream, RfftPlanner, StftStream, batch,
                  batch_inverse, frame, hann, inverse_frame, inverse_parallel, irfft_packed, istft, multi_channel, multi_channel_inverse, new_fft_impl, parallel, rfft_packed, stft, stft_magnitudes, stft_multi)
from ._lib import LibraryMissing, load as load_library

__all__ = ["DeviceError", "fft2d_inplace", "fft3d_inplace", "flatten_2d", "flatten_3d", "FftError", "FftPlan", "FftPlanner", "FftStrategy", "HipFftImpl", "HipMulti", "IstftStream", "RfftPlanner", "StftStream",
           "batch", "batch_inverse", "frame", "hann", "inverse_frame", "inverse_parallel", "irfft_packed", "istft", "multi_channel", "multi_channel_inverse", "new_fft_impl",
           "parallel", "rfft_packed", "stft", "stft_magnitudes", "stft_multi", "LibraryMissing", "load_library"]
__version__ = "0.1.0"
