// k_big_f32.hip -- Complex<float> transforms beyond one workgroup (fft.rs:961-1037 at n >= 2^15 / 2^14): the two- / three-factor path,
// its windowed form (rfft_direct's packed rows) and ndfft's long strided axes in two column-tile passes.
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
template int fft_big_dev<float, false>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
template int fft_big_dev<float, true>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
template int fft_axis2_dev<float>(kofft_hip_ctx *, float *, int, int, size_t, int);
template int fft_big_windowed_dev<float>(kofft_hip_ctx *, const float *, float *, const float *, size_t, size_t);
// factor kernels the Bluestein arm shares with this unit (k_blue_f32.hip declares them extern: one copy in the library)
template int launch_sub<float, BigColsIO<float, false, 0>>(kofft_hip_ctx *, const BigColsIO<float, false, 0> &, const cpx<float> *, int, size_t, bool);
template int launch_mid<float>(kofft_hip_ctx *, const BigMidIO<float> &, const cpx<float> *, int, size_t);
// the column-tile pass of the fused 2-D route (k_nd_fused.hip declares these extern: one copy of the kernels in the library)
#define KOFFT_CASE(LL)                                                                                                       \
    template int launch_tile_persist<float, LL, AxisLastIO<float, false>>(kofft_hip_ctx *, const AxisLastIO<float, false> &, \
                                                                          const cpx<float> *, size_t);                      \
    template int launch_tile_persist<float, LL, AxisLastIO<float, true>>(kofft_hip_ctx *, const AxisLastIO<float, true> &,   \
                                                                         const cpx<float> *, size_t);
KOFFT_CASE(8)
KOFFT_CASE(9)
KOFFT_CASE(10)
KOFFT_CASE(11)
#undef KOFFT_CASE
}  // namespace host
}  // namespace kofft
