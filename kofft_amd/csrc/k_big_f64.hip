// k_big_f64.hip -- Complex<double> transforms beyond one workgroup (fft.rs:961-1037 at n >= 2^15 / 2^14): the two- / three-factor path,
// its windowed form (rfft_direct's packed rows) and ndfft's long strided axes in two column-tile passes.
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
template int fft_big_dev<double, false>(kofft_hip_ctx *, const double *, double *, size_t, size_t);
template int fft_big_dev<double, true>(kofft_hip_ctx *, const double *, double *, size_t, size_t);
template int fft_axis2_dev<double>(kofft_hip_ctx *, double *, int, int, size_t, int);
template int fft_big_windowed_dev<double>(kofft_hip_ctx *, const double *, double *, const double *, size_t, size_t);
// factor kernels the Bluestein arm shares with this unit (k_blue_f64.hip declares them extern: one copy in the library)
template int launch_sub<double, BigColsIO<double, false, 0>>(kofft_hip_ctx *, const BigColsIO<double, false, 0> &, const cpx<double> *, int, size_t, bool);
template int launch_mid<double>(kofft_hip_ctx *, const BigMidIO<double> &, const cpx<double> *, int, size_t);
}  // namespace host
}  // namespace kofft
