// k_blue_f64.hip -- Complex<double> transforms of any other length: the Bluestein arm (fft.rs:411-433, 1088-1132).
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
extern template int fft_dev<double>(kofft_hip_ctx *, const double *, double *, size_t, size_t, int);  // k_complex_f64.hip
// factor kernels shared with the plain factor path (k_big_f64.hip holds the one copy)
extern template int launch_sub<double, BigColsIO<double, false, 0>>(kofft_hip_ctx *, const BigColsIO<double, false, 0> &, const cpx<double> *, int, size_t, bool);
extern template int launch_mid<double>(kofft_hip_ctx *, const BigMidIO<double> &, const cpx<double> *, int, size_t);
template int fft_bluestein_dev<double, false>(kofft_hip_ctx *, const double *, double *, size_t, size_t);
template int fft_bluestein_dev<double, true>(kofft_hip_ctx *, const double *, double *, size_t, size_t);
}  // namespace host
}  // namespace kofft
