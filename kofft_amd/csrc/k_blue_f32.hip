// k_blue_f32.hip -- Complex<float> transforms of any other length: the Bluestein arm (fft.rs:411-433, 1088-1132) and the STFT of frames whose length is not a power of two.
#define KOFFT_BLUE_STFT_UNIT 1  // stft_bluestein_dev (float only) is defined in this translation unit
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
extern template int fft_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);  // k_complex_f32.hip
// factor kernels shared with the plain factor path (k_big_f32.hip holds the one copy)
extern template int launch_sub<float, BigColsIO<float, false, 0>>(kofft_hip_ctx *, const BigColsIO<float, false, 0> &, const cpx<float> *, int, size_t, bool);
extern template int launch_mid<float>(kofft_hip_ctx *, const BigMidIO<float> &, const cpx<float> *, int, size_t);
template int fft_bluestein_dev<float, false>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
template int fft_bluestein_dev<float, true>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
}  // namespace host
}  // namespace kofft
