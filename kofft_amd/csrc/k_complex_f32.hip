// k_complex_f32.hip -- Complex<float> transforms (fft.rs:1054-1174): every kernel instance of the family.
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
template int fft_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);
template int fft_radix4_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
template int fft_big_windowed_dev<float>(kofft_hip_ctx *, const float *, float *, const float *, size_t, size_t);
}  // namespace host
}  // namespace kofft
