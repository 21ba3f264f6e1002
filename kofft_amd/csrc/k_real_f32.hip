// k_real_f32.hip -- real transforms of float rows (rfft.rs:425-508): every kernel instance of the family.
#include "real_impl.hip.h"

namespace kofft {
namespace host {
template int rfft_dev<float>(kofft_hip_ctx *, const float *, float *, const float *, size_t, size_t);
template int irfft_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
}  // namespace host
}  // namespace kofft
