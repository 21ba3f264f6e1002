// k_real_f64.hip -- real transforms of double rows (rfft.rs:425-508): every kernel instance of the family.
#include "real_impl.hip.h"

namespace kofft {
namespace host {
template int rfft_dev<double>(kofft_hip_ctx *, const double *, double *, const double *, size_t, size_t);
template int irfft_dev<double>(kofft_hip_ctx *, const double *, double *, size_t, size_t);
}  // namespace host
}  // namespace kofft
