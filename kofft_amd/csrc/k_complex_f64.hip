// k_complex_f64.hip -- Complex<double> transforms (fft.rs:1054-1174): every kernel instance of the family.
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
template int fft_dev<double>(kofft_hip_ctx *, const double *, double *, size_t, size_t, int);
template int fft_axis2_dev<double>(kofft_hip_ctx *, double *, int, int, size_t, int);
template int fft_radix4_dev<double>(kofft_hip_ctx *, const double *, double *, size_t, size_t, int);
template int fft_big_windowed_dev<double>(kofft_hip_ctx *, const double *, double *, const double *, size_t, size_t);
}  // namespace host
}  // namespace kofft

#if defined(KOFFT_RF_STAMPS)
extern "C" int kofft_hip_exp_rf_stamps_f64(void *out, size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kofft::g_rf_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
