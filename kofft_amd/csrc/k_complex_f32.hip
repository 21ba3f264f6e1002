// k_complex_f32.hip -- Complex<float> transforms (fft.rs:1054-1174): the single-pass kernels (one workgroup per transform, the CU's
// register file) and the Radix4 arm.  The factor path lives in k_big_f32.hip and the Bluestein arm in k_blue_f32.hip: three translation
// units per element type so that `make -j` builds them side by side (round 6: one unit took 70 s of a 90 s build).
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
extern template int fft_big_dev<float, false>(kofft_hip_ctx *, const float *, float *, size_t, size_t);       // k_big_f32.hip
extern template int fft_big_dev<float, true>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
extern template int fft_bluestein_dev<float, false>(kofft_hip_ctx *, const float *, float *, size_t, size_t);  // k_blue_f32.hip
extern template int fft_bluestein_dev<float, true>(kofft_hip_ctx *, const float *, float *, size_t, size_t);
template int fft_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);
template int fft_radix4_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);
}  // namespace host
}  // namespace kofft

#if defined(KOFFT_RF_STAMPS)
// diagnostic builds only: the s_memtime stamps of fft_regfile_persist_kernel<float, ...> (this translation unit's copy)
extern "C" int kofft_hip_exp_rf_stamps_f32(void *out, size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kofft::g_rf_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
