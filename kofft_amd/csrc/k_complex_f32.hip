// k_complex_f32.hip -- Complex<float> transforms (fft.rs:1054-1174): every kernel instance of the family.
#define KOFFT_BLUE_STFT_UNIT 1  // stft_bluestein_dev (float only) is defined in this translation unit
#include "complex_impl.hip.h"

namespace kofft {
namespace host {
template int fft_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);
template int fft_axis2_dev<float>(kofft_hip_ctx *, float *, int, int, size_t, int);
template int fft_radix4_dev<float>(kofft_hip_ctx *, const float *, float *, size_t, size_t, int);
template int fft_big_windowed_dev<float>(kofft_hip_ctx *, const float *, float *, const float *, size_t, size_t);
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

#if defined(KOFFT_RF_STAMPS)
// diagnostic builds only: the s_memtime stamps of fft_regfile_persist_kernel<float, ...> (this translation unit's copy)
extern "C" int kofft_hip_exp_rf_stamps_f32(void *out, size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kofft::g_rf_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
