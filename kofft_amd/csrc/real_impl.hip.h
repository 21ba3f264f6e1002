// real_impl.hip.h -- rfft_direct / irfft_direct (rfft.rs:425-508) on device pointers for one element type.
#pragma once

#include "host_common.hip.h"

namespace kofft {
namespace host {

template <typename T>
int rfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;   // rfft.rs:434
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;  // rfft.rs:437
    const size_t m = n / 2;
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    RfftIO<T> io{{}, d_in, d_window, reinterpret_cast<cpx<T> *>(d_out), rtab, (int)m};
    return dispatch<T, EPI_RFFT>(ctx, io, m, batch);
}

template <typename T>
int irfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;   // rfft.rs:477
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;  // rfft.rs:480
    const size_t m = n / 2;
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    // threads per transform of the persistent kernels (m/16; m/8 up to m = 512)
    const int tpt = (int)(m <= 64 ? m / 4 : m <= 512 ? m / 8 : m / 16);
    IrfftIO<T> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), rtab, (int)m,
                  (T)1 / (T)(float)m, tpt};
    return dispatch<T, EPI_STORE>(ctx, io, m, batch);
}


}  // namespace host
}  // namespace kofft
