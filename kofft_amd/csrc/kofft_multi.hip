// kofft_multi.hip -- the multi-GPU entry points of include/kofft_hip.h.
//
// stft::parallel (stft.rs:232-263) runs rayon over frames: frames are the parallel unit.  The device analogue is a
// single process that owns G devices: device r computes the contiguous frame range
//     [r * ceil(F / G), min((r + 1) * ceil(F / G), F))
// from its own slice of the signal (the slice plus the win_len - hop halo, cut on the host: no halo exchange), on its own
// context and stream (kofft_hip_create per device), all G devices running concurrently.  The only exchange is the
// OPTIONAL all-gather of the spectra (BASELINE config #4): one ncclAllGather per device inside ncclGroupStart / End on
// communicators from ncclCommInitAll, in place (every device writes its shard straight into its slot of the gathered
// buffer).  Batched complex / real transforms shard the same way with no exchange at all (kofft_hip_multi_fft_c32).
//
// This file only uses the public C ABI of the single-device library plus the HIP runtime; RCCL is bound at run time
// (dlopen) so that a process which never gathers has no RCCL dependency and one that already carries an RCCL (PyTorch
// bundles its own) keeps using that copy.
#include "../../include/kofft_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

// ---- RCCL, bound lazily ------------------------------------------------------------------------------------------
typedef void *nccl_comm_t;
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int kNcclFloat32 = 7;  // ncclFloat32 (rccl.h: ncclDataType_t)

void rccl_bind(Rccl &r);
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;  // the first gathers of two handles may come from two threads
    std::call_once(once, [] { rccl_bind(r); });
    return r;
}
void rccl_bind(Rccl &r)
{
    const char *env = getenv("KOFFT_HIP_RCCL_LIB");
    // a copy that is already mapped (PyTorch's, or the application's) wins; otherwise the ROCm one
    const char *noload[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : noload) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    }
    const char *load[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : load) {
        if (r.handle) break;
        if (n && *n) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) {
        const char *e = dlerror();
        r.error = std::string("RCCL not found (set KOFFT_HIP_RCCL_LIB): ") + (e ? e : "?");
        return;
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.handle, name);
        if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.error.empty()) r.handle = nullptr;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct kofft_hip_multi {
    int ngpu = 0;
    std::vector<int> devices;
    std::vector<kofft_hip_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<DevBuf> sig, win, out;       // per-device signal slice, window, spectra (shard or gathered)
    std::vector<hipEvent_t> ev0, ev1, ev2;   // start, compute done, gather done
    std::vector<nccl_comm_t> comms;          // created at the first gather
    std::string last_error;
    float compute_ms = 0.0f, gather_ms = 0.0f;
};

namespace {

#define KOFFT_MULTI_TRY(m, expr)                                                                  \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (m)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return KOFFT_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

int ensure(kofft_hip_multi *m, DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return KOFFT_OK;
    if (b.p) KOFFT_MULTI_TRY(m, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    KOFFT_MULTI_TRY(m, hipMalloc(&b.p, bytes ? bytes : 1));
    b.bytes = bytes ? bytes : 1;
    return KOFFT_OK;
}

// rank r owns [r * ceil(total / world), ...), the tail short or empty
void shard_range(size_t total, int r, int world, size_t *lo, size_t *hi)
{
    const size_t per = (total + (size_t)world - 1) / (size_t)world;
    size_t a = (size_t)r * per, b = a + per;
    if (a > total) a = total;
    if (b > total) b = total;
    *lo = a;
    *hi = b;
}

int ensure_comms(kofft_hip_multi *m)
{
    if (!m->comms.empty()) return KOFFT_OK;
    Rccl &r = rccl();
    if (!r.handle) {
        m->last_error = r.error;
        return KOFFT_ERR_RCCL;
    }
    std::vector<nccl_comm_t> comms(m->ngpu, nullptr);
    const int rc = r.CommInitAll(comms.data(), m->ngpu, m->devices.data());
    if (rc != 0) {
        m->last_error = std::string("ncclCommInitAll: ") + r.GetErrorString(rc);
        return KOFFT_ERR_RCCL;
    }
    m->comms = comms;
    return KOFFT_OK;
}

}  // namespace

extern "C" {

int kofft_hip_multi_create(int ngpu, const int *devices, kofft_hip_multi **out)
{
    if (!out) return KOFFT_ERR_NULL;
    *out = nullptr;
    if (ngpu <= 0) return KOFFT_ERR_INVALID_VALUE;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return KOFFT_ERR_HIP;
    kofft_hip_multi *m = new (std::nothrow) kofft_hip_multi();
    if (!m) return KOFFT_ERR_ALLOC;
    m->ngpu = ngpu;
    for (int r = 0; r < ngpu; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= have) {
            kofft_hip_multi_destroy(m);
            return KOFFT_ERR_INVALID_VALUE;
        }
        m->devices.push_back(d);
    }
    m->ctx.assign(ngpu, nullptr);
    m->stream.assign(ngpu, nullptr);
    m->sig.resize(ngpu);
    m->win.resize(ngpu);
    m->out.resize(ngpu);
    m->ev0.assign(ngpu, nullptr);
    m->ev1.assign(ngpu, nullptr);
    m->ev2.assign(ngpu, nullptr);
    for (int r = 0; r < ngpu; ++r) {
        int rc = kofft_hip_create(m->devices[r], &m->ctx[r]);
        if (rc == KOFFT_OK && hipSetDevice(m->devices[r]) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc == KOFFT_OK && hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc == KOFFT_OK) rc = kofft_hip_set_stream(m->ctx[r], m->stream[r]);
        if (rc == KOFFT_OK && (hipEventCreate(&m->ev0[r]) != hipSuccess || hipEventCreate(&m->ev1[r]) != hipSuccess ||
                               hipEventCreate(&m->ev2[r]) != hipSuccess))
            rc = KOFFT_ERR_HIP;
        if (rc != KOFFT_OK) {
            kofft_hip_multi_destroy(m);
            return rc;
        }
    }
    *out = m;
    return KOFFT_OK;
}

int kofft_hip_multi_destroy(kofft_hip_multi *m)
{
    if (!m) return KOFFT_ERR_NULL;
    for (size_t r = 0; r < m->ctx.size(); ++r) {  // nothing may still be running when the communicators go
        (void)hipSetDevice(m->devices[r]);
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
    }
    if (!m->comms.empty()) {
        Rccl &r = rccl();
        for (nccl_comm_t c : m->comms)
            if (c && r.CommDestroy) (void)r.CommDestroy(c);
    }
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        for (DevBuf *b : {&m->sig[r], &m->win[r], &m->out[r]})
            if (b->p) (void)hipFree(b->p);
        for (hipEvent_t e : {m->ev0[r], m->ev1[r], m->ev2[r]})
            if (e) (void)hipEventDestroy(e);
        if (m->ctx[r]) {
            (void)kofft_hip_set_stream(m->ctx[r], nullptr);
            (void)kofft_hip_destroy(m->ctx[r]);
        }
        if (m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    delete m;
    return KOFFT_OK;
}

const char *kofft_hip_multi_last_error(const kofft_hip_multi *m) { return m ? m->last_error.c_str() : ""; }

int kofft_hip_multi_ngpu(const kofft_hip_multi *m) { return m ? m->ngpu : 0; }

int kofft_hip_multi_shard(const kofft_hip_multi *m, size_t total, int rank, size_t *first, size_t *count)
{
    if (!m || !first || !count) return KOFFT_ERR_NULL;
    if (rank < 0 || rank >= m->ngpu) return KOFFT_ERR_INVALID_VALUE;
    size_t lo, hi;
    shard_range(total, rank, m->ngpu, &lo, &hi);
    *first = lo;
    *count = hi - lo;
    return KOFFT_OK;
}

int kofft_hip_multi_last_timing(const kofft_hip_multi *m, float *compute_ms, float *gather_ms)
{
    if (!m) return KOFFT_ERR_NULL;
    if (compute_ms) *compute_ms = m->compute_ms;
    if (gather_ms) *gather_ms = m->gather_ms;
    return KOFFT_OK;
}

namespace {
// A failure on one device must not return while the others still have copies in flight that read or write the CALLER's
// host buffers: drain every stream first (errors of the drain itself are ignored, the first error is the one reported).
void drain_all(kofft_hip_multi *m)
{
    for (int r = 0; r < m->ngpu; ++r) {
        if (hipSetDevice(m->devices[r]) == hipSuccess && m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
    }
    (void)hipGetLastError();
}
int multi_stft_impl(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len, size_t hop, float *out,
                    size_t frames, int allgather, float **d_out_per_gpu);
int multi_fft_impl(kofft_hip_multi *m, float *data, size_t n, size_t batch, int inverse);
}  // namespace

int kofft_hip_multi_stft_f32(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len,
                             size_t hop, float *out, size_t frames, int allgather, float **d_out_per_gpu)
{
    const int rc = multi_stft_impl(m, signal, len, window, win_len, hop, out, frames, allgather, d_out_per_gpu);
    if (rc != KOFFT_OK && m) drain_all(m);  // possibly with work of other devices already enqueued
    return rc;
}

namespace {
int multi_stft_impl(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len, size_t hop, float *out,
                    size_t frames, int allgather, float **d_out_per_gpu)
{
    // stft::stft's checks in its order (stft.rs:83-87), then the transform's (fft.rs:1056)
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    if (frames < (len + hop - 1) / hop) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || (!signal && len) || !window) return KOFFT_ERR_NULL;
    if (!out && !d_out_per_gpu) return KOFFT_ERR_NULL;  // nowhere to leave the result
    const int G = m->ngpu;
    const size_t per = (frames + (size_t)G - 1) / (size_t)G;  // slot size of the gather
    const size_t frame_bytes = win_len * 2 * sizeof(float);
    if (allgather) {
        const int rc = ensure_comms(m);
        if (rc) return rc;
    }
    // 1) every device: upload its slice + halo and the window, launch its frames.  All asynchronous per device, so the
    //    devices run concurrently; a slot of `per` frames in the gathered layout, or just the shard.
    for (int r = 0; r < G; ++r) {
        size_t f0, f1;
        shard_range(frames, r, G, &f0, &f1);
        const size_t count = f1 - f0;
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        int rc = ensure(m, m->out[r], allgather ? (size_t)G * per * frame_bytes : count * frame_bytes);
        if (rc) return rc;
        rc = ensure(m, m->win[r], win_len * sizeof(float));
        if (rc) return rc;
        size_t lo = f0 * hop < len ? f0 * hop : len;
        size_t hi = count ? (f1 - 1) * hop + win_len : lo;
        if (hi > len) hi = len;
        if (hi < lo) hi = lo;
        rc = ensure(m, m->sig[r], (hi - lo) * sizeof(float));
        if (rc) return rc;
        hipStream_t s = m->stream[r];
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev0[r], s));
        KOFFT_MULTI_TRY(m, hipMemcpyAsync(m->win[r].p, window, win_len * sizeof(float), hipMemcpyHostToDevice, s));
        if (hi > lo) KOFFT_MULTI_TRY(m, hipMemcpyAsync(m->sig[r].p, signal + lo, (hi - lo) * sizeof(float), hipMemcpyHostToDevice, s));
        float *dst = static_cast<float *>(m->out[r].p) + (allgather ? (size_t)r * per * win_len * 2 : 0);
        if (allgather && count < per)  // zero the padding of a short last slot: the gathered buffer is fully defined
            KOFFT_MULTI_TRY(m, hipMemsetAsync(dst + count * win_len * 2, 0, (per - count) * frame_bytes, s));
        if (count) {
            // frames f0 .. f1-1 of the whole STFT = frames 0 .. count-1 of the slice that starts at sample f0*hop
            rc = kofft_hip_stft_f32_dev(m->ctx[r], static_cast<const float *>(m->sig[r].p), hi - lo,
                                        static_cast<const float *>(m->win[r].p), win_len, hop, dst, 0, count);
            if (rc) {
                m->last_error = std::string("device ") + std::to_string(m->devices[r]) + ": " + kofft_hip_last_error(m->ctx[r]);
                return rc;
            }
        }
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev1[r], s));
    }
    // 2) the exchange: every device contributes its slot, in place
    if (allgather) {
        Rccl &rc = rccl();
        int st = rc.GroupStart();
        for (int r = 0; r < G && st == 0; ++r) {
            float *base = static_cast<float *>(m->out[r].p);
            st = rc.AllGather(base + (size_t)r * per * win_len * 2, base, per * win_len * 2, kNcclFloat32, m->comms[r], m->stream[r]);
        }
        const int st_end = rc.GroupEnd();
        if (st == 0) st = st_end;
        if (st != 0) {
            m->last_error = std::string("ncclAllGather: ") + rc.GetErrorString(st);
            return KOFFT_ERR_RCCL;
        }
    }
    for (int r = 0; r < G; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev2[r], m->stream[r]));
    }
    // 3) results: host copy of every shard from the device that computed it (G PCIe links in parallel)
    if (out) {
        for (int r = 0; r < G; ++r) {
            size_t f0, f1;
            shard_range(frames, r, G, &f0, &f1);
            if (f1 == f0) continue;
            KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
            const float *src = static_cast<const float *>(m->out[r].p) + (allgather ? (size_t)r * per * win_len * 2 : 0);
            KOFFT_MULTI_TRY(m, hipMemcpyAsync(out + f0 * win_len * 2, src, (f1 - f0) * frame_bytes, hipMemcpyDeviceToHost, m->stream[r]));
        }
    }
    float cmax = 0.0f, gmax = 0.0f;
    for (int r = 0; r < G; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        KOFFT_MULTI_TRY(m, hipStreamSynchronize(m->stream[r]));
        float c = 0.0f, g = 0.0f;
        if (hipEventElapsedTime(&c, m->ev0[r], m->ev1[r]) == hipSuccess && c > cmax) cmax = c;
        if (hipEventElapsedTime(&g, m->ev1[r], m->ev2[r]) == hipSuccess && g > gmax) gmax = g;
        if (d_out_per_gpu) d_out_per_gpu[r] = static_cast<float *>(m->out[r].p);
    }
    m->compute_ms = cmax;  // upload + kernel of the slowest device
    m->gather_ms = allgather ? gmax : 0.0f;
    return KOFFT_OK;
}
}  // namespace

int kofft_hip_stft_f32_multi(int ngpu, const float *signal, size_t len, const float *window, size_t win_len, size_t hop,
                             float *out, size_t frames, int allgather)
{
    // argument checks first: they need no device (and keep the reference's order, stft.rs:83-87)
    if (ngpu <= 0) return KOFFT_ERR_INVALID_VALUE;
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    if (frames < (len + hop - 1) / hop) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if ((!signal && len) || !window || !out) return KOFFT_ERR_NULL;
    kofft_hip_multi *m = nullptr;
    int rc = kofft_hip_multi_create(ngpu, nullptr, &m);
    if (rc) return rc;
    rc = kofft_hip_multi_stft_f32(m, signal, len, window, win_len, hop, out, frames, allgather, nullptr);
    (void)kofft_hip_multi_destroy(m);
    return rc;
}

int kofft_hip_multi_fft_c32(kofft_hip_multi *m, float *data, size_t n, size_t batch, int inverse)
{
    const int rc = multi_fft_impl(m, data, n, batch, inverse);
    if (rc != KOFFT_OK && m) drain_all(m);
    return rc;
}

namespace {
int multi_fft_impl(kofft_hip_multi *m, float *data, size_t n, size_t batch, int inverse)
{
    // fft::batch (fft.rs:2156-2175) with the batch split into G contiguous blocks: no exchange of any kind
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || !data) return KOFFT_ERR_NULL;
    const int G = m->ngpu;
    const size_t row = n * 2 * sizeof(float);
    for (int r = 0; r < G; ++r) {
        size_t b0, b1;
        shard_range(batch, r, G, &b0, &b1);
        if (b1 == b0) continue;
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        int rc = ensure(m, m->out[r], (b1 - b0) * row);
        if (rc) return rc;
        float *d = static_cast<float *>(m->out[r].p);
        KOFFT_MULTI_TRY(m, hipMemcpyAsync(d, data + b0 * n * 2, (b1 - b0) * row, hipMemcpyHostToDevice, m->stream[r]));
        rc = kofft_hip_fft_c32_dev(m->ctx[r], d, n, b1 - b0, inverse);
        if (rc) {
            m->last_error = std::string("device ") + std::to_string(m->devices[r]) + ": " + kofft_hip_last_error(m->ctx[r]);
            return rc;
        }
        KOFFT_MULTI_TRY(m, hipMemcpyAsync(data + b0 * n * 2, d, (b1 - b0) * row, hipMemcpyDeviceToHost, m->stream[r]));
    }
    for (int r = 0; r < G; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        KOFFT_MULTI_TRY(m, hipStreamSynchronize(m->stream[r]));
    }
    return KOFFT_OK;
}
}  // namespace

}  // extern "C"
