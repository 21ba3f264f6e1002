// kofft_multi.hip -- the multi-GPU entry points of include/kofft_hip.h.
//
// stft::parallel (stft.rs:232-263) runs rayon over frames: frames are the parallel unit; fft::batch (fft.rs:2156-2175)
// and RfftPlanner::rfft_with_scratch over rows (rfft.rs:264-282) are loops over independent transforms.  The device
// analogue is a single process that owns G devices: device r works on the contiguous block
//     [r * ceil(U / G), min((r + 1) * ceil(U / G), U))
// of the U units (frames / transforms / rows) on its own context and stream (kofft_hip_create per device).
//
// Who issues what (round 3):
//   * HOST-pointer entries: every device has its own parked WORKER THREAD.  A copy from or to pageable caller memory
//     blocks the thread that issues it, so one thread feeding G devices would run the G uploads -- and the G downloads,
//     922 MB for BASELINE config #4 -- one after another.  Worker r does upload -> launch -> download for device r and
//     nothing else; the G PCIe links are busy together.  The calling thread only runs the one step that spans devices,
//     the grouped RCCL all-gather, between the workers' launch and download phases.
//   * DEVICE-pointer entries (*_dev): nothing blocks -- kernel launches and the gather are enqueued by the calling thread,
//     device after device, and the call returns without synchronising (the *_dev convention of the single-device ABI).
// The only exchange is the OPTIONAL all-gather of the STFT spectra (BASELINE config #4): one ncclAllGather per device
// inside ncclGroupStart / End on communicators from ncclCommInitAll, in place (every device writes its shard straight
// into its slot of the gathered buffer).  Batched complex / real transforms have no exchange at all.
//
// Timing: four HIP events per device and call -- t0 | uploads | t1 | kernels | t2 | gather | t3 | download | t4 -- and the
// slowest device's span of each phase is reported, separately (kofft_hip_multi_last_timing_ex).  kernel_ms is kernels
// only: no copy is inside it.
//
// This file only uses the public C ABI of the single-device library plus the HIP runtime; RCCL is bound at run time
// (dlopen) so that a process which never gathers has no RCCL dependency and one that already carries an RCCL (PyTorch
// bundles its own) keeps using that copy.
#include "../../include/kofft_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// (see kNcclFloat32 below)
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define KOFFT_HAVE_RCCL_HEADER 1
#endif
#endif

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
// ncclFloat32 of ncclDataType_t: taken from RCCL's own header when the build machine has it (only the enumerator -- the
// functions stay dlsym-bound so that the library loads, and everything but the exchange works, without RCCL installed).
#ifdef KOFFT_HAVE_RCCL_HEADER
constexpr int kNcclFloat32 = (int)ncclFloat32;
static_assert((int)ncclFloat32 == 7, "ncclFloat32 moved: the header-less build's literal below must follow");
#else
constexpr int kNcclFloat32 = 7;  // ncclFloat32 (rccl.h: ncclDataType_t) -- header absent at build time
#endif

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

// The calling thread's current device is part of ITS state (PyTorch allocates on it): every entry that walks the devices
// puts it back on every exit path.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard()
    {
        if (hipGetDevice(&saved) != hipSuccess) {
            saved = -1;
            (void)hipGetLastError();
        }
    }
    ~DeviceGuard()
    {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

// One parked thread per device.  run(fn) executes fn(r) on worker r for every r and returns when all are done.
class Workers {
public:
    bool start(const std::vector<int> &devices)
    {
        {   // a failed start() ran stop(): begin from a clean state, or the new threads would see quit_ and leave at once
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = false;
            generation_ = 0;
            pending_ = 0;
            job_ = nullptr;
        }
        n_ = (int)devices.size();
        rc_.assign(n_, 0);
        try {
            for (int r = 0; r < n_; ++r) threads_.emplace_back([this, r, d = devices[r]] { loop(r, d); });
        } catch (...) {
            stop();
            return false;
        }
        return true;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }
    // first non-zero status in rank order (the others have finished too)
    int run(const std::function<int(int)> &fn)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &fn;
            pending_ = n_;
            ++generation_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        for (int r = 0; r < n_; ++r)
            if (rc_[r]) return rc_[r];
        return 0;
    }
    ~Workers() { stop(); }

private:
    void loop(int r, int device)
    {
        // thread-local: the caller's current device is never touched by a worker.  If it fails, every job of this worker fails
        // (its HIP calls would land on another device).
        const bool device_ok = hipSetDevice(device) == hipSuccess;
        if (!device_ok) (void)hipGetLastError();
        unsigned long long seen = 0;
        for (;;) {
            const std::function<int(int)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                job = job_;
            }
            int rc;
            try {
                rc = device_ok ? (*job)(r) : KOFFT_ERR_HIP;
            } catch (...) {
                rc = KOFFT_ERR_ALLOC;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_[r] = rc;
                --pending_;
            }
            done_cv_.notify_one();
        }
    }
    int n_ = 0;
    std::vector<std::thread> threads_;
    std::vector<int> rc_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<int(int)> *job_ = nullptr;
    unsigned long long generation_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

}  // namespace

struct kofft_hip_multi {
    int ngpu = 0;
    std::vector<int> devices;
    std::vector<kofft_hip_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<DevBuf> sig, win, out;       // per-device input (slice / rows), window, output (shard or gathered)
    std::vector<hipEvent_t> ev[5];           // t0 .. t4 of the header comment
    std::vector<nccl_comm_t> comms;          // created at the first RCCL gather
    // the exchange of the STFT spectra: 1 = one grouped in-place ncclAllGather per device, 2 = direct: every device pushes its slot
    // to every peer with hipMemcpyPeerAsync on its own per-peer stream (KOFFT_HIP_MULTI_GATHER=direct / kofft_hip_multi_set_gather)
    int gather_mode = 1, last_gather = 0;
    std::vector<hipStream_t> peer_stream;    // [r * ngpu + p], on device r; created at the first direct gather
    std::vector<hipEvent_t> peer_done;       // [r * ngpu + p]: the copy r -> p has finished
    bool peers_up = false;
    std::vector<std::string> dev_error;      // written by worker r only
    std::string last_error;
    Workers workers;
    bool workers_up = false;
    // the last call: which phases it had, and (host forms) the wall time between entry and return
    bool timed = false, had_upload = false, had_gather = false, had_download = false;
    float wall_ms = 0.0f;
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
// inside worker r: errors go to the worker's own slot
#define KOFFT_WORKER_TRY(m, r, expr)                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (m)->dev_error[r] = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return KOFFT_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

// (current device = the buffer's device: the workers' own, or set by the caller)
int ensure(std::string &err, DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return KOFFT_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    const hipError_t e = hipMalloc(&b.p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        err = std::string("hipMalloc: ") + hipGetErrorString(e);
        b.p = nullptr;
        return KOFFT_ERR_HIP;
    }
    b.bytes = bytes ? bytes : 1;
    return KOFFT_OK;
}

// rank r owns [r * ceil(total / world), ...), the tail short or empty
void shard_range(size_t total, int r, int world, size_t *lo, size_t *hi)
{
    const size_t per = total / (size_t)world + (total % (size_t)world != 0);
    size_t a = (size_t)r * per, b = a + per;
    if (a > total) a = total;
    if (b > total) b = total;
    *lo = a;
    *hi = b;
}

// samples the frames [f0, f1) of an STFT can see: [lo, hi) of the signal
void slice_range(size_t len, size_t win_len, size_t hop, size_t f0, size_t f1, size_t *lo, size_t *hi)
{
    size_t a = f0 * hop < len ? f0 * hop : len;
    size_t b = f1 > f0 ? (f1 - 1) * hop + win_len : a;
    if (b > len) b = len;
    if (b < a) b = a;
    *lo = a;
    *hi = b;
}

inline size_t ceil_div(size_t a, size_t b) { return a / b + (a % b != 0); }  // no wrap for a + b > SIZE_MAX

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

int ensure_workers(kofft_hip_multi *m)
{
    if (m->workers_up) return KOFFT_OK;
    if (!m->workers.start(m->devices)) {
        m->last_error = "could not start the per-device worker threads";
        return KOFFT_ERR_ALLOC;
    }
    m->workers_up = true;
    return KOFFT_OK;
}

// A failure on one device must not return while the others still have copies in flight that read or write the CALLER's
// host buffers: drain every stream first (errors of the drain itself are ignored, the first error is the one reported).
void drain_all(kofft_hip_multi *m)
{
    for (hipStream_t ps : m->peer_stream)
        if (ps) (void)hipStreamSynchronize(ps);
    for (int r = 0; r < m->ngpu; ++r)
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
    (void)hipGetLastError();
}

// first worker error text, prefixed with the device
void collect_error(kofft_hip_multi *m)
{
    for (int r = 0; r < m->ngpu; ++r)
        if (!m->dev_error[r].empty()) {
            m->last_error = std::string("device ") + std::to_string(m->devices[r]) + ": " + m->dev_error[r];
            return;
        }
}

void begin_call(kofft_hip_multi *m, bool upload, bool gather, bool download)
{
    m->timed = false;
    m->had_upload = upload;
    m->had_gather = gather;
    m->had_download = download;
    m->wall_ms = 0.0f;
    m->last_gather = 0;
    m->last_error.clear();
    for (std::string &s : m->dev_error) s.clear();
}

// the exchange: every device contributes its slot of `per_floats` floats, in place in `base[r]`; enqueued by the caller
int gather_all(kofft_hip_multi *m, float *const *base, size_t per_floats)
{
    Rccl &rc = rccl();
    int st = rc.GroupStart();
    for (int r = 0; r < m->ngpu && st == 0; ++r)
        st = rc.AllGather(base[r] + (size_t)r * per_floats, base[r], per_floats, kNcclFloat32, m->comms[r], m->stream[r]);
    const int st_end = rc.GroupEnd();
    if (st == 0) st = st_end;
    if (st != 0) {
        m->last_error = std::string("ncclAllGather: ") + rc.GetErrorString(st);
        return KOFFT_ERR_RCCL;
    }
    return KOFFT_OK;
}

// The same exchange without RCCL: device r PUSHES its slot into every peer's buffer, the G - 1 copies of a device concurrent on
// G - 1 streams of their own (xGMI is point to point: seven links per device, one per peer; SURVEY 8e: ~0.75 ms for config #4's
// 115 MB slots against ~5 ms for a ring).  Ordering is by events only: a copy starts behind its source's kernels (ev[2][r]), and
// device p's stream goes on (ev[3][p], the download) behind every copy into AND out of p's buffer.
int ensure_peers(kofft_hip_multi *m)
{
    if (m->peers_up) return KOFFT_OK;
    const int G = m->ngpu;
    // restartable: a call that failed part-way left its streams / events in the vectors; only the missing ones are created
    if (m->peer_stream.size() != (size_t)G * G) m->peer_stream.assign((size_t)G * G, nullptr);
    if (m->peer_done.size() != (size_t)G * G) m->peer_done.assign((size_t)G * G, nullptr);
    for (int r = 0; r < G; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        for (int p = 0; p < G; ++p) {
            if (p == r) continue;
            if (m->devices[p] != m->devices[r]) {  // (logical devices on one card: nothing to enable)
                const hipError_t e = hipDeviceEnablePeerAccess(m->devices[p], 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled, or no direct path: the copy is then staged by the runtime
            }
            if (!m->peer_stream[(size_t)r * G + p])
                KOFFT_MULTI_TRY(m, hipStreamCreateWithFlags(&m->peer_stream[(size_t)r * G + p], hipStreamNonBlocking));
            if (!m->peer_done[(size_t)r * G + p])
                KOFFT_MULTI_TRY(m, hipEventCreateWithFlags(&m->peer_done[(size_t)r * G + p], hipEventDisableTiming));
        }
    }
    m->peers_up = true;
    return KOFFT_OK;
}

int gather_direct(kofft_hip_multi *m, float *const *base, size_t per_floats)
{
    const int G = m->ngpu;
    const size_t bytes = per_floats * sizeof(float);
    int rc = ensure_peers(m);
    if (rc) return rc;
    for (int r = 0; r < G; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        for (int p = 0; p < G; ++p) {
            if (p == r) continue;
            hipStream_t ps = m->peer_stream[(size_t)r * G + p];
            KOFFT_MULTI_TRY(m, hipStreamWaitEvent(ps, m->ev[2][r], 0));  // behind r's kernels (and the zero fill of a short slot)
            // ... and behind whatever the caller had queued on the DESTINATION's stream when this call began (ev[1][p] is recorded on
            // stream[p] before p's kernels): a consumer of the previous call's gathered buffer ordered on stream[p] -- the stream
            // kofft_hip_multi_context hands out for exactly that -- must have finished reading slot r before it is overwritten.
            // The RCCL form runs the all-gather ON stream[p] and has this ordering by construction.
            KOFFT_MULTI_TRY(m, hipStreamWaitEvent(ps, m->ev[1][p], 0));
            KOFFT_MULTI_TRY(m, hipMemcpyPeerAsync(base[p] + (size_t)r * per_floats, m->devices[p], base[r] + (size_t)r * per_floats,
                                                  m->devices[r], bytes, ps));
            KOFFT_MULTI_TRY(m, hipEventRecord(m->peer_done[(size_t)r * G + p], ps));
        }
    }
    for (int p = 0; p < G; ++p) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[p]));
        for (int r = 0; r < G; ++r) {
            if (r == p) continue;
            KOFFT_MULTI_TRY(m, hipStreamWaitEvent(m->stream[p], m->peer_done[(size_t)r * G + p], 0));  // into p's buffer
            KOFFT_MULTI_TRY(m, hipStreamWaitEvent(m->stream[p], m->peer_done[(size_t)p * G + r], 0));  // out of p's buffer
        }
    }
    return KOFFT_OK;
}

// the exchange in the handle's mode; records which form ran
int gather_any(kofft_hip_multi *m, float *const *base, size_t per_floats)
{
    m->last_gather = m->gather_mode;
    return m->gather_mode == 2 ? gather_direct(m, base, per_floats) : gather_all(m, base, per_floats);
}

int record_all(kofft_hip_multi *m, int which)
{
    for (int r = 0; r < m->ngpu; ++r) {
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev[which][r], m->stream[r]));
    }
    return KOFFT_OK;
}

struct WallClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    float ms() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

int stft_checks(size_t len, size_t win_len, size_t hop, size_t frames)
{
    // stft::stft's checks in its order (stft.rs:83-87), then the transform's (fft.rs:1056)
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    if (frames < ceil_div(len, hop)) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    return -100;  // go on
}

int multi_stft_host(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len, size_t hop, float *out,
                    size_t frames, int allgather, float **d_out_per_gpu)
{
    const int chk = stft_checks(len, win_len, hop, frames);
    if (chk != -100) return chk;
    if (!m || (!signal && len) || !window) return KOFFT_ERR_NULL;
    if (!out && !d_out_per_gpu) return KOFFT_ERR_NULL;  // nowhere to leave the result
    WallClock wall;
    begin_call(m, true, allgather != 0, out != nullptr);
    const int G = m->ngpu;
    const size_t per = ceil_div(frames, (size_t)G);  // slot size of the gather
    const size_t frame_bytes = win_len * 2 * sizeof(float);
    int rc = ensure_workers(m);
    if (rc) return rc;
    if (allgather && m->gather_mode == 1) {
        DeviceGuard guard;  // ncclCommInitAll walks the devices
        rc = ensure_comms(m);
        if (rc) return rc;
    }
    // phase A on worker r: upload slice + halo and window, launch the frames.  A slot of `per` frames in the gathered
    // layout, or just the shard.
    auto launch = [&](int r) -> int {
        size_t f0, f1, lo, hi;
        shard_range(frames, r, G, &f0, &f1);
        slice_range(len, win_len, hop, f0, f1, &lo, &hi);
        const size_t count = f1 - f0;
        std::string &err = m->dev_error[r];
        int st = ensure(err, m->out[r], allgather ? (size_t)G * per * frame_bytes : count * frame_bytes);
        if (st == KOFFT_OK) st = ensure(err, m->win[r], win_len * sizeof(float));
        if (st == KOFFT_OK) st = ensure(err, m->sig[r], (hi - lo) * sizeof(float));
        if (st) return st;
        hipStream_t s = m->stream[r];
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[0][r], s));
        KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(m->win[r].p, window, win_len * sizeof(float), hipMemcpyHostToDevice, s));
        if (hi > lo) KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(m->sig[r].p, signal + lo, (hi - lo) * sizeof(float), hipMemcpyHostToDevice, s));
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[1][r], s));
        float *dst = static_cast<float *>(m->out[r].p) + (allgather ? (size_t)r * per * win_len * 2 : 0);
        if (allgather && count < per)  // zero the padding of a short last slot: the gathered buffer is fully defined
            KOFFT_WORKER_TRY(m, r, hipMemsetAsync(dst + count * win_len * 2, 0, (per - count) * frame_bytes, s));
        if (count) {
            // frames f0 .. f1-1 of the whole STFT = frames 0 .. count-1 of the slice that starts at sample f0*hop
            st = kofft_hip_stft_f32_dev(m->ctx[r], static_cast<const float *>(m->sig[r].p), hi - lo,
                                        static_cast<const float *>(m->win[r].p), win_len, hop, dst, 0, count);
            if (st) {
                err = kofft_hip_last_error(m->ctx[r]);
                return st;
            }
        }
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[2][r], s));
        return KOFFT_OK;
    };
    // phase B on worker r: its shard back to the caller's memory, over its own PCIe link
    auto finish = [&](int r) -> int {
        size_t f0, f1;
        shard_range(frames, r, G, &f0, &f1);
        hipStream_t s = m->stream[r];
        if (!allgather) KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[3][r], s));  // (with a gather the caller recorded it)
        if (out && f1 > f0) {
            const float *src = static_cast<const float *>(m->out[r].p) + (allgather ? (size_t)r * per * win_len * 2 : 0);
            KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(out + f0 * win_len * 2, src, (f1 - f0) * frame_bytes, hipMemcpyDeviceToHost, s));
        }
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[4][r], s));
        KOFFT_WORKER_TRY(m, r, hipStreamSynchronize(s));
        return KOFFT_OK;
    };
    if (!allgather) {
        rc = m->workers.run([&](int r) {
            const int st = launch(r);
            return st ? st : finish(r);
        });
    } else {
        rc = m->workers.run(launch);
        if (rc == KOFFT_OK) {
            DeviceGuard guard;
            std::vector<float *> base(G);
            for (int r = 0; r < G; ++r) base[r] = static_cast<float *>(m->out[r].p);
            rc = gather_any(m, base.data(), per * win_len * 2);
            if (rc == KOFFT_OK) rc = record_all(m, 3);
        }
        if (rc == KOFFT_OK) rc = m->workers.run(finish);
    }
    if (rc != KOFFT_OK) {
        if (m->last_error.empty()) collect_error(m);
        return rc;
    }
    if (d_out_per_gpu)
        for (int r = 0; r < G; ++r) d_out_per_gpu[r] = static_cast<float *>(m->out[r].p);
    m->timed = true;
    m->wall_ms = wall.ms();
    return KOFFT_OK;
}

// fft::batch (fft.rs:2156-2175) / rfft rows (rfft.rs:264-282) with the batch split into G contiguous blocks: no exchange
// of any kind.  Worker r: upload its block, transform, download.
template <typename T, class Kernel>
int multi_rows_host(kofft_hip_multi *m, const T *in, T *out, size_t in_row, size_t out_row, size_t batch, bool in_place,
                    const T *window, size_t win_len, Kernel kernel)
{
    WallClock wall;
    begin_call(m, true, false, true);
    const int G = m->ngpu;
    int rc = ensure_workers(m);
    if (rc) return rc;
    rc = m->workers.run([&](int r) -> int {
        size_t b0, b1;
        shard_range(batch, r, G, &b0, &b1);
        const size_t rows = b1 - b0;
        hipStream_t s = m->stream[r];
        std::string &err = m->dev_error[r];
        if (!rows) {  // more devices than rows: empty spans
            for (int e = 0; e < 5; ++e) KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[e][r], s));
            return KOFFT_OK;
        }
        int st = ensure(err, m->sig[r], rows * in_row * sizeof(T));
        if (st == KOFFT_OK && !in_place) st = ensure(err, m->out[r], rows * out_row * sizeof(T));
        if (st == KOFFT_OK && window) st = ensure(err, m->win[r], win_len * sizeof(T));
        if (st) return st;
        T *d_in = static_cast<T *>(m->sig[r].p);
        T *d_out = in_place ? d_in : static_cast<T *>(m->out[r].p);
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[0][r], s));
        if (window) KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(m->win[r].p, window, win_len * sizeof(T), hipMemcpyHostToDevice, s));
        KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(d_in, in + b0 * in_row, rows * in_row * sizeof(T), hipMemcpyHostToDevice, s));
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[1][r], s));
        st = kernel(m->ctx[r], d_in, d_out, window ? static_cast<const T *>(m->win[r].p) : nullptr, rows);
        if (st) {
            err = kofft_hip_last_error(m->ctx[r]);
            return st;
        }
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[2][r], s));
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[3][r], s));
        KOFFT_WORKER_TRY(m, r, hipMemcpyAsync(out + b0 * out_row, d_out, rows * out_row * sizeof(T), hipMemcpyDeviceToHost, s));
        KOFFT_WORKER_TRY(m, r, hipEventRecord(m->ev[4][r], s));
        KOFFT_WORKER_TRY(m, r, hipStreamSynchronize(s));
        return KOFFT_OK;
    });
    if (rc != KOFFT_OK) {
        collect_error(m);
        return rc;
    }
    m->timed = true;
    m->wall_ms = wall.ms();
    return KOFFT_OK;
}

// Device-resident rows: everything is an asynchronous enqueue, issued by the caller's thread device after device.
template <class Launch>
int multi_rows_dev(kofft_hip_multi *m, size_t batch, Launch launch)
{
    begin_call(m, false, false, false);
    DeviceGuard guard;
    const int G = m->ngpu;
    for (int r = 0; r < G; ++r) {
        size_t b0, b1;
        shard_range(batch, r, G, &b0, &b1);
        KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev[1][r], m->stream[r]));
        if (b1 > b0) {
            const int st = launch(r, b1 - b0);
            if (st) {
                m->last_error = std::string("device ") + std::to_string(m->devices[r]) + ": " + kofft_hip_last_error(m->ctx[r]);
                return st;
            }
        }
        KOFFT_MULTI_TRY(m, hipEventRecord(m->ev[2][r], m->stream[r]));
    }
    m->timed = true;
    return KOFFT_OK;
}

template <typename P>
bool any_null(P *const *ptrs, int n)
{
    if (!ptrs) return true;
    for (int r = 0; r < n; ++r)
        if (!ptrs[r]) return true;
    return false;
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
    DeviceGuard guard;
    kofft_hip_multi *m = new (std::nothrow) kofft_hip_multi();
    if (!m) return KOFFT_ERR_ALLOC;
    m->ngpu = ngpu;
    for (int r = 0; r < ngpu; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= have) {
            delete m;
            return KOFFT_ERR_INVALID_VALUE;
        }
        m->devices.push_back(d);
    }
    m->ctx.assign(ngpu, nullptr);
    m->stream.assign(ngpu, nullptr);
    m->sig.resize(ngpu);
    m->win.resize(ngpu);
    m->out.resize(ngpu);
    m->dev_error.resize(ngpu);
    for (auto &v : m->ev) v.assign(ngpu, nullptr);
    for (int r = 0; r < ngpu; ++r) {
        int rc = kofft_hip_create(m->devices[r], &m->ctx[r]);
        if (rc == KOFFT_OK && hipSetDevice(m->devices[r]) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc == KOFFT_OK && hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc == KOFFT_OK) rc = kofft_hip_set_stream(m->ctx[r], m->stream[r]);
        for (auto &v : m->ev)
            if (rc == KOFFT_OK && hipEventCreate(&v[r]) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc != KOFFT_OK) {
            kofft_hip_multi_destroy(m);
            return rc;
        }
    }
    if (const char *e = getenv("KOFFT_HIP_MULTI_GATHER")) m->gather_mode = (e[0] == 'd' || e[0] == '2') ? 2 : 1;
    *out = m;
    return KOFFT_OK;
}

int kofft_hip_multi_set_gather(kofft_hip_multi *m, int mode)
{
    if (!m) return KOFFT_ERR_NULL;
    if (mode != KOFFT_MULTI_GATHER_RCCL && mode != KOFFT_MULTI_GATHER_DIRECT) return KOFFT_ERR_INVALID_VALUE;
    m->gather_mode = mode;
    return KOFFT_OK;
}

int kofft_hip_multi_gather_mode(const kofft_hip_multi *m, int *configured, int *last)
{
    if (!m) return KOFFT_ERR_NULL;
    if (configured) *configured = m->gather_mode;
    if (last) *last = m->last_gather;
    return KOFFT_OK;
}

int kofft_hip_multi_destroy(kofft_hip_multi *m)
{
    if (!m) return KOFFT_ERR_NULL;
    DeviceGuard guard;
    m->workers.stop();
    for (size_t r = 0; r < m->ctx.size(); ++r)  // nothing may still be running when the communicators go
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
    if (!m->comms.empty()) {
        Rccl &r = rccl();
        for (nccl_comm_t c : m->comms)
            if (c && r.CommDestroy) (void)r.CommDestroy(c);
    }
    for (size_t i = 0; i < m->peer_stream.size(); ++i) {
        (void)hipSetDevice(m->devices[i / (size_t)m->ngpu]);
        if (m->peer_stream[i]) {
            (void)hipStreamSynchronize(m->peer_stream[i]);
            (void)hipStreamDestroy(m->peer_stream[i]);
        }
        if (m->peer_done[i]) (void)hipEventDestroy(m->peer_done[i]);
    }
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        for (DevBuf *b : {&m->sig[r], &m->win[r], &m->out[r]})
            if (b->p) (void)hipFree(b->p);
        for (auto &v : m->ev)
            if (v[r]) (void)hipEventDestroy(v[r]);
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

int kofft_hip_multi_stft_slice(const kofft_hip_multi *m, size_t len, size_t win_len, size_t hop, size_t frames, int rank,
                               size_t *first_sample, size_t *count)
{
    if (!m || !first_sample || !count) return KOFFT_ERR_NULL;
    if (rank < 0 || rank >= m->ngpu) return KOFFT_ERR_INVALID_VALUE;
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    size_t f0, f1, lo, hi;
    shard_range(frames, rank, m->ngpu, &f0, &f1);
    slice_range(len, win_len, hop, f0, f1, &lo, &hi);
    *first_sample = lo;
    *count = hi - lo;
    return KOFFT_OK;
}

int kofft_hip_multi_context(const kofft_hip_multi *m, int rank, kofft_hip_ctx **ctx, void **hip_stream)
{
    if (!m) return KOFFT_ERR_NULL;
    if (rank < 0 || rank >= m->ngpu) return KOFFT_ERR_INVALID_VALUE;
    if (ctx) *ctx = m->ctx[rank];
    if (hip_stream) *hip_stream = m->stream[rank];
    return KOFFT_OK;
}

int kofft_hip_multi_synchronize(kofft_hip_multi *m)
{
    if (!m) return KOFFT_ERR_NULL;
    for (int r = 0; r < m->ngpu; ++r) KOFFT_MULTI_TRY(m, hipStreamSynchronize(m->stream[r]));
    return KOFFT_OK;
}

int kofft_hip_multi_last_timing_ex(const kofft_hip_multi *m, float *upload_ms, float *kernel_ms, float *gather_ms,
                                   float *download_ms, float *wall_ms)
{
    if (!m) return KOFFT_ERR_NULL;
    float span[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (m->timed) {
        const bool have[4] = {m->had_upload, true, m->had_gather, m->had_download};
        for (int r = 0; r < m->ngpu; ++r) {
            // a *_dev call returns with its work still enqueued: wait for the last event it recorded
            (void)hipEventSynchronize(m->ev[m->had_download ? 4 : (m->had_gather ? 3 : 2)][r]);
            for (int p = 0; p < 4; ++p) {
                float t = 0.0f;
                if (have[p] && hipEventElapsedTime(&t, m->ev[p][r], m->ev[p + 1][r]) == hipSuccess && t > span[p]) span[p] = t;
            }
        }
        (void)hipGetLastError();
    }
    if (upload_ms) *upload_ms = span[0];
    if (kernel_ms) *kernel_ms = span[1];
    if (gather_ms) *gather_ms = span[2];
    if (download_ms) *download_ms = span[3];
    if (wall_ms) *wall_ms = m->wall_ms;
    return KOFFT_OK;
}

int kofft_hip_multi_last_timing(const kofft_hip_multi *m, float *compute_ms, float *gather_ms)
{
    return kofft_hip_multi_last_timing_ex(m, nullptr, compute_ms, gather_ms, nullptr, nullptr);
}

int kofft_hip_multi_stft_f32(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len,
                             size_t hop, float *out, size_t frames, int allgather, float **d_out_per_gpu)
{
    const int rc = multi_stft_host(m, signal, len, window, win_len, hop, out, frames, allgather, d_out_per_gpu);
    if (rc != KOFFT_OK && m) drain_all(m);  // possibly with work of other devices already enqueued
    return rc;
}

int kofft_hip_multi_stft_f32_dev(kofft_hip_multi *m, const float *const *d_signal_per_gpu, size_t len,
                                 const float *const *d_window_per_gpu, size_t win_len, size_t hop, size_t frames, int allgather,
                                 float **d_out_per_gpu)
{
    const int chk = stft_checks(len, win_len, hop, frames);
    if (chk != -100) return chk;
    if (!m || !d_out_per_gpu || any_null(d_window_per_gpu, m->ngpu) || !d_signal_per_gpu) return KOFFT_ERR_NULL;
    begin_call(m, false, allgather != 0, false);
    DeviceGuard guard;
    const int G = m->ngpu;
    const size_t per = ceil_div(frames, (size_t)G);
    const size_t frame_floats = win_len * 2;
    // every per-device argument is checked, and every buffer exists, BEFORE anything is enqueued: an error return must not leave
    // earlier devices writing the caller's buffers
    for (int r = 0; r < G; ++r) {
        size_t f0, f1, lo, hi;
        shard_range(frames, r, G, &f0, &f1);
        slice_range(len, win_len, hop, f0, f1, &lo, &hi);
        if (hi > lo && !d_signal_per_gpu[r]) return KOFFT_ERR_NULL;
    }
    if (allgather && m->gather_mode == 1) {
        const int rc = ensure_comms(m);
        if (rc) return rc;
    }
    std::vector<float *> base(G);
    for (int r = 0; r < G; ++r) {
        size_t f0, f1;
        shard_range(frames, r, G, &f0, &f1);
        if (!d_out_per_gpu[r]) {  // no caller buffer: the handle's
            KOFFT_MULTI_TRY(m, hipSetDevice(m->devices[r]));
            const int rc = ensure(m->last_error, m->out[r], (allgather ? (size_t)G * per : f1 - f0) * frame_floats * sizeof(float));
            if (rc) return rc;
            d_out_per_gpu[r] = static_cast<float *>(m->out[r].p);
        }
        base[r] = d_out_per_gpu[r];
    }
    // from here on work is in flight: every error exit drains the streams first
#define KOFFT_MULTI_TRY_DRAIN(m, expr)                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (m)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            drain_all(m);                                                                         \
            return KOFFT_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)
    for (int r = 0; r < G; ++r) {
        size_t f0, f1, lo, hi;
        shard_range(frames, r, G, &f0, &f1);
        slice_range(len, win_len, hop, f0, f1, &lo, &hi);
        const size_t count = f1 - f0;
        KOFFT_MULTI_TRY_DRAIN(m, hipSetDevice(m->devices[r]));
        hipStream_t s = m->stream[r];
        float *dst = base[r] + (allgather ? (size_t)r * per * frame_floats : 0);
        KOFFT_MULTI_TRY_DRAIN(m, hipEventRecord(m->ev[1][r], s));
        if (allgather && count < per)
            KOFFT_MULTI_TRY_DRAIN(m, hipMemsetAsync(dst + count * frame_floats, 0, (per - count) * frame_floats * sizeof(float), s));
        if (count) {
            const int rc = kofft_hip_stft_f32_dev(m->ctx[r], d_signal_per_gpu[r], hi - lo, d_window_per_gpu[r], win_len, hop, dst, 0, count);
            if (rc) {
                m->last_error = std::string("device ") + std::to_string(m->devices[r]) + ": " + kofft_hip_last_error(m->ctx[r]);
                drain_all(m);
                return rc;
            }
        }
        KOFFT_MULTI_TRY_DRAIN(m, hipEventRecord(m->ev[2][r], s));
    }
#undef KOFFT_MULTI_TRY_DRAIN
    if (allgather) {
        int rc = gather_any(m, base.data(), per * frame_floats);
        if (rc == KOFFT_OK) rc = record_all(m, 3);
        if (rc) {
            drain_all(m);
            return rc;
        }
    }
    m->timed = true;
    return KOFFT_OK;
}

int kofft_hip_stft_f32_multi(int ngpu, const float *signal, size_t len, const float *window, size_t win_len, size_t hop,
                             float *out, size_t frames, int allgather)
{
    // argument checks first: they need no device (and keep the reference's order, stft.rs:83-87)
    if (ngpu <= 0) return KOFFT_ERR_INVALID_VALUE;
    const int chk = stft_checks(len, win_len, hop, frames);
    if (chk != -100) return chk;
    if ((!signal && len) || !window || !out) return KOFFT_ERR_NULL;
    kofft_hip_multi *m = nullptr;
    int rc = kofft_hip_multi_create(ngpu, nullptr, &m);
    if (rc) return rc;
    rc = kofft_hip_multi_stft_f32(m, signal, len, window, win_len, hop, out, frames, allgather, nullptr);
    (void)kofft_hip_multi_destroy(m);
    return rc;
}

// ---- batched transforms: no exchange ---------------------------------------------------------------------------------
int kofft_hip_multi_fft_c32(kofft_hip_multi *m, float *data, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || !data) return KOFFT_ERR_NULL;
    const int rc = multi_rows_host<float>(m, data, data, 2 * n, 2 * n, batch, true, nullptr, 0,
                                          [&](kofft_hip_ctx *c, float *d, float *, const float *, size_t rows) {
                                              return kofft_hip_fft_c32_dev(c, d, n, rows, inverse);
                                          });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

int kofft_hip_multi_fft_c64(kofft_hip_multi *m, double *data, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || !data) return KOFFT_ERR_NULL;
    const int rc = multi_rows_host<double>(m, data, data, 2 * n, 2 * n, batch, true, nullptr, 0,
                                           [&](kofft_hip_ctx *c, double *d, double *, const double *, size_t rows) {
                                               return kofft_hip_fft_c64_dev(c, d, n, rows, inverse);
                                           });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

int kofft_hip_multi_rfft_f32(kofft_hip_multi *m, const float *in, float *out, const float *window, size_t n, size_t batch)
{
    // rfft_direct's checks (rfft.rs:431-440)
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;
    if (!m || !in || !out) return KOFFT_ERR_NULL;
    const int rc = multi_rows_host<float>(m, in, out, n, (n / 2 + 1) * 2, batch, false, window, n,
                                          [&](kofft_hip_ctx *c, float *di, float *dout, const float *dw, size_t rows) {
                                              return kofft_hip_rfft_f32_dev(c, di, dout, dw, n, rows);
                                          });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

int kofft_hip_multi_fft_c32_dev(kofft_hip_multi *m, float *const *d_data_per_gpu, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || !d_data_per_gpu) return KOFFT_ERR_NULL;
    const int rc = multi_rows_dev(m, batch, [&](int r, size_t rows) {
        if (!d_data_per_gpu[r]) return KOFFT_ERR_NULL;
        return kofft_hip_fft_c32_dev(m->ctx[r], d_data_per_gpu[r], n, rows, inverse);
    });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

int kofft_hip_multi_fft_c64_dev(kofft_hip_multi *m, double *const *d_data_per_gpu, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!m || !d_data_per_gpu) return KOFFT_ERR_NULL;
    const int rc = multi_rows_dev(m, batch, [&](int r, size_t rows) {
        if (!d_data_per_gpu[r]) return KOFFT_ERR_NULL;
        return kofft_hip_fft_c64_dev(m->ctx[r], d_data_per_gpu[r], n, rows, inverse);
    });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

int kofft_hip_multi_rfft_f32_dev(kofft_hip_multi *m, const float *const *d_in_per_gpu, float *const *d_out_per_gpu,
                                 const float *const *d_window_per_gpu, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;
    if (!m || !d_in_per_gpu || !d_out_per_gpu) return KOFFT_ERR_NULL;
    const int rc = multi_rows_dev(m, batch, [&](int r, size_t rows) {
        if (!d_in_per_gpu[r] || !d_out_per_gpu[r]) return KOFFT_ERR_NULL;
        return kofft_hip_rfft_f32_dev(m->ctx[r], d_in_per_gpu[r], d_out_per_gpu[r], d_window_per_gpu ? d_window_per_gpu[r] : nullptr, n, rows);
    });
    if (rc != KOFFT_OK) drain_all(m);
    return rc;
}

}  // extern "C"
