// ubench_fused.hip -- would ONE persistent launch that runs both factors of the large-n path, software-pipelined over
// small chunks, get the intermediate out of the 256 MiB Infinity Cache?  (development tool, round 2; copies only, no FFT)
//
// Model of config 5 (2^20-point c64 transforms, 16 MiB each): task A copies a column tile (8 columns x 1024 rows, 128-byte
// segments at a 16 KiB stride) X -> M, task B copies a row tile of M (8 rows x 1024) to a transposed column tile of Y.
// One 512-thread workgroup per CU walks: A(0), [A(1) B(0)], [A(2) B(1)], ... -- B(c) needs every workgroup's A(c), which
// ended one iteration earlier, so the counter wait is (nearly) free; M is a ring of NB chunks.  The next task's 16 loads are
// issued before the current task's stores.  Hand-off forms (MI355X_MICROARCH.md, inter-workgroup visibility):
//   MODE 0: plain M stores + agent release fence, agent acquire fence + plain M loads
//   MODE 1: sc1 (write-through) M stores drained with vmcnt(0), sc1 M loads, no fences
// phase 1 / 2 = only the A / only the B tasks (no hand-off): the two-launch baseline in the same code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static constexpr int G = 256;              // workgroups = CUs
static constexpr size_t XFE = size_t(1) << 20;  // points per transform
static constexpr int SPIN_LIMIT = 400000;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(XFE * 16), 0x00020000);
}


// dependent VALU work standing in for a tile's butterflies: value-preserving (x ^ k ^ k), opaque to the optimizer
__device__ __forceinline__ void burn(v4u (&v)[16], int work)
{
    for (int i = 0; i < work; ++i) {
        const unsigned k = (unsigned)i * 2654435761u;
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(v[u].x) : "v"(k));
    }
}

struct Task {
    int kind, chunk, j;
    bool valid;
};
__device__ __forceinline__ Task decode(int n, int tpw, int nchunks, int phase)
{
    const int it = n / (2 * tpw), s = n % (2 * tpw);
    Task t;
    if (s < tpw) {
        t.kind = 0, t.chunk = it, t.j = s, t.valid = it < nchunks && (phase & 1);
    } else {
        t.kind = 1, t.chunk = it - 1, t.j = s - tpw, t.valid = it >= 1 && (phase & 2);
    }
    return t;
}

__device__ __forceinline__ void wait_count(const unsigned *ctr, unsigned target, unsigned *timeout)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) {  // every wave leaves: a timeout word, never a hang
                __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            if ((spins & 255) == 0 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void fused_copy(const char *X, char *Y, char *M, unsigned *doneA, unsigned *doneB, unsigned *timeout, int C,
                                                  int nchunks, int NB, int phase, int work = 0)
{
    const int tid = threadIdx.x, w = blockIdx.x;
    const int tpw = C * 128 / G;
    const int tau = tid / 8, col = tid % 8;
    // per-lane byte offsets inside a transform
    const int offA = (tau * 1024 + col) * 16;   // + c * 64 * 1024 * 16 + tile * 8 * 16   (column tile, both sides of A and the Y side of B)
    const int offBrow = (col * 1024 + tau) * 16;  // + c * 64 * 16 + tile * 8 * 1024 * 16  (row tile of M)
    constexpr int AUXM = MODE == 1 ? 16 : 0;    // sc1
    const bool handoff = phase == 3;

    v4u cur[16], nxt[16];
    auto issue = [&](const Task &t, v4u(&v)[16]) {
        const int id = w + G * t.j, xl = id / 128, tile = id % 128;
        if (t.kind == 0) {
            const rsrc_t r = make_rsrc(X + ((size_t)t.chunk * C + xl) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(r, offA + tile * 128, c * (64 * 1024 * 16), 2);
        } else {
            const rsrc_t r = make_rsrc(M + ((size_t)(t.chunk % NB) * C + xl) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(r, offBrow + tile * (8 * 1024 * 16), c * (64 * 16), AUXM);
        }
    };
    auto store = [&](const Task &t, v4u(&v)[16]) {
        const int id = w + G * t.j, xl = id / 128, tile = id % 128;
        if (t.kind == 0) {
            const rsrc_t r = make_rsrc(M + ((size_t)(t.chunk % NB) * C + xl) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) __builtin_amdgcn_raw_buffer_store_b128(v[c], r, offA + tile * 128 + c * (64 * 1024 * 16), 0, AUXM);
        } else {
            const rsrc_t r = make_rsrc(Y + ((size_t)t.chunk * C + xl) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) __builtin_amdgcn_raw_buffer_store_b128(v[c], r, offA + tile * 128 + c * (64 * 1024 * 16), 0, 2);
        }
    };
    auto before_loads = [&](const Task &t) {  // B(c) reads what every workgroup's A(c) wrote
        if (handoff && t.kind == 1 && t.j == 0) {
            wait_count(doneA + t.chunk, G, timeout);
            if (MODE == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    };
    auto before_stores = [&](const Task &t) {  // A(c) overwrites the ring slot B(c - NB) read
        if (handoff && t.kind == 0 && t.j == 0 && t.chunk >= NB) wait_count(doneB + (t.chunk - NB), G, timeout);
    };
    auto after_stores = [&](const Task &t) {
        if (handoff && t.j == tpw - 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (MODE == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add((t.kind == 0 ? doneA : doneB) + t.chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    const int ntask = (nchunks + 1) * 2 * tpw;
    int n = 0;
    Task t = decode(n, tpw, nchunks, phase);
    while (!t.valid && n + 1 < ntask) t = decode(++n, tpw, nchunks, phase);
    if (!t.valid) return;
    before_loads(t);
    issue(t, cur);
    for (;;) {
        int m = n + 1;
        Task u;
        u.valid = false;
        while (m < ntask) {
            u = decode(m, tpw, nchunks, phase);
            if (u.valid) break;
            ++m;
        }
        const bool more = m < ntask && u.valid;
        if (more) {
            before_loads(u);
            issue(u, nxt);
        }
        burn(cur, work);
        before_stores(t);
        store(t, cur);
        after_stores(t);
        if (!more) break;
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
        t = u;
        n = m;
    }
}


// ---- the fused walk again, as the product kernel would run it: set P always holds the A tile, set Q the B tile (no
// copies), one poll (both counters) and one signal (both counters) per iteration, signals lagged so that no wait drains
// the queue.  C = 2 transforms per chunk: every workgroup has ONE A tile and ONE B tile per chunk.  `work` = dependent
// VALU work per tile standing in for the butterflies (0 = pure copy).
__global__ __launch_bounds__(512) void fused2_copy(const char *X, char *Y, char *M, unsigned *doneA, unsigned *doneB, unsigned *timeout, int nchunks,
                                                   int NB, int work, int LAG)
{
    const int tid = threadIdx.x, w = blockIdx.x;
    const int xl = w / 128, tile = w % 128;
    const int tau = tid / 8, col = tid % 8;
    const int offA = (tau * 1024 + col) * 16 + tile * 128;
    const int offBrow = (col * 1024 + tau) * 16 + tile * (8 * 1024 * 16);
    v4u P[16], Q[16];
    auto load_A = [&](int c, bool valid) {
        const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(X + ((size_t)(valid ? c : 0) * 2 + xl) * XFE * 16), 0,
                                                           valid ? (int)(XFE * 16) : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < 16; ++u) P[u] = __builtin_amdgcn_raw_buffer_load_b128(r, offA, u * (64 * 1024 * 16), 2);
    };
    auto load_B = [&](int c) {
        const rsrc_t r = make_rsrc(M + ((size_t)(c % NB) * 2 + xl) * XFE * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) Q[u] = __builtin_amdgcn_raw_buffer_load_b128(r, offBrow, u * (64 * 16), 16);
    };
    auto store_A = [&](int c) {
        const rsrc_t r = make_rsrc(M + ((size_t)(c % NB) * 2 + xl) * XFE * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_amdgcn_raw_buffer_store_b128(P[u], r, offA + u * (64 * 1024 * 16), 0, 16);
    };
    auto store_B = [&](int c) {
        const rsrc_t r = make_rsrc(Y + ((size_t)c * 2 + xl) * XFE * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_amdgcn_raw_buffer_store_b128(Q[u], r, offA + u * (64 * 1024 * 16), 0, 2);
    };
    load_A(0, true);
    for (int it = 0; it < nchunks + LAG; ++it) {
        const bool hasA = it < nchunks, hasB = it >= LAG;
        // one poll for both dependencies: every workgroup's A(it-1) is stored; every workgroup's B(it-NB) has been read
        if (tid == 0) {
            int spins = 0;
            for (;;) {
                const bool okA = !hasB || __hip_atomic_load(doneA + (it - LAG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)G;
                const bool okB = !hasA || it < NB || __hip_atomic_load(doneB + (it - NB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)G;
                if (okA && okB) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                if ((spins & 255) == 0 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
        }
        __syncthreads();
        if (hasB) load_B(it - LAG);
        if (hasA) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // P has landed (Q's 16 loads are younger)
            burn(P, work);
            store_A(it);
        }
        load_A(it + 1, it + 1 < nchunks);  // an empty descriptor past the end: 16 loads that touch nothing
        if (hasB) {
            burn(Q, work);
            store_B(it - LAG);
        }
        // signals: A(it)'s stores are complete once at most the 16 loads and 16 stores issued after them are outstanding
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (hasA) __hip_atomic_fetch_add(doneA + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (hasB) __hip_atomic_fetch_add(doneB + (it - LAG), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- fused3: the fully pipelined walk.  Per iteration two sync points, each placed where the queue has had a whole compute
// phase to empty (so the vmcnt(0) there is free): (1) after computing A(it): signal "B(it-1-LAG) stored/read", check the ring
// slot is free; then store A(it), prefetch A(it+1).  (2) after computing B(it-LAG): signal "A(it) stored" -- its stores were
// issued a compute phase ago --, check A(it+1-LAG) is complete everywhere; then store B(it-LAG), prefetch B(it+1-LAG).
// The polls are issued one sync point early and consumed at the next (no exposed round trip unless the answer was "not yet").
__global__ __launch_bounds__(512) void fused3_copy(const char *X, char *Y, char *M, unsigned *doneA, unsigned *doneB, unsigned *timeout, int nchunks,
                                                   int NB, int work, int LAG)
{
    const int tid = threadIdx.x, w = blockIdx.x;
    const int xl = w / 128, tile = w % 128;
    const int tau = tid / 8, col = tid % 8;
    const int offA = (tau * 1024 + col) * 16 + tile * 128;
    const int offBrow = (col * 1024 + tau) * 16 + tile * (8 * 1024 * 16);
    v4u P[16], Q[16];
    auto load_A = [&](int c, bool valid) {
        const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(X + ((size_t)(valid ? c : 0) * 2 + xl) * XFE * 16), 0,
                                                           valid ? (int)(XFE * 16) : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < 16; ++u) P[u] = __builtin_amdgcn_raw_buffer_load_b128(r, offA, u * (64 * 1024 * 16), 2);
    };
    auto load_B = [&](int c, bool valid) {
        const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(M + ((size_t)((valid ? c : 0) % NB) * 2 + xl) * XFE * 16, 0, valid ? (int)(XFE * 16) : 0,
                                                           0x00020000);
#pragma unroll
        for (int u = 0; u < 16; ++u) Q[u] = __builtin_amdgcn_raw_buffer_load_b128(r, offBrow, u * (64 * 16), 16);
    };
    auto store_A = [&](int c) {
        const rsrc_t r = make_rsrc(M + ((size_t)(c % NB) * 2 + xl) * XFE * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_amdgcn_raw_buffer_store_b128(P[u], r, offA + u * (64 * 1024 * 16), 0, 16);
    };
    auto store_B = [&](int c) {
        const rsrc_t r = make_rsrc(Y + ((size_t)c * 2 + xl) * XFE * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) __builtin_amdgcn_raw_buffer_store_b128(Q[u], r, offA + u * (64 * 1024 * 16), 0, 2);
    };
    // wave 0's lane 0 keeps the early poll results; `need` = counter must have reached G
    auto spin = [&](const unsigned *ctr) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255) == 0 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
    };
    unsigned early_ring = G, early_a = G;  // results of polls issued one sync point earlier (lane 0 of wave 0)
    load_A(0, true);
    for (int it = 0; it < nchunks + LAG; ++it) {
        const bool hasA = it < nchunks, hasB = it >= LAG;
        const int cb = it - LAG;  // the B chunk of this iteration
        if (hasA) burn(P, work);
        // ---- sync point 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (cb - 1 >= 0) __hip_atomic_fetch_add(doneB + (cb - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // B(cb-1): read, stored, drained
            if (hasA && it >= NB && early_ring < (unsigned)G) spin(doneB + (it - NB));
            // early poll for sync point 2 of this iteration: is A(it + 1 - LAG) complete everywhere?
            const int ca = it + 1 - LAG;
            early_a = (ca >= 0 && ca < nchunks) ? __hip_atomic_load(doneA + ca, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)G;
        }
        __syncthreads();
        if (hasA) store_A(it);
        load_A(it + 1, it + 1 < nchunks);
        if (hasB) burn(Q, work);
        // ---- sync point 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (hasA) __hip_atomic_fetch_add(doneA + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ca = it + 1 - LAG;
            if (ca >= 0 && ca < nchunks && early_a < (unsigned)G) spin(doneA + ca);
            // early poll for sync point 1 of the next iteration: has everybody read ring slot of chunk it + 1 - NB?
            const int cr = it + 1 - NB;
            early_ring = (cr >= 0 && it + 1 < nchunks) ? __hip_atomic_load(doneB + cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)G;
        }
        __syncthreads();
        if (hasB) store_B(cb);
        load_B(cb + 1, cb + 1 >= 0 && cb + 1 < nchunks);
    }
}

// ---- two CONCURRENT launches (two streams), 128 workgroups each: kernel A walks the transforms doing its column tile of
// each, kernel B its row tile of each; A runs ahead of B by up to NB-1 chunks of C transforms.  Signals are LAGGED so that no
// wait drains the queue: A signals chunk c when the stores of its last tile are complete while the next tile's loads and
// stores are already queued (vmcnt counts in issue order); B signals "slot read" as soon as its loads have landed.
template <int KIND>
__global__ __launch_bounds__(512) void stream_copy(const char *X, char *Y, char *M, unsigned *doneA, unsigned *doneB, unsigned *timeout, int C, int nxf,
                                                   int NB)
{
    const int tid = threadIdx.x, tile = blockIdx.x;  // 0..127
    const int tau = tid / 8, col = tid % 8;
    const int offA = (tau * 1024 + col) * 16 + tile * 128;
    const int offBrow = (col * 1024 + tau) * 16 + tile * (8 * 1024 * 16);
    const int ringx = NB * C;
    v4u cur[16], nxt[16];
    auto issue = [&](int xf, v4u(&v)[16]) {
        if (KIND == 0) {
            const rsrc_t r = make_rsrc(X + (size_t)xf * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(r, offA, c * (64 * 1024 * 16), 2);
        } else {
            if (xf % C == 0) wait_count(doneA + xf / C, 128, timeout);
            const rsrc_t r = make_rsrc(M + (size_t)(xf % ringx) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(r, offBrow, c * (64 * 16), 16);
        }
    };
    auto store = [&](int xf, v4u(&v)[16]) {
        if (KIND == 0) {
            if (xf % C == 0 && xf / C >= NB) wait_count(doneB + (xf / C - NB), 128, timeout);
            const rsrc_t r = make_rsrc(M + (size_t)(xf % ringx) * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) __builtin_amdgcn_raw_buffer_store_b128(v[c], r, offA + c * (64 * 1024 * 16), 0, 16);
        } else {
            const rsrc_t r = make_rsrc(Y + (size_t)xf * XFE * 16);
#pragma unroll
            for (int c = 0; c < 16; ++c) __builtin_amdgcn_raw_buffer_store_b128(v[c], r, offA + c * (64 * 1024 * 16), 0, 2);
        }
    };
    auto signal = [&](unsigned *ctr) {
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    issue(0, cur);
    for (int xf = 0; xf < nxf; ++xf) {
        const bool more = xf + 1 < nxf;
        if (more) issue(xf + 1, nxt);
        if (KIND == 1 && xf % C == C - 1) {  // the chunk's last tile has been read once `cur` has landed
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            signal(doneB + xf / C);
        }
        store(xf, cur);
        if (KIND == 0) {
            // stores of tile xf-1 are complete when at most this tile's 16 stores and the next tile's 16 loads are outstanding
            if (xf >= 1 && (xf - 1) % C == C - 1) {
                if (more) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                signal(doneA + (xf - 1) / C);
            }
            if (!more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                signal(doneA + xf / C);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
    }
}

int main()
{
    const size_t nxf = 256;  // 4 GiB per side
    const size_t bytes = nxf * XFE * 16;
    char *x, *y, *m;
    unsigned *ctr;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMalloc(&m, size_t(32) * XFE * 16));  // 512 MiB: the two-launch baseline's chunk; rings use the front of it
    CK(hipMalloc(&ctr, 4096 * 4));
    std::vector<unsigned> hx(bytes / 4);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(x, hx.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0, bytes));
    CK(hipMemset(m, 0, size_t(32) * XFE * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned *doneA = ctr, *doneB = ctr + 1024, *timeout = ctr + 2048;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t ea, eb, ef;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    auto concurrent = [&](int C, int NB) {
        // fork from the null stream, join back: timed by the events on the null stream
        (void)hipMemsetAsync(ctr, 0, 4096 * 4, 0);
        (void)hipEventRecord(ef, 0);
        (void)hipStreamWaitEvent(sa, ef, 0);
        (void)hipStreamWaitEvent(sb, ef, 0);
        hipLaunchKernelGGL(stream_copy<0>, dim3(128), dim3(512), 0, sa, x, y, m, doneA, doneB, timeout, C, (int)nxf, NB);
        hipLaunchKernelGGL(stream_copy<1>, dim3(128), dim3(512), 0, sb, x, y, m, doneA, doneB, timeout, C, (int)nxf, NB);
        (void)hipEventRecord(ea, sa);
        (void)hipEventRecord(eb, sb);
        (void)hipStreamWaitEvent(0, ea, 0);
        (void)hipStreamWaitEvent(0, eb, 0);
    };

    auto baseline = [&]() {
        for (size_t c0 = 0; c0 + 32 <= nxf; c0 += 32) {
            hipLaunchKernelGGL(fused_copy<0>, dim3(G), dim3(512), 0, 0, x + c0 * XFE * 16, y + c0 * XFE * 16, m, doneA, doneB, timeout, 32, 1, 1, 1);
            hipLaunchKernelGGL(fused_copy<0>, dim3(G), dim3(512), 0, 0, x + c0 * XFE * 16, y + c0 * XFE * 16, m, doneA, doneB, timeout, 32, 1, 1, 2);
        }
    };
    auto fused = [&](int mode, int C, int NB, int work = 0) {
        hipMemsetAsync(ctr, 0, 4096 * 4, 0);
        const int nchunks = (int)(nxf / C);
        if (mode == 0)
            hipLaunchKernelGGL(fused_copy<0>, dim3(G), dim3(512), 0, 0, x, y, m, doneA, doneB, timeout, C, nchunks, NB, 3, work);
        else
            hipLaunchKernelGGL(fused_copy<1>, dim3(G), dim3(512), 0, 0, x, y, m, doneA, doneB, timeout, C, nchunks, NB, 3, work);
    };
    // expected Y: Y[xf][q*1024 + K] = X[xf][K*1024 + q]  (A is the identity into M, B transposes) -- check a sample after each variant
    std::vector<unsigned> hy(XFE * 4);
    auto check = [&](const char *what) -> int {
        unsigned to = 0;
        if (hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        size_t bad = 0;
        for (size_t xf : {size_t(0), size_t(101), nxf - 1}) {
            if (hipMemcpy(hy.data(), y + xf * XFE * 16, XFE * 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            for (size_t q = 0; q < 1024; ++q)
                for (size_t K = 0; K < 1024; ++K)
                    for (int d = 0; d < 4; ++d) bad += hy[(q * 1024 + K) * 4 + d] != hx[(xf * XFE + K * 1024 + q) * 4 + d];
        }
        printf("   check %-28s timeout=%u  wrong words=%zu\n", what, to, bad);
        fflush(stdout);
        return (to || bad) ? 1 : 0;
    };
    auto timeit = [&](const char *name, auto &&fn) {
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) {
            hipEventRecord(e0);
            fn();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-44s %.3f ms per 256 transforms -> x4 = %.2f ms per 1024\n", name, ms[ms.size() / 2], ms[ms.size() / 2] * 4);
        fflush(stdout);
    };
    for (int r = 0; r < 10; ++r) baseline();
    CK(hipDeviceSynchronize());
    timeit("two launches per 512 MiB chunk", baseline);
    if (check("baseline")) return 1;
    for (int mode = 1; mode < 2; ++mode)
        for (int C : {2, 4})
            for (int NB : {3}) {
                if ((size_t)C * NB > 32) continue;
                char name[96];
                snprintf(name, sizeof name, "fused %s C=%d (%3d MiB) ring %d (%3d MiB)", mode ? "sc1  " : "fence", C, C * 16, NB, C * NB * 16);
                CK(hipMemset(y, 0, bytes));
                timeit(name, [&] { fused(mode, C, NB); });
                CK(hipDeviceSynchronize());
                if (check(name)) return 1;
            }
    for (int work : {0, 24, 48, 72}) {
        char nm[64];
        snprintf(nm, sizeof nm, "two launches per 512 MiB chunk, work %d", work);
        timeit(nm, [&] {
            for (size_t c0 = 0; c0 + 32 <= nxf; c0 += 32) {
                hipLaunchKernelGGL(fused_copy<0>, dim3(G), dim3(512), 0, 0, x + c0 * XFE * 16, y + c0 * XFE * 16, m, doneA, doneB, timeout, 32, 1, 1, 1, work);
                hipLaunchKernelGGL(fused_copy<0>, dim3(G), dim3(512), 0, 0, x + c0 * XFE * 16, y + c0 * XFE * 16, m, doneA, doneB, timeout, 32, 1, 1, 2, work);
            }
        });
        for (int C : {2, 4})
            for (int NB : {3, 4}) {
                char name[96];
                snprintf(name, sizeof name, "fused v1 sc1 C=%d ring %d (%3d MiB) work %d", C, NB, C * NB * 16, work);
                CK(hipMemset(y, 0, bytes));
                timeit(name, [&] { fused(1, C, NB, work); });
                CK(hipDeviceSynchronize());
                if (check(name)) return 1;
            }
        {
            char name[96];
            snprintf(name, sizeof name, "fused3 C=2 lag 2 ring 5 work %d", work);
            timeit(name, [&] {
                (void)hipMemsetAsync(ctr, 0, 4096 * 4, 0);
                hipLaunchKernelGGL(fused3_copy, dim3(G), dim3(512), 0, 0, x, y, m, doneA, doneB, timeout, (int)(nxf / 2), 5, work, 2);
            });
            CK(hipDeviceSynchronize());
            if (check(name)) return 1;
        }
    }
    for (int work : {0})
      for (int LAG : {2})
        for (int NB : {LAG + 2}) {
            char name[96];
            snprintf(name, sizeof name, "fused2 C=2 lag %d ring %d (%3d MiB) work %d", LAG, NB, 2 * NB * 16, work);
            CK(hipMemset(y, 0, bytes));
            timeit(name, [&] {
                (void)hipMemsetAsync(ctr, 0, 4096 * 4, 0);
                hipLaunchKernelGGL(fused2_copy, dim3(G), dim3(512), 0, 0, x, y, m, doneA, doneB, timeout, (int)(nxf / 2), NB, work, LAG);
            });
            CK(hipDeviceSynchronize());
            if (check(name)) return 1;
        }
    for (int C : {4})
        for (int NB : {3}) {  // 2 deadlocks: A signals one tile late, B prefetches one tile ahead
            if ((size_t)C * NB > 32) continue;
            char name[96];
            snprintf(name, sizeof name, "concurrent A|B C=%d (%3d MiB) ring %d (%3d MiB)", C, C * 16, NB, C * NB * 16);
            CK(hipMemset(y, 0, bytes));
            timeit(name, [&] { concurrent(C, NB); });
            CK(hipDeviceSynchronize());
            if (check(name)) return 1;
        }
    timeit("two launches per 512 MiB chunk (again)", baseline);
    return 0;
}
