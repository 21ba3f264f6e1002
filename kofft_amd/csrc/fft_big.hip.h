// fft_big.hip.h -- power-of-two transforms too large for one workgroup's LDS (n > 16384 f32 / 8192 f64),
// e.g. BASELINE config #5: 2^20-point Complex64.  Two factors up to 2^21, three (BigMidIO below) from 2^22.
//
// The reference runs L = log2(n) radix-2 Stockham stages over the whole array (fft.rs:834-898).  In index bits,
// stage s maps [k : s bits][b][j] -> [b'][k][j], so the stages split cleanly into two factors, L = LA + LB:
//
//   factor A = stages 0 .. LA-1.  No frequency prefix yet (k empty); the LB low bits j are untouched.
//       For each j it is a complete 2^LA-point Stockham transform of the column  x_c = in[c * 2^LB + j]
//       whose result q (natural order) lands at  q * 2^LB + j.
//   factor B = stages LA .. L-1.  The prefix K (LA bits) is fixed per row; no j bits remain.
//       For each K it is a 2^LB-point Stockham transform of the contiguous row  x_c = mid[K * 2^LB + c]
//       whose result q lands at  q * 2^LA + K   -- natural order of the full transform.
//
// Every butterfly is the reference's butterfly with the reference's table entry: global stage s, group kk uses
// T_n[kk * 2^(L-1-s)].  For a sub-transform that handles global stages S_off + s_local with local group index
// kk_local, kk = K + 2^S_off * kk_local, hence
//   index = (K << (L-1-S_off-s_local)) + (idx_local << (L - L_sub)),   idx_local = kk_local << (L_sub-1-s_local)
// (TwSub in fft_device.hip.h).  T_n != a subsampling of T_{2^L_sub} bitwise (each table is its own recurrence),
// so both factors index the ONE table T_n.  Only the order of independent butterflies changes: results are
// bit-identical to the reference's 20 sweeps, with 2 passes over HBM instead of 20.
//
// The two factors run as the generic workgroup kernel (fft_wg.hip.h) with these IO policies; the intermediate
// lives in a scratch buffer of `chunk` transforms so that factor B reads it back from the Infinity Cache.
#pragma once

#include <type_traits>

#include "fft_persist.hip.h"
#include "fft_wg.hip.h"

namespace kofft {

// factor A: unit xf = (b, j) with j in [0, 2^LB): column j of transform b.  Adjacent units are adjacent columns,
// so lanes run over units first (kSlotMinor) and a wave touches whole 128-byte lines.
//
// PRE (round 3): a pointwise factor folded into the first factor's LOAD, so that transforms composed around the factor path
// need no separate pass over HBM for it (fft.rs:1088-1132, rfft.rs:444-447 with stft.rs:96's product):
//   PRE_NONE   the plain transform;
//   PRE_CHIRP  Bluestein's a = x * chirp, zero-padded: the input rows are n_in values long (stride n_in), element i >= n_in is
//              exactly (0, 0) (the descriptor's bounds check supplies the zero, the select keeps its sign), INVERSE = ifft's
//              conj on the way in, BEFORE the product -- the expressions of BlueFirstIO / bluestein_pre_kernel;
//   PRE_WINDOW the real transform's pack: the row read as m complex values z[i] = (x[2i], x[2i+1]), each part times its own
//              window sample (pre_tab read as pairs): real_window_kernel's product.
enum : int { PRE_NONE = 0, PRE_CHIRP = 1, PRE_WINDOW = 2 };
template <typename T, bool INVERSE, int PRE = PRE_NONE>
struct BigColsIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;  // c64: 8-byte exchange elements (re / im in two rounds), 8-column tiles fit twice per CU
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;  // two 512-thread workgroups per CU need <= 128 VGPRs
    static constexpr int kPre = PRE;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int LB;     // log2 of the column count
    int shift;  // L - LA
    size_t n;   // full transform length
    bool nt_in_pieces = true;  // streaming loads: off for tiles narrower than half a line (narrow tiles of a single transform)
    const cpx<T> *__restrict__ pre_tab = nullptr;  // PRE_CHIRP: chirp (n_in entries); PRE_WINDOW: the window as n pairs
    unsigned n_in = 0;                             // PRE_CHIRP: values per input row (= its stride)
    static constexpr bool nt = false;  // the intermediate is read back by the next factor: plain stores
    __device__ __forceinline__ TwSubFirst tw_map(size_t) const { return TwSubFirst{shift}; }
    // tile forms (fft_tile_persist_kernel): local index i of unit xf lives at  transform(xf) * n + off(xf) + (i << sl)
    static constexpr bool kConjIn = INVERSE, kConjScaleOut = false, kNtOut = false;
    static constexpr bool kTileInvariantTw = true;  // no frequency prefix yet: every tile uses the same 2^(L_sub-1) table entries
    __device__ __forceinline__ bool nt_in() const { return nt_in_pieces; }
    __device__ __forceinline__ T out_scale() const { return T(1); }
    __device__ __forceinline__ size_t in_row() const { return PRE == PRE_CHIRP ? (size_t)n_in : n; }     // input row stride = valid values
    __device__ __forceinline__ size_t out_row() const { return n; }
    __device__ __forceinline__ size_t out_valid() const { return n; }
    // the value the transform sees for element `elem` of its row, from the loaded (and, for ifft, conjugated) value
    __device__ __forceinline__ cpx<T> pre(cpx<T> v, unsigned elem) const
    {
        if constexpr (PRE == PRE_CHIRP) {
            const unsigned ec = elem < n_in ? elem : n_in - 1;  // branch-free: clamp the address, select the value
            const cpx<T> a = cmul(v, pre_tab[ec]);
            return elem < n_in ? a : mk<T>(T(0), T(0));
        } else if constexpr (PRE == PRE_WINDOW) {
            const cpx<T> w = pre_tab[elem];
            return mk<T>(v.re * w.re, v.im * w.im);
        } else {
            return v;
        }
    }
    // pre() in two steps for the persistent tile kernel (round 6): the table entry through a descriptor -- requested BEFORE the next tile's
    // prefetch is issued, see fft_tile_persist_kernel -- and the product
    __device__ __forceinline__ rsrc_t pre_desc() const
    {
        return make_rsrc(pre_tab, (unsigned)((PRE == PRE_CHIRP ? (size_t)n_in : n) * sizeof(cpx<T>)));
    }
    __device__ __forceinline__ cpx<T> pre_fetch(const rsrc_t d, const unsigned elem) const
    {
        unsigned ec = elem;
        if constexpr (PRE == PRE_CHIRP) ec = elem < n_in ? elem : n_in - 1;  // (as pre(): clamp the address, select the value)
        return buf_load_cpx<T, AUX_DEFAULT>(d, (int)(ec * (unsigned)sizeof(cpx<T>)), 0);
    }
    __device__ __forceinline__ cpx<T> pre_apply(const cpx<T> v, const cpx<T> w, const unsigned elem) const
    {
        if constexpr (PRE == PRE_CHIRP) {
            const cpx<T> a = cmul(v, w);
            return elem < n_in ? a : mk<T>(T(0), T(0));
        } else if constexpr (PRE == PRE_WINDOW) {
            return mk<T>(v.re * w.re, v.im * w.im);
        } else {
            return v;
        }
    }
    __device__ __forceinline__ size_t xf_transform(size_t xf) const { return xf >> LB; }
    __device__ __forceinline__ unsigned in_off(size_t xf) const { return (unsigned)(xf & ((size_t(1) << LB) - 1)); }
    __device__ __forceinline__ int in_sl() const { return LB; }
    __device__ __forceinline__ unsigned out_off(size_t xf) const { return in_off(xf); }
    __device__ __forceinline__ int out_sl() const { return LB; }
    // Block-interleaved intermediate (round 4; fft_tile_persist_kernel + fft_rows_persist_kernel only): element (q, j) of the
    // 2^LA x 2^LB matrix lives at  [q >> br][j >> bc][q & (2^br - 1)][j & (2^bc - 1)], br = log2(rows of a last-factor tile),
    // bc = log2(columns of a first-factor tile).  A wavefront's store then covers ONE contiguous run of 2^(br+bc) values
    // (1 KiB for 8 x 8 c64) instead of 2^br runs of 2^bc values a row apart, and the last factor reads its row tile as one
    // contiguous 2^(LB+br)-value stream.  br = 0 is the natural layout.  The store's thread part, in elements (the register part
    // out_index(0, u) << LB is the same in both layouts: its low br bits are zero).
    int blk_r = 0, blk_c = 0;
    static constexpr bool kBlockedOut = true;
    __device__ __forceinline__ unsigned out_lane(size_t xf, int tau) const
    {
        const unsigned j = in_off(xf);
        return (((unsigned)tau >> blk_r) << (LB + blk_r)) + ((j >> blk_c) << (blk_r + blk_c)) + (((unsigned)tau & ((1u << blk_r) - 1)) << blk_c) +
               (j & ((1u << blk_c) - 1));
    }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> LB, j = xf & ((size_t(1) << LB) - 1);
        const size_t elem = ((size_t)c << LB) + j;
        size_t ec = elem;
        if constexpr (PRE == PRE_CHIRP) ec = elem < n_in ? elem : n_in - 1;
        const cpx<T> *p = in + b * in_row() + ec;
        cpx<T> v = nt_in_pieces ? ld_stream(p) : *p;  // read once
        if (INVERSE) v.im = -v.im;  // ifft: conj on the way in (fft.rs:1163-1165)
        return pre(v, (unsigned)elem);
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> LB, j = xf & ((size_t(1) << LB) - 1);
        out[b * n + ((size_t)q << LB) + j] = v;  // read back by factor B: plain store (non-temporal measured no better)
    }
};

// factor B: unit xf = (b, K) with K in [0, 2^LA): row K of transform b (contiguous), output transposed.
//
// POST (round 3): a pointwise factor folded into the last factor's STORE:
//   POST_NONE      the plain transform (INVERSE: ifft's conj, * 1/n);
//   POST_BLUE_MID  Bluestein's a *= fft(b), then ifft's conj on the way in (fft.rs:1119-1121, 1163-1165): BlueFirstIO::store;
//   POST_BLUE_OUT  Bluestein's way out: conj, * 1/m, * chirp, INVERSE: conj, * 1/n_out; only the first n_out outputs exist,
//                  in rows of n_out values (the descriptor's bounds check drops the rest): BlueSecondIO::store.
enum : int { POST_NONE = 0, POST_BLUE_MID = 1, POST_BLUE_OUT = 2 };
template <typename T, bool INVERSE, int POST = POST_NONE>
struct BigRowsIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;  // lanes run over adjacent rows K: 64-byte segments for loads and stores
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;
    static constexpr int kPost = POST;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int LA, LB;
    int shift;  // L - LB
    int kbase;  // L - 1 - LA
    size_t n;
    T scale;    // 1 / (n as f32 as T), fft.rs:1167 (POST_BLUE_OUT: 1 / (m as f32 as T))
    bool nt;    // non-temporal stores: only when a workgroup's adjacent rows fill at least 64-byte segments
    bool nt_load = true;  // the intermediate is read once: streaming hint, unless it is meant to be served by the Infinity Cache
    const cpx<T> *__restrict__ post_tab = nullptr;  // POST_BLUE_MID: fft(b) (n entries); POST_BLUE_OUT: chirp (n_out entries)
    unsigned n_out = 0;                              // POST_BLUE_OUT: values per output row (= its stride)
    T scale_out = T(1);                              // POST_BLUE_OUT, INVERSE: 1 / (n_out as f32 as T)
    __device__ __forceinline__ TwSub tw_map(size_t xf) const { return TwSub{shift, (int)(xf & ((size_t(1) << LA) - 1)), kbase}; }
    static constexpr bool kConjIn = false, kConjScaleOut = INVERSE && POST == POST_NONE, kNtOut = true;
    static constexpr bool kTileInvariantTw = false;  // the table index carries the row's prefix K
    __device__ __forceinline__ bool nt_in() const { return nt_load; }
    __device__ __forceinline__ T out_scale() const { return scale; }
    __device__ __forceinline__ size_t in_row() const { return n; }
    __device__ __forceinline__ size_t out_row() const { return POST == POST_BLUE_OUT ? (size_t)n_out : n; }
    __device__ __forceinline__ size_t out_valid() const { return out_row(); }
    // the value stored at output index o (after the plain transform's own conj / scale, which POST modes do not use)
    __device__ __forceinline__ cpx<T> post(cpx<T> v, unsigned o) const
    {
        if constexpr (POST == POST_BLUE_MID) {
            cpx<T> w = cmul(v, post_tab[o]);
            w.im = -w.im;
            return w;
        } else if constexpr (POST == POST_BLUE_OUT) {
            v.im = -v.im;  // ifft: conj, * 1/m (fft.rs:1168-1172)
            v = mk<T>(v.re * scale, v.im * scale);
            cpx<T> r = cmul(v, post_tab[o < n_out ? o : n_out - 1]);
            if (INVERSE) {
                const T im = -r.im;
                r = mk<T>(r.re * scale_out, im * scale_out);
            }
            return r;
        } else {
            return v;
        }
    }
    __device__ __forceinline__ size_t xf_transform(size_t xf) const { return xf >> LA; }
    __device__ __forceinline__ unsigned in_off(size_t xf) const { return (unsigned)((xf & ((size_t(1) << LA) - 1)) << LB); }
    __device__ __forceinline__ int in_sl() const { return 0; }
    __device__ __forceinline__ unsigned out_off(size_t xf) const { return (unsigned)(xf & ((size_t(1) << LA) - 1)); }
    __device__ __forceinline__ int out_sl() const { return LA; }
    // the block-interleaved intermediate of BigColsIO (blk_r = 0: natural): thread part of the load of row K, elements tau + U
    // (U = in_index(0, u), a multiple of 2^blk_c: its part of the address is U << blk_r)
    int blk_r = 0, blk_c = 0;
    __device__ __forceinline__ unsigned in_lane(unsigned K, int tau) const
    {
        return ((K >> blk_r) << (LB + blk_r)) + (((unsigned)tau >> blk_c) << (blk_r + blk_c)) + ((K & ((1u << blk_r) - 1)) << blk_c) +
               ((unsigned)tau & ((1u << blk_c) - 1));
    }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> LA, K = xf & ((size_t(1) << LA) - 1);
        const cpx<T> *p = in + b * n + (K << LB) + (size_t)c;
        return nt_load ? ld_stream(p) : *p;  // the intermediate is read exactly once
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> LA, K = xf & ((size_t(1) << LA) - 1);
        const size_t o = ((size_t)q << LA) + K;
        if (kConjScaleOut) {  // conj, then scale (fft.rs:1168-1172)
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        if constexpr (POST == POST_BLUE_OUT) {
            if (o >= n_out) return;
        }
        v = post(v, (unsigned)o);
        if (nt) st_stream(out + b * out_row() + o, v);
        else out[b * out_row() + o] = v;
    }
};

// Middle factor of a three-factor split (n >= 2^21): global stages S .. S+LS-1.  The index bits at that point read
// [K : S bits][c : LS bits][j : JB bits] and the stage group maps them to [q : LS][K : S][j : JB] (same algebra as
// above with both a frequency prefix K and untouched low bits j).  Unit xf = (b, K, j); adjacent units are adjacent j.
template <typename T>
struct BigMidIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int S, LS, JB;
    int shift;  // L - LS
    int kbase;  // L - 1 - S
    size_t n;
    static constexpr bool nt = false;
    __device__ __forceinline__ TwSub tw_map(size_t xf) const
    {
        return TwSub{shift, (int)((xf >> JB) & ((size_t(1) << S) - 1)), kbase};
    }
    static constexpr bool kConjIn = false, kConjScaleOut = false, kNtOut = false;
    static constexpr bool kTileInvariantTw = false;
    // The table entries of a tile depend on its prefix K only, and the 2^JB / XPB tiles of one (transform, K) are consecutive:
    // the persistent kernel walks them in runs of `tpg` and keeps the run's 2^LS - 1 entries in LDS (kTileGroupTw).
    static constexpr bool kTileGroupTw = true;
    int tpg = 1;  // tiles per table load (divides 2^JB / XPB); set by launch_tile_persist
    __device__ __forceinline__ bool nt_in() const { return true; }
    __device__ __forceinline__ T out_scale() const { return T(1); }
    __device__ __forceinline__ size_t xf_transform(size_t xf) const { return xf >> (S + JB); }
    __device__ __forceinline__ unsigned in_off(size_t xf) const
    {
        const size_t K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return (unsigned)((K << (LS + JB)) + j);
    }
    __device__ __forceinline__ int in_sl() const { return JB; }
    __device__ __forceinline__ unsigned out_off(size_t xf) const
    {
        const size_t K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return (unsigned)((K << JB) + j);
    }
    __device__ __forceinline__ int out_sl() const { return S + JB; }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return ld_stream(in + b * n + (K << (LS + JB)) + ((size_t)c << JB) + j);
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        out[b * n + ((size_t)q << (S + JB)) + (K << JB) + j] = v;
    }
};

// ndfft's long strided axes in TWO column-tile passes (round 4): a dense [2^LT][2^I] block, every column a stand-alone 2^LT-point
// transform with ITS table T_{2^LT} (what fft_strided's gather / fft / scatter computes, ndfft.rs:89-98, 131-151).  Stages
// 0 .. L1-1 are BigColsIO above with LB = (LT - L1) + I untouched low bits and the table shift LT - L1; the remaining LS = LT - L1
// stages are this policy: BigMidIO's index algebra -- [K : S = L1][c : LS][j : JB = I] -> [q : LS][K][j], entries
// T[(idx_local << (LT - LS)) + (K << (LT - 1 - S - s_local))] -- as the LAST pass: ifft's conj, * 1/len on the way out
// (fft.rs:1168-1172) and streaming stores.  Three passes over the image for a 2-D transform instead of four (transpose, rows,
// transpose), 128-byte segments in each.
template <typename T, bool INVERSE>
struct AxisLastIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int S, LS, JB;
    int shift;  // LT - LS
    int kbase;  // LT - 1 - S
    size_t n;   // elements per block: 2^(LT + I)
    T scale;    // 1 / (len as f32 as T)
    static constexpr bool nt = true;
    __device__ __forceinline__ TwSub tw_map(size_t xf) const { return TwSub{shift, (int)((xf >> JB) & ((size_t(1) << S) - 1)), kbase}; }
    static constexpr bool kConjIn = false, kConjScaleOut = INVERSE, kNtOut = true;
    static constexpr bool kTileInvariantTw = false;
    static constexpr bool kTileGroupTw = true;
    int tpg = 1;
    bool nt_load = true;  // streaming loads of the intermediate; off (round 5) where it is meant to be served by the Infinity Cache
    __device__ __forceinline__ bool nt_in() const { return nt_load; }
    __device__ __forceinline__ T out_scale() const { return scale; }
    __device__ __forceinline__ size_t xf_transform(size_t xf) const { return xf >> (S + JB); }
    __device__ __forceinline__ unsigned in_off(size_t xf) const
    {
        const size_t K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return (unsigned)((K << (LS + JB)) + j);
    }
    __device__ __forceinline__ int in_sl() const { return JB; }
    __device__ __forceinline__ unsigned out_off(size_t xf) const
    {
        const size_t K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return (unsigned)((K << JB) + j);
    }
    __device__ __forceinline__ int out_sl() const { return S + JB; }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return ld_stream(in + b * n + (K << (LS + JB)) + ((size_t)c << JB) + j);
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        if (INVERSE) {  // conj, then scale (fft.rs:1168-1172)
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        st_stream(out + b * n + ((size_t)q << (S + JB)) + (K << JB) + j, v);
    }
};

// ---- persistent, prefetching form of the factor kernels --------------------------------------------------------------
// One workgroup per CU walks the tiles (a tile = XPB adjacent units = XPB adjacent columns / rows of ONE transform) with a
// stride of the grid.  The next tile's loads are issued into a second register set before the current tile is computed,
// so the CU always has a tile of HBM reads in flight while it runs butterflies, LDS exchanges and the previous tile's
// stores; with one 512-thread workgroup per CU the register budget is 256 per thread: both sets and the twiddles of a
// pass fit, nothing spills (the one-tile-per-workgroup form ran at 128 registers with 36..116 bytes of scratch per lane
// and waited on memory for 45 % of its wave cycles).  Addressing is "one buffer descriptor per transform + one per-lane
// byte offset + a wave-uniform offset per register" (local index i of a unit sits at  off(unit) + (i << sl)  inside its
// transform), so the 16 loads / stores of a set cost no 64-bit address arithmetic.
// Same butterflies, same table entries as fft_wg_kernel with the same policy: bit-identical results.
// ---- whole-value (16-byte) exchange for the persistent c64 kernels of 2^10-point factors ---------------------------------
// With the CU's LDS to itself (one workgroup per CU) a c64 tile fits as whole values: 8 x 1024 x 16 B = 128 KiB.  One
// scatter, one barrier, one gather per exchange (the split form: two of each plus two more barriers), no temporaries.
// Cells of 16 bytes, slot-minor:  cell(idx, slot) = (idx ^ bit2(idx)) * 8 + slot.
//   ds_write_b128 serves lanes in groups of 8 = the 8 slots of one thread index: 8 consecutive cells, all 32 banks once.
//   ds_read_b128 serves lanes in four groups of 16 -- {0-3, 12-15, 20-27}, ... = slots 0-3 of thread 4g and 4g+3 with
//   slots 4-7 of 4g+1 and 4g+2, or the complement: conflict-free iff the cells' low index bit differs between threads
//   4g / 4g+3 and between 4g+1 / 4g+2.  The middle pass gathers idx = [tau >> 2][c][tau & 3]: bit 0 is the thread's
//   bit 0 (bit 2 is constant per instruction); the last pass gathers idx = [tau][c]: bit 0 is constant, bit 2 is the
//   thread's bit 0 -- the XOR makes both shapes differ where they must.  (Derived for L = 10, RL = 4 only.)
__host__ __device__ constexpr int lds_cell_b128(int idx, int slot) { return ((idx ^ ((idx >> 2) & 1)) << 3) + slot; }

// compile-time proof that the base + constant forms used in wg_exchange_b128 are lds_cell_b128(index(tau, u), slot)
template <int P>
constexpr bool b128_forms_match()
{
    using Gs = WgGeom<10, 4, P>;
    using Gg = WgGeom<10, 4, P + 1>;
    for (int tau = 0; tau < 64; ++tau)
        for (int u = 0; u < 16; ++u) {
            const int slot = (tau + u) & 7;
            if ((((tau ^ ((tau >> 2) & 1)) << 3) + slot) + (Gs::out_index(0, u) << 3) != lds_cell_b128(Gs::out_index(tau, u), slot)) return false;
            int g = 0;
            if (P == 0) {
                const int it = ((tau >> 2) << 6) | (tau & 3);
                g = ((((u & 1) ? (it ^ 1) : it) << 3) + slot) + (u << 5);
            } else {
                const int t0 = tau & 1, gg = u >> 2, c = u & 3;
                g = (((((c & 1) ? (tau << 2) - t0 : (tau << 2) + t0)) << 3) + slot) + ((((64 * gg) << 2) + c) << 3);
            }
            if (g != lds_cell_b128(Gg::in_index(tau, u), slot)) return false;
        }
    return true;
}

template <typename T, int L, int RL, int P, int XPB>
__device__ __forceinline__ void wg_exchange_b128(cpx<T> *v, char *base, const int tau, const int slot)
{
    static_assert(b128_forms_match<P>(), "explicit-base addressing must equal lds_cell_b128(index(tau, u), slot)");
    static_assert(L == 10 && RL == 4 && XPB == 8 && sizeof(cpx<T>) == 16, "layout derived for 2^10-point c64 tiles of 8 units");
    using Gs = WgGeom<L, RL, P>;
    constexpr int R = 1 << RL;
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(base);
    if (P > 0) __syncthreads();  // every gather of the previous exchange is done
    // Every access below is "one per-thread base + a compile-time constant" (an immediate offset).  Written as
    // lds_cell_b128(index(tau, u), slot) the XOR hides that from the compiler, which then keeps one address register PER
    // REGISTER u alive across the tile loop (32 VGPRs for the two gathers) -- in the last-factor kernel, next to the 48
    // registers of resident twiddles, three of them spilled and every reload's s_waitcnt vmcnt(0) also waited for the
    // next tile's prefetch (found in the ISA; round 2).
    static_assert(Gs::TPT == 64 && (P == 0 || P == 1), "2^10 points, 16 per thread, three passes");
    {   // scatter: index = (bits >= 6 from u) | tau, and bit 2 of the index is bit 2 of tau
        cpx<T> *p = buf + (((tau ^ ((tau >> 2) & 1)) << 3) + slot);
#pragma unroll
        for (int u = 0; u < R; ++u) p[Gs::out_index(0, u) << 3] = v[u];
    }
    __syncthreads();
    if constexpr (P == 0) {
        // gather of the middle pass: index = [tau >> 2][u][tau & 3]; bit 2 of the index is bit 0 of u, the XOR flips bit 0
        // (a tau bit): one base for even u, one for odd u
        const int it = ((tau >> 2) << 6) | (tau & 3);
        const cpx<T> *pe = buf + ((it << 3) + slot), *po = buf + (((it ^ 1) << 3) + slot);
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = ((u & 1) ? po : pe)[u << 5];
    } else {
        // gather of the last pass: u = 4g + c, index = [tau + 64g][c]; bit 2 of the index is bit 0 of tau, the XOR turns c
        // into c ^ t0 = c + t0 (c even) or c - t0 (c odd): one base for even c, one for odd c
        const int t0 = tau & 1;
        const cpx<T> *pe = buf + ((((tau << 2) + t0) << 3) + slot), *po = buf + ((((tau << 2) - t0) << 3) + slot);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int g = u >> 2, c = u & 3;
            v[u] = ((c & 1) ? po : pe)[(((64 * g) << 2) + c) << 3];
        }
    }
}

// which exchange a persistent factor kernel uses, and how many bytes it needs
template <typename T, int L, int RL, int XPB, bool SPLIT>
struct TileExchange {
#ifdef KOFFT_TILE_SPLIT_ONLY
    static constexpr bool kWhole = false;
#else
    static constexpr bool kWhole = SPLIT && L == 10 && RL == 4 && XPB == 8 && sizeof(cpx<T>) == 16;
#endif
    static constexpr size_t bytes = kWhole ? (size_t)XPB * (1 << L) * sizeof(cpx<T>) : lds_wg_bytes<T, SPLIT, true, XPB>(1 << L);
    template <int P>
    __device__ static __forceinline__ void run(cpx<T> *v, char *base, const int tau, const int slot)
    {
        if constexpr (kWhole) wg_exchange_b128<T, L, RL, P, XPB>(v, base, tau, slot);
        else wg_exchange<T, L, RL, P, SPLIT, true, XPB>(v, base, tau, slot);
    }
};

template <class IO, class = void>
struct io_pre { static constexpr int value = 0; };
template <class IO>
struct io_pre<IO, decltype((void)IO::kPre)> { static constexpr int value = IO::kPre; };
template <class IO, class = void>
struct io_post { static constexpr int value = 0; };
template <class IO>
struct io_post<IO, decltype((void)IO::kPost)> { static constexpr int value = IO::kPost; };

template <class IO, class = void>
struct io_blocked_out { static constexpr bool value = false; };
template <class IO>
struct io_blocked_out<IO, decltype((void)IO::kBlockedOut)> { static constexpr bool value = IO::kBlockedOut; };
template <class IO, class = void>
struct io_tile_group_tw { static constexpr bool value = false; };
template <class IO>
struct io_tile_group_tw<IO, decltype((void)IO::kTileGroupTw)> { static constexpr bool value = IO::kTileGroupTw; };
// LDS copy of a sub-transform's table entries by stage: entry (1 << s) - 1 + kk holds the entry of stage s, group kk
// (reg_pass asks for (idx_local, s) with idx_local = kk << (L - 1 - s)).
template <int L>
struct TwLdsStage {
    __device__ __forceinline__ int operator()(int idx_local, int s_local) const { return (1 << s_local) - 1 + (idx_local >> (L - 1 - s_local)); }
};

template <typename T, int L, int RL, int BLOCK, class IO>
__global__ __launch_bounds__(BLOCK, (BLOCK >= 1024 ? 4 : 2)) void fft_tile_persist_kernel(const IO io, const cpx<T> *__restrict__ tw,
                                                                              const size_t ntiles)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 5, "pass count");
    constexpr bool SPLIT = IO::kSplitLds;
    static_assert(IO::kSlotMinor, "tile kernel: lanes run over adjacent units");
    static_assert((SPLIT ? sizeof(T) : sizeof(cpx<T>)) == 8, "slot-minor layout is built for 8-byte exchange elements");
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;
    constexpr int ES = (int)sizeof(cpx<T>);

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int tau_launch = tid / XPB, slot_launch = tid % XPB;
    const int tau = tau_launch, slot = slot_launch;

    constexpr bool GROUPED = io_tile_group_tw<IO>::value;
    size_t tpg = 1;
    if constexpr (GROUPED) tpg = (size_t)io.tpg;
    size_t tile = (size_t)blockIdx.x * tpg;
    if (tile >= ntiles) return;  // the whole workgroup leaves together
    // tile walk: runs of tpg consecutive tiles (one table load each), runs strided by the grid
    auto next_tile = [&](const size_t t) -> size_t {
        if constexpr (GROUPED) return (((t + 1) & (tpg - 1)) != 0) ? t + 1 : (t + 1 - tpg) + (size_t)gridDim.x * tpg;  // tpg: a power of two
        else return t + gridDim.x;
    };
    // First factor: the sub-transform's table entries T_n[idx << shift], idx < N/2, do not depend on the tile.  One LDS copy
    // per workgroup (8 KiB for 2^10 c64 points) replaces 27 global loads per thread and tile; the 8 lanes of a unit group
    // read one address (broadcast).  Behind the exchange region.
    const cpx<T> *tw_lds = nullptr;
    if constexpr (IO::kTileInvariantTw) {
        cpx<T> *tl = reinterpret_cast<cpx<T> *>(smem_raw + TileExchange<T, L, RL, XPB, SPLIT>::bytes);
        for (int i = tid; i < N / 2; i += BLOCK) tl[i] = tw[io.tw_map(0)(i, 0)];
        tw_lds = tl;
        __syncthreads();
    }
    auto fill_group_table = [&](const size_t t) {  // GROUPED: the 2^L - 1 entries of the run that starts at tile t
        if constexpr (GROUPED) {
            cpx<T> *tl = reinterpret_cast<cpx<T> *>(smem_raw + TileExchange<T, L, RL, XPB, SPLIT>::bytes);
            const auto map = io.tw_map(t * XPB);
            for (int e = tid; e < N - 1; e += BLOCK) {
                const int s = 31 - __builtin_clz(e + 1), kk = e + 1 - (1 << s);
                tl[e] = tw[map(kk << (L - 1 - s), s)];
            }
            tw_lds = tl;
            __syncthreads();
        }
    };
    fill_group_table(tile);
    auto compute = [&](auto pass, cpx<T> *v, const size_t xf) {
        constexpr int P = decltype(pass)::value;
        if constexpr (GROUPED) {
            using Gm = WgGeom<L, RL, P>;
#pragma unroll
            for (int g = 0; g < Gm::G; ++g)
                reg_pass<T, L, Gm::S0, Gm::Q, false>(&v[g * (1 << Gm::Q)], (tau + g * Gm::TPT) >> Gm::JB, tw_lds, TwLdsStage<L>{});
        } else if constexpr (IO::kTileInvariantTw) {
            using Gm = WgGeom<L, RL, P>;
#pragma unroll
            for (int g = 0; g < Gm::G; ++g)
                reg_pass<T, L, Gm::S0, Gm::Q, false>(&v[g * (1 << Gm::Q)], (tau + g * Gm::TPT) >> Gm::JB, tw_lds, TwPlain{});
        } else {
            wg_compute<T, L, RL, P>(v, io, tw, xf, tau);
        }
    };
    const int in_sl = io.in_sl(), out_sl = io.out_sl();
    const unsigned xf_bytes = (unsigned)(io.n * sizeof(cpx<T>));  // n <= 2^26 points: below 4 GiB
    // policies with a folded pointwise factor (PRE / POST) may read or write rows of another length than the transform's
    size_t in_row = io.n, out_row = io.n;
    unsigned in_bytes = xf_bytes, out_bytes = xf_bytes;
    if constexpr (io_pre<IO>::value != 0) {
        in_row = io.in_row();
        in_bytes = (unsigned)(in_row * sizeof(cpx<T>));
    }
    if constexpr (io_post<IO>::value != 0) {
        out_row = io.out_row();
        out_bytes = (unsigned)(io.out_valid() * sizeof(cpx<T>));
    }

    auto issue_loads = [&](cpx<T> *dst, const size_t t, const bool valid) {
        // an EMPTY descriptor when there is no next tile: the loads return zeros without touching memory, and no branch
        // sits between them (see fft_persist.hip.h)
        const size_t xf0 = (valid ? t : 0) * XPB;
        const rsrc_t d = make_rsrc(io.in + io.xf_transform(xf0) * in_row, valid ? in_bytes : 0u);
        const int lane = (int)((io.in_off(xf0 + slot) + ((unsigned)tau << in_sl)) * (unsigned)ES);
        if (io.nt_in()) {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_NT>(d, lane, (G0::in_index(0, u) << in_sl) * ES);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_DEFAULT>(d, lane, (G0::in_index(0, u) << in_sl) * ES);
        }
    };

    // (Measured and dropped: real and imaginary parts in two separate LDS cell arrays so that one barrier separates all
    // scatters from all gathers -- 2 barriers per exchange instead of 4.  The 32 gathers then in flight pushed the
    // kernels over 256 registers (52 / 140 bytes of scratch) and config 5 went 12.3 -> 14.7 ms.)
    // Two register sets, A and B, swap roles every tile (2x unrolled, no copies): while the tile held in one set is
    // computed and stored, the other set receives the next tile's loads.  (A copy `cur = nxt` after the stores would
    // put VALU writes to the store-data registers right behind 16-byte buffer stores that use an SGPR offset -- a
    // sequence the compiler does not pad, and one that stored the NEXT tile's values from a few lanes about once in
    // ten launches on gfx950.)
    // A folded pointwise factor on the loads (PRE_CHIRP, PRE_WINDOW) needs a table entry per value.  Requested where it is used -- at the top
    // of run_tile, BEHIND the next tile's prefetch -- waiting for it waited for the whole prefetch too (vmcnt counts in order): the windowed
    // first factor of rfft32 2^18 ran 259 us per 512 MiB chunk against the plain one's 186, no load overlapping any arithmetic (round 6).
    // Now the entries of tile t are requested BEFORE the prefetch of tile t + 1 is issued:
    //   mode 1 (registers allow R more values: f32 at 8 points per thread / 256-thread blocks): they stay in flight beside the prefetch,
    //          the product opens run_tile;
    //   mode 2 (1024 threads at 128 registers with 16 points, every f64 kernel): the product is done first -- the prefetch leaves one L2
    //          round trip later, and still runs under the tile's passes.
    constexpr int PRE_MODE = io_pre<IO>::value == 0 ? 0 : ((sizeof(T) == 4 && (R <= 8 || BLOCK <= 256)) ? 1 : 2);
    //   ... and in mode 1 they are usually RESIDENT: the walk advances by the grid size, the column tile of tile t is t mod (tiles per
    //   transform), so with a grid that is a multiple of the tiles per transform (a power of two against CUs x workgroups per CU: the rule
    //   unless the batch is tiny) a workgroup keeps its column tile for the whole launch, and the entries depend on the column and the row only.
    cpx<T> pw[PRE_MODE != 0 ? R : 1];
    bool pre_resident = false;
    if constexpr (PRE_MODE == 1) {
        const size_t tiles_per_xf = (size_t(1) << io.in_sl()) / XPB;
        pre_resident = tpg == 1 && tiles_per_xf != 0 && (gridDim.x % tiles_per_xf) == 0;  // (workgroup-uniform)
    }
    auto pre_request = [&](const size_t t) {
        if constexpr (PRE_MODE != 0) {
            int tau = tau_launch, slot = slot_launch;
            if constexpr (BLOCK >= 1024) asm volatile("" : "+v"(tau), "+v"(slot));
            const rsrc_t pd = io.pre_desc();
            const unsigned e0 = io.in_off(t * XPB + slot) + ((unsigned)tau << in_sl);
#pragma unroll
            for (int u = 0; u < R; ++u) pw[u] = io.pre_fetch(pd, e0 + ((unsigned)G0::in_index(0, u) << in_sl));
        }
    };
    auto pre_product = [&](cpx<T> *cur, const size_t t) {
        if (IO::kConjIn) {
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].im = -cur[u].im;  // ifft: conj on the way in (fft.rs:1163-1165)
        }
        if constexpr (PRE_MODE != 0) {
            int tau = tau_launch, slot = slot_launch;
            if constexpr (BLOCK >= 1024) asm volatile("" : "+v"(tau), "+v"(slot));
            const unsigned e0 = io.in_off(t * XPB + slot) + ((unsigned)tau << in_sl);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = io.pre_apply(cur[u], pw[u], e0 + ((unsigned)G0::in_index(0, u) << in_sl));
        }
    };
    auto run_tile = [&](cpx<T> *cur, const size_t t, const bool do_store) {
        // 1024-thread instances (128 registers per thread): opaque copies of the thread's coordinates, taken per tile.  Every LDS base
        // of the exchanges and the thread part of the store offset depend on the thread only, so the compiler computes them once and
        // keeps them for the whole launch -- a handful of registers the kernel does not have: 3 .. 7 of them were spilled, and every
        // reload from scratch is a vector-memory load whose s_waitcnt also waits for the prefetched next tile (round 5).
        int tau = tau_launch, slot = slot_launch;
        if constexpr (BLOCK >= 1024) asm volatile("" : "+v"(tau), "+v"(slot));
        const size_t xf = t * XPB + slot;
        if constexpr (PRE_MODE != 2) pre_product(cur, t);  // (mode 2: done before the prefetch was issued)
        compute(std::integral_constant<int, 0>{}, cur, xf);
        if constexpr (NP > 1) { TileExchange<T, L, RL, XPB, SPLIT>::template run<0>(cur, smem_raw, tau, slot); compute(std::integral_constant<int, 1>{}, cur, xf); }
        if constexpr (NP > 2) { TileExchange<T, L, RL, XPB, SPLIT>::template run<1>(cur, smem_raw, tau, slot); compute(std::integral_constant<int, 2>{}, cur, xf); }
        if constexpr (NP > 3) { TileExchange<T, L, RL, XPB, SPLIT>::template run<2>(cur, smem_raw, tau, slot); compute(std::integral_constant<int, 3>{}, cur, xf); }
        if constexpr (NP > 4) { TileExchange<T, L, RL, XPB, SPLIT>::template run<3>(cur, smem_raw, tau, slot); compute(std::integral_constant<int, 4>{}, cur, xf); }
        const size_t xf0 = t * XPB;
        const rsrc_t d = make_rsrc(io.out + io.xf_transform(xf0) * out_row, out_bytes);
        int lane;
        if constexpr (io_blocked_out<IO>::value) {
            static_assert(TPT >= 16, "the block's row bits are thread bits of the last pass's output index");
            lane = (int)(io.out_lane(xf, tau) * (unsigned)ES);
        } else {
            lane = (int)((io.out_off(xf) + ((unsigned)tau << out_sl)) * (unsigned)ES);
        }
        const T scale = io.out_scale();
        // 16-byte stores: the register offset goes into the VGPR offset, NOT into the SGPR offset field.  hipcc pads the
        // "store of more than 8 bytes, then a VALU write to its data registers" hazard only when the instruction has no
        // SGPR offset (it assumes the hazard away otherwise); on gfx950 the SGPR-offset form did store, about once in
        // ten launches, the NEXT value of the data registers' first half from the last lanes of each 16-lane row.
        constexpr bool nt_out = IO::kNtOut;  // tiles of this kernel are at least 64 bytes wide: streaming stores for the last factor
        if (!do_store) return;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            cpx<T> v = cur[u];
            if (IO::kConjScaleOut) {  // conj, then scale (fft.rs:1168-1172)
                const T im = -v.im;
                v = mk<T>(v.re * scale, im * scale);
            }
            const int off = lane + (GL::out_index(0, u) << out_sl) * ES;
            if constexpr (io_post<IO>::value != 0) v = io.post(v, (unsigned)off / (unsigned)ES);
            if (nt_out) buf_store_cpx_aux<T, AUX_NT>(v, d, off, 0);
            else buf_store_cpx_aux<T, AUX_DEFAULT>(v, d, off, 0);
        }
    };

    cpx<T> ra[R], rb[R];
    if (pre_resident) pre_request(tile);
    issue_loads(ra, tile, true);
#ifdef KOFFT_TILE_NO_MEM /* measurement only: the arithmetic and the exchanges without global traffic inside the loop */
#define KOFFT_TILE_PREFETCH(CUR, NXT)
#define KOFFT_TILE_RUN(CUR) run_tile(ra, tile, !more)
#else
#define KOFFT_TILE_PREFETCH(CUR, NXT) issue_loads(NXT, ntile, more);
#define KOFFT_TILE_RUN(CUR) run_tile(CUR, tile, true)
#endif
#define KOFFT_TILE_STEP(CUR, NXT)                                                                        \
    {                                                                                                    \
        const size_t ntile = next_tile(tile);                                                            \
        const bool more = ntile < ntiles; /* workgroup-uniform */                                        \
        if constexpr (PRE_MODE != 0) {                                                                   \
            if (!pre_resident) pre_request(tile);                                                        \
            if constexpr (PRE_MODE == 2) pre_product(CUR, tile);                                         \
            __builtin_amdgcn_sched_barrier(0); /* the table entries are requested ahead of the prefetch */ \
        }                                                                                                \
        KOFFT_TILE_PREFETCH(CUR, NXT)                                                                    \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ahead of the first use of CUR */         \
        KOFFT_TILE_RUN(CUR);                                                                             \
        if (!more) break;                                                                                \
        tile = ntile;                                                                                    \
        __syncthreads(); /* the last gathers of this tile are done before the next tile's first scatter */ \
        if constexpr (GROUPED) {                                                                         \
            if ((tile & (tpg - 1)) == 0) fill_group_table(tile); /* a new run: its table (uniform) */    \
        }                                                                                                \
    }
    do {  // first step peeled: see fft_rows_persist_kernel
        KOFFT_TILE_STEP(ra, rb)
        for (;;) {
            KOFFT_TILE_STEP(rb, ra)
            KOFFT_TILE_STEP(ra, rb)
        }
    } while (false);
#undef KOFFT_TILE_RUN
#undef KOFFT_TILE_STEP
#undef KOFFT_TILE_PREFETCH
}

// ---- last factor, persistent, twiddles RESIDENT per row tile ---------------------------------------------------------
// The last factor's table index carries the row's frequency prefix K (TwSub), so every tile of XPB rows needs its own
// 2^L - 1 entries per row: 46 global loads per thread and tile in the generic tile kernel, issued behind the next tile's
// prefetch -- and "vmcnt" counts in order, so waiting for a twiddle also waits for the whole prefetch.  But the entries
// depend on K only, not on the transform: a workgroup that stays on ONE row tile (K0 .. K0+XPB-1) and walks the
// transforms of the launch needs them once.  Passes before the last read them from an LDS copy ([entry][row] so the
// lanes of a row group read consecutive cells), the last pass -- every thread its own 2^Q - 1 per group -- keeps them in
// registers.  No global load but the data prefetch is left inside the loop.
// Work split: `groups` workgroups per row tile when the grid has at least one per tile (each takes every groups-th
// transform), otherwise every workgroup walks several row tiles and reloads the tables when it moves on.
// Table entry of a last-factor butterfly.  Row pairs (T = f32x2: rows 2 K and 2 K + 1, K = map.K the PAIR's index): each row its own entry
// of the Complex<f32> table -- (idx << shift) + (row << (kbase - s)) -- side by side.
template <typename T>
__device__ __forceinline__ cpx<T> tw_at(const cpx<T> *__restrict__ tw, const TwSub map, const int idx, const int s)
{
    if constexpr (std::is_same<T, f32x2>::value) {
        const cpx<float> *t = reinterpret_cast<const cpx<float> *>(tw);
        const int i0 = (idx << map.shift) + ((2 * map.K) << (map.kbase - s));
        const cpx<float> a = t[i0], b = t[i0 + (1 << (map.kbase - s))];
        f32x2 re, im;
        re.x = a.re; re.y = b.re;
        im.x = a.im; im.y = b.im;
        return mk<T>(re, im);
    } else {
        return tw[map(idx, s)];
    }
}

template <typename T, int Q, int STRIDE>
__device__ __forceinline__ void reg_pass_lds(cpx<T> *v, const cpx<T> *tl)
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const int pos = Q - 1 - t;
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const cpx<T> w = tl[((1 << t) - 1 + h) * STRIDE];
#pragma unroll
            for (int lo = 0; lo < (1 << pos); ++lo) {
                const int c = (h << (pos + 1)) | lo;
                bfly(v[c], v[c | (1 << pos)], w);
            }
        }
    }
}

template <typename T, int L, int S0, int Q, class TwMap>
__device__ __forceinline__ void load_pass_twiddles_map(cpx<T> *twr, const int k, const cpx<T> *__restrict__ tw, const TwMap map)
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const int idx = (k << (L - 1 - S0 - t)) + (bitrev(h, t) << (L - 1 - t));
            twr[(1 << t) - 1 + h] = tw_at<T>(tw, map, idx, S0 + t);
        }
    }
}

// LDS cells of the row-resident tables: passes 0 .. NP-2, pass P holding 2^(P*RL) groups of 2^RL - 1 entries
template <int L, int RL>
__host__ __device__ constexpr int rows_tw_entries()
{
    constexpr int NP = (L + RL - 1) / RL;
    int e = 0;
    for (int P = 0; P + 1 < NP; ++P) e += (1 << (P * RL)) * ((1 << RL) - 1);
    return e;
}

template <typename T, int L, int RL, int BLOCK, class IO>
__global__ __launch_bounds__(BLOCK, (BLOCK >= 1024 ? 4 : 2)) void fft_rows_persist_kernel(const IO io, const cpx<T> *__restrict__ tw,
                                                                                         const unsigned nb, const unsigned kt_base,
                                                                                         const unsigned kt_count)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 4, "pass count");
    constexpr bool SPLIT = IO::kSplitLds;
    static_assert((SPLIT ? sizeof(T) : sizeof(cpx<T>)) == 8, "slot-minor layout is built for 8-byte exchange elements");
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;
    constexpr int QL = GL::Q, GRP = GL::G;  // last pass: GRP groups of 2^QL registers per thread
    constexpr int ES = (int)sizeof(cpx<T>);
    constexpr int FULL = R - 1;  // entries per group of a full pass

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *tw_lds = reinterpret_cast<cpx<T> *>(smem_raw + TileExchange<T, L, RL, XPB, SPLIT>::bytes);
    const int tid = threadIdx.x;
    const int tau = tid / XPB;
    const int slot = tid % XPB;

    // ---- which row tile and which transforms this workgroup owns: the launch covers row tiles kt_base .. kt_base +
    // kt_count - 1 with `groups` = gridDim.x / kt_count workgroups each, a workgroup taking every groups-th transform.
    // (When a transform has more row tiles than the chip has CUs the host launches once per slice of row tiles: ONE row
    // tile per workgroup keeps table loads out of the pipelined loop -- see load_tables.)
    const unsigned G = gridDim.x, w = blockIdx.x;
    const unsigned groups = G / kt_count;
    if (groups == 0 || w >= kt_count * groups) return;
    const unsigned kt = kt_base + w % kt_count;
    const unsigned b_first = w / kt_count;
    if (b_first >= nb) return;
    unsigned b = b_first;

    const unsigned xf_bytes = (unsigned)(io.n * sizeof(cpx<T>));
    size_t out_row = io.n;
    unsigned out_bytes = xf_bytes;
    if constexpr (io_post<IO>::value != 0) {  // a folded pointwise factor may write rows of another length
        out_row = io.out_row();
        out_bytes = (unsigned)(io.out_valid() * sizeof(cpx<T>));
    }
    const int out_sl = io.LA;
    auto issue_loads = [&](cpx<T> *dst, const unsigned tkt, const unsigned tb, const bool valid) {
        // (walking the transforms last to first, so that the most recently written part of the intermediate is read first,
        // measured no gain from the Infinity Cache: 12.4 vs 12.1 ms on config 5)
        const rsrc_t d = make_rsrc(io.in + (size_t)(valid ? tb : 0) * io.n, valid ? xf_bytes : 0u);
        static_assert(G0::JB >= 4, "the block's column bits are thread bits of the first pass's input index");
        const int lane = (int)(io.in_lane((valid ? tkt : 0) * XPB + slot, tau) * (unsigned)ES);
        const int bsh = io.blk_r;  // register part: in_index(0, u) << blk_r elements (wave-uniform)
        if (io.nt_in()) {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_NT>(d, lane, (G0::in_index(0, u) << bsh) * ES);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_DEFAULT>(d, lane, (G0::in_index(0, u) << bsh) * ES);
        }
    };

    // one group's 2^RL - 1 entries of a staged pass, straight from the table into the LDS cells [entry][row]
    auto stage_group = [&](auto s0, cpx<T> *cells, const int k, const TwSub map) {
        constexpr int S0 = decltype(s0)::value;
#pragma unroll
        for (int t = 0; t < RL; ++t) {
#pragma unroll
            for (int h = 0; h < (1 << t); ++h) {
                const int idx = (k << (L - 1 - S0 - t)) + (bitrev(h, t) << (L - 1 - t));
                cells[((1 << t) - 1 + h) * XPB] = tw_at<T>(tw, map, idx, S0 + t);
            }
        }
    };
    cpx<T> twl[GRP * ((1 << QL) - 1)];  // the last pass's twiddles of this thread, for the current row tile
    int lds_base[NP > 1 ? NP - 1 : 1];  // this thread's group of entries in each staged pass (cells)
    auto load_tables = [&](const unsigned tkt) {
        const TwSub map = io.tw_map((size_t)tkt * XPB + slot);  // K = tkt * XPB + slot
        int off = 0;
        // staged passes: one representative thread per (group k, row) fetches the group's entries
#define KOFFT_ROWS_STAGE(P)                                                                                           \
        if constexpr (P + 1 < NP) {                                                                                   \
            using Gm = WgGeom<L, RL, P>;                                                                              \
            static_assert(Gm::G == 1 && Gm::Q == RL, "staged passes are full passes");                                \
            const int k = tau >> Gm::JB;                                                                              \
            lds_base[P] = (off + k * FULL) * XPB + slot;                                                              \
            if ((tau & ((1 << Gm::JB) - 1)) == 0) stage_group(std::integral_constant<int, Gm::S0>{}, tw_lds + lds_base[P], k, map);                  \
            off += (1 << Gm::S0) * FULL;                                                                              \
        }
        KOFFT_ROWS_STAGE(0)
        KOFFT_ROWS_STAGE(1)
        KOFFT_ROWS_STAGE(2)
#undef KOFFT_ROWS_STAGE
#pragma unroll
        for (int g = 0; g < GRP; ++g)
            load_pass_twiddles_map<T, L, GL::S0, QL>(twl + g * ((1 << QL) - 1), (tau + g * TPT) >> GL::JB, tw, map);
        // The table loads must have LANDED before the tile loop is (re-)entered.  Left pending, they are "maybe in flight"
        // at the loop header on one incoming path, and the compiler then guards every use of twl[] inside the loop with
        // s_waitcnt vmcnt(4) .. vmcnt(0) -- which, vmcnt counting in order, also waits for the just-issued prefetch of
        // the next tile before the current tile may finish and store (found in the ISA; it cost the overlap).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
        __syncthreads();
    };

    auto run_tile = [&](cpx<T> *cur, const unsigned tkt, const unsigned tb, const bool do_store) {
        // passes 0 .. NP-2 from the LDS tables, last pass from registers
        reg_pass_lds<T, RL, XPB>(cur, tw_lds + lds_base[0]);
        if constexpr (NP > 2) {
            TileExchange<T, L, RL, XPB, SPLIT>::template run<0>(cur, smem_raw, tau, slot);
            reg_pass_lds<T, RL, XPB>(cur, tw_lds + lds_base[NP > 2 ? 1 : 0]);
        }
        if constexpr (NP > 3) {
            TileExchange<T, L, RL, XPB, SPLIT>::template run<1>(cur, smem_raw, tau, slot);
            reg_pass_lds<T, RL, XPB>(cur, tw_lds + lds_base[NP > 3 ? 2 : 0]);
        }
        TileExchange<T, L, RL, XPB, SPLIT>::template run<NP - 2>(cur, smem_raw, tau, slot);
#pragma unroll
        for (int g = 0; g < GRP; ++g) reg_pass_r<T, QL>(cur + g * (1 << QL), twl + g * ((1 << QL) - 1));
        const rsrc_t d = make_rsrc(io.out + (size_t)tb * out_row, out_bytes);
        int lane = (int)((tkt * XPB + slot + ((unsigned)tau << out_sl)) * (unsigned)ES);
        // A folded pointwise factor reads its table at the store's own element index: with the row tile fixed per workgroup those R
        // 64-bit addresses do not depend on the transform, the compiler keeps all of them for the whole launch (32 registers) and the
        // kernel spilled 14 .. 28.  An opaque copy per transform: recomputed, a few VALU operations each (round 5).
        if constexpr (io_post<IO>::value != 0) asm volatile("" : "+v"(lane));
        const T scale = io.out_scale();
        if (!do_store) return;
        // (offsets in the VGPR field: see fft_tile_persist_kernel)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            cpx<T> v = cur[u];
            if (IO::kConjScaleOut) {  // conj, then scale (fft.rs:1168-1172)
                const T im = -v.im;
                v = mk<T>(v.re * scale, im * scale);
            }
            const int off = lane + (GL::out_index(0, u) << out_sl) * ES;
            if constexpr (io_post<IO>::value != 0) v = io.post(v, (unsigned)off / (unsigned)ES);
            buf_store_cpx_aux<T, AUX_NT>(v, d, off, 0);
        }
    };

    cpx<T> ra[R], rb[R];
    load_tables(kt);
    issue_loads(ra, kt, b, true);
#define KOFFT_ROWS_STEP(CUR, NXT)                                                                          \
    {                                                                                                      \
        const unsigned nbb = b + groups;                                                                   \
        const bool more = nbb < nb; /* workgroup-uniform */                                                \
        issue_loads(NXT, kt, nbb, more);                                                                   \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ahead of the first use of CUR */           \
        run_tile(CUR, kt, b, true);                                                                        \
        if (!more) break;                                                                                  \
        b = nbb;                                                                                           \
        __syncthreads(); /* this tile's last gathers and table reads are done */                          \
    }
    // The first step is peeled off the loop so that both predecessors of the loop header carry the same pending
    // operations (loads of one set, stores of the other): with the first step inside, the header merges "no stores yet"
    // with the back edge and the compiler's conservative vmcnt makes every tile wait for the previous tile's stores.
    do {
        KOFFT_ROWS_STEP(ra, rb)
        for (;;) {
            KOFFT_ROWS_STEP(rb, ra)
            KOFFT_ROWS_STEP(ra, rb)
        }
    } while (false);
#undef KOFFT_ROWS_STEP
}

// ndfft (ndfft.rs:74-155): FftImpl::fft_strided over every line of one axis.  Unit xf is one line:
// element i lives at  (xf / inner) * outer_stride + (xf % inner) + i * stride.  Adjacent lines are adjacent in memory,
// so lanes run over lines first (kSlotMinor), as for the column factor above.  Each line is a stand-alone transform
// with its own table T_len (TwPlain), exactly what fft_strided's gather / fft / scatter computes.
template <typename T, bool INVERSE>
struct StridedIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = false;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = 1;
    static constexpr bool kLen1 = false;  // fft_axis_dev returns for an axis of one point
    cpx<T> *__restrict__ data;  // in place
    size_t inner, outer_stride, stride;
    T scale;  // 1 / (len as f32 as T)
    __device__ __forceinline__ TwPlain tw_map(size_t) const { return {}; }
    __device__ __forceinline__ size_t base(size_t xf) const { return (xf / inner) * outer_stride + (xf % inner); }
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const
    {
        cpx<T> v = data[base(xf) + (size_t)i * stride];
        if (INVERSE) v.im = -v.im;
        return v;
    }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (INVERSE) {
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        data[base(xf) + (size_t)o * stride] = v;
    }
};

// ---- tiled transpose for the long strided axes of ndfft ------------------------------------------------------------
// dst[b][c][r] = src[b][r][c] for r < rows, c < cols: 32 x 32 tiles through LDS (row padded by one cell), both sides
// coalesced.  Leading dimensions and per-batch strides in elements.  The tile grid is flattened onto grid.x
// (tile = blockIdx.x: column tile fastest): either side of a panel can be millions of elements long (a 3 x 2^22 transform, an
// axis of 2^22 points), far beyond the 65535 of grid.y, while the panel as a whole is at most 2^27 elements.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const cpx<T> *__restrict__ src, cpx<T> *__restrict__ dst, const size_t rows,
                                                        const size_t cols, const size_t src_ld, const size_t dst_ld,
                                                        const size_t src_bs, const size_t dst_bs, const unsigned col_tiles)
{
    __shared__ cpx<T> tile[32][33];
    const size_t b = blockIdx.y;
    const cpx<T> *s = src + b * src_bs;
    cpx<T> *d = dst + b * dst_bs;
    const size_t c0 = (size_t)(blockIdx.x % col_tiles) * 32, r0 = (size_t)(blockIdx.x / col_tiles) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const size_t r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = ld_stream(s + r * src_ld + c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const size_t c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) st_stream(d + c * dst_ld + r, tile[tx][ty + k]);
    }
}

// ---- Bluestein arm for non-power-of-two lengths (fft.rs:1088-1132) --------------------------------
// a = x * chirp (zero-padded to m = next_pow2(2n-1)); fft_m; a *= fft(b); conj; fft_m; conj; * 1/m; out = a * chirp.
// The three pointwise steps below use Complex::mul's un-fused form; the two m-point transforms are the ordinary
// power-of-two kernels.  INVERSE folds ifft's conj-in / conj-scale-out (fft.rs:1163-1172) around the same steps.
// Fused forms for m <= one workgroup's transform (n <= 8192 f32 / 4096 f64): the three pointwise steps ride on the two
// m-point transforms' loads and stores -- x is read once (n values), the padded intermediate is written and read once
// (m values), the result is written once (n values): 2n + 2m complex values of traffic instead of 2n + 8m.
// Same operations per element in the same order as the kernels below.
template <typename T, bool INVERSE>
struct BlueFirstIO : PlainTw {  // load: a = x * chirp, zero-padded to m; store: (a * fft(b)), conjugated (ifft's way in)
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ tmp;
    const cpx<T> *__restrict__ chirp;
    const cpx<T> *__restrict__ bfft;
    int n, m;
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const
    {
        const int ic = i < n ? i : n - 1;  // branch-free: clamp the address, select the value
        cpx<T> x = ld_stream(in + xf * (size_t)n + ic);
        if (INVERSE) x.im = -x.im;
        const cpx<T> v = cmul(x, chirp[ic]);
        return i < n ? v : mk<T>(T(0), T(0));
    }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        cpx<T> w = cmul(v, bfft[o]);
        w.im = -w.im;
        tmp[xf * (size_t)m + o] = w;  // read back by the second transform: plain store
    }
};
template <typename T, bool INVERSE>
struct BlueSecondIO : PlainTw {  // store: conj, * 1/m (ifft's way out), * chirp; only the first n values exist
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    const cpx<T> *__restrict__ tmp;
    cpx<T> *__restrict__ out;
    const cpx<T> *__restrict__ chirp;
    int n, m;
    T scale_m, scale_n;
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const { return ld_stream(tmp + xf * (size_t)m + i); }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (o >= n) return;
        v.im = -v.im;
        v = mk<T>(v.re * scale_m, v.im * scale_m);
        cpx<T> r = cmul(v, chirp[o]);
        if (INVERSE) {
            const T im = -r.im;
            r = mk<T>(r.re * scale_n, im * scale_n);
        }
        st_stream(out + xf * (size_t)n + o, r);
    }
};

// ---- the whole Bluestein arm in ONE kernel (m = 2^L up to the single-workgroup sizes) -------------------------------------
// fft.rs:1088-1132 for one transform per TPT threads: a = x * chirp (zero-padded to m), fft(a), a *= fft(b), ifft(a) = conj,
// fft, conj, * 1/m, out = a * chirp -- the m-point intermediate never leaves the CU (registers + the LDS exchange buffer), so
// HBM sees n points in and n points out instead of n + 3m + n (BlueFirstIO / BlueSecondIO: two launches through a scratch).
// Between the two transforms the values go through the exchange buffer once more: the first transform leaves element
// out_index(tau, u) in register u, the second one wants in_index(tau, u).  Same butterflies, same table entries, same
// pointwise expressions as the two policies above: bit-identical results.
template <typename T, int L, int RL, int BLOCK, bool INVERSE>
__global__ __launch_bounds__(BLOCK) void bluestein_wg_kernel(const cpx<T> *__restrict__ in, cpx<T> *__restrict__ out,
                                                             const cpx<T> *__restrict__ chirp, const cpx<T> *__restrict__ bfft,
                                                             const cpx<T> *__restrict__ tw, const int n, const T scale_m, const T scale_n,
                                                             const size_t batch)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 5, "pass count");
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int tau = tid % TPT, slot = tid / TPT;
    const size_t xf = (size_t)blockIdx.x * XPB + slot;
    const bool active = xf < batch;
    const PlainTw io{};

    // m >= 2n (see bluestein_persist_kernel): registers R/2 .. R-1 of the first pass hold elements m/2 .. m-1, zeros whatever n is
    static_assert(G0::in_index(0, R / 2) == N / 2, "pass 0: the register number is the index's top bits");
    cpx<T> v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = mk<T>(T(0), T(0));
    if (active) {  // one branch around all loads
        const cpx<T> *row = in + xf * (size_t)n;
#pragma unroll
        for (int u = 0; u < R / 2; ++u) {
            const int i = G0::in_index(tau, u);
            const int ic = i < n ? i : n - 1;  // branch-free: clamp the address, select the value
            cpx<T> x = ld_stream(row + ic);
            if (INVERSE) x.im = -x.im;  // ifft: conj on the way in (fft.rs:1163-1165)
            const cpx<T> a = cmul(x, chirp[ic]);
            v[u] = i < n ? a : mk<T>(T(0), T(0));
        }
    }
    auto transform = [&]() {
        wg_compute<T, L, RL, 0>(v, io, tw, xf, tau);
        if constexpr (NP > 1) { wg_exchange<T, L, RL, 0, false, false, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 1>(v, io, tw, xf, tau); }
        if constexpr (NP > 2) { wg_exchange<T, L, RL, 1, false, false, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 2>(v, io, tw, xf, tau); }
        if constexpr (NP > 3) { wg_exchange<T, L, RL, 2, false, false, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 3>(v, io, tw, xf, tau); }
        if constexpr (NP > 4) { wg_exchange<T, L, RL, 3, false, false, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 4>(v, io, tw, xf, tau); }
    };
    transform();
    // a *= fft(b) (fft.rs:1119-1121), then ifft's conj on the way in; natural order -> the first pass's register layout: the same element
    // stays in the same thread (bluestein_persist_kernel), register u' = out_index(0, u) / TPT -- a renaming (round 2 went through LDS)
    {
        cpx<T> w[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            static_assert(GL::out_index(0, 1) >= TPT && G0::in_index(0, 1) == TPT, "register bits above the thread bits");
            cpx<T> p = cmul(v[u], bfft[GL::out_index(tau, u)]);
            p.im = -p.im;
            w[GL::out_index(0, u) / TPT] = p;
        }
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = w[u];
    }
    __syncthreads();  // the last gathers of the first transform are done before the second transform's first scatter
    transform();
    if (active) {
        cpx<T> *orow = out + xf * (size_t)n;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (GL::out_index(0, u) >= N / 2) continue;  // outputs m/2 .. m-1 are never stored (n <= m/2)
            const int o = GL::out_index(tau, u);
            if (o < n) {
                cpx<T> a = v[u];
                a.im = -a.im;  // ifft: conj, * 1/m (fft.rs:1168-1172)
                a = mk<T>(a.re * scale_m, a.im * scale_m);
                cpx<T> r = cmul(a, chirp[o]);
                if (INVERSE) {
                    const T im = -r.im;
                    r = mk<T>(r.re * scale_n, im * scale_n);
                }
                st_stream(orow + o, r);
            }
        }
    }
}

// ---- the same arm, persistent (round 4) -------------------------------------------------------------------------------------
// bluestein_wg_kernel starts a workgroup per XPB transforms, and each of them reads its table entries again: per transform and
// thread 2 x (R - 1) x (NP - 1) twiddles (gathers with strides of 2^s entries: up to 64 cache lines per instruction), R entries of
// fft(b) and 2 x R of the chirp -- at m = 2048 five times the bytes of the transform's own input, all through the vector L1, and
// every pass waits for them (SQ_WAIT_ANY 65 % of the wave cycles, VALU 44 % busy: profiles/r04_bluestein1000.md as first collected).
// Here the grid is sized to the chip, a workgroup loops over transforms, and what depends on the thread but not on the transform is
// read ONCE: the twiddles of passes 1.. and fft(b) live in registers (pass 0's table indices are compile-time constants: scalar
// loads), the next transform's input is in flight while this one is computed.  The chirp entries stay loads (coalesced, 8 KiB,
// L1 hits): keeping them would cost 2 x R more registers and the second wavefront per SIMD.
// Same butterflies, same table entries, same pointwise expressions: bit-identical to bluestein_wg_kernel.
// Where the n input values of transform xf come from: rows of n complex values (fft / ifft), or STFT frames of a real signal
// (stft.rs:91-103 for a window length that is not a power of two: frame[i] = (signal[start + i] * window[i], 0), zeros past the end --
// what stft_frame_kernel writes out for the composed route, here on this kernel's loads: one pass over HBM instead of three).
template <typename T, bool INVERSE>
struct BlueRowsSrc {
    using Raw = cpx<T>;
    const cpx<T> *__restrict__ in;
    int n;
    size_t batch;
    __device__ __forceinline__ Raw fetch(size_t xf, int i) const
    {
        const size_t xc = xf < batch ? xf : batch - 1;  // past the end: a valid address, the values are never stored
        return ld_stream(in + xc * (size_t)n + (i < n ? i : n - 1));
    }
    __device__ __forceinline__ cpx<T> finish(Raw x, size_t, int) const
    {
        if (INVERSE) x.im = -x.im;  // ifft: conj on the way in (fft.rs:1163-1165)
        return x;
    }
};
struct BlueStftSrc {
    using Raw = float;
    const float *__restrict__ signal;
    const float *__restrict__ window;
    size_t len, hop, start0;
    int n;
    __device__ __forceinline__ Raw fetch(size_t xf, int i) const
    {
        const size_t pos = start0 + xf * hop + (size_t)(i < n ? i : n - 1);
        return pos < len ? signal[pos] : 0.0f;
    }
    __device__ __forceinline__ cpx<float> finish(Raw x, size_t xf, int i) const
    {
        const int ic = i < n ? i : n - 1;
        const bool in = start0 + xf * hop + (size_t)ic < len;
        return mk<float>(in ? x * window[ic] : 0.0f, 0.0f);  // stft.rs:95-100
    }
};

#ifndef KOFFT_BLUE_CHIRP_REG
#define KOFFT_BLUE_CHIRP_REG 1 /* 0: measurement builds -- the input product's chirp entries read per transform, as in rounds 4-5 */
#endif
template <typename T>
__host__ __device__ constexpr bool bluestein_chirp_resident(int L, int wg_per_cu)
{
    // f32, two workgroups of 256 threads per CU (256 registers each: m = 2048 / 4096 use 208 / 194 + 16)
    return KOFFT_BLUE_CHIRP_REG && sizeof(T) == 4 && wg_per_cu <= 2 && (L == 11 || L == 12);
}
template <typename T, int L, int RL, int BLOCK, int WG_PER_CU, bool INVERSE, class SRC = BlueRowsSrc<T, INVERSE>>
__global__ __launch_bounds__(BLOCK, WG_PER_CU * BLOCK / 256 /* wavefronts per SIMD */) void bluestein_persist_kernel(const SRC src, cpx<T> *__restrict__ out,
                                                                                        const cpx<T> *__restrict__ chirp,
                                                                                        const cpx<T> *__restrict__ bfft,
                                                                                        const cpx<T> *__restrict__ tw, const int n, const T scale_m,
                                                                                        const T scale_n, const size_t batch)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 4, "pass count");
    constexpr bool WAVE = TPT <= 64;  // a transform inside one wavefront: its exchanges need no s_barrier
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int tau = tid % TPT, slot = tid / TPT;
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)slot * lds_elems(N);

    // per-thread invariants
    cpx<T> tw1[R - 1], tw2[NP >= 3 ? R - 1 : 1], tw3[NP >= 4 ? R - 1 : 1];
    persist_load_tw<T, L, RL, 1>(tw1, tau, tw);
    if constexpr (NP >= 3) persist_load_tw<T, L, RL, 2>(tw2, tau, tw);
    if constexpr (NP >= 4) persist_load_tw<T, L, RL, 3>(tw3, tau, tw);
    constexpr bool BF_REG = RL <= 3 && NP <= 3;  // otherwise fft(b)'s entries are read while the first transform's last pass computes
    cpx<T> bf[BF_REG ? R : 1];
    if constexpr (BF_REG) {
#pragma unroll
        for (int u = 0; u < R; ++u) bf[u] = bfft[GL::out_index(tau, u)];
    }
    const int sc = lds_pad(tau);
    const int g1 = lds_pad(WgGeom<L, RL, 1>::in_index(tau, 0));
    const int g2 = NP >= 3 ? lds_pad(WgGeom<L, RL, (NP >= 3 ? 2 : 0)>::in_index(tau, 0)) : 0;
    const int g3 = NP >= 4 ? lds_pad(WgGeom<L, RL, (NP >= 4 ? 3 : 0)>::in_index(tau, 0)) : 0;
    // The chirp entries of the INPUT product (R / 2 of them: the upper half of the padded input is zeros) depend on the thread only.  Read per
    // transform at the top of its work they cost an exposed L2 round trip each time; where the register budget allows (KOFFT_BLUE_CHIRP_REG)
    // they are read once (round 6).
    constexpr bool CH_REG = bluestein_chirp_resident<T>(L, WG_PER_CU);
    cpx<T> chin[CH_REG ? R / 2 : 1];
    if constexpr (CH_REG) {
#pragma unroll
        for (int u = 0; u < R / 2; ++u) {
            const int i = G0::in_index(tau, u);
            chin[u] = chirp[i < n ? i : n - 1];
        }
    }

    // m = (2n - 1).next_power_of_two() >= 2n: elements m/2 .. m-1 of the padded input are zeros and outputs m/2 .. m-1 are never stored, for
    // every n of this m -- half of the loads, chirp entries and stores are decided at compile time (RH registers of input in flight)
    constexpr int RH = R / 2;
    static_assert(G0::in_index(0, RH) == N / 2, "pass 0: the register number is the index's top bits");
    using Raw = typename SRC::Raw;
    auto fetch = [&](Raw *raw, size_t xf) {
#pragma unroll
        for (int u = 0; u < RH; ++u) raw[u] = src.fetch(xf, G0::in_index(tau, u));
    };
    auto transform = [&](cpx<T> *v, auto &&before_last) {  // before_last(): table loads that ride under the last pass's butterflies
        persist_compute_p0<T, L, RL>(v, tw);
        exchange_sync<WAVE>();  // the buffer's previous gathers are done
        persist_lds_scatter<T, L, RL, 0>(v, buf, sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 1>(v, buf, g1);
        if constexpr (NP == 2) before_last();
        persist_compute<T, L, RL, 1>(v, tw1);
        if constexpr (NP >= 3) {
            exchange_sync<WAVE>();
            persist_lds_scatter<T, L, RL, 1>(v, buf, sc);
            exchange_sync<WAVE>();
            persist_lds_gather<T, L, RL, 2>(v, buf, g2);
            if constexpr (NP == 3) before_last();
            persist_compute<T, L, RL, 2>(v, tw2);
        }
        if constexpr (NP >= 4) {
            exchange_sync<WAVE>();
            persist_lds_scatter<T, L, RL, 2>(v, buf, sc);
            exchange_sync<WAVE>();
            persist_lds_gather<T, L, RL, 3>(v, buf, g3);
            before_last();
            persist_compute<T, L, RL, 3>(v, tw3);
        }
    };

    const size_t stride = (size_t)gridDim.x * XPB;
    size_t xf = (size_t)blockIdx.x * XPB + slot;
    Raw raw[RH];
    fetch(raw, xf);
    for (size_t base = (size_t)blockIdx.x * XPB; base < batch; base += stride, xf += stride) {
        // the table addresses below depend on the thread only: left visible, the compiler hoists the LOADS out of the transform loop
        // (3 x R values in registers, spilled); an opaque copy of tau per transform keeps them loads
        int tau_t = tau;
        asm volatile("" : "+v"(tau_t));
        cpx<T> v[R];
#pragma unroll
        for (int u = 0; u < RH; ++u) {
            const int i = G0::in_index(tau_t, u);
            const cpx<T> x = src.finish(raw[u], xf, i);
            cpx<T> ce;
            if constexpr (CH_REG) ce = chin[u];
            else ce = chirp[i < n ? i : n - 1];
            const cpx<T> a = cmul(x, ce);
            v[u] = i < n ? a : mk<T>(T(0), T(0));
        }
#pragma unroll
        for (int u = RH; u < R; ++u) v[u] = mk<T>(T(0), T(0));
        fetch(raw, xf + stride);
        cpx<T> tab[R];  // fft(b)'s entries, then the chirp's for the output
        transform(v, [&]() {
            if constexpr (!BF_REG) {
#pragma unroll
                for (int u = 0; u < R; ++u) tab[u] = bfft[GL::out_index(tau_t, u)];
            }
        });
        // a *= fft(b) (fft.rs:1119-1121), ifft's conj on the way in; natural order -> the first pass's register layout
#pragma unroll
        for (int u = 0; u < R; ++u) {
            cpx<T> w = cmul(v[u], BF_REG ? bf[BF_REG ? u : 0] : tab[u]);
            w.im = -w.im;
            v[u] = w;
        }
        // natural order -> the first pass's register layout.  Both are "RL index bits from the register number, the rest from the
        // thread": out_index(tau, u) = [bitrev(c) | g | tau] with u = (g, c), in_index(tau, u') = [u' | tau] -- the same element sits
        // in the same THREAD, register u' = bitrev(c) << (RL - Q) | g: a renaming, no exchange (bluestein_wg_kernel goes through LDS)
        {
            cpx<T> w[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                static_assert(GL::out_index(0, 1) >= TPT && G0::in_index(0, 1) == TPT, "register bits above the thread bits");
                w[GL::out_index(0, u) / TPT] = v[u];
            }
#pragma unroll
            for (int u = 0; u < R; ++u) v[u] = w[G0::in_index(0, u) / TPT];
        }
        // (without the exchange's barrier the two transforms' passes interleave and registers spill; pinning the values here: m = 2048 0.65 -> 0.58 ms)
#pragma unroll
        for (int u = 0; u < R; ++u) asm volatile("" : "+v"(v[u].re), "+v"(v[u].im));
        transform(v, [&]() {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (GL::out_index(0, u) >= N / 2) continue;
                const int o = GL::out_index(tau_t, u);
                tab[u] = chirp[o < n ? o : n - 1];
            }
        });
        if (xf < batch) {
            cpx<T> *orow = out + xf * (size_t)n;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (GL::out_index(0, u) >= N / 2) continue;
                const int o = GL::out_index(tau, u);
                if (o < n) {
                    cpx<T> a = v[u];
                    a.im = -a.im;  // ifft: conj, * 1/m (fft.rs:1168-1172)
                    a = mk<T>(a.re * scale_m, a.im * scale_m);
                    cpx<T> r = cmul(a, tab[u]);
                    if (INVERSE) {
                        const T im = -r.im;
                        r = mk<T>(r.re * scale_n, im * scale_n);
                    }
                    st_stream(orow + o, r);
                }
            }
        }
    }
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void bluestein_pre_kernel(const cpx<T> *__restrict__ in, cpx<T> *__restrict__ a,
                                                            const cpx<T> *__restrict__ chirp, const size_t n, const size_t m,
                                                            const size_t total /* batch * m */)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx / m, i = idx % m;
    cpx<T> v = mk<T>(T(0), T(0));
    if (i < n) {
        cpx<T> x = in[b * n + i];
        if (INVERSE) x.im = -x.im;
        v = cmul(x, chirp[i]);
    }
    a[idx] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void bluestein_mid_kernel(cpx<T> *__restrict__ a, const cpx<T> *__restrict__ bfft, const size_t m,
                                                            const size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    cpx<T> v = cmul(a[idx], bfft[idx % m]);
    v.im = -v.im;
    a[idx] = v;
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void bluestein_post_kernel(const cpx<T> *__restrict__ a, cpx<T> *__restrict__ out,
                                                             const cpx<T> *__restrict__ chirp, const size_t n, const size_t m,
                                                             const size_t total /* batch * n */, const T scale_m, const T scale_n)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx / n, i = idx % n;
    cpx<T> v = a[b * m + i];
    v.im = -v.im;
    v = mk<T>(v.re * scale_m, v.im * scale_m);
    cpx<T> o = cmul(v, chirp[i]);
    if (INVERSE) {
        const T im = -o.im;
        o = mk<T>(o.re * scale_n, im * scale_n);
    }
    out[idx] = o;
}

}  // namespace kofft
