// fft_device.hip.h -- device-side building blocks shared by every kernel of the hot path.
//
// The arithmetic contract (what makes results identical to kofft's ScalarFftImpl):
//   * every radix-2 butterfly is  t = o * w  (4 mul, 1 sub, 1 add, un-fused),
//     out0 = e + t, out1 = e - t            -- fft.rs:881-893 (f32) / 1020-1032 (f64)
//   * w is looked up from the reference-recipe table T_n (uploaded from the host),
//     entry k*n2 for group k of a stage with half-span n2 -- fft.rs:839
//   * lengths 2,4,8,16 use the straight-line kernels of fft_kernels.rs with their
//     f32-literal constants.
// The translation unit is compiled with -ffp-contract=off so no FMA is formed.
//
// Stage/bit bookkeeping used throughout (derived from fft.rs:834-898):
//   stage s (s = 0..L-1, n2 = 2^(L-1-s)) reads  src[(2k+b)*n2 + j]  and writes
//   dst[(k + b'*2^s)*n2 + j]:  in index bits, [k : s bits][b][j] -> [b'][k][j].
//   A register pass that runs stages S0..S0+Q-1 therefore needs, per (k, j), the 2^Q
//   elements  i = k*2^(L-S0) + c*2^(L-S0-Q) + j  (c = 0..2^Q-1) and produces
//   o = rev_Q(c')*2^(L-Q) + (k*2^(L-S0-Q) + j), where c' is the in-place register index.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace kofft {

template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T re, im;
};

template <typename T>
__device__ __forceinline__ cpx<T> mk(T re, T im)
{
    cpx<T> c;
    c.re = re;
    c.im = im;
    return c;
}

// Streaming (read-once) global load of one complex value: the non-temporal hint keeps the batch from
// displacing the tables in L2 and measured +7 % on a read+write stream (5.41 -> 5.84 TB/s, tools/ubench_copy2).
template <typename T>
__device__ __forceinline__ cpx<T> ld_stream(const cpx<T> *p)
{
    typedef T vec2 __attribute__((ext_vector_type(2)));
    const vec2 v = __builtin_nontemporal_load(reinterpret_cast<const vec2 *>(p));
    return mk<T>(v.x, v.y);
}

// Streaming (write-once) global store: same hint on the way out.  Measured on the bench kernels (buffer form,
// one box, back to back): STFT 0.216 -> 0.193 ms, 65536 x 4096 c32 0.80 -> 0.775 ms, rfft unchanged.
template <typename T>
__device__ __forceinline__ void st_stream(cpx<T> *p, cpx<T> v)
{
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 f;
    f.x = v.re;
    f.y = v.im;
#ifdef KOFFT_PLAIN_STORES
    *reinterpret_cast<vec2 *>(p) = f;
#else
    __builtin_nontemporal_store(f, reinterpret_cast<vec2 *>(p));
#endif
}
__device__ __forceinline__ void st_stream(float *p, float v)
{
#ifdef KOFFT_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

// ---- buffer (SRSRC) addressing for the streaming kernels ---------------------------------------------
// A per-transform descriptor built from wave-uniform values (base pointer + byte count) lets every access be
// "descriptor (SGPRs) + one per-thread byte offset (a single VGPR) + a constant", instead of a 64-bit address
// per access: that is what keeps the prefetching kernels inside the register budget.  Reads past the byte
// count return 0 without touching memory (used for STFT frames that run off the end of the signal).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
enum : int { AUX_DEFAULT = 0, AUX_NT = 2 };  // gfx950 cache-policy bits of the buffer instructions

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// Two Complex<f32> values of ADJACENT ROWS as one 16-byte value (round 4: the c32 last factor on row pairs, fft_big.hip.h): the "scalar"
// is a 2-vector of f32, every arithmetic operation is elementwise (v_pk_*_f32: the same IEEE operations on each row's value with that
// row's table entry).  In memory such a value is {row0.re, row0.im, row1.re, row1.im}; in registers {re: {row0, row1}, im: {row0, row1}}.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NOTE: the loaded vector is bit-cast as a WHOLE.  Extracting .x/.y from the <2 x i32> result and casting
// each to float makes hipcc (ROCm 7.2) narrow the load to one dword and duplicate it.
template <typename T, int AUX>
__device__ __forceinline__ cpx<T> buf_load_cpx(rsrc_t r, int voff, int coff)
{
    if constexpr (std::is_same<T, f32x2>::value) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f f4 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, coff, AUX));
        f32x2 re, im;
        re.x = f4.x; re.y = f4.z;
        im.x = f4.y; im.y = f4.w;
        return mk<T>(re, im);
    } else {
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 f;
    if constexpr (sizeof(T) == 4) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, coff, AUX);
        f = __builtin_bit_cast(vec2, v);
    } else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, coff, AUX);
        f = __builtin_bit_cast(vec2, v);
    }
    return mk<T>(f.x, f.y);
    }
}

#ifndef KOFFT_STORE_AUX
#define KOFFT_STORE_AUX AUX_NT  // outputs are written once and not read back by the kernel (see st_stream)
#endif
// A 16-byte store of a Complex<f64> whose data registers the next f64 VALU instruction overwrites (round 6).  The compiler reuses ONE
// register quadruple for a loop of "value * scale -> store" (ifft's scaling, the rfft post-pass, Bluestein's products):
//     v_mul_f64 v[66:67], ..   v_mul_f64 v[68:69], ..   buffer_store_dwordx4 v[66:69], v1, s[12:15], s61 offen
//     v_mul_f64 v[66:67], ..   <- overwrites the first half of the data the store is still reading
// LLVM knows this hazard (GCNHazardRecognizer: a store of more than 64 bits followed by a VALU write of its data registers) and
// inserts `s_nop 1` -- but only when the store's soffset field is NOT an SGPR, as the SI-era documents say; every store here carries
// its compile-time offset in an SGPR soffset (one VGPR of address for all of a thread's stores).  On gfx950 with f64 VALU
// instructions the hazard is real with an SGPR soffset too: large c64 batches returned, in a few transforms per thousand, ONE
// register's real part replaced by the next store's in lanes 12-15 of every row of 16 (tools/dbg_f64.py; the imaginary part, written
// one instruction later, always arrived).  Found when tools/kernel_coverage.sh showed that the parity tests had stopped reaching
// these kernels (the host pipeline cut their batches into eight).  The wait states are pinned to the store by an asm statement that
// READS the stored registers: side effects keep it behind the store, the anti-dependence keeps every overwrite behind it.
// Packed-f32 data (the c32 row pairs' 16-byte stores) never showed it in the every-transform tests; it is the same instruction with the
// same soffset form and gets the same two wait states.
#ifndef KOFFT_F64_STORE_NOP
#define KOFFT_F64_STORE_NOP 1 /* s_nop argument: wait states - 1 (LLVM's own choice for the hazard it knows on gfx940+: 2 wait states); < 0: none */
#endif
template <typename V>
__device__ __forceinline__ void b128_store_guard(const V bits)
{
#if KOFFT_F64_STORE_NOP >= 0
    asm volatile("s_nop %1" ::"v"(bits), "n"(KOFFT_F64_STORE_NOP));
#endif
}
template <typename T>
__device__ __forceinline__ void buf_store_cpx(cpx<T> c, rsrc_t r, int voff, int coff)
{
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 f;
    f.x = c.re;
    f.y = c.im;
    if constexpr (sizeof(T) == 4) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, f), r, voff, coff, KOFFT_STORE_AUX);
    } else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u bits = __builtin_bit_cast(v4u, f);
        __builtin_amdgcn_raw_buffer_store_b128(bits, r, voff, coff, KOFFT_STORE_AUX);
        b128_store_guard(bits);
    }
}

template <typename T, int AUX>
__device__ __forceinline__ void buf_store_cpx_aux(cpx<T> c, rsrc_t r, int voff, int coff)
{
    if constexpr (std::is_same<T, f32x2>::value) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f f4;
        f4.x = c.re.x; f4.y = c.im.x; f4.z = c.re.y; f4.w = c.im.y;
        const v4u bits = __builtin_bit_cast(v4u, f4);
        __builtin_amdgcn_raw_buffer_store_b128(bits, r, voff, coff, AUX);
        b128_store_guard(bits);  // (same instruction, same soffset form: guarded too, although packed-f32 arithmetic never showed the hazard)
        return;
    } else {
    typedef T vec2 __attribute__((ext_vector_type(2)));
    vec2 f;
    f.x = c.re;
    f.y = c.im;
    if constexpr (sizeof(T) == 4) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, f), r, voff, coff, AUX);
    } else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u bits = __builtin_bit_cast(v4u, f);
        __builtin_amdgcn_raw_buffer_store_b128(bits, r, voff, coff, AUX);
        b128_store_guard(bits);
    }
    }
}

__device__ __forceinline__ void buf_store_f32(float v, rsrc_t r, int voff, int coff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, coff, KOFFT_STORE_AUX);
}

template <int AUX>
__device__ __forceinline__ float buf_load_f32(rsrc_t r, int voff, int coff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, coff, AUX));
}

// num.rs:127-141, 161-166 (non-FMA arm)
template <typename T>
__device__ __forceinline__ cpx<T> cadd(cpx<T> a, cpx<T> b) { return mk<T>(a.re + b.re, a.im + b.im); }
template <typename T>
__device__ __forceinline__ cpx<T> csub(cpx<T> a, cpx<T> b) { return mk<T>(a.re - b.re, a.im - b.im); }
template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b)
{
    return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}

// One Stockham butterfly, in place on (e, o).  fft.rs:881-893:
//   t_re = o.re*w.re - o.im*w.im ; t_im = o.re*w.im + o.im*w.re   (4 mul, 1 sub, 1 add, un-fused)
//   e' = e + t ; o' = e - t
// Generic form (f64, and f32 when KOFFT_BFLY_NOASM is defined):
template <typename T>
__device__ __forceinline__ void bfly_generic(cpx<T> &e, cpx<T> &o, const cpx<T> w)
{
    const T t_re = o.re * w.re - o.im * w.im;
    const T t_im = o.re * w.im + o.im * w.re;
    const T e_re = e.re, e_im = e.im;
    e.re = e_re + t_re;
    e.im = e_im + t_im;
    o.re = e_re - t_re;
    o.im = e_im - t_im;
}

// f32 form: five packed instructions on aligned {re, im} register pairs, operand swizzles and sign modifiers
// doing the shuffling that hipcc otherwise spends v_mov's and splatted twiddle copies on (7 instructions and
// 4 registers per twiddle when left to the SLP vectoriser):
//   p1 = (o.re*w.re, o.im*w.re)            v_pk_mul  op_sel:[0,0] op_sel_hi:[1,0]
//   p2 = (o.im*w.im, o.re*w.im)            v_pk_mul  op_sel:[1,1] op_sel_hi:[0,1]
//   t  = (p1.x - p2.x, p1.y + p2.y)        v_pk_add  neg_lo:[0,1]
//   o' = e - t ; e' = e + t                v_pk_add  (neg_lo/neg_hi on t), v_pk_add
// Each instruction is IEEE mul/add/sub on f32 with round-to-nearest-even: the same operations, in the same
// order per output value, as the reference (a + (-b) is a - b exactly; t_im's operands commute).
// One asm statement per instruction, no "volatile": the scheduler stays free to interleave butterflies.
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool W_IN_SGPR>
__device__ __forceinline__ void bfly_f32_pk(v2f &e, v2f &o, const v2f w)
{
    v2f p1, p2, t, d, s;
    if constexpr (W_IN_SGPR) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p1) : "v"(o), "s"(w));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(p2) : "v"(o), "s"(w));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p1) : "v"(o), "v"(w));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(p2) : "v"(o), "v"(w));
    }
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(t) : "v"(p1), "v"(p2));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(e), "v"(t));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(e), "v"(t));
    e = s;
    o = d;
}

// The real-FFT post-pass (rfft.rs:454-463) in packed form, 7 instructions (the compiler's own packing takes 10-11):
//   b = conj(ymk); sum = a + b; diff = a - b; t = w * diff (Complex::mul, un-fused); X = (sum + (t.im, -t.re)) * 0.5
// The conjugation and the (t.im, -t.re) swizzle are folded into neg_* / op_sel modifiers, which are exact.
__device__ __forceinline__ v2f rfft_post_f32_pk(const v2f w, const v2f a, const v2f ymk)
{
    v2f s, d, p1, p2, t, x;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(s) : "v"(a), "v"(ymk));  // (a.re + y.re, a.im - y.im)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(ymk));  // (a.re - y.re, a.im + y.im)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p1) : "v"(w), "v"(d));  // (w.re*d.re, w.re*d.im)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p2) : "v"(w), "v"(d));  // (w.im*d.im, w.im*d.re)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(t) : "v"(p1), "v"(p2));    // (t.re, t.im)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(x) : "v"(s), "v"(t));  // (s.re + t.im, s.im - t.re)
    asm("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(x) : "v"(x));
    return x;
}

template <typename T, bool W_UNIFORM = false>
__device__ __forceinline__ void bfly(cpx<T> &e, cpx<T> &o, const cpx<T> w)
{
#ifndef KOFFT_BFLY_NOASM
    if constexpr (sizeof(T) == 4) {
        v2f ev = {e.re, e.im}, ov = {o.re, o.im};
        const v2f wv = {w.re, w.im};
        bfly_f32_pk<W_UNIFORM>(ev, ov, wv);
        e.re = ev.x;
        e.im = ev.y;
        o.re = ov.x;
        o.im = ov.y;
        return;
    }
#endif
    bfly_generic(e, o, w);
}

__host__ __device__ constexpr int bitrev(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// Q Stockham stages (global stages S0 .. S0+Q-1 of a 2^L-point transform) on the 2^Q
// registers v[], for group index k (the S0-bit frequency prefix built so far).
// Register index bit (Q-1-t) is the bit consumed (and replaced by b') at local stage t.
// Twiddle for local stage t, prefix bits h (register bits Q-1..Q-t):
//   kk = k + 2^S0 * rev_t(h),  index = kk * 2^(L-1-S0-t).
// Where a sub-transform's twiddles live when it is one factor of a larger 2^LT-point transform (fft_big.hip.h).
// A sub-transform that performs global stages S_off .. S_off+L-1 for the outer frequency prefix K reads
//   T_{2^LT}[ (idx_local << (LT - L)) + (K << (LT - 1 - S_off - s_local)) ]
// where idx_local is the index the same butterfly would use in a stand-alone 2^L-point transform and s_local its
// stage there (derivation: fft_big.hip.h).  Stand-alone transforms use TwPlain (shift 0, no K term).
struct TwPlain {
    __device__ __forceinline__ int operator()(int idx_local, int) const { return idx_local; }
};
struct TwSub {
    int shift;  // LT - L
    int K;      // outer prefix
    int kbase;  // LT - 1 - S_off  (>= every s_local of the sub-transform)
    __device__ __forceinline__ int operator()(int idx_local, int s_local) const
    {
        return (idx_local << shift) + (K << (kbase - s_local));
    }
};
// first factor: no prefix yet (S_off = 0, K = 0) -- a separate type so that no shift by (0 - s_local) is ever formed
struct TwSubFirst {
    int shift;  // LT - L
    __device__ __forceinline__ int operator()(int idx_local, int) const { return idx_local << shift; }
};

// UNIFORM: the caller guarantees k is the same in every lane of the wave AND that the compiler can see it
// (k == 0 in pass 0): the twiddles then come from scalar loads and are used straight from SGPRs.
template <typename T, int L, int S0, int Q, bool UNIFORM = false, class TwMap = TwPlain>
__device__ __forceinline__ void reg_pass(cpx<T> *v, const int k, const cpx<T> *__restrict__ tw, const TwMap map = TwMap{})
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const int pos = Q - 1 - t;
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const int idx = (k << (L - 1 - S0 - t)) + (bitrev(h, t) << (L - 1 - t));
            const cpx<T> w = tw[map(idx, S0 + t)];
#pragma unroll
            for (int lo = 0; lo < (1 << pos); ++lo) {
                const int c = (h << (pos + 1)) | lo;
                bfly<T, UNIFORM>(v[c], v[c | (1 << pos)], w);
            }
        }
    }
}

// ---- straight-line kernels for n = 2, 4, 8, 16 (fft_kernels.rs) -----------------

// fft_kernels.rs:4-10
template <typename T>
__device__ __forceinline__ void small_fft2(cpx<T> *x)
{
    const cpx<T> a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}

// fft_kernels.rs:13-29 ; the (0,-1) factor goes through the general complex multiply.
template <typename T>
__device__ __forceinline__ void small_fft4(cpx<T> *x)
{
    const cpx<T> mi = mk<T>(T(0), -T(1));
    const cpx<T> s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
    const cpx<T> s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
    const cpx<T> r = cmul(d13, mi);
    x[0] = cadd(s02, s13);
    x[2] = csub(s02, s13);
    x[1] = cadd(d02, r);
    x[3] = csub(d02, r);
}

// The 4-point sub-transform used inside fft8/fft16: inputs p0..p3, outputs q0..q3 with
// q0 = (p0+p2)+(p1+p3), q2 = (p0+p2)-(p1+p3), q1 = (p0-p2)+(p1-p3)*(0,-1), q3 = ... - ...
template <typename T>
__device__ __forceinline__ void quad(const cpx<T> p0, const cpx<T> p1, const cpx<T> p2, const cpx<T> p3,
                                     cpx<T> &q0, cpx<T> &q1, cpx<T> &q2, cpx<T> &q3)
{
    const cpx<T> mi = mk<T>(T(0), -T(1));
    const cpx<T> a0 = cadd(p0, p2), a1 = csub(p0, p2);
    const cpx<T> a2 = cadd(p1, p3), a3 = csub(p1, p3);
    const cpx<T> r = cmul(a3, mi);
    q0 = cadd(a0, a2);
    q2 = csub(a0, a2);
    q1 = cadd(a1, r);
    q3 = csub(a1, r);
}

// 8-point combine used by fft8 and by each half of fft16: e[0..3] from the even comb,
// o[0..3] from the odd comb; o1*(s,-s), o2*(0,-1), o3*(-s,-s); y[i] = e+o', y[i+4] = e-o'.
// fft_kernels.rs:69-85 and 133-144 / 167-178 (s = 0.70710677 as an f32 literal).
template <typename T>
__device__ __forceinline__ void oct_combine(const cpx<T> *e, const cpx<T> *o, cpx<T> *y)
{
    const T s = T(0.70710677f);
    const cpx<T> t0 = o[0];
    const cpx<T> t1 = cmul(o[1], mk<T>(s, -s));
    const cpx<T> t2 = cmul(o[2], mk<T>(T(0), -T(1)));
    const cpx<T> t3 = cmul(o[3], mk<T>(-s, -s));
    y[0] = cadd(e[0], t0);
    y[1] = cadd(e[1], t1);
    y[2] = cadd(e[2], t2);
    y[3] = cadd(e[3], t3);
    y[4] = csub(e[0], t0);
    y[5] = csub(e[1], t1);
    y[6] = csub(e[2], t2);
    y[7] = csub(e[3], t3);
}

// fft_kernels.rs:32-86
template <typename T>
__device__ __forceinline__ void small_fft8(cpx<T> *x)
{
    cpx<T> e[4], o[4], y[8];
    quad(x[0], x[2], x[4], x[6], e[0], e[1], e[2], e[3]);
    quad(x[1], x[3], x[5], x[7], o[0], o[1], o[2], o[3]);
    oct_combine(e, o, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = y[i];
}

// fft_kernels.rs:89-224
template <typename T>
__device__ __forceinline__ void small_fft16(cpx<T> *x)
{
    cpx<T> e[4], o[4], E[8], O[8];
    // even comb x0,x2,..,x14: its own even comb is x0,x4,x8,x12, odd comb x2,x6,x10,x14
    quad(x[0], x[4], x[8], x[12], e[0], e[1], e[2], e[3]);
    quad(x[2], x[6], x[10], x[14], o[0], o[1], o[2], o[3]);
    oct_combine(e, o, E);
    quad(x[1], x[5], x[9], x[13], e[0], e[1], e[2], e[3]);
    quad(x[3], x[7], x[11], x[15], o[0], o[1], o[2], o[3]);
    oct_combine(e, o, O);
    // fft_kernels.rs:181-197: f32 literals, widened for f64
    const T c1 = T(0.9238795f), s1 = T(-0.38268343f);
    const T c2 = T(0.70710677f), s2 = T(-0.70710677f);
    const T c3 = T(0.38268343f), s3 = T(-0.9238795f);
    cpx<T> q[8];
    q[0] = O[0];
    q[1] = cmul(O[1], mk<T>(c1, s1));
    q[2] = cmul(O[2], mk<T>(c2, s2));
    q[3] = cmul(O[3], mk<T>(c3, s3));
    q[4] = cmul(O[4], mk<T>(T(0), T(-1.0f)));
    q[5] = cmul(O[5], mk<T>(-c3, s3));
    q[6] = cmul(O[6], mk<T>(-c2, s2));
    q[7] = cmul(O[7], mk<T>(-c1, s1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = cadd(E[i], q[i]);
        x[i + 8] = csub(E[i], q[i]);
    }
}

}  // namespace kofft
