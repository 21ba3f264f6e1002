// C++ restatement of the reference's own tests for the hot path, against the C++ host mirror
// (include/kofft_hip.hpp) of kofft's interface.  Each block cites the reference test it mirrors.
// The oracle (oracle/kofft_oracle.h) is linked as the checker only.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/kofft_hip.hpp"
#include "../../oracle/kofft_oracle.h"

using namespace kofft;
static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++g_checks;                                                                  \
        if (!(cond)) { ++g_fail; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static std::vector<Complex32> oracle_fft(std::vector<Complex32> x, bool inverse = false)
{
    ko_fft_batch_f32(reinterpret_cast<float *>(x.data()), x.size(), 1, inverse ? 1 : 0);
    return x;
}
static bool same_bits(const std::vector<Complex32> &a, const std::vector<Complex32> &b)
{
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(Complex32)) == 0;
}
static bool same_bits(const std::vector<Complex64> &a, const std::vector<Complex64> &b)
{
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(Complex64)) == 0;
}

int main()
{
    HipFftImpl<float> fft;  // ScalarFftImpl::<f32>::default()
    HipFftImpl<double> fft64;

    {   // lib.rs:178-199 test_fft_ifft_f32
        std::vector<Complex32> data = {{1, 0}, {0, 0}, {0, 0}, {0, 0}};
        fft.fft(data).unwrap();
        for (auto &c : data) { CHECK(std::fabs(c.re - 1.0f) < 1e-6f); CHECK(std::fabs(c.im) < 1e-6f); }
        fft.ifft(data).unwrap();
        CHECK(std::fabs(data[0].re - 1.0f) < 1e-6f);
        for (size_t i = 1; i < 4; ++i) { CHECK(std::fabs(data[i].re) < 1e-6f); CHECK(std::fabs(data[i].im) < 1e-6f); }
    }
    {   // lib.rs:254-264 test_fft_all_ones
        std::vector<Complex32> data(8, Complex32(1, 0));
        fft.fft(data).unwrap();
        CHECK(std::fabs(data[0].re - 8.0f) < 1e-6f);
        for (size_t i = 1; i < 8; ++i) { CHECK(std::fabs(data[i].re) < 1e-6f); CHECK(std::fabs(data[i].im) < 1e-6f); }
    }
    {   // lib.rs:322-326 test_fft_empty ; lib.rs:352-358 single element
        std::vector<Complex32> empty;
        CHECK(fft.fft(empty) == Result::Err(FftError::EmptyInput));
        CHECK(fft.ifft(empty).unwrap_err() == FftError::EmptyInput);
        std::vector<Complex32> one = {{1, 0}};
        fft.fft(one).unwrap();
        CHECK(one[0].re == 1.0f && one[0].im == 0.0f);
    }
    {   // lib.rs:329-338 test_fft_out_of_place_mismatched_lengths
        std::vector<Complex32> in = {{1, 0}, {2, 0}}, out(3);
        CHECK(fft.fft_out_of_place(in, out) == Result::Err(FftError::MismatchedLengths));
    }
    {   // tests/stockham_parity.rs + tests/stockham_large.rs + tests/parallel_stockham.rs inputs, vs the oracle, bit for bit
        for (size_t n : {32u, 64u, 128u, 256u, 512u, 1024u, 4096u}) {
            std::vector<Complex32> data(n);
            for (size_t i = 0; i < n; ++i) data[i] = Complex32((float)i, -(float)i * 0.25f);
            auto expected = oracle_fft(data);
            fft.stockham_fft(data).unwrap();
            CHECK(same_bits(data, expected));
        }
    }
    {   // examples/basic_usage.rs:232-241 (BASELINE config #1): 1024-pt FFT of sin(0.1 i)
        std::vector<Complex32> sig(1024);
        for (size_t i = 0; i < 1024; ++i) sig[i] = Complex32(std::sin(0.1f * (float)i), 0.0f);
        auto expected = oracle_fft(sig);
        fft.fft(sig).unwrap();
        CHECK(same_bits(sig, expected));
    }
    {   // fft_strided error order (fft.rs:1181-1190) and result
        std::vector<Complex32> buf(8), scratch(4), none;
        CHECK(fft.fft_strided(buf, 0, scratch) == Result::Err(FftError::InvalidStride));
        std::vector<Complex32> small(6);
        CHECK(fft.fft_strided(small, 2, scratch) == Result::Err(FftError::MismatchedLengths));
        CHECK(fft.fft_strided(buf, 2, none).is_ok());
        for (size_t i = 0; i < 8; ++i) buf[i] = Complex32((float)i, 1.0f);
        std::vector<Complex32> picked = {buf[0], buf[2], buf[4], buf[6]};
        auto want = oracle_fft(picked);
        fft.fft_strided(buf, 2, scratch).unwrap();
        CHECK(buf[0] == want[0] && buf[2] == want[1] && buf[4] == want[2] && buf[6] == want[3]);
        CHECK(buf[1] == Complex32(1, 1));
    }
    {   // rfft.rs:892-907 rfft_irfft_roundtrip + lib.rs:470-478 mismatched lengths + rfft.rs:433-443 order
        std::vector<float> input = {1, 2, 3, 4, 5, 6, 7, 8}, orig = input, out(8);
        std::vector<Complex32> freq(5), scratch(4);
        fft.rfft_with_scratch(input, freq, scratch).unwrap();
        fft.irfft_with_scratch(freq, out, scratch).unwrap();
        for (size_t i = 0; i < 8; ++i) CHECK(std::fabs(orig[i] - out[i]) < 1e-5f);
        CHECK(std::fabs(freq[0].im) < 1e-6f && std::fabs(freq[4].im) < 1e-6f);  // lib.rs:451-467
        std::vector<float> four = {1, 2, 3, 4}, none, odd = {1, 2, 3};
        std::vector<Complex32> wrong(4), two(2);
        CHECK(fft.rfft(four, wrong) == Result::Err(FftError::MismatchedLengths));
        CHECK(fft.rfft(none, two) == Result::Err(FftError::EmptyInput));
        CHECK(fft.rfft(odd, two) == Result::Err(FftError::InvalidValue));
        // f64 (rfft.rs:921-936)
        std::vector<double> in64 = {1, 2, 3, 4, 5, 6, 7, 8}, o64(8);
        std::vector<Complex64> f64(5), s64(4);
        fft64.rfft_with_scratch(in64, f64, s64).unwrap();
        fft64.irfft_with_scratch(f64, o64, s64).unwrap();
        for (size_t i = 0; i < 8; ++i) CHECK(std::fabs(o64[i] - (double)(i + 1)) < 1e-10);
    }
    {   // tests/rfft_twiddles.rs + tests/twiddle.rs table entries
        auto tw = RfftPlanner<float>().get_twiddles(8);
        CHECK(tw.size() == 8);
        CHECK(std::fabs(tw[1].re - std::cos(-3.14159274f / 8.0f)) < 1e-6f && std::fabs(tw[1].im - std::sin(-3.14159274f / 8.0f)) < 1e-6f);
        auto t32 = FftPlanner<float>().get_twiddles(8);
        CHECK(std::fabs(t32[1].re - std::cos(-2.0f * 3.14159274f / 8.0f)) < 1e-6f);
        CHECK(FftPlanner<float>().plan_strategy(4096) == FftStrategy::SplitRadix);
    }
    {   // tests/stft.rs:6-14 + stft.rs:560-580 batch round-trip shape + stft.rs:690-697 zero hop + window.rs:104-109
        auto w = hann(8);
        CHECK(std::fabs(w[0]) < 1e-6f && std::fabs(w[4] - 1.0f) < 1e-6f);
        std::vector<float> signal(10, 0.0f);
        auto window = hann(4);
        std::vector<std::vector<Complex32>> frames(2);
        CHECK(stft(signal, window, 4, frames, fft) == Result::Err(FftError::MismatchedLengths));
        std::vector<std::vector<Complex32>> four(4);
        CHECK(stft(signal, window, 0, four, fft) == Result::Err(FftError::InvalidHopSize));
        std::vector<float> sig = {1, 2, 3, 4, 5, 6, 7, 8}, ones(4, 1.0f);
        stft(sig, ones, 2, four, fft).unwrap();
        std::vector<float> want(4 * 4 * 2);
        ko_stft_f32(sig.data(), 8, ones.data(), 4, 2, want.data(), 4);
        for (size_t f = 0; f < 4; ++f) CHECK(four[f].size() == 4 && std::memcmp(four[f].data(), &want[f * 8], 32) == 0);
        {   // stft.rs:560-580 test_stft_istft_batch_roundtrip
            auto fr = four;
            std::vector<float> output(8, 0.0f), scr(8, 0.0f), bad(7);
            istft(fr, ones, 2, output, scr, fft).unwrap();
            for (size_t i = 0; i < 8; ++i) CHECK(std::fabs(output[i] - sig[i]) < 1e-4f);
            CHECK(istft(fr, ones, 0, output, scr, fft) == Result::Err(FftError::InvalidHopSize));
            CHECK(istft(fr, ones, 2, output, bad, fft) == Result::Err(FftError::MismatchedLengths));
        }
        std::vector<std::vector<Complex32>> two(2);
        parallel(sig, ones, 2, two, fft).unwrap();  // fewer frames than stft() demands: parallel() does not check
        CHECK(std::memcmp(two[1].data(), &want[8], 32) == 0);
        StftStream *stream = nullptr;
        CHECK(StftStream::create(sig, ones, 0, fft, stream) == Result::Err(FftError::InvalidHopSize));
        StftStream::create(sig, ones, 2, fft, stream).unwrap();
        std::vector<Complex32> buf(4), bad(3);
        bool more = false;
        CHECK(stream->next_frame(bad, more) == Result::Err(FftError::MismatchedLengths));
        size_t count = 0;
        while (stream->next_frame(buf, more).is_ok() && more) {
            CHECK(std::memcmp(buf.data(), &want[count * 8], 32) == 0);
            ++count;
        }
        CHECK(count == 4);
        delete stream;
    }
    {   // fft::batch over ragged slices (fft.rs:2156-2164) + contiguous batch entry
        std::vector<std::vector<Complex32>> vs(3);
        vs[0].assign(64, Complex32(1, -1)); vs[1].assign(64, Complex32(0.5f, 2)); vs[2].assign(256, Complex32(3, 0));
        vs[0][3] = Complex32(7, 7);
        auto w0 = oracle_fft(vs[0]), w2 = oracle_fft(vs[2]);
        batch<float>(fft, vs).unwrap();
        CHECK(same_bits(vs[0], w0) && same_bits(vs[2], w2));
        std::vector<Complex32> flat(4 * 4096);
        for (size_t i = 0; i < flat.size(); ++i) flat[i] = Complex32(std::sin((float)i), std::cos((float)i));
        std::vector<Complex32> row(flat.begin() + 2 * 4096, flat.begin() + 3 * 4096);
        auto wr = oracle_fft(row);
        fft.fft_batch(flat, 4096).unwrap();
        CHECK(std::memcmp(&flat[2 * 4096], wr.data(), 4096 * 8) == 0);
    }
    {   // non-power-of-two lengths take the Bluestein arm (fft.rs:1088-1132): tests/bluestein.rs n = 15, lib.rs:267-282 n = 3
        std::vector<Complex32> fifteen(15);
        for (size_t i = 0; i < 15; ++i) fifteen[i] = Complex32((float)i, (float)i * 0.5f);
        auto want = oracle_fft(fifteen);
        fft.fft(fifteen).unwrap();
        CHECK(same_bits(fifteen, want));
        std::vector<Complex32> three = {{1, 0}, {2, 0}, {3, 0}};
        fft.fft(three).unwrap();
        fft.ifft(three).unwrap();
        for (size_t i = 0; i < 3; ++i) { CHECK(std::fabs(three[i].re - (float)(i + 1)) < 1e-5f); CHECK(std::fabs(three[i].im) < 1e-5f); }
        // a real FFT whose half length is not a power of two takes the composed path (rfft.rs:447: fft.fft for any m):
        // a constant row has X[0] = sum, everything else 0 up to the Bluestein arm's rounding
        std::vector<float> twelve(12, 1.0f);
        std::vector<Complex32> seven(7);
        fft.rfft(twelve, seven).unwrap();
        CHECK(std::fabs(seven[0].re - 12.0f) < 1e-4f && std::fabs(seven[0].im) < 1e-4f);
        for (size_t k = 1; k < 7; ++k) CHECK(std::fabs(seven[k].re) < 1e-4f && std::fabs(seven[k].im) < 1e-4f);
    }
    {   // stft.rs:526-557 frame / inverse_frame streaming round trip; stft.rs:289-343 inverse_parallel vs istft
        std::vector<float> sig = {1, 2, 3, 4, 5, 6, 7, 8}, win(4, 1.0f), output(8, 0.0f), norm(8, 0.0f);
        std::vector<Complex32> buf(4);
        for (size_t pos = 0; pos < 8; pos += 2) {
            frame(sig, win, pos, buf, fft).unwrap();
            inverse_frame(buf, win, pos, output, fft).unwrap();
            for (size_t i = 0; i < 4 && pos + i < 8; ++i) norm[pos + i] += win[i] * win[i];
        }
        for (size_t i = 0; i < 8; ++i) CHECK(std::fabs(output[i] / norm[i] - sig[i]) < 1e-4f);
        std::vector<float> hw = hann(8), out_a(16, 0.0f), out_b(16, 0.0f), scratch(16, 0.0f), sig16(16);
        for (size_t i = 0; i < 16; ++i) sig16[i] = std::sin(0.37f * (float)i);
        std::vector<std::vector<Complex32>> frames(8, std::vector<Complex32>(8)), copy;
        stft(sig16, hw, 2, frames, fft).unwrap();
        copy = frames;
        inverse_parallel(frames, hw, 2, out_b, fft).unwrap();
        CHECK(frames == copy);  // untouched
        istft(copy, hw, 2, out_a, scratch, fft).unwrap();
        for (size_t i = 0; i < 16; ++i) CHECK(scratch[i] > 1e-8f ? std::memcmp(&out_a[i], &out_b[i], 4) == 0 : out_b[i] == 0.0f);
        CHECK(inverse_parallel(frames, hw, 0, out_b, fft) == Result::Err(FftError::InvalidHopSize));
    }
    {   // visual/spectrogram.rs:52-76 stft_magnitudes against the mirror's own stft (same kernel arithmetic, sqrt of re*re + im*im)
        std::vector<float> sig(300);
        for (size_t i = 0; i < sig.size(); ++i) sig[i] = std::sin(0.05f * (float)i) + 0.25f * std::cos(0.31f * (float)i);
        std::vector<std::vector<float>> mags;
        float mx = -1.0f;
        stft_magnitudes(sig, 64, 16, mags, mx, fft).unwrap();
        CHECK(mags.size() == 19 && mags[0].size() == 32);
        std::vector<std::vector<Complex32>> fr(19, std::vector<Complex32>(64));
        stft(sig, hann(64), 16, fr, fft).unwrap();
        float want_max = 0.0f;
        bool same = true;
        for (size_t f = 0; f < 19; ++f)
            for (size_t k = 0; k < 32; ++k) {
                const float m = std::sqrt(fr[f][k].re * fr[f][k].re + fr[f][k].im * fr[f][k].im);
                same = same && std::memcmp(&m, &mags[f][k], 4) == 0;
                if (m > want_max) want_max = m;
            }
        CHECK(same);
        CHECK(mx == want_max);
        CHECK(stft_magnitudes(sig, 64, 0, mags, mx, fft) == Result::Err(FftError::InvalidHopSize));
    }
    {   // ndfft.rs:158-225 tests: 2-D impulse -> all ones; length checks; 3-D round trip through rows/columns/tubes
        std::vector<Complex32> img(8 * 16, Complex32(0, 0)), col(8), bad(7);
        img[0] = Complex32(1, 0);
        fft2d_inplace<float>(img, 8, 16, fft, col).unwrap();
        bool ones = true;
        for (auto &c : img) ones = ones && c.re == 1.0f && c.im == 0.0f;
        CHECK(ones);
        CHECK(fft2d_inplace<float>(img, 8, 16, fft, bad) == Result::Err(FftError::MismatchedLengths));
        CHECK(fft2d_inplace<float>(img, 8, 15, fft, col) == Result::Err(FftError::MismatchedLengths));
        // 2-D against the oracle: rows, then columns through gather / fft / scatter (ndfft.rs:88-99)
        std::vector<Complex32> a(4 * 8), want;
        for (size_t i = 0; i < a.size(); ++i) a[i] = Complex32(std::sin((float)i), 0.5f * std::cos(2.0f * (float)i));
        want = a;
        for (size_t r = 0; r < 4; ++r) {
            std::vector<Complex32> row(want.begin() + r * 8, want.begin() + (r + 1) * 8);
            row = oracle_fft(row);
            std::copy(row.begin(), row.end(), want.begin() + r * 8);
        }
        for (size_t c = 0; c < 8; ++c) {
            std::vector<Complex32> cc(4);
            for (size_t r = 0; r < 4; ++r) cc[r] = want[r * 8 + c];
            cc = oracle_fft(cc);
            for (size_t r = 0; r < 4; ++r) want[r * 8 + c] = cc[r];
        }
        std::vector<Complex32> col4(4);
        fft2d_inplace<float>(a, 4, 8, fft, col4).unwrap();
        CHECK(same_bits(a, want));
        std::vector<Complex64> vol(2 * 4 * 8, Complex64(0, 0)), t2(2), r4(4), c8(8);
        vol[0] = Complex64(2, 0);
        fft3d_inplace<double>(vol, 2, 4, 8, fft64, t2, r4, c8).unwrap();
        bool twos = true;
        for (auto &c : vol) twos = twos && c.re == 2.0 && c.im == 0.0;
        CHECK(twos);
    }
    {   // tests/istft_stream.rs:4-52 istft_stream_reconstructs_and_flushes; stft.rs:679-690 frame size mismatch
        std::vector<float> signal = {1, 2, 3, 4, 5, 6, 7, 8}, window(4, 1.0f);
        StftStream *ss = nullptr;
        IstftStream *is = nullptr;
        CHECK(StftStream::create(signal, window, 2, fft, ss).is_ok());
        CHECK(IstftStream::create(4, 2, window, fft, is).is_ok());
        std::vector<Complex32> fr(4);
        std::vector<std::vector<Complex32>> frames;
        std::vector<float> stream_out, chunk, tail;
        bool more = false;
        while (ss->next_frame(fr, more).is_ok() && more) {
            frames.push_back(fr);
            is->push_frame(fr, chunk).unwrap();
            stream_out.insert(stream_out.end(), chunk.begin(), chunk.end());
        }
        is->flush(tail);
        std::vector<float> offline(signal.size() + 2, 0.0f), scratch(offline.size(), 0.0f);
        istft(frames, window, 2, offline, scratch, fft).unwrap();
        CHECK(stream_out.size() == signal.size() && std::memcmp(stream_out.data(), offline.data(), signal.size() * 4) == 0);
        CHECK(tail.size() == 2 && std::memcmp(tail.data(), &offline[signal.size()], 8) == 0);
        is->flush(tail);
        CHECK(tail.empty());
        std::vector<Complex32> short_frame(3);
        CHECK(is->push_frame(short_frame, chunk) == Result::Err(FftError::MismatchedLengths));
        IstftStream *bad = nullptr;
        CHECK(IstftStream::create(4, 0, window, fft, bad) == Result::Err(FftError::InvalidHopSize));
        delete ss;
        delete is;
    }
    {   // fft.rs:2361-2387 plan fft / ifft / out of place; 2581-2610 length mismatches
        FftPlan<float> plan(4, FftStrategy::SplitRadix, fft);
        std::vector<Complex32> data = {{1, 0}, {2, 0}, {3, 0}, {4, 0}}, orig = data;
        plan.fft(data).unwrap();
        CHECK(same_bits(data, oracle_fft(orig)));
        plan.ifft(data).unwrap();
        for (size_t i = 0; i < 4; ++i) CHECK(std::fabs(data[i].re - orig[i].re) < 1e-4f);
        std::vector<Complex32> ones(4, Complex32(1, 0)), out(4), out2(4), three(3);
        plan.fft_out_of_place(ones, out).unwrap();
        plan.ifft_out_of_place(out, out2).unwrap();
        CHECK(same_bits(out2, oracle_fft(out, true)));
        FftPlan<float> p2(4, FftStrategy::Radix2, fft);
        CHECK(p2.fft(three) == Result::Err(FftError::MismatchedLengths));
        CHECK(p2.ifft(three) == Result::Err(FftError::MismatchedLengths));
        CHECK(p2.fft_out_of_place(ones, three) == Result::Err(FftError::MismatchedLengths));
        CHECK(p2.ifft_out_of_place(three, out) == Result::Err(FftError::MismatchedLengths));
        std::vector<Complex32> x64(64);
        for (size_t i = 0; i < 64; ++i) x64[i] = Complex32(std::sin(0.3f * (float)i), std::cos(1.1f * (float)i));
        auto want = oracle_fft(x64);
        FftPlan<float>(64, FftStrategy::Radix4, fft).fft(x64).unwrap();  // f32 plan: the *_with_twiddles shortcut = stockham_fft (fft.rs:2016-2035)
        CHECK(same_bits(x64, want));
        // fft_with_strategy(.., Radix4) on a DEFAULT-constructed implementation: the reference's own fft_radix4 bytes
        // (fft.rs:1356, 1455-1548), which are not that transform
        std::vector<Complex32> r4(64), r4want(64), r4in(64);
        for (size_t i = 0; i < 64; ++i) r4[i] = r4want[i] = r4in[i] = Complex32(std::sin(0.3f * (float)i), std::cos(1.1f * (float)i));
        ko_fft_radix4_batch_f32(reinterpret_cast<float *>(r4want.data()), 64, 1);
        CHECK(fft.radix4_compat);
        fft.fft_with_strategy(r4, FftStrategy::Radix4).unwrap();
        CHECK(same_bits(r4, r4want));
        CHECK(!same_bits(r4, want));
        // the opt-out: the true transform for every strategy
        HipFftImpl<float> plain;
        plain.radix4_compat = false;
        r4 = r4in;
        plain.fft_with_strategy(r4, FftStrategy::Radix4).unwrap();
        CHECK(same_bits(r4, want));
        // an f64 plan has no shortcut: Radix4 -> fft_radix4, and ifft = conj, fft_radix4, conj * 1/n (fft.rs:2040-2055)
        std::vector<Complex64> d4(64), d4want(64), d4inv(64);
        for (size_t i = 0; i < 64; ++i) d4[i] = d4want[i] = d4inv[i] = Complex64(std::sin(0.3 * (double)i), std::cos(1.1 * (double)i));
        ko_fft_radix4_batch_f64(reinterpret_cast<double *>(d4want.data()), 64, 1);
        FftPlan<double> p4(64, FftStrategy::Radix4, fft64);
        p4.fft(d4).unwrap();
        CHECK(same_bits(d4, d4want));
        std::vector<Complex64> d4ref = d4inv;
        for (auto &c : d4ref) c.im = -c.im;
        ko_fft_radix4_batch_f64(reinterpret_cast<double *>(d4ref.data()), 64, 1);
        const double sc = 1.0 / (double)(float)64;
        for (auto &c : d4ref) { c.im = -c.im; c.re = c.re * sc; c.im = c.im * sc; }
        p4.ifft(d4inv).unwrap();
        CHECK(same_bits(d4inv, d4ref));
    }
    {   // multi-GPU STFT through the C++ mirror: one device here, same path as G devices
        std::vector<float> sig(3000), window = hann(256);
        for (size_t i = 0; i < sig.size(); ++i) sig[i] = std::sin(0.05f * (float)i);
        const size_t frames = (sig.size() + 63) / 64;
        std::vector<std::vector<Complex32>> a(frames), b(frames), c(frames);
        stft(sig, window, 64, a, fft).unwrap();
        HipMulti multi(1);
        CHECK(multi.ngpu() == 1);
        multi.stft(sig, window, 64, b, false).unwrap();
        multi.stft(sig, window, 64, c, true).unwrap();  // with the RCCL all-gather
        for (size_t f = 0; f < frames; ++f) CHECK(same_bits(a[f], b[f]) && same_bits(a[f], c[f]));
        std::vector<std::vector<Complex32>> few(2);
        CHECK(multi.stft(sig, window, 64, few) == Result::Err(FftError::MismatchedLengths));
        CHECK(multi.stft(sig, window, 0, b) == Result::Err(FftError::InvalidHopSize));
        // batches and real rows shard the same way, no exchange (fft.rs:2156-2175, rfft.rs:264-282)
        std::vector<Complex32> flat(5 * 64);
        for (size_t i = 0; i < flat.size(); ++i) flat[i] = Complex32(std::sin(0.7f * (float)i), std::cos(0.2f * (float)i));
        std::vector<Complex32> flat_want;
        for (size_t r = 0; r < 5; ++r) {
            auto w = oracle_fft(std::vector<Complex32>(flat.begin() + r * 64, flat.begin() + (r + 1) * 64));
            flat_want.insert(flat_want.end(), w.begin(), w.end());
        }
        multi.fft_batch(flat, 64).unwrap();
        CHECK(same_bits(flat, flat_want));
        const HipMulti::Timing t = multi.last_timing();
        CHECK(t.kernel_ms > 0.0f && t.upload_ms > 0.0f && t.download_ms > 0.0f && t.gather_ms == 0.0f);
        std::vector<float> rows(3 * 256);
        for (size_t i = 0; i < rows.size(); ++i) rows[i] = std::sin(0.11f * (float)i);
        std::vector<Complex32> spec;
        multi.rfft_batch(rows, 256, spec, &window).unwrap();
        CHECK(spec.size() == 3 * 129);
        std::vector<float> odd(3 * 255);
        CHECK(multi.rfft_batch(odd, 255, spec) == Result::Err(FftError::InvalidValue));
        CHECK(multi.shard(10, 0).first == 0 && multi.shard(10, 0).second == 10);
    }
    std::printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail == 0 ? 0 : 1;
}
