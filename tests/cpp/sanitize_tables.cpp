// sanitize_tables.cpp -- the product's host-side planner recipes (kofft_amd/csrc/tables.cpp) and the C oracle under
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the device pool).
// Built and run by tests/test_sanitizers.py:  g++ -fsanitize=address,undefined ... tables.cpp kofft_oracle.c
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../kofft_amd/csrc/tables.h"
extern "C" {
#include "../../oracle/kofft_oracle.h"
}

int main()
{
    int bad = 0;
    // tables: every size class the library asks for, written into exactly-sized buffers (ASan guards the ends)
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(8), size_t(1024), size_t(4096), size_t(1) << 16}) {
        std::vector<float> t32(n ? n : 1), r32(2 * (n ? n : 1));
        std::vector<double> t64(n ? n : 1), r64(2 * (n ? n : 1));
        kofft_tables::twiddles_f32(n, t32.data());   // n/2 complex = n floats
        kofft_tables::twiddles_f64(n, t64.data());
        kofft_tables::rfft_table_f32(n, r32.data());  // n complex
        kofft_tables::rfft_table_f64(n, r64.data());
        std::vector<float> o32(n ? n : 1);
        if (ko_twiddles_f32(n, o32.data()) == 0 && n >= 2 && std::memcmp(o32.data(), t32.data(), n * sizeof(float)) != 0) ++bad;
        std::vector<float> w(n ? n : 1);
        kofft_tables::hann_f32(n, w.data());
    }
    for (size_t n : {size_t(3), size_t(12), size_t(1000)}) {
        size_t m = 1;
        while (m < 2 * n - 1) m <<= 1;
        std::vector<float> c(2 * n), b(2 * m);
        kofft_tables::bluestein_f32(n, m, c.data(), b.data());
        std::vector<double> cd(2 * n), bd(2 * m);
        kofft_tables::bluestein_f64(n, m, cd.data(), bd.data());
    }
    // oracle: power-of-two, Bluestein, real and STFT entry points on tight buffers
    for (size_t n : {size_t(1), size_t(2), size_t(16), size_t(32), size_t(1024), size_t(12), size_t(1000)}) {
        std::vector<float> x(2 * n, 0.25f);
        if (ko_fft_batch_f32(x.data(), n, 1, 0) != 0) ++bad;
        if (ko_fft_batch_f32(x.data(), n, 1, 1) != 0) ++bad;
        std::vector<double> xd(2 * n, 0.25);
        if (ko_fft_batch_f64(xd.data(), n, 1, 0) != 0) ++bad;
    }
    for (size_t n : {size_t(2), size_t(8), size_t(30), size_t(2048)}) {
        std::vector<float> in(n, 1.0f), out(2 * (n / 2 + 1)), back(n);
        if (ko_rfft_batch_f32(in.data(), out.data(), nullptr, n, 1) != 0) ++bad;
        if (ko_irfft_batch_f32(out.data(), back.data(), n, 1) != 0) ++bad;
    }
    {
        std::vector<float> sig(1000, 0.5f), win(256), spec(2 * 256 * 16), outp(1000, 0.0f), scr(1000), mags(16 * 128);
        ko_hann_f32(256, win.data());
        if (ko_stft_f32(sig.data(), sig.size(), win.data(), 256, 64, spec.data(), 16) != 0) ++bad;
        if (ko_istft_f32(spec.data(), 16, win.data(), 256, 64, outp.data(), 1000, scr.data(), 1000) != 0) ++bad;
        float mx = 0;
        if (ko_stft_magnitudes_f32(sig.data(), sig.size(), 256, 64, mags.data(), &mx) != 0) ++bad;
        if (ko_stft_f32(sig.data(), sig.size(), win.data(), 256, 0, spec.data(), 16) != 5) ++bad;  // InvalidHopSize
    }
    std::printf("sanitize_tables: %d problems\n", bad);
    return bad ? 1 : 0;
}
