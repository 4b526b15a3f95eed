// Host-only: the unit-root tables of the Iterative-F0 summary-spectrum kernels at the tuned frame sizes (if0_plan, mpx_if0.hip).
// Plain C++ (no HIP): tests/test_if0_tables_cpu.py compiles it with g++ -fsanitize=address and holds it to the direct formulas
// bit for bit -- the first version of the shortcut below read past the end of W_2NF, and nothing that runs without a GPU saw it.
#pragma once
#include <cmath>
#include <cstddef>
#include <thread>

namespace mpx {

// tw[2 j], tw[2 j + 1] = cos, sin of -2 pi j / NF for j < NF; twn[2 k], twn[2 k + 1] = cos, sin of -2 pi k / (2 NF) for k <= NF:
// long double arguments and functions, rounded to double -- exactly what rounds 1-5 computed entry by entry (32 770 long double
// sines and cosines at the default frame size: 3.3 of the 5.6 ms by which the FIRST call of a process at a new sample rate was
// slower than the second).  NF a power of two >= 8.  For j < NF/2, W_NF^j = W_2NF^(2 j): the same long double angle (the factor 2
// scales numerator and denominator exactly), hence the same bits -- the lower half of W_NF is copied.  Its upper half is
// computed (a reflection would not round the same way): 1.5 NF angles instead of 2 NF, on four threads.
inline void if0_unit_roots(int NF, double* tw, double* twn) {
    auto fill = [=](int q) {   // quarter q of both tables
        for (int k = q * (NF / 4); k < (q + 1) * (NF / 4) + (q == 3 ? 1 : 0); ++k) {
            const long double ang = -2.0L * M_PIl * k / (long double)(2 * NF);
            twn[2 * (size_t)k] = (double)cosl(ang);
            twn[2 * (size_t)k + 1] = (double)sinl(ang);
        }
        for (int j = NF / 2 + q * (NF / 8); j < NF / 2 + (q + 1) * (NF / 8); ++j) {
            const long double ang = -2.0L * M_PIl * j / (long double)NF;
            tw[2 * (size_t)j] = (double)cosl(ang);
            tw[2 * (size_t)j + 1] = (double)sinl(ang);
        }
    };
    std::thread th[3];
    for (int q = 1; q < 4; ++q) th[q - 1] = std::thread(fill, q);
    fill(0);
    for (auto& t : th) t.join();
    for (int j = 0; j < NF / 2; ++j) {
        tw[2 * (size_t)j] = twn[4 * (size_t)j];
        tw[2 * (size_t)j + 1] = twn[4 * (size_t)j + 1];
    }
}

}  // namespace mpx
