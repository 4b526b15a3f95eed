// |X|^0.67 from x = |X|^2 (reference esacf.py:95-101: np.abs(X) ** 0.67 with k fixed at 0.67):  x^K, K = 0.67 / 2.
//
// Until round 3 this was exp(0.335 * log(x)) through the device math library: ~110 fp64 instructions per value, two values per
// bin, 1.18 of the 6.1 ms of sacf_pfa_kernel.  Here: log2 by a 64-entry reciprocal table and a degree-7 log1p, the product
// with K carried with its rounding error (K * e is the big part of the exponent), exp2 by a 65-entry table and a degree-5
// polynomial: ~35 instructions, relative error <= 1.2e-15 against long-double powl over 1e-30 .. 1e30 (tests/test_pow067.py).
// Branch-free: an argument below 1e-290 (|X| < 1e-145: an exact zero, or nothing an audio spectrum can hold) gives the
// exact zero |X|^0.67 of a zero is; inf and NaN are handed through.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef MPX_HD
#define MPX_HD __host__ __device__
#endif

namespace mpx {
namespace p067 {

constexpr double K = 0.5 * 0.67;          // the reference's exponent on |X|, halved exactly for |X|^2
constexpr int TAB_DOUBLES = 64 + 64 + 65; // rc | lc | t2

// rc[i] ~ 1 / (1 + (i + 0.5) / 64) rounded to double; lc[i] = -log2(rc[i]) of the ROUNDED value; t2[j] = 2^((j - 32) / 64)
inline void build_tables(double* tab) {
    for (int i = 0; i < 64; ++i) {
        const double rc = (double)(1.0L / (1.0L + ((long double)i + 0.5L) / 64.0L));
        tab[i] = rc;
        tab[64 + i] = (double)(-log2l((long double)rc));
    }
    for (int j = 0; j <= 64; ++j) tab[128 + j] = (double)exp2l(((long double)j - 32.0L) / 64.0L);
}

MPX_HD inline double pow067(double x, const double* __restrict__ tab) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const int e = (int)(bits >> 52) - 1023;
    const int i = (int)(bits >> 46) & 63;
    const uint64_t mb = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    const double r = fma(m, tab[i], -1.0);                       // m * rc - 1, |r| <= 1/128
    // log2(1 + r) = r * (c1 + r (c2 + ... c7 r^6)), c_k = (-1)^(k+1) / (k ln 2)
    double p = 0.20609929155694092;                              //  1 / (7 ln 2)
    p = fma(p, r, -0.24044917348309775);                         // -1 / (6 ln 2)
    p = fma(p, r, 0.28853900817779268);                          //  1 / (5 ln 2)
    p = fma(p, r, -0.36067376022224085);                         // -1 / (4 ln 2)
    p = fma(p, r, 0.48089834696298783);                          //  1 / (3 ln 2)
    p = fma(p, r, -0.72134752044448170);                         // -1 / (2 ln 2)
    p = fma(p, r, 1.4426950408889634);                           //  1 / ln 2
    const double lm = fma(p, r, tab[64 + i]);                    // log2(m) in [0, 1)
    // y = K (e + lm): K e with its rounding error, so that the integer part can be taken off exactly
    const double ed = (double)e;
    const double ph = K * ed, pl = fma(K, ed, -ph);
    const double klm = K * lm;
    const double n = rint(ph + klm);
    const double f = (ph - n) + (pl + klm);                      // |f| <= 0.5 (+ rounding)
    const double jd = rint(f * 64.0);
    const double g = fma(jd, -0.015625, f);                      // f - j / 64, |g| <= 1/128
    int ji = (int)jd;
    ji = ji < -32 ? -32 : (ji > 32 ? 32 : ji);                   // (only garbage arguments can leave the table: keep the read inside it)
    const double t = tab[128 + 32 + ji];
    // 2^g - 1 = g ln2 (1 + g ln2 / 2 (1 + ...)): degree 5 in g
    double q = 1.3333558146428443e-3;                            // ln2^5 / 120
    q = fma(q, g, 9.6181291076284772e-3);                        // ln2^4 / 24
    q = fma(q, g, 5.5504108664821580e-2);                        // ln2^3 / 6
    q = fma(q, g, 2.4022650695910071e-1);                        // ln2^2 / 2
    q = fma(q, g, 6.9314718055994531e-1);                        // ln2
    const double res = ldexp(fma(t, q * g, t), (int)n);
    return x >= 1e-290 ? (x <= 1e290 ? res : x) : (x == x ? 0.0 : x);   // (selects; NaN fails every comparison and is returned)
}

}  // namespace p067
}  // namespace mpx
