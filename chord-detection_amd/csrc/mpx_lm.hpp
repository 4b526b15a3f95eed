// Levenberg-Marquardt gaussian peak fit, one GPU thread per peak, fp64.
//
// Replaces peakutils.interpolate -> gaussian_fit -> scipy.optimize.curve_fit
// (reference esacf.py:60), i.e. MINPACK lmdif with forward-difference
// jacobian, ftol = xtol = 1.49012e-8, gtol = 0, maxfev = 200*(n+1), factor 100.
// Restated from MINPACK's published algorithm (More 1978: lmdif / lmpar /
// qrfac / qrsolv / fdjac2); the same restatement in NumPy is the oracle
// (oracle/thirdparty.py) and that one is cross-checked against scipy's MINPACK.
#pragma once
#include <hip/hip_runtime.h>

#define MPX_HD __host__ __device__

// Everything in this header is compiled with floating-point contraction OFF and its multiply-adds spelled as fma():
// the two GPU kernels that inline these functions (peakfit_kernel and coopfit_kernel, csrc/mpx_esacf.hip) must round
// identically, and which a*b+c the compiler fuses must not depend on the code around the inlined copy.
#pragma clang fp contract(off)

namespace mpx {
namespace lm {

constexpr int MAXM = 21;  // 2*width+1 samples, width = 10 (peakutils default)
constexpr int NP = 3;     // ampl, center, dev
constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;

// Quotients and square roots of the device code.  The compiler's IEEE sequences are 12 (division: two v_div_scale, v_rcp, seven
// multiply-adds, v_div_fmas, v_div_fixup) and 17 instructions (square root: range scaling, v_rsq, nine multiply-adds, special cases);
// a trip of the fit kernels runs ~90 divisions and ~35 square roots, a quarter of its instructions (a third in the cooperative
// kernels, where they also sit on the dependent path).  These are the hardware estimate (25 bits), ONE Newton step and the
// residual step -- q = a y, then q + (a - b q) y; g = x y, then g + (x - g^2) h -- 7 and 10 instructions: correctly rounded on
// every operand pair tried over 2^-300 ... 2^300 (tests/test_gpu_lm_div_sqrt.py holds that bit for bit), with zero, infinite and
// NaN operands answered as IEEE does (v_div_fixup; the select in lm_sqrt).
// NO RANGE SCALING.  What that costs, measured on MI355X and asserted (not masked) by the same test:
//   (1) a finite a / b whose quotient overflows: the residual step computes inf - inf = NaN, but v_div_fixup_f64 looks at the
//       operands' exponents and returns +-inf -- as IEEE does;
//   (2) a SUBNORMAL divisor (|b| < 2^-1022): v_rcp_f64 returns +-inf, the Newton step NaN, and v_div_fixup_f64 treats the divisor
//       as zero: lm_div returns +-inf (the sign is IEEE's) where IEEE returns a finite quotient when the numerator is small
//       enough (1e-300 / 5e-310 = 2e9).  THE ONE DIVERGENCE FROM IEEE DIVISION;
//   and operands within ~2^-970 of the ends of the exponent range lose bits (the intermediate 1/b or x y underflows).
// (2) needs the range scaling this form exists to avoid (7 instructions where the compiler's IEEE sequence is 12, ~90 quotients on
// the dependent path of a trip).  It is not reachable from a fit that MINPACK would accept: amplitudes are lag-domain heights
// <= 1, widths and centres are lags (|x| < 1e6 even in the runaway fits, which stop at maxfev), the smallest divisors are the
// jacobian steps eps |x| >= 1e-8 |x| and 2 s^2 + 2.2e-16; a fit whose state reaches 1e-308 is one the reference drops, and an
// infinite quotient drops it here as well (no convergence test passes on inf or NaN).
MPX_HD inline double lm_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);       // 25 bits
    y = fma(y, fma(-b, y, 1.0), y);           // 1 / b to ~11 ulp
    const double q = a * y;
    return __builtin_amdgcn_div_fixup(fma(fma(-b, q, a), y, q), b, a);   // + the residual's share
#else
    return a / b;
#endif
}
MPX_HD inline double lm_rcp(double b) { return lm_div(1.0, b); }
MPX_HD inline double lm_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return (x == 0.0 || x == __builtin_inf()) ? x : g;   // (x < 0 and NaN: v_rsq returns NaN, and so does this)
#else
    return sqrt(x);
#endif
}

struct Problem {
    double xs[MAXM];
    double ys[MAXM];
    int m;
};

MPX_HD inline void residual(const Problem& pr, const double* p, double* out) {
    const double den = 2.0 * p[2] * p[2] + EPSMCH;  // peakutils: 2*dev**2 + eps
    for (int i = 0; i < pr.m; ++i) {
        const double d = pr.xs[i] - p[1];
        out[i] = p[0] * exp(-(d * d) / den) - pr.ys[i];
    }
}

MPX_HD inline double enorm(const double* v, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}
MPX_HD inline double enorm3(const double* v) { return lm_sqrt(fma(v[2], v[2], fma(v[1], v[1], v[0] * v[0]))); }

// a: [m][NP] row-major, modified in place.
MPX_HD inline void qrfac(double* a, int m, int* ipvt, double* rdiag, double* acnorm) {
    double wa[NP];
    for (int j = 0; j < NP; ++j) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += a[i * NP + j] * a[i * NP + j];
        acnorm[j] = sqrt(s);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    const int minmn = m < NP ? m : NP;
    for (int j = 0; j < minmn; ++j) {
        int kmax = j;
        for (int k = j; k < NP; ++k)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; ++i) {
                const double t = a[i * NP + j];
                a[i * NP + j] = a[i * NP + kmax];
                a[i * NP + kmax] = t;
            }
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            const int t = ipvt[j];
            ipvt[j] = ipvt[kmax];
            ipvt[kmax] = t;
        }
        double s = 0.0;
        for (int i = j; i < m; ++i) s += a[i * NP + j] * a[i * NP + j];
        double ajnorm = sqrt(s);
        if (ajnorm != 0.0) {
            if (a[j * NP + j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) a[i * NP + j] /= ajnorm;
            a[j * NP + j] += 1.0;
            for (int k = j + 1; k < NP; ++k) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += a[i * NP + j] * a[i * NP + k];
                const double temp = sum / a[j * NP + j];
                for (int i = j; i < m; ++i) a[i * NP + k] -= temp * a[i * NP + j];
                if (rdiag[k] != 0.0) {
                    const double t = a[j * NP + k] / rdiag[k];
                    const double u = 1.0 - t * t;
                    rdiag[k] *= sqrt(u > 0.0 ? u : 0.0);
                    const double q = rdiag[k] / wa[k];
                    if (0.05 * q * q <= EPSMCH) {
                        double s2 = 0.0;
                        for (int i = j + 1; i < m; ++i) s2 += a[i * NP + k] * a[i * NP + k];
                        rdiag[k] = sqrt(s2);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// Register-only helpers: every array below is indexed with compile-time constants after unrolling; the
// few data-dependent indices (pivot permutation, rank) go through selects.  On the GPU a dynamically
// indexed private array lives in scratch memory, and lmpar's dependent chains then run at DRAM latency.
template <typename T>
MPX_HD inline T sel3(const T* a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
template <typename T>
MPX_HD inline void put3(T* a, int i, T v) {
    a[0] = i == 0 ? v : a[0];
    a[1] = i == 1 ? v : a[1];
    a[2] = i == 2 ? v : a[2];
}

// r: [NP][NP] row-major (upper triangle = R); lower triangle is scratch.
MPX_HD inline void qrsolv(double* r, const int* ipvt, const double* diag, const double* qtb, double* x,
                          double* sdiag) {
    double wa[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int i = j; i < NP; ++i) r[i * NP + j] = r[j * NP + i];
        x[j] = r[j * NP + j];
        wa[j] = qtb[j];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const double dl = sel3(diag, ipvt[j]);
        if (dl != 0.0) {
#pragma unroll
            for (int k = j; k < NP; ++k) sdiag[k] = 0.0;
            sdiag[j] = dl;
            double qtbpj = 0.0;
#pragma unroll
            for (int k = j; k < NP; ++k) {
                if (sdiag[k] != 0.0) {
                    // MINPACK's two cases (cotangent when |r_kk| < |sdiag_k|, tangent otherwise) as ONE instruction
                    // stream with the operands selected: the same operations per case, but a wave whose lanes
                    // disagree no longer runs both (2 divisions + 1 square root each)
                    const double rkk = r[k * NP + k], sk = sdiag[k];
                    const bool small = fabs(rkk) < fabs(sk);
                    const double t = lm_div(small ? rkk : sk, small ? sk : rkk);
#if defined(__HIP_DEVICE_COMPILE__)
                    // |t| <= 1, so the radicand lies in [0.25, 0.5]: no range scaling needed, and the hardware
                    // reciprocal square root + two Newton steps (~1 ulp) replaces a full sqrt and a full division
                    const double wq = fma(0.25 * t, t, 0.25);
                    double yq = __builtin_amdgcn_rsq(wq);
                    yq = fma(0.5 * yq, fma(-wq * yq, yq, 1.0), yq);
                    yq = fma(0.5 * yq, fma(-wq * yq, yq, 1.0), yq);
                    const double c0 = 0.5 * yq;
#else
                    const double c0 = 0.5 / sqrt(0.25 + 0.25 * t * t);
#endif
                    const double c1 = c0 * t;
                    const double sn = small ? c0 : c1, cs = small ? c1 : c0;
                    r[k * NP + k] = fma(cs, r[k * NP + k], sn * sdiag[k]);
                    const double temp = fma(cs, wa[k], sn * qtbpj);
                    qtbpj = fma(-sn, wa[k], cs * qtbpj);
                    wa[k] = temp;
#pragma unroll
                    for (int i = k + 1; i < NP; ++i) {
                        const double t = fma(cs, r[i * NP + k], sn * sdiag[i]);
                        sdiag[i] = fma(-sn, r[i * NP + k], cs * sdiag[i]);
                        r[i * NP + k] = t;
                    }
                }
            }
        }
        sdiag[j] = r[j * NP + j];
        r[j * NP + j] = x[j];
    }
    int nsing = NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (sdiag[j] == 0.0 && nsing == NP) nsing = j;
        if (nsing < NP) wa[j] = 0.0;
    }
#pragma unroll
    for (int j = NP - 1; j >= 0; --j) {
        if (j < nsing) {
            double s = 0.0;
#pragma unroll
            for (int i = j + 1; i < NP; ++i)
                if (i < nsing) s = fma(r[i * NP + j], wa[i], s);
            wa[j] = lm_div(wa[j] - s, sdiag[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) put3(x, ipvt[j], wa[j]);
}

MPX_HD inline double lmpar(double* r, const int* ipvt, const double* diag, const double* qtb, double delta,
                           double par, double* x, double* sdiag, int* iters = nullptr) {
    double wa1[NP], wa2[NP];
    int nsing = NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        wa1[j] = qtb[j];
        if (r[j * NP + j] == 0.0 && nsing == NP) nsing = j;
        if (nsing < NP) wa1[j] = 0.0;
    }
#pragma unroll
    for (int j = NP - 1; j >= 0; --j) {
        if (j < nsing) {
            wa1[j] = lm_div(wa1[j], r[j * NP + j]);
            const double temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; ++i) wa1[i] = fma(-r[i * NP + j], temp, wa1[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) put3(x, ipvt[j], wa1[j]);
#pragma unroll
    for (int j = 0; j < NP; ++j) sdiag[j] = 0.0;
    int it = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm3(wa2);
    double fp = dxnorm - delta;
    if (iters) *iters = 0;   // (statistics of development builds; dead code otherwise)
    if (fp <= 0.1 * delta) return 0.0;
    double parl = 0.0;
    if (nsing >= NP) {
        {
#if defined(__HIP_DEVICE_COMPILE__)
            const double inv_dx = lm_rcp(dxnorm);  // one division for the three quotients (1 ulp apart)
#endif
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int l = ipvt[j];
#if defined(__HIP_DEVICE_COMPILE__)
                wa1[j] = sel3(diag, l) * (sel3(wa2, l) * inv_dx);
#else
                wa1[j] = sel3(diag, l) * (sel3(wa2, l) / dxnorm);
#endif
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < j; ++i) s = fma(r[i * NP + j], wa1[i], s);
            wa1[j] = lm_div(wa1[j] - s, r[j * NP + j]);
        }
        const double temp = enorm3(wa1);
#if defined(__HIP_DEVICE_COMPILE__)
        parl = lm_div(fp, delta * temp * temp);
#else
        parl = ((fp / delta) / temp) / temp;
#endif
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i <= j; ++i) s = fma(r[i * NP + j], qtb[i], s);
        wa1[j] = lm_div(s, sel3(diag, ipvt[j]));
    }
    const double gnorm = enorm3(wa1);
    double paru = lm_div(gnorm, delta);
    if (paru == 0.0) paru = lm_div(DWARF, delta < 0.1 ? delta : 0.1);
    par = par > parl ? par : parl;
    par = par < paru ? par : paru;
    if (par == 0.0) par = lm_div(gnorm, dxnorm);
    for (;;) {
        ++it;
        if (par == 0.0) par = DWARF > 0.001 * paru ? DWARF : 0.001 * paru;
        double temp = lm_sqrt(par);
#pragma unroll
        for (int j = 0; j < NP; ++j) wa1[j] = temp * diag[j];
        qrsolv(r, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < NP; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm3(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || it == 10) break;
        {
#if defined(__HIP_DEVICE_COMPILE__)
            const double inv_dx = lm_rcp(dxnorm);  // one division for the three quotients (1 ulp apart)
#endif
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int l = ipvt[j];
#if defined(__HIP_DEVICE_COMPILE__)
                wa1[j] = sel3(diag, l) * (sel3(wa2, l) * inv_dx);
#else
                wa1[j] = sel3(diag, l) * (sel3(wa2, l) / dxnorm);
#endif
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            wa1[j] = lm_div(wa1[j], sdiag[j]);
            const double t = wa1[j];
#pragma unroll
            for (int i = j + 1; i < NP; ++i) wa1[i] = fma(-r[i * NP + j], t, wa1[i]);
        }
        temp = enorm3(wa1);
#if defined(__HIP_DEVICE_COMPILE__)
        const double parc = lm_div(fp, delta * temp * temp);
#else
        const double parc = ((fp / delta) / temp) / temp;
#endif
        if (fp > 0.0) parl = parl > par ? parl : par;
        if (fp < 0.0) paru = paru < par ? paru : par;
        par = parl > par + parc ? parl : par + parc;
    }
    if (iters) *iters = it;
    return par;
}

// Returns MINPACK's info code (1..4 = converged); *center = fitted centre.
MPX_HD inline int gaussian_fit(const Problem& pr, double* center) {
    const int m = pr.m;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (NP + 1);
    double x[NP];
    double ymax = pr.ys[0];
    for (int i = 1; i < m; ++i) ymax = pr.ys[i] > ymax ? pr.ys[i] : ymax;
    x[0] = ymax;                           // peakutils initial guess: [max(y), x[0], 5*(x[1]-x[0])]
    x[1] = pr.xs[0];
    x[2] = (pr.xs[1] - pr.xs[0]) * 5.0;
    double fvec[MAXM], fnew[MAXM], wa4[MAXM], fjac[MAXM * NP];
    residual(pr, x, fvec);
    int nfev = 1;
    double fnorm = enorm(fvec, m);
    double par = 0.0;
    int it = 1, info = 0;
    const double eps = sqrt(EPSMCH);
    double diag[NP] = {1.0, 1.0, 1.0};
    double delta = 0.0, xnorm = 0.0;
    int ipvt[NP];
    double wa1[NP], wa2[NP], wa3[NP], qtf[NP], p[NP], xnew[NP], sd[NP];
    for (;;) {
        for (int j = 0; j < NP; ++j) {
            const double temp = x[j];
            double h = eps * fabs(temp);
            if (h == 0.0) h = eps;
            x[j] = temp + h;
            residual(pr, x, wa4);
            x[j] = temp;
            for (int i = 0; i < m; ++i) fjac[i * NP + j] = (wa4[i] - fvec[i]) / h;
        }
        nfev += NP;
        qrfac(fjac, m, ipvt, wa1, wa2);
        if (it == 1) {
            for (int j = 0; j < NP; ++j) {
                diag[j] = wa2[j] != 0.0 ? wa2[j] : 1.0;
                wa3[j] = diag[j] * x[j];
            }
            xnorm = enorm(wa3, NP);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        for (int j = 0; j < NP; ++j) {
            if (fjac[j * NP + j] != 0.0) {
                double s = 0.0;
                for (int i = j; i < m; ++i) s += fjac[i * NP + j] * wa4[i];
                const double temp = -s / fjac[j * NP + j];
                for (int i = j; i < m; ++i) wa4[i] += fjac[i * NP + j] * temp;
            }
            fjac[j * NP + j] = wa1[j];
            qtf[j] = wa4[j];
        }
        double gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < NP; ++j) {
                const int l = ipvt[j];
                if (wa2[l] != 0.0) {
                    double s = 0.0;
                    for (int i = 0; i <= j; ++i) s += fjac[i * NP + j] * (qtf[i] / fnorm);
                    const double g = fabs(s / wa2[l]);
                    gnorm = g > gnorm ? g : gnorm;
                }
            }
        }
        if (gnorm <= gtol) {
            info = 4;
            break;
        }
        for (int j = 0; j < NP; ++j) diag[j] = diag[j] > wa2[j] ? diag[j] : wa2[j];
        double r[NP * NP], rr[NP * NP];
        for (int i = 0; i < NP; ++i)
            for (int j = 0; j < NP; ++j) r[i * NP + j] = fjac[i * NP + j];
        for (;;) {
            for (int i = 0; i < NP * NP; ++i) rr[i] = r[i];
            par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd);
            for (int j = 0; j < NP; ++j) {
                p[j] = -p[j];
                xnew[j] = x[j] + p[j];
                wa3[j] = diag[j] * p[j];
            }
            const double pnorm = enorm(wa3, NP);
            if (it == 1) delta = delta < pnorm ? delta : pnorm;
            residual(pr, xnew, fnew);
            ++nfev;
            const double fnorm1 = enorm(fnew, m);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
            for (int j = 0; j < NP; ++j) {
                const double temp = p[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += r[i * NP + j] * temp;
            }
            const double temp1 = enorm(wa3, NP) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            const double ratio = prered != 0.0 ? actred / prered : 0.0;
            if (ratio <= 0.25) {
                double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                const double dm = delta < pnorm / 0.1 ? delta : pnorm / 0.1;
                delta = temp * dm;
                par = par / temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < NP; ++j) {
                    x[j] = xnew[j];
                    wa3[j] = diag[j] * x[j];
                }
                for (int i = 0; i < m; ++i) fvec[i] = fnew[i];
                xnorm = enorm(wa3, NP);
                fnorm = fnorm1;
                ++it;
            }
            const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (c1) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (c1 && info == 2) info = 3;
            if (info != 0) break;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) break;
            if (ratio >= 1e-4) break;
        }
        if (info != 0) break;
    }
    *center = x[1];
    return info;
}

}  // namespace lm
}  // namespace mpx

#pragma clang fp contract(fast)  // the compiler's default for HIP, for whatever follows the include
