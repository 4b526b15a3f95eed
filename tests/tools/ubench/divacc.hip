// Accuracy of v_rcp_f64 / v_rsq_f64 and of the Newton-refined quotients and square roots the ESACF fit kernels use
// (csrc/mpx_lm.hpp: lm_div, lm_sqrt), against the correctly rounded results, in ulp, over 2^24 random operands.
//   hipcc --offload-arch=gfx950 -O3 -o build_tmp/divacc tests/tools/ubench/divacc.hip && build_tmp/divacc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
#pragma clang fp contract(off)
__device__ double ulps(double got, double want) {
    if (got == want) return 0.0;
    int e;
    frexp(want, &e);
    return fabs(got - want) / ldexp(1.0, e - 53);
}
__global__ void k(const double* a, const double* b, int n, double* worst) {
    // worst[0..2]: rcp raw / 1 step / 2 steps (error of 1/b);  [3..4]: a/b with 1 / 2 steps, no residual correction; [8]: one step and the residual step (lm_div);
    // [5..7]: rsq raw, sqrt with one coupled step, with the residual step on top (lm_sqrt)
    double w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = a[i], d = b[i];
        double y = __builtin_amdgcn_rcp(d);
        w[0] = fmax(w[0], ulps(y, 1.0 / d));
        y = fma(y, fma(-d, y, 1.0), y);
        w[1] = fmax(w[1], ulps(y, 1.0 / d));
        w[3] = fmax(w[3], ulps(x * y, x / d));
        {
            const double q = x * y;
            w[8] = fmax(w[8], ulps(fma(fma(-d, q, x), y, q), x / d));   // lm_div: one step + the residual's share
        }
        y = fma(y, fma(-d, y, 1.0), y);
        w[2] = fmax(w[2], ulps(y, 1.0 / d));
        w[4] = fmax(w[4], ulps(x * y, x / d));
        const double p = fabs(d);
        const double r0 = __builtin_amdgcn_rsq(p);
        w[5] = fmax(w[5], ulps(r0, 1.0 / sqrt(p)));
        double g = p * r0, h = 0.5 * r0;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        w[6] = fmax(w[6], ulps(g, sqrt(p)));
        g = fma(fma(-g, g, p), h, g);
        w[7] = fmax(w[7], ulps(g, sqrt(p)));
    }
    for (int j = 0; j < 9; ++j) {
        // (positive doubles order like their bit patterns)
        atomicMax(reinterpret_cast<unsigned long long*>(worst + j), (unsigned long long)__double_as_longlong(w[j]));
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<double> a(n), b(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        a[i] = ldexp(1.0 + (double)(rnd() >> 11) / 9007199254740992.0, (int)(rnd() % 200) - 100) * ((rnd() & 1) ? 1 : -1);
        b[i] = ldexp(1.0 + (double)(rnd() >> 11) / 9007199254740992.0, (int)(rnd() % 200) - 100) * ((rnd() & 1) ? 1 : -1);
    }
    double *da, *db, *dw;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dw, 72);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(dw, 0, 72);
    k<<<1024, 256>>>(da, db, n, dw);
    double w[9];
    hipMemcpy(w, dw, 72, hipMemcpyDeviceToHost);
    printf("worst error in ulp over %d operands\n", n);
    printf("v_rcp_f64 %.3g   + 1 Newton step %.3g   + 2 steps %.3g\n", w[0], w[1], w[2]);
    printf("a * rcp(b): 1 step %.3g   2 steps %.3g   1 step + residual step (lm_div) %.3g\n", w[3], w[4], w[8]);
    printf("v_rsq_f64 %.3g   sqrt: coupled step %.3g   + residual step (lm_sqrt) %.3g\n", w[5], w[6], w[7]);
    return 0;
}
