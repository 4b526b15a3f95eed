// Standalone check + timing of he_wave_kernel (csrc/mpx_he_wave.hpp) against a long-double host DFT of the same frames.
// Not part of the library or the test suite; build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/he_wave_check tests/tools/he_wave_check.hip && /tmp/he_wave_check
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../chord-detection_amd/csrc/mpx_he_wave.hpp"

namespace mpx {
int set_error(mpx_ctx*, int code, const char*, ...) { return code; }
}
using namespace mpx;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                          \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void swap_probe(unsigned* out) { out[threadIdx.x] = threadIdx.x; }

// the library's second launch (sum_all_kernel in mpx_he.hip): rows summed in a fixed order
__global__ void sum_rows(const double* rows, long long n, double* out) {
    if (threadIdx.x < 12) {
        double t = 0.0;
        for (long long r = 0; r < n; ++r) t += rows[r * 12 + threadIdx.x];
        out[threadIdx.x] = t;
    }
}

static void host_fft(std::vector<long double>& re, std::vector<long double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1)
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * M_PIl * k / len;
                const long double c = cosl(ang), s = sinl(ang);
                const long double xr = re[i + k + len / 2] * c - im[i + k + len / 2] * s;
                const long double xi = re[i + k + len / 2] * s + im[i + k + len / 2] * c;
                re[i + k + len / 2] = re[i + k] - xr; im[i + k + len / 2] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
}

int main(int argc, char** argv) {
    #ifndef HWC_WAVES
#define HWC_WAVES 8
#endif
    constexpr int N = 4096, M = 2048, HOP = 1024, WAVES = HWC_WAVES;
    const long long F = argc > 1 ? atoll(argv[1]) : 8192;
    const int fs = getenv("HWC_FS") ? atoi(getenv("HWC_FS")) : 44100;
    unsigned* d_probe; CK(hipMalloc(&d_probe, 128 * 4));
    hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, d_probe);

    // plan (harmonic_energy.py:33-57, defaults)
    std::vector<int> k0, k1; std::vector<double> ww;
    const double c3 = 440.0 * std::pow(2.0, (48.0 - 69.0) / 12.0), div = (fs / 4.0) / N;
    for (int n = 0; n < 12; ++n)
        for (int oct = 1; oct <= 2; ++oct)
            for (int h = 1; h <= 2; ++h) {
                const double kp = std::nearbyint(c3 * std::pow(2.0, n / 12.0) * oct * h / div);
                k0.push_back((int)(kp - 2 * h)); k1.push_back((int)(kp + 2 * h)); ww.push_back(1.0 / h);
            }
    std::vector<int> bins;
    for (size_t w = 0; w < k0.size(); ++w) for (int k = k0[w]; k < k1[w]; ++k) bins.push_back(k);
    std::sort(bins.begin(), bins.end()); bins.erase(std::unique(bins.begin(), bins.end()), bins.end());
    std::vector<int> c0(k0.size()), c1(k0.size());
    for (size_t w = 0; w < k0.size(); ++w) {
        c0[w] = (int)(std::lower_bound(bins.begin(), bins.end(), k0[w]) - bins.begin());
        c1[w] = c0[w] + (k1[w] - k0[w]);
    }
    const int nb = (int)bins.size(), nwin = (int)k0.size();
    printf("nb %d nwin %d bins [%d, %d]\n", nb, nwin, bins.front(), bins.back());
    std::vector<unsigned> slots(2 * nb); std::vector<cx<double>> twnb(nb), tw(M);
    for (int i = 0; i < nb; ++i) {
        const int kp = bins[i] & 1023, km = (1024 - kp) & 1023;
        slots[2 * i] = (unsigned)hw_slot(0, kp) | ((unsigned)hw_slot(0, km) << 16);
        slots[2 * i + 1] = (unsigned)hw_slot(1, kp) | ((unsigned)hw_slot(1, km) << 16);
        const long double ang = -2.0L * M_PIl * bins[i] / N;
        twnb[i] = {(double)cosl(ang), (double)sinl(ang)};
    }
    for (int j = 0; j < M; ++j) { const long double ang = -2.0L * M_PIl * j / M; tw[j] = {(double)cosl(ang), (double)sinl(ang)}; }
    std::vector<double> whalf(M);
    for (int i = 0; i < M; ++i) whalf[i] = (double)(0.54L - 0.46L * cosl(2.0L * M_PIl * i / (N - 1)));

    const long long n = (F - 1) * HOP + N;
    std::vector<float> x(n);
    unsigned s = 12345;
    for (long long i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        x[i] = (float)(0.3 * sin(2 * M_PI * 440.0 * i / fs) + 0.2 * sin(2 * M_PI * 277.18 * i / fs) + 0.05 * ((s >> 8) / 16777216.0 - 0.5));
    }
    auto up = [](const void* h, size_t b) { void* d = nullptr; hipMalloc(&d, b); hipMemcpy(d, h, b, hipMemcpyHostToDevice); return d; };
    HeWaveArgs a{};
    a.sig = (const float*)up(x.data(), n * 4); a.n = n; a.desc = nullptr; a.num_frames = F; a.hop = HOP;
    a.whalf = (const double*)up(whalf.data(), M * 8); a.tw = (const cx<double>*)up(tw.data(), M * 16);
    a.wk0 = (const int*)up(c0.data(), nwin * 4); a.wk1 = (const int*)up(c1.data(), nwin * 4); a.ww = (const double*)up(ww.data(), nwin * 8);
    a.slots = (const unsigned*)up(slots.data(), nb * 8); a.twnb = (const cx<double>*)up(twnb.data(), nb * 16);
    unsigned k2 = 0;
    for (int k : bins) k2 |= hw_k2_bits(k);
#ifndef HWC_K2
#define HWC_K2 HW_K2_44K
#endif
    printf("k2mask of the plan %08x (%d of 32 rows), kernel instantiated for %08x\n", k2, __builtin_popcount(k2), (unsigned)HWC_K2);
    if (k2 & ~(unsigned)HWC_K2) { printf("plan needs rows outside the instantiated mask\n"); return 1; }
    a.nb = nb; a.nwin = nwin; a.wins_per_note = 4; a.num_harmonic = 2; a.quad_tail = getenv("HWC_GENERIC_TAIL") ? 0 : 1;
    double* d_out; CK(hipMalloc(&d_out, F * 12 * 8)); a.out = d_out; a.partial = nullptr;
    const long long G = std::min<long long>(256, (F + WAVES - 1) / WAVES) * (getenv("HWC_GMUL") ? atoi(getenv("HWC_GMUL")) : 1);
    const size_t lds = hw_shared_bytes(4, nwin) + (size_t)WAVES * HW_XBUF;
    printf("grid %lld x %d, LDS %zu B\n", G, WAVES * 64, lds);
    const long long FD = std::min<long long>(F, 24);
    cx<double>* d_dbg; CK(hipMalloc(&d_dbg, (size_t)F * M * 16));
    auto kd = he_wave_kernel<WAVES, 4, true, true, HWC_K2>; auto kr = he_wave_kernel<WAVES, 4, false, true, HWC_K2>;
    CK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kd, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, d_dbg);
    CK(hipDeviceSynchronize());
    std::vector<cx<double>> Z((size_t)F * M); CK(hipMemcpy(Z.data(), d_dbg, Z.size() * 16, hipMemcpyDeviceToHost));
    std::vector<double> out(F * 12); CK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
    double zerr = 0, zmax = 0, cerr = 0;
    const long long check[] = {0, 1, 3, 4, 31, 32, 33, F / 2, F - 2, F - 1};
    for (long long f : check) {
        if (f < 0 || f >= F) continue;
        std::vector<long double> re(M), im(M);
        for (int m = 0; m < M; ++m) {
            const int n0 = 2 * m, n1 = 2 * m + 1;
            const double w0 = n0 < M ? whalf[n0] : whalf[N - 1 - n0], w1 = n1 < M ? whalf[n1] : whalf[N - 1 - n1];
            re[m] = (double)x[f * HOP + n0] * w0; im[m] = (double)x[f * HOP + n1] * w1;
        }
        {
            // the kernel's intermediate: ZA = FFT_1024(x[4m] + i x[4m+1]), ZB = FFT_1024(x[4m+2] + i x[4m+3])
            for (int par = 0; par < 2; ++par) {
                std::vector<long double> r2(1024), i2(1024);
                for (int m = 0; m < 1024; ++m) { r2[m] = re[2 * m + par]; i2[m] = im[2 * m + par]; }
                host_fft(r2, i2);
                for (int k = 0; k < 1024; ++k) {
                    const cx<double> g = Z[(f * 2 + par) * 1024 + k];
                    zerr = std::max<double>(zerr, std::hypot((double)(r2[k] - g.x), (double)(i2[k] - g.y)));
                    zmax = std::max<double>(zmax, std::hypot((double)r2[k], (double)i2[k]));
                }
            }
        }
        host_fft(re, im);
        // chroma from the host spectrum
        std::vector<double> mag(nb);
        for (int i = 0; i < nb; ++i) {
            const int k = bins[i], ka = k & (M - 1), kb = (M - k) & (M - 1);
            const long double ax = re[ka], ay = im[ka], bx = re[kb], by = -im[kb];
            const long double ex = 0.5L * (ax + bx), ey = 0.5L * (ay + by), dx = 0.5L * (ax - bx), dy = 0.5L * (ay - by);
            const long double ang = -2.0L * M_PIl * k / N, c = cosl(ang), sn = sinl(ang);
            const long double tx = c * dx - sn * dy, ty = c * dy + sn * dx;   // W D
            const long double X = ex + ty, Y = ey - tx;                         // E - i W D
            mag[i] = (double)sqrtl(sqrtl(X * X + Y * Y));
        }
        for (int nn = 0; nn < 12; ++nn) {
            double chroma = 0;
            for (int oc = 0; oc < 4; oc += 2) {
                double ns = 0;
                for (int h = 0; h < 2; ++h) {
                    const int wi = nn * 4 + oc + h;
                    double m = -INFINITY;
                    for (int k = c0[wi]; k < c1[wi]; ++k) m = std::max(m, mag[k]);
                    ns += m * ww[wi];
                }
                chroma += ns;
            }
            cerr = std::max(cerr, std::fabs(chroma - out[f * 12 + nn]) / std::max(1e-300, std::fabs(chroma)));
        }
    }
    printf("spectrum: max |Z - Zref| = %.3e (max |Z| %.3e); chroma max rel err %.3e\n", zerr, zmax, cerr);
    {
        // run to run: the frames go to whichever wave asks first, the rows must not care
        std::vector<double> out2(F * 12), out3(F * 12);
        CK(hipMemset(d_out, 0, F * 96));
        hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, nullptr);
        CK(hipMemcpy(out2.data(), d_out, out2.size() * 8, hipMemcpyDeviceToHost));
        size_t diff = 0;
        double rel = 0;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipMemset(d_out, 0, F * 96));
            hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, nullptr);
            CK(hipMemcpy(out3.data(), d_out, out3.size() * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < out2.size(); ++i) diff += out3[i] != out2[i];
        }
        for (size_t i = 0; i < out.size(); ++i) rel = std::max(rel, std::fabs(out[i] - out2[i]) / std::fabs(out[i]));
        printf("20 more launches: %zu values differ from the first; debug build vs release build max rel %.2e\n", diff, rel);
    }
    // timing
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, nullptr);
    const int R = 200;
    hipEventRecord(e0, 0);
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, nullptr);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("he_wave_kernel: %.2f us per launch of %lld frames (with per-frame rows)\n", 1e3 * ms / R, F);
#ifdef HW_TRACE
    {
        long long* d_t; CK(hipMalloc(&d_t, 8 * 32 * 16 * 8)); CK(hipMemset(d_t, 0, 8 * 32 * 16 * 8));
        hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, (cx<double>*)d_t);
        CK(hipDeviceSynchronize());
        std::vector<long long> tl(8 * 32 * 16); CK(hipMemcpy(tl.data(), d_t, tl.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = tl[0];
        for (int wv = 0; wv < WAVES; ++wv) t0 = std::min(t0, tl[wv * 512]);
        for (int wv = 0; wv < WAVES; ++wv) {
            int nf = 0;
            while (nf < 32 && tl[(wv * 32 + nf) * 16]) ++nf;
            printf("FRAMES wave %d start %lld, %d frames, durations:", wv, tl[wv * 512] - t0, nf);
            for (int fr = 0; fr + 1 < nf; ++fr) printf(" %lld", tl[(wv * 32 + fr + 1) * 16] - tl[(wv * 32 + fr) * 16]);
            printf(" | last frame ends %lld\n", nf ? tl[(wv * 32 + nf - 1) * 16 + 9] - t0 : 0);
        }
        // phase durations (clocks) of frames 1 .. 3 of waves 0, 4: 0 window+level-1 | 1 transform A | 2 deferred tail |
        // 3 (nothing) | 4 transpose | 5 transform B | 6 bin copy + gathers + prefetch issue | 7 split | 9 magnitudes + window maxima
        for (int wv = 0; wv < WAVES; wv += 4)
            for (int fr = 1; fr < 4; ++fr) {
                const long long* q = &tl[(wv * 32 + fr) * 16];
                if (!q[0]) continue;
                printf("PHASES wave %d frame %d: window %lld | fftA %lld | tail %lld | transpose %lld | fftB %lld | bincopy %lld | split %lld | maxima %lld | total %lld\n",
                       wv, fr, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[7] - q[6], q[9] - q[7], q[9] - q[0]);
            }
    }
#endif
    a.num_frames = 0;
    hipEventRecord(e0, 0);
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kr, dim3((unsigned)G), dim3(WAVES * 64), lds, 0, a, nullptr);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
    printf("he_wave_kernel: %.2f us per launch of 0 frames (prologue only)\n", 1e3 * ms / R);
    hipEventRecord(e0, 0);
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, d_probe);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
    printf("trivial kernel: %.2f us per launch\n", 1e3 * ms / R);
    return 0;
}
