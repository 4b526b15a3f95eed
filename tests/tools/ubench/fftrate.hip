#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../../chord-detection_amd/csrc/mpx_he_wave.hpp"
namespace mpx { int set_error(mpx_ctx*, int code, const char*, ...) { return code; } }
using namespace mpx;
// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (the radix-2 step of the 32 x 2 x 32 variant)
__device__ __forceinline__ void hw_swap_halves(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}
#define ITER 200
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(double* out, const float* in, const cx<double>* tw) {
    cx<double> z[32];
    for (int i = 0; i < 32; ++i) z[i] = {(double)in[threadIdx.x + i], (double)in[threadIdx.x + 32 + i]};
    const cx<double> wl = tw[threadIdx.x & 63];
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0 || MODE == 2) hw_fft32(z);
        if (MODE == 1 || MODE == 2) {
            cx<double> w1 = wl;
            asm volatile("" : "+v"(w1.x), "+v"(w1.y));
            cx<double> pw_ = w1;
#pragma unroll
            for (int kk = 1; kk < 32; ++kk) {
                z[hw_br5(kk)] = cmul(z[hw_br5(kk)], pw_);
                if (kk < 31) pw_ = cmul(pw_, w1);
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                cx<double> X = z[p], Y = z[p + 16];
                hw_swap_halves(X.x, Y.x);
                hw_swap_halves(X.y, Y.y);
                z[p] = cadd(X, Y);
                z[p + 16] = cmul(csub(X, Y), wl);
            }
        }
        hw_phase();
    }
    double s = 0; for (int i = 0; i < 32; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int threads, int instr, double* out, const float* in, const cx<double>* tw) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, in, tw);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, in, tw);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s waves/SIMD %d: %.3f ms -> %.3f us per pass per wave (%d instr at 2 ns = %.3f us)\n", name, threads / 256, ms, 1e3 * ms / ITER, instr, instr * 2e-3);
}
int main() {
    double* out; float* in; cx<double>* tw; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 8192); hipMemset(in, 0, 8192); hipMalloc(&tw, 64 * 16); hipMemset(tw, 0, 64 * 16);
    run<0>("fft32", 256, 456, out, in, tw); run<0>("fft32", 512, 456, out, in, tw);
    run<1>("twiddle", 256, 244, out, in, tw); run<1>("twiddle", 512, 244, out, in, tw);
    run<2>("fft+tw", 256, 700, out, in, tw); run<2>("fft+tw", 512, 700, out, in, tw);
    run<3>("C1", 256, 192, out, in, tw); run<3>("C1", 512, 192, out, in, tw);
    return 0;
}
