#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../../chord-detection_amd/csrc/mpx_he_wave.hpp"
namespace mpx { int set_error(mpx_ctx*, int code, const char*, ...) { return code; } }
using namespace mpx;
constexpr int HW_ROW = 264;   // the first transpose layout: one 264-byte row per (k1, parity)
#define ITER 200
// MODE bit0: VALU block (fft32), bit1: LDS transposition block
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(double* out, const float* in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* xbuf = smem + wave * HW_XBUF;
    const int half = lane >> 5, lp = lane & 31;
    const int wr_base = half * HW_ROW + lp * 8, rd_base = lane * HW_ROW;
    cx<double> z[32];
    for (int i = 0; i < 32; ++i) z[i] = {(double)in[threadIdx.x + i], (double)in[threadIdx.x + 32 + i]};
    for (int it = 0; it < ITER; ++it) {
        if (MODE & 1) hw_fft32(z);
        hw_phase();
        if (MODE & 2) {
            cx<double> b[32];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                *reinterpret_cast<double*>(xbuf + wr_base + hw_br5(p) * HW_ROW) = z[p].x;
                *reinterpret_cast<double*>(xbuf + wr_base + (hw_br5(p) + 32) * HW_ROW) = z[p + 16].x;
            }
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < 32; ++c) b[c].x = *reinterpret_cast<const double*>(xbuf + rd_base + 8 * c);
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                *reinterpret_cast<double*>(xbuf + wr_base + hw_br5(p) * HW_ROW) = z[p].y;
                *reinterpret_cast<double*>(xbuf + wr_base + (hw_br5(p) + 32) * HW_ROW) = z[p + 16].y;
            }
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < 32; ++c) b[c].y = *reinterpret_cast<const double*>(xbuf + rd_base + 8 * c);
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < 32; ++c) z[c] = b[c];
        }
        hw_phase();
    }
    double s = 0; for (int i = 0; i < 32; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int threads, double* out, const float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = k<MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * HW_XBUF);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 8 * HW_XBUF, 0, out, in);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 8 * HW_XBUF, 0, out, in);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1e3f * ms / ITER;
}
int main() {
    double* out; float* in; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 8192); hipMemset(in, 0, 8192);
    for (int t : {64, 256, 512}) {
        const float v = run<1>(t, out, in), l = run<2>(t, out, in), b = run<3>(t, out, in);
        printf("%d waves/CU: VALU block %.3f us, LDS block %.3f us, both %.3f us per iteration (sum %.3f)\n", t / 64, v, l, b, v + l);
    }
    return 0;
}
