#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
// CH independent dependent-chains per wave, W waves per workgroup (one workgroup per CU)
template <int CH>
__global__ __launch_bounds__(512, 1) void k(double* out, const float* in) {
    double a[CH];
    for (int i = 0; i < CH; ++i) a[i] = in[threadIdx.x + i];
    const double c = in[3], d = in[5];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = __builtin_fma(a[i], c, d);
    }
    double s = 0; for (int i = 0; i < CH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> void run(int threads, double* out, const float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, in);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, in);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wps = threads / 256.0;
    printf("chains %2d, waves/SIMD %.0f: %.3f ms -> %.2f ns per instruction per wave, SIMD busy %.0f%% of 2 ns/instr\n", CH, wps, ms, 1e6 * ms / (ITER * 16.0),
           100.0 * 2.0 * wps * ITER * 16 / (1e6 * ms));
}
int main() {
    double* out; float* in; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    run<1>(256, out, in); run<2>(256, out, in); run<4>(256, out, in); run<8>(256, out, in);
    run<1>(512, out, in); run<2>(512, out, in); run<4>(512, out, in); run<8>(512, out, in);
    return 0;
}
