#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
template <int OP>
__global__ __launch_bounds__(512, 1) void k(double* out, const float* in) {
    double a[16];
    float f[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + i]; f[i] = in[threadIdx.x + 16 + i]; u[i] = threadIdx.x * (i + 3); }
    const double c = in[3], d = in[5];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], c, d);
            if (OP == 1) a[i] = a[i] * c;
            if (OP == 2) a[i] = a[i] + c;
            if (OP == 3) { a[i] += (double)f[i]; asm volatile("" : "+v"(f[i])); }   // cvt + add
            if (OP == 4) { auto r = __builtin_amdgcn_permlane32_swap(u[i], u[(i + 1) & 15], false, false); u[i] = r[0]; u[(i + 1) & 15] = r[1]; }
            if (OP == 5) { f[i] = f[i] * (float)c + 1.0f; }
            if (OP == 6) { u[i] = __builtin_amdgcn_mov_dpp(u[i], 0x141, 0xf, 0xf, false) + 1; }   // row_half_mirror
            if (OP == 7) { a[i] = __builtin_fma(a[i], c, d); f[i] = f[i] * (float)c + 1.0f; }   // f64 + f32 interleaved
        }
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i] + f[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* out, const float* in, int per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, in);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, in);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x ITER x per_iter instructions
    printf("%-28s %.3f ms -> %.2f ns per wave-instruction per SIMD (x2.4 = %.2f clk)\n", name, ms, 1e6 * ms / (2.0 * ITER * per_iter), 2.4e6 * ms / (2.0 * ITER * per_iter));
}
int main() {
    double* out; float* in; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    run<0>("v_fma_f64", out, in, 16); run<1>("v_mul_f64", out, in, 16); run<2>("v_add_f64", out, in, 16);
    run<3>("v_cvt_f64_f32 + v_add_f64", out, in, 32); run<4>("v_permlane32_swap", out, in, 16); run<5>("v_fma_f32", out, in, 16);
    run<6>("v_mov_dpp + add", out, in, 32); run<7>("fma_f64 + fma_f32", out, in, 32);
    return 0;
}
