#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
template <int OP>
__global__ __launch_bounds__(512, 1) void k(double* out, const float* in) {
    double a[16], v[4];
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + i];
    for (int i = 0; i < 4; ++i) v[i] = in[threadIdx.x + 16 + i];
    const double c = in[3];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(v[i & 3]), "v"(c));
            if (OP == 1) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v[i & 3])); a[i] = __builtin_fma(t, c, a[i]); }
            if (OP == 2) { int lo = __builtin_amdgcn_mov_dpp(__double2loint(v[i & 3]), 0x125, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v[i & 3]), 0x125, 0xf, 0xf, true); a[i] = __builtin_fma(__hiloint2double(hi, lo), c, a[i]); }
            if (OP == 3) a[i] = __builtin_fma(v[i & 3], c, a[i]);
            if (OP == 4) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i & 3]) : "v"(v[i & 3]), "v"(c));   // 4 chains
            if (OP == 5) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i & 1]) : "v"(v[i & 3]), "v"(c));   // 2 chains
            if (OP == 6) a[i & 3] = __builtin_fma(v[i & 3], c, a[i & 3]);
        }
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* out, const float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, in);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, in);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms -> %.2f ns per accumulate per wave per SIMD\n", name, ms, 1e6 * ms / (2.0 * ITER * 16));
}
int main() {
    double* out; float* in; hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    run<3>("v_fma_f64 (no cross-lane)", out, in);
    run<0>("v_fmac_f64_dpp row_newbcast", out, in);
    run<1>("v_mov_b64_dpp row_newbcast + v_fma_f64", out, in);
    run<2>("2 x v_mov_b32_dpp row_ror + v_fma_f64", out, in);
    run<4>("v_fmac_f64_dpp, 4 dependent chains", out, in);
    run<5>("v_fmac_f64_dpp, 2 dependent chains", out, in);
    run<6>("v_fma_f64, 4 dependent chains", out, in);
    return 0;
}
