#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 400
// 8 waves per CU, each wave owns 16.5 KB.  OP: 0 write_b64 consecutive (32 per iter), 1 read_b64 stride 264, 2 write_b128 consecutive, 3 read_b128 consecutive,
// 4 read_b128 stride 272, 5 write_b64 x2 rows (transposition writer), 6 read_b64 consecutive
template <int OP>
__global__ __launch_bounds__(512, 1) void k(double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* xb = smem + wave * 16896;
    double v[32];
    for (int i = 0; i < 32; ++i) v[i] = lane + i;
    double acc = 0;
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<double*>(xb + lane * 8 + i * 512) = v[i];
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<double*>(xb + (lane >> 5) * 264 + (lane & 31) * 8 + (i ^ (i >> 1)) * 528 % 16368) = v[i];
        }
        if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] += *reinterpret_cast<const double*>(xb + lane * 264 + i * 8);
        }
        if (OP == 6) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] += *reinterpret_cast<const double*>(xb + lane * 8 + i * 512);
        }
        if (OP == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<double2*>(xb + lane * 16 + i * 1024) = double2{v[2 * i], v[2 * i + 1]};
        }
        if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double2 t = *reinterpret_cast<const double2*>(xb + lane * 16 + i * 1024); v[2 * i] += t.x; v[2 * i + 1] += t.y; }
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double2 t = *reinterpret_cast<const double2*>(xb + lane * 256 + ((i ^ (lane & 15)) * 16)); v[2 * i] += t.x; v[2 * i + 1] += t.y; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 32; ++i) acc += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP> void run(const char* name, double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = k<OP>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16896);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * 16896, 0, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * 16896, 0, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: 8 waves x 16 KB per iteration
    printf("%-34s %.3f ms: %.1f B/ns per CU (16 KB per wave-iteration in %.3f us of CU time)\n", name, ms, 8.0 * 16384 * ITER / (1e6 * ms), 1e3 * ms / ITER / 8);
}
int main() {
    double* out; hipMalloc(&out, 256 * 512 * 8);
    run<0>("write_b64 consecutive", out); run<5>("write_b64 two rows", out); run<2>("write_b128 consecutive", out);
    run<6>("read_b64 consecutive", out); run<1>("read_b64 lane stride 264", out); run<3>("read_b128 consecutive", out); run<4>("read_b128 stride 256 xor-swizzled", out);
    return 0;
}
