// How long does it take to get a large device buffer: one hipMalloc vs one virtual range backed by several physical chunks
// (hipMemAddressReserve / hipMemCreate / hipMemMap)?   usage: vmm_alloc <total GiB> <chunk GiB>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(char* p, size_t n, size_t stride) { size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main(int argc, char** argv) {
    const double tot_g = argc > 1 ? atof(argv[1]) : 32, chunk_g = argc > 2 ? atof(argv[2]) : 8;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;   // 0: VMM, 1: hipMalloc
    CK(hipSetDevice(0)); CK(hipFree(0));
    size_t total = (size_t)(tot_g * (1ull << 30)), chunk = (size_t)(chunk_g * (1ull << 30));
    if (mode == 1) {
        void* p; double t0 = now(); CK(hipMalloc(&p, total)); CK(hipDeviceSynchronize()); double t1 = now();
        touch<<<(unsigned)((total / (2 << 20) + 255) / 256), 256>>>((char*)p, total, 2 << 20); CK(hipDeviceSynchronize()); double t2 = now();
        printf("hipMalloc %.0f GiB: alloc %.3f s, touch %.3f s\n", tot_g, t1 - t0, t2 - t1);
        return 0;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran; total = (total + chunk - 1) / chunk * chunk;
    double t0 = now();
    void* base = nullptr; CK(hipMemAddressReserve(&base, total, 0, nullptr, 0));
    double t1 = now();
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); hs.push_back(h);
        CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, total, &acc, 1)); CK(hipDeviceSynchronize());
    double t2 = now();
    touch<<<(unsigned)((total / (2 << 20) + 255) / 256), 256>>>((char*)base, total, 2 << 20); CK(hipDeviceSynchronize());
    double t3 = now();
    printf("VMM %.0f GiB in %.0f GiB chunks (granularity %zu): reserve %.3f s, create+map+access %.3f s, touch %.3f s\n", tot_g, chunk_g, gran, t1 - t0, t2 - t1, t3 - t2);
    return 0;
}
