// Dev: where do workgroups land?  Records HW_REG_XCC_ID / HW_REG_HW_ID per block for single and concurrent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_probe(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);          // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | 4);           // HW_REG_HW_ID[15:0]
        out[blockIdx.x] = (xcc << 16) | hw;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
static void show(const char *what, const std::vector<unsigned> &v) {
    printf("%s: xcc per block:", what);
    int ok = 0;
    for (size_t i = 0; i < v.size(); ++i) { if (i < 48) printf(" %u", v[i] >> 16); ok += ((v[i] >> 16) == (i & 7)); }
    printf("  | block%%8 == xcc for %d of %zu\n", ok, v.size());
}
int main() {
    unsigned *d; CK(hipMalloc(&d, 1 << 20));
    std::vector<unsigned> h(4096);
    hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(64), dim3(256), 0, 0, d, 1000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost)); h.resize(64); show("single 64 blocks", h);
    }
    // a 20-block kernel first, then 64 blocks on the same stream
    hipLaunchKernelGGL(k_probe, dim3(20), dim3(256), 0, s1, d + 1024, 1000);
    hipLaunchKernelGGL(k_probe, dim3(64), dim3(256), 0, s1, d, 1000); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost)); show("after a 20-block kernel, same stream", h);
    // three streams at once: long 20-block kernel on s1, 2048-block kernel on s2, 64 blocks on s3
    hipLaunchKernelGGL(k_probe, dim3(20), dim3(256), 0, s1, d + 1024, 2000000);
    hipLaunchKernelGGL(k_probe, dim3(2048), dim3(256), 0, s2, d + 2048, 200000);
    hipLaunchKernelGGL(k_probe, dim3(64), dim3(256), 0, s3, d, 1000); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost)); show("64 blocks beside two running kernels", h);
    h.resize(2048); CK(hipMemcpy(h.data(), d + 2048, 2048 * 4, hipMemcpyDeviceToHost)); show("the 2048-block kernel", h);
    // 2048 blocks with 140 KB of LDS each (one per CU): placement of a CU-filling kernel
    return 0;
}
