// Dev timing of the fp16-split projection kernels, one vs two row tiles per wave (not part of the product).
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
template <typename K> static float run(K kern, int threads, DfxPhArgs A, int bm = DFX_PH_BM, size_t smem = DFX_PH_SMEM) {
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    const int nblk = (int)((A.M + bm - 1) / bm);
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), smem, 0, A);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}
int main() {
    const int64_t Mmax = 256512; const int N = 768;
    float *a, *bias, *out; dfx_h8 *wf;
    CK(hipMalloc(&a, Mmax * 256 * 4)); CK(hipMalloc(&wf, (size_t)(N / 64) * DFX_PH_CHUNK_H8 * 16)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&out, Mmax * N * 4));
    CK(hipMemset(wf, 0x11, (size_t)(N / 64) * DFX_PH_CHUNK_H8 * 16)); CK(hipMemset(bias, 0, N * 4));
    std::vector<float> h(Mmax * 256); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int64_t M : {(int64_t)256512, (int64_t)21248, (int64_t)8192}) {
        DfxPhArgs A; A.a = a; A.wf = wf; A.bias = bias; A.out = out; A.M = M; A.N = N; A.unscale = 1.f; A.rm = DfxRowMap{0, 0, 0};
        const float t1 = run(dfx_k_proj256_h3, DFX_PH_THREADS, A), t2 = run(dfx_k_proj256_h3x2<4, 4>, 256, A, 128), t3 = run(dfx_k_proj256_h3x2<8, 4>, 512, A, 256),
                    t4 = run(dfx_k_proj256_h3x2<4, 2>, 256, A, 128, DFX_PH_SMEM / 2);
        printf("M=%lld: one tile %.4f ms, two tiles x 4 waves %.4f ms, two tiles x 8 waves %.4f ms, two tiles x 4 waves on 32-column chunks (2 workgroups per CU) %.4f ms  (HBM floor %.4f ms at 6 TB/s)\n", (long long)M, t1, t2, t3, t4, M * 4096.0 / 6e12 * 1e3);
    }
    return 0;
}
