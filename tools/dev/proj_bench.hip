// Dev ablation of the weight-stationary projection kernel (not part of the product).
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
template <int MODE> static void run(const char *name, DfxPjArgs A, int nblk) {
    CK(hipFuncSetAttribute((const void *)dfx_k_proj256<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_PJ_SMEM));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(dfx_k_proj256<MODE>, dim3(nblk), dim3(DFX_PJ_THREADS), DFX_PJ_SMEM, 0, A);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-28s %.3f ms  (%.1f TFLOP/s)\n", name, best, 2.0 * A.M * 256 * A.N / best / 1e9);
}
int main() {
    const int64_t M = 256512; const int N = 768;
    float *a, *w, *bias, *out;
    CK(hipMalloc(&a, M * 256 * 4)); CK(hipMalloc(&w, 256 * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&out, M * N * 4));
    CK(hipMemset(a, 0, M * 256 * 4)); CK(hipMemset(w, 0, 256 * N * 4)); CK(hipMemset(bias, 0, N * 4));
    std::vector<float> h(M * 256); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 256 * N * 4, hipMemcpyHostToDevice));
    DfxPjArgs A; A.a = a; A.w = w; A.bias = bias; A.out = out; A.M = M; A.N = N; A.ncol = N / DFX_PJ_BN;
    for (int rg : {40, 80}) {
        A.rgroups = rg; const int nblk = ((rg + 7) / 8) * 8 * A.ncol;
        printf("rgroups=%d blocks=%d\n", rg, nblk);
        run<0>("full", A, nblk); run<1>("no activation loads", A, nblk); run<2>("no LDS fragment reads", A, nblk);
    }
    return 0;
}
