// Dev timing of dfx_k_gru_rec_h3 for one (FR, FL, D) configuration given with -DDFX_GH_FR=.. etc. (random weights, no check).
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 256, T = 1002;
    float *gi, *y, *bhn; dfx_h8 *w;
    CK(hipMalloc(&gi, B * T * 768 * 4)); CK(hipMalloc(&y, B * T * 256 * 4)); CK(hipMalloc(&bhn, 1024)); CK(hipMalloc(&w, 768 * 256 * 4));
    std::vector<float> h(768 * 256); for (auto &v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    std::vector<uint16_t> hw(768 * 256 * 2); for (size_t i = 0; i < hw.size(); ++i) hw[i] = dfx_f32_to_f16_bits(h[i / 2] * 64.f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(gi, 0, B * T * 768 * 4)); CK(hipMemset(bhn, 0, 1024));
    DfxGhArgs A; A.gi = gi; A.whf = w; A.bhn = bhn; A.h_in = nullptr; A.h_out = nullptr; A.y = y; A.B = B; A.T = T; A.unscale = 1.f / 64.f; A.xcd_mask = 0;
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_GH_SMEM));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(dfx_k_gru_rec_h3, dim3((B + 15) / 16), dim3(DFX_GH_THREADS), DFX_GH_SMEM, 0, A);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("gru_h3 NW=%d FR=%d FL=%d D=%d FS=%d B=%lld: %.3f ms  %.3f us/step (%d KB streamed per step)\n", DFX_GH_NW, DFX_GH_FR, DFX_GH_FL, DFX_GH_D, DFX_GH_FS, (long long)B, best, best * 1e3 / T, DFX_GH_FS * 2 * DFX_GH_NW);
    return 0;
}
