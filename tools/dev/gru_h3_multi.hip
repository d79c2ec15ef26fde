// Dev: N concurrent dfx_k_gru_rec_h3 launches (different weights/buffers) on N streams, optional XCD confinement experiment.
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
int main(int argc, char **argv) {
    const int64_t B = 256, T = argc > 2 ? atoll(argv[2]) : 167;
    const int NK = argc > 1 ? atoi(argv[1]) : 5;
    std::vector<DfxGhArgs> args(NK);
    std::vector<hipStream_t> st(NK);
    std::vector<float> h(768 * 256);
    for (int i = 0; i < NK; ++i) {
        float *gi, *y, *bhn; dfx_h8 *w;
        CK(hipMalloc(&gi, B * T * 768 * 4)); CK(hipMalloc(&y, B * T * 256 * 4)); CK(hipMalloc(&bhn, 1024)); CK(hipMalloc(&w, 768 * 256 * 4));
        for (auto &v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
        std::vector<uint16_t> hw(768 * 256 * 2); for (size_t j = 0; j < hw.size(); ++j) hw[j] = dfx_f32_to_f16_bits(h[j / 2] * 64.f);
        CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(gi, 0, B * T * 768 * 4)); CK(hipMemset(bhn, 0, 1024));
        DfxGhArgs A; A.gi = gi; A.whf = w; A.bhn = bhn; A.h_in = nullptr; A.h_out = nullptr; A.y = y; A.B = B; A.T = T; A.t0 = 0; A.t1 = T; A.unscale = 1.f / 64.f;
        args[i] = A;
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_GH_SMEM));
    for (int n = 1; n <= NK; ++n) {
        float best = 1e9;
        for (int it = 0; it < 3; ++it) {
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            std::vector<hipEvent_t> done(n);
            for (int i = 0; i < n; ++i) {
                CK(hipStreamWaitEvent(st[i], a, 0));
                hipLaunchKernelGGL(dfx_k_gru_rec_h3, dim3((B + 15) / 16), dim3(DFX_GH_THREADS), DFX_GH_SMEM, st[i], args[i]);
                CK(hipEventCreate(&done[i])); CK(hipEventRecord(done[i], st[i])); CK(hipStreamWaitEvent(0, done[i], 0));
            }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("%d concurrent gru_h3 kernels (16 blocks each), %lld steps: %.3f ms -> %.3f us/step\n", n, (long long)T, best, best * 1e3 / T);
    }
    return 0;
}
