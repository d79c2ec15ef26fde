// Dev timing of dfx_k_gru_step_h3 (one GRU time step of many streams): 64 vs 32 hidden units per workgroup (not part of the product).
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
template <typename K> static float run(K kern, DfxGstArgs A, int nu, size_t smem) {
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    const int nblk = (int)(((A.B + DFX_PH_BM - 1) / DFX_PH_BM + 7) / 8 * 8) * nu;
    for (int it = 0; it < 10; ++it) {
        CK(hipEventRecord(a, 0));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(nblk), dim3(DFX_PH_THREADS), smem, 0, A);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best * 100.f;   // us per launch
}
int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
    float *x, *hin, *ho4, *ho2, *y, *bi, *bhn; dfx_h8 *wi, *wh;
    const size_t wbytes = (size_t)12 * DFX_PH_CHUNK_H8 * 16;
    CK(hipMalloc(&x, B * 1024)); CK(hipMalloc(&hin, B * 1024)); CK(hipMalloc(&ho4, B * 1024)); CK(hipMalloc(&ho2, B * 1024)); CK(hipMalloc(&y, B * 1024));
    CK(hipMalloc(&bi, 3072)); CK(hipMalloc(&bhn, 1024)); CK(hipMalloc(&wi, wbytes)); CK(hipMalloc(&wh, wbytes));
    std::vector<float> h(B * 256); for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(hin, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> w(wbytes / 2); for (size_t i = 0; i < w.size(); ++i) w[i] = dfx_f32_to_f16_bits(((float)rand() / RAND_MAX - 0.5f) * ((i / 512) & 1 ? 1e-3f : 1.f));
    CK(hipMemcpy(wi, w.data(), wbytes, hipMemcpyHostToDevice));
    for (size_t i = 0; i < w.size(); ++i) w[i] = dfx_f32_to_f16_bits(((float)rand() / RAND_MAX - 0.5f) * ((i / 512) & 1 ? 1e-3f : 1.f));
    CK(hipMemcpy(wh, w.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemset(bi, 0, 3072)); CK(hipMemset(bhn, 0, 1024));
    DfxGstArgs A; A.x = x; A.h_in = hin; A.h_out = ho4; A.y = y; A.wif = wi; A.whf = wh; A.bias_i = bi; A.bhn = bhn; A.unscale_i = 1.f / 16; A.unscale_h = 1.f / 16;
    A.B = B; A.xrm = DfxRowMap{0, 0, 0}; A.yrm = DfxRowMap{0, 0, 0};
    const float t4 = run(dfx_k_gru_step_h3<4>, A, 4, DFX_PH_SMEM);
    A.h_out = ho2;
    const float t2 = run(dfx_k_gru_step_h3<2>, A, 8, DFX_PH_SMEM / 2);
    std::vector<float> a4(B * 256), a2(B * 256);
    CK(hipMemcpy(a4.data(), ho4, B * 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(a2.data(), ho2, B * 1024, hipMemcpyDeviceToHost));
    size_t diff = 0; double sum = 0; for (size_t i = 0; i < a4.size(); ++i) { diff += memcmp(&a4[i], &a2[i], 4) != 0; sum += fabs(a4[i]); }
    printf("B=%lld: 64 units per workgroup %.2f us, 32 units per workgroup %.2f us per launch (back to back); outputs differ in %zu of %zu values, mean |h| %.4f\n",
           (long long)B, t4, t2, diff, a4.size(), sum / a4.size());
    return 0;
}
