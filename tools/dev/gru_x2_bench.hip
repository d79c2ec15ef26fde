// Dev timing of dfx_k_gru_rec_h3x2 (two-CU GRU recurrence), N concurrent launches on N streams.
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
int main(int argc, char **argv) {
    const int64_t B = 256, T = argc > 2 ? atoll(argv[2]) : 334;
    const int NK = argc > 1 ? atoi(argv[1]) : 5;
    std::vector<DfxG2Args> args(NK);
    std::vector<hipStream_t> st(NK);
    std::vector<float> h(768 * 256);
    unsigned int *err; CK(hipMalloc(&err, 256)); CK(hipMemset(err, 0, 256));
    for (int i = 0; i < NK; ++i) {
        float *gi, *y, *bhn; dfx_h8 *w; unsigned long long *xb;
        CK(hipMalloc(&gi, B * T * 768 * 4)); CK(hipMalloc(&y, B * T * 256 * 4)); CK(hipMalloc(&bhn, 1024)); CK(hipMalloc(&w, 768 * 256 * 4));
        CK(hipMalloc(&xb, 16 * 2 * 2 * 2048 * 8)); CK(hipMemset(xb, 0, 16 * 2 * 2 * 2048 * 8));
        for (auto &v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
        std::vector<uint16_t> hw(768 * 256 * 2); for (size_t j = 0; j < hw.size(); ++j) hw[j] = dfx_f32_to_f16_bits(h[j / 2] * 64.f);
        CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(gi, 0, B * T * 768 * 4)); CK(hipMemset(bhn, 0, 1024));
        DfxG2Args A; A.gi = gi; A.whf = w; A.bhn = bhn; A.h_in = nullptr; A.h_out = nullptr; A.y = y; A.xbuf = xb; A.err = err; A.B = B; A.T = T;
        A.t0 = 0; A.t1 = T; A.groups = 16; A.epoch = 1; A.unscale = 1.f / 64.f;
        args[i] = A;
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_h3x2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_G2_SMEM));
    unsigned int epoch = 1;
    for (int n = 1; n <= NK; ++n) {
        float best = 1e9;
        for (int it = 0; it < 3; ++it) {
            CK(hipDeviceSynchronize());
            ++epoch;
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            std::vector<hipEvent_t> done(n);
            for (int i = 0; i < n; ++i) {
                args[i].epoch = epoch;
                CK(hipStreamWaitEvent(st[i], a, 0));
                hipLaunchKernelGGL(dfx_k_gru_rec_h3x2, dim3(32), dim3(256), DFX_G2_SMEM, st[i], args[i]);
                CK(hipEventCreate(&done[i])); CK(hipEventRecord(done[i], st[i])); CK(hipStreamWaitEvent(0, done[i], 0));
            }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        unsigned int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("%d concurrent gru_h3x2 kernels (32 blocks each), %lld steps: %.3f ms -> %.3f us/step  err=%u\n", n, (long long)T, best, best * 1e3 / T, e);
    }
    return 0;
}
