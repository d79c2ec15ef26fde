// Dev (round 6): the GRU recurrence on a PAIR of CUs (dfx_k_gru_rec_p2, csrc/dfx_gru_pair.h) against the one-CU form (dfx_k_gru_rec_h3):
// same random weights / inputs, y compared bit for bit, us per step alone, with N layers at once and beside a streaming kernel.
// usage: gru_p2_bench <layers> <steps> <stream blocks, 0 = none> [B]
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
__global__ void __launch_bounds__(256) k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n, int reps) {
    __shared__ float pad_lds[256];
    pad_lds[threadIdx.x] = 0.f;
    for (int r = 0; r < reps; ++r)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}
int main(int argc, char **argv) {
    const int NK = argc > 1 ? atoi(argv[1]) : 5;
    const int64_t T = argc > 2 ? atoll(argv[2]) : 200;
    const int sblocks = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t B = argc > 4 ? atoll(argv[4]) : 256;
    const int pairs = (int)((B + 31) / 32);
    std::vector<DfxGhArgs> a1(NK);
    std::vector<DfxGpArgs> a2(NK);
    std::vector<float *> y1(NK), y2(NK);
    std::vector<hipStream_t> st(NK);
    unsigned int *err; CK(hipMalloc(&err, 256)); CK(hipMemset(err, 0, 256));
    srand(1);
    for (int i = 0; i < NK; ++i) {
        float *gi, *bhn; dfx_h8 *w; unsigned int *sync;
        CK(hipMalloc(&gi, B * T * 768 * 4)); CK(hipMalloc(&y1[i], B * T * 256 * 4)); CK(hipMalloc(&y2[i], B * T * 256 * 4)); CK(hipMalloc(&bhn, 1024)); CK(hipMalloc(&w, 768 * 256 * 4));
        CK(hipMalloc(&sync, (size_t)pairs * 48 * 4)); CK(hipMemset(sync, 0, (size_t)pairs * 48 * 4));
        std::vector<uint16_t> hw(768 * 256 * 2);
        for (size_t j = 0; j < hw.size(); ++j) hw[j] = dfx_f32_to_f16_bits(((float)rand() / RAND_MAX - 0.5f) * 0.08f * 64.f);
        std::vector<float> hg((size_t)B * T * 768), hb(256);
        for (auto &v : hg) v = ((float)rand() / RAND_MAX - 0.5f) * 2.f;
        for (auto &v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
        CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(gi, hg.data(), hg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bhn, hb.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemset(y1[i], 0xff, B * T * 256 * 4)); CK(hipMemset(y2[i], 0xee, B * T * 256 * 4));
        DfxGhArgs A; A.gi = gi; A.whf = w; A.bhn = bhn; A.h_in = nullptr; A.h_out = nullptr; A.y = y1[i]; A.B = B; A.T = T; A.t0 = 0; A.t1 = T; A.unscale = 1.f / 64.f; A.xcd_mask = 0;
        a1[i] = A;
        DfxGpArgs P; P.g = A; P.g.y = y2[i]; P.sync = sync; P.pbase = 0; P.tag = 0; P.err = err; P.stat = err + 16; P.spin_limit = 1 << 20;
        a2[i] = P;
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_GH_SMEM));
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_p2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_GP_SMEM));
    const int64_t NS = (int64_t)1 << 25;
    float4 *sin_, *sout; CK(hipMalloc(&sin_, NS * 16)); CK(hipMalloc(&sout, NS * 16)); CK(hipMemset(sin_, 0, NS * 16));
    hipStream_t ss; CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    unsigned long long *ptr; CK(hipMalloc(&ptr, 64 * 12 * 8)); CK(hipMemset(ptr, 0, 64 * 12 * 8));
    a2[0].ptrace = ptr;
    unsigned int pass = 0;
    auto run = [&](int form, int n, bool load) -> float {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            ++pass;
            for (int i = 0; i < n; ++i) {
                CK(hipStreamWaitEvent(st[i], a, 0));
                if (form == 0) hipLaunchKernelGGL(dfx_k_gru_rec_h3, dim3((unsigned)((B + 15) / 16)), dim3(DFX_GH_THREADS), DFX_GH_SMEM, st[i], a1[i]);
                else {
                    a2[i].pbase = pass * (unsigned int)(T + 1); a2[i].tag = pass << 4;
                    hipLaunchKernelGGL(dfx_k_gru_rec_p2, dim3(dfx_gp_grid(pairs)), dim3(DFX_GP_THREADS), DFX_GP_SMEM, st[i], a2[i]);
                }
                hipEvent_t d; CK(hipEventCreate(&d)); CK(hipEventRecord(d, st[i])); CK(hipStreamWaitEvent(0, d, 0));
            }
            CK(hipEventRecord(b, 0));
            if (load) { CK(hipStreamWaitEvent(ss, a, 0)); hipLaunchKernelGGL(k_stream, dim3(sblocks), dim3(256), 0, ss, sin_, sout, NS, (int)(T / 25 + 1)); }
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            CK(hipDeviceSynchronize());
            if (ms < best) best = ms;
        }
        return best;
    };
    // correctness first: all layers, both forms
    run(0, NK, false); run(1, NK, false);
    unsigned int herr[32]; CK(hipMemcpy(herr, err, 128, hipMemcpyDeviceToHost));
    size_t nbad = 0; double amax = 0, dmax = 0;
    std::vector<float> h1((size_t)B * T * 256), h2((size_t)B * T * 256);
    for (int i = 0; i < NK; ++i) {
        CK(hipMemcpy(h1.data(), y1[i], h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), y2[i], h2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < h1.size(); ++j) { if (memcmp(&h1[j], &h2[j], 4)) ++nbad; if (fabs((double)h1[j] - (double)h2[j]) > dmax) dmax = fabs((double)h1[j] - (double)h2[j]); if (fabs(h1[j]) > amax && fabs(h1[j]) < 1e30) amax = fabs(h1[j]); }
    }
    printf("y of the pair form vs the one-CU form, %d layers x %lld clips x %lld steps: %zu values differ (max |diff| %.3g, max |y| %.4f); timeouts raised %u, pairs on two XCDs %u\n", NK, (long long)B, (long long)T, nbad, dmax, amax, herr[2], herr[17]);
    for (int n = 1; n <= NK; n += (NK > 1 ? NK - 1 : 1)) {
        const float t0 = run(0, n, false), t1 = run(1, n, false);
        printf("%d layer(s) alone: one CU per 16 clips %.3f us/step, pair per 32 clips %.3f us/step\n", n, t0 * 1e3f / T, t1 * 1e3f / T);
        if (sblocks) {
            const float u0 = run(0, n, true), u1 = run(1, n, true);
            printf("%d layer(s) beside %d streaming blocks: one CU %.3f us/step, pair %.3f us/step\n", n, sblocks, u0 * 1e3f / T, u1 * 1e3f / T);
        }
    }
#if DFX_GP_TRACE
    {
        run(1, 1, false);
        unsigned long long h[64 * 12]; CK(hipMemcpy(h, ptr, sizeof(h), hipMemcpyDeviceToHost));
        double tot[2] = {0, 0};
        for (int i = 0; i < 12; ++i) tot[0] += h[i], tot[1] += h[8 * 12 + i];
        printf("phases of a step, thread 0 of block 0 / block 8 (shader-clock ticks per step; %.0f / %.0f in all):\n", tot[0] / T, tot[1] / T);
        for (int i = 0; i < 12; ++i) printf("  phase %2d %8.1f %8.1f\n", i, (double)h[i] / T, (double)h[8 * 12 + i] / T);
    }
#endif
    return 0;
}
