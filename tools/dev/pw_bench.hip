// Dev: the separable-conv kernel (dfx_k_pwconv) stand-alone at config-2 size against a frame-staged candidate (coalesced 16-byte loads
// of whole frames into a wave-private LDS strip, dfx_chain_stage on the strip, output strip -> coalesced stores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I deepfilternet_amd/csrc/env_hip -I deepfilternet_amd/csrc tools/dev/pw_bench.hip -o tools/dev/_build/pw_bench
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}

template <typename K> static float time_it(K &&launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < reps; ++it) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

static float w_unscale = 1.f;
template <int MODE, bool SKIP>
static void run(const char *name, int Fin, int Fout, int stride, int64_t R, const float *x, const float *skip, const float *w, float *o1, float *o2, int wgs) {
    constexpr int C = 64;
    DfxPwArgs A;
    A.x = x; A.skip = SKIP ? skip : nullptr; A.sk_a = w; A.sk_b = w + 64; A.dw = w + 128; A.wt = w + 512; A.bias = w + 512 + C * C;
    A.R = R; A.Fin = Fin; A.Fout = Fout; A.stride = stride; A.rm = DfxRowMap{0, 0, 0};
    A.out = o1;
    const int grid1 = 256 * 8;
    const float t1 = time_it([&] { hipLaunchKernelGGL((dfx_k_pwconv<C, MODE, SKIP>), dim3(grid1), dim3(256), 0, 0, A); });
    A.out = o2;
    const int G = dfx_pwf_group(C, Fin, Fout);
    const size_t smem = dfx_pwf_smem(C, Fin, Fout);
    const int nvi = dfx_pwf_nvi(C, Fin, Fout);
    CK(hipFuncSetAttribute((const void *)dfx_k_pwconv_f<C, MODE, SKIP, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void *)dfx_k_pwconv_f<C, MODE, SKIP, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const float t2 = time_it([&] {
        if (nvi == 4) hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 4>), dim3(wgs), dim3(256), smem, 0, A);
        else hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 8>), dim3(wgs), dim3(256), smem, 0, A);
    });
    A.wt_h3 = reinterpret_cast<const dfx_h8 *>(w + 5000); A.unscale = w_unscale;
    CK(hipFuncSetAttribute((const void *)dfx_k_pwconv_f<C, MODE, SKIP, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CK(hipFuncSetAttribute((const void *)dfx_k_pwconv_f<C, MODE, SKIP, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const float t3 = time_it([&] {
        if (nvi == 4) hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 4, true>), dim3(wgs), dim3(256), smem, 0, A);
        else hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 8, true>), dim3(wgs), dim3(256), smem, 0, A);
    });
    const size_t n = (size_t)R * Fout * C;
    {
        std::vector<float> h1(n), h3(n);
        CK(hipMemcpy(h1.data(), o1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h3.data(), o2, n * 4, hipMemcpyDeviceToHost));
        double md = 0, se = 0, sr = 0;
        for (size_t i = 0; i < n; ++i) { const double d = (double)h1[i] - h3[i]; md = fmax(md, fabs(d)); se += d * d; sr += (double)h1[i] * h1[i]; }
        printf("%-8s fp16-split staged %.3f ms (%.2f TB/s)  maxdiff vs fp32 %.3g  rel rms %.3g\n", name, t3,
               (double)R * C * 4 * (Fin * (SKIP ? 2 : 1) + Fout) / t3 / 1e9, md, sqrt(se / (sr + 1e-30)));
        hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 8, false>), dim3(wgs), dim3(256), smem, 0, A);   // o2 back to the fp32 staged result
        if (nvi == 4) hipLaunchKernelGGL((dfx_k_pwconv_f<C, MODE, SKIP, 4, false>), dim3(wgs), dim3(256), smem, 0, A);
        CK(hipDeviceSynchronize());
    }
    std::vector<float> h1(n), h2(n);
    CK(hipMemcpy(h1.data(), o1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; size_t nz = 0;
    for (size_t i = 0; i < n; ++i) { md = fmax(md, fabs((double)h1[i] - h2[i])); mx = fmax(mx, fabs((double)h1[i])); nz += h1[i] != 0.f; }
    const double bytes = (double)R * C * 4 * (Fin * (SKIP ? 2 : 1) + Fout);
    printf("%-8s G=%d smem=%zu  old %.3f ms (%.2f TB/s)   staged %.3f ms (%.2f TB/s)   maxdiff %.3g (max %.3g, nonzero %.2f)\n", name, G, smem, t1,
           bytes / t1 / 1e9, t2, bytes / t2 / 1e9, md, mx, (double)nz / n);
}

int main(int argc, char **argv) {
    const int64_t R = 256512;
    const int wgs = argc > 1 ? atoi(argv[1]) : 512;
    const size_t nin = (size_t)R * 16 * 64, nout = (size_t)R * 16 * 64;
    float *x, *skip, *w, *o1, *o2;
    CK(hipMalloc(&x, nin * 4)); CK(hipMalloc(&skip, nin * 4)); CK(hipMalloc(&w, 16384 * 4)); CK(hipMalloc(&o1, nout * 4)); CK(hipMalloc(&o2, nout * 4));
    {
        std::vector<float> h(nin);
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.3f;
        CK(hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice));
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(skip, h.data(), nin * 4, hipMemcpyHostToDevice));
        std::vector<float> hw(16384);
        for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.25f;
        {   // f16 hi/lo fragments of wt (w + 512) at w + 5000: [nt*KC + kc][hi, lo][64 lanes][8]
            const int C = 64, KC = 2;
            float mx = 0; for (int i = 0; i < C * C; ++i) mx = fmaxf(mx, fabsf(hw[512 + i]));
            int ex; frexpf(mx, &ex); const int e = 14 - ex;
            w_unscale = ldexpf(1.f, -e);
            uint16_t *dst = reinterpret_cast<uint16_t *>(&hw[5000]);
            for (int fr = 0; fr < 4 * KC; ++fr) for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) {
                const int nt = fr / KC, kc = fr % KC;
                const float v = ldexpf(hw[512 + (16 * (l >> 4) + 8 * kc + i) * C + 16 * nt + (l & 15)], e);
                const uint16_t hb = dfx_f32_to_f16_bits(v), lb = dfx_f32_to_f16_bits(v - dfx_f16_bits_to_f32(hb));
                dst[((fr * 2 + 0) * 64 + l) * 8 + i] = hb; dst[((fr * 2 + 1) * 64 + l) * 8 + i] = lb;
            }
        }
        CK(hipMemcpy(w, hw.data(), 16384 * 4, hipMemcpyHostToDevice));
    }
    printf("workgroups (staged) = %d\n", wgs);
    run<DFX_PW_MODE_DW3, false>("conv2", 16, 8, 2, R, x, skip, w, o1, o2, wgs);
    run<DFX_PW_MODE_DW3, false>("conv3", 8, 8, 1, R, x, skip, w, o1, o2, wgs);
    run<DFX_PW_MODE_DW3, true>("convt3", 8, 8, 1, R, x, skip, w, o1, o2, wgs);
    run<DFX_PW_MODE_DWT3, true>("convt2", 8, 16, 2, R, x, skip, w, o1, o2, wgs);
    run<DFX_PW_MODE_DWT3, true>("convt1", 16, 32, 2, R / 2, x, skip, w, o1, o2, wgs);
    return 0;
}
