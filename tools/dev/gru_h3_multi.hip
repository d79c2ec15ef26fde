// Dev: N concurrent dfx_k_gru_rec_h3 launches (different weights/buffers) on N streams, optional XCD confinement experiment.
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}
// streaming traffic beside the recurrences: mode 0 plain float4 copy, 1 non-temporal loads+stores, 2 read-only sum
// xmask != 0: only blocks whose blockIdx % 8 (= XCD under the observed round-robin dispatch) is in the mask work
template <int MODE> __global__ void __launch_bounds__(256) k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n, int reps, int xmask) {
    float4 acc = {0, 0, 0, 0};
    int64_t bid = blockIdx.x, nb = gridDim.x;
    if (xmask) {
        const int x = dfx_xcc_id(), pc = __builtin_popcount(xmask);
        if (!((xmask >> x) & 1)) return;
        bid = (int64_t)(blockIdx.x >> 3) * pc + __builtin_popcount(xmask & ((1 << x) - 1));
        nb = (int64_t)(gridDim.x >> 3) * pc;
    }
    for (int r = 0; r < reps; ++r)
        for (int64_t i = bid * 256 + threadIdx.x; i < n; i += nb * 256) {
            float4 v;
            if (MODE == 1) { v.x = __builtin_nontemporal_load(&in[i].x); v.y = __builtin_nontemporal_load(&in[i].y); v.z = __builtin_nontemporal_load(&in[i].z); v.w = __builtin_nontemporal_load(&in[i].w); }
            else v = in[i];
            if (MODE == 2) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else if (MODE == 1) { __builtin_nontemporal_store(v.x, &out[i].x); __builtin_nontemporal_store(v.y, &out[i].y); __builtin_nontemporal_store(v.z, &out[i].z); __builtin_nontemporal_store(v.w, &out[i].w); }
            else out[i] = v;
        }
    if (MODE == 2 && acc.x == 12345.678f) out[0] = acc;
}
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
        DfxGhArgs A; A.gi = gi; A.whf = w; A.bhn = bhn; A.h_in = nullptr; A.h_out = nullptr; A.y = y; A.B = B; A.T = T; A.t0 = 0; A.t1 = T; A.unscale = 1.f / 64.f; A.xcd_mask = 0;
        args[i] = A;
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    CK(hipFuncSetAttribute((const void *)dfx_k_gru_rec_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DFX_GH_SMEM));
    const int64_t NS = (int64_t)1 << 26;  // 1 GiB in, 1 GiB out
    float4 *sin_, *sout; CK(hipMalloc(&sin_, NS * 16)); CK(hipMalloc(&sout, NS * 16)); CK(hipMemset(sin_, 0, NS * 16));
    hipStream_t ss; CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    const int smode = argc > 3 ? atoi(argv[3]) : -1, sblocks = argc > 4 ? atoi(argv[4]) : 2048, sreps = argc > 5 ? atoi(argv[5]) : 2;
    // argv[6]: XCD mask of the recurrence kernels (hex; 0 = plain grid, "p" = the product's per-layer 4-XCD windows), argv[7]: XCD mask of the stream
    const bool gprod = argc > 6 && argv[6][0] == 'p';
    const int gmask = (argc > 6 && !gprod) ? (int)strtol(argv[6], nullptr, 16) : 0, smask = argc > 7 ? (int)strtol(argv[7], nullptr, 16) : 0;
    const int nfirst = argc > 8 ? atoi(argv[8]) : 1;
    const bool stream_last = argc > 9 && atoi(argv[9]) == 1;
    const int npend = argc > 10 ? atoi(argv[10]) : 0;   // streams that hold pending packets (blocked on an event recorded behind the recurrences)
    std::vector<hipStream_t> pst(npend);
    for (auto &q : pst) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));   // launch the stream kernel AFTER the recurrences (they then own their CUs first)
    for (int i = 0; i < NK; ++i) args[i].xcd_mask = gprod ? ((0xF << ((i * 4) % 8)) & 0xff) : gmask;
    printf("# recurrence XCD mask %s, stream mode %d XCD mask 0x%x\n", gprod ? "product" : (argc > 6 ? argv[6] : "0"), smode, smask);
    for (int n = nfirst; n <= NK; ++n) {
        float best = 1e9;
        for (int it = 0; it < 3; ++it) {
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            std::vector<hipEvent_t> done(n);
            std::vector<hipEvent_t> kb(n), ke(n);
            auto launch_stream = [&]() {
                CK(hipStreamWaitEvent(ss, a, 0));
                if (smode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(sblocks), dim3(256), 0, ss, sin_, sout, NS, sreps, smask);
                if (smode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(sblocks), dim3(256), 0, ss, sin_, sout, NS, sreps, smask);
                if (smode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(sblocks), dim3(256), 0, ss, sin_, sout, NS, sreps, smask);
            };
            if (smode >= 0 && !stream_last) launch_stream();
            for (int i = 0; i < n; ++i) {
                CK(hipStreamWaitEvent(st[i], a, 0));
                CK(hipEventCreate(&kb[i])); CK(hipEventCreate(&ke[i])); CK(hipEventRecord(kb[i], st[i]));
                const int pc = args[i].xcd_mask ? __builtin_popcount(args[i].xcd_mask) : 8;
                const unsigned nblk = args[i].xcd_mask ? (unsigned)(((B + 15) / 16 + pc - 1) / pc * 8) : (unsigned)((B + 15) / 16);
                hipLaunchKernelGGL(dfx_k_gru_rec_h3, dim3(nblk), dim3(DFX_GH_THREADS), DFX_GH_SMEM, st[i], args[i]);
                CK(hipEventRecord(ke[i], st[i]));
                CK(hipEventCreate(&done[i])); CK(hipEventRecord(done[i], st[i])); CK(hipStreamWaitEvent(0, done[i], 0));
            }
            if (smode >= 0 && stream_last) launch_stream();
            CK(hipEventRecord(b, 0));
            if (npend) {   // pending work on other queues: waits for b, i.e. for everything timed here
                hipEvent_t pe; CK(hipEventCreateWithFlags(&pe, hipEventDisableTiming));
                for (int i = 0; i < npend; ++i) {
                    CK(hipStreamWaitEvent(pst[i], b, 0));
                    hipLaunchKernelGGL(k_stream<2>, dim3(8), dim3(256), 0, pst[i], sin_, sout, (int64_t)4096, 1, 0);
                    CK(hipEventRecord(pe, pst[i])); CK(hipStreamWaitEvent(pst[(i + 1) % npend], pe, 0));
                }
            } CK(hipEventSynchronize(b)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            if (it == 2) {  // per-kernel spans (start offset from the common start event, duration)
                printf("   kernel spans:");
                for (int i = 0; i < n; ++i) { float o, d; CK(hipEventElapsedTime(&o, a, kb[i])); CK(hipEventElapsedTime(&d, kb[i], ke[i])); printf(" [+%.3f %.3f]", o, d); }
                printf(" ms\n");
            }
        }
        printf("%d concurrent gru_h3 kernels (16 blocks each), %lld steps: %.3f ms -> %.3f us/step\n", n, (long long)T, best, best * 1e3 / T);
    }
    return 0;
}
