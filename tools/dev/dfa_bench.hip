// Dev: HBM streaming microbenchmarks (SURVEY §8d "measure achievable BW with a stream-copy microbench") and the deep-filter kernels
// stand-alone at config-2 size (256 clips x 1002 frames, F = 481, nb_df = 96, E = 32), O = 5 and 10.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I deepfilternet_amd/csrc/env_hip -I deepfilternet_amd/csrc tools/dev/dfa_bench.hip -o tools/dev/_build/dfa_bench
#include "dfx_dsp_kernels.h"
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}

// ---- plain streams.  MODE 0 copy, 1 copy with non-temporal loads + stores, 2 read only, 3 write only, 4 out = a * b (2 reads : 1 write)
template <int MODE, int U>
__global__ void __launch_bounds__(256) k_stream(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *__restrict__ out, int64_t n) {
    f32x4 acc = {0, 0, 0, 0};
    const int64_t step = (int64_t)gridDim.x * 256 * U;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += step) {
        f32x4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256;
            if (MODE != 3) {
                if (i < n) v[u] = MODE == 1 ? __builtin_nontemporal_load(a + i) : a[i];
                if (MODE == 4 && i < n) w[u] = b[i];
            } else {
                v[u] = acc;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256;
            if (i >= n) continue;
            if (MODE == 2) acc += v[u];
            else if (MODE == 1) __builtin_nontemporal_store(v[u], out + i);
            else if (MODE == 4) out[i] = v[u] * w[u];
            else out[i] = v[u];
        }
    }
    if (MODE == 2 && acc[0] == 12345.678f) out[0] = acc;
}
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_set(unsigned int *flag, unsigned int v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_wait(const unsigned int *flag, unsigned int target) {
    int spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(16);
}
// one contiguous slab per workgroup instead of a grid-stride walk
template <int U>
__global__ void __launch_bounds__(256) k_copy_slab(const f32x4 *__restrict__ a, f32x4 *__restrict__ out, int64_t n, int64_t per_wg) {
    const int64_t lo = (int64_t)blockIdx.x * per_wg, hi = lo + per_wg < n ? lo + per_wg : n;
    for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i0 + u * 256 < hi) v[u] = a[i0 + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i0 + u * 256 < hi) out[i0 + u * 256] = v[u];
    }
}

static float time_it(int iters, const std::function<void()> &f) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char **argv) {
    const int64_t B = 256, T = 1002, F = 481, Fs = 482, nd = 96, E = 32;
    const int OM = 10;
    float *spec_d, *spec_p, *out_d, *out_p, *coefs, *gains;
    CK(hipMalloc(&spec_d, B * T * F * 8)); CK(hipMalloc(&out_d, B * T * F * 8));
    CK(hipMalloc(&spec_p, B * T * Fs * 8)); CK(hipMalloc(&out_p, B * T * Fs * 8));
    CK(hipMalloc(&coefs, B * OM * T * nd * 8)); CK(hipMalloc(&gains, B * T * E * 4));
    CK(hipMemset(spec_d, 0, B * T * F * 8)); CK(hipMemset(spec_p, 0, B * T * Fs * 8));
    CK(hipMemset(coefs, 0, B * OM * T * nd * 8)); CK(hipMemset(gains, 0, B * T * E * 4));
    {   // something non-trivial in the inputs (values do not change the timing of these kernels)
        std::vector<float> h(1 << 20);
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
        for (int64_t o = 0; o + (int64_t)h.size() * 4 <= B * T * F * 8; o += (int64_t)h.size() * 4 * 64) {
            CK(hipMemcpy((char *)spec_d + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy((char *)spec_p + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
    }
    // band table of the 48 kHz / 960 / 32-band configuration (min_nb_freqs = 2)
    const int widths[32] = {2,2,2,2,2,2,2,2,2,2,2,2,2,5,5,7,7,8,10,12,13,15,18,20,24,28,31,37,42,50,56,67};
    std::vector<unsigned char> b2b(F);
    { int f = 0; for (int e = 0; e < 32; ++e) for (int k = 0; k < widths[e]; ++k) b2b[f++] = (unsigned char)e; }
    unsigned char *d_b2b; CK(hipMalloc(&d_b2b, F)); CK(hipMemcpy(d_b2b, b2b.data(), F, hipMemcpyHostToDevice));

    auto report = [&](const char *name, double bytes, float ms) { printf("%-44s %8.4f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    if (argc > 1) {
    // ---------------------------------------------------------------- streams
    const int64_t n4 = B * T * Fs * 8 / 16;   // float4s in one padded spectrum buffer (0.99 GB)
    const f32x4 *a4 = (const f32x4 *)spec_p, *b4 = (const f32x4 *)out_d;
    f32x4 *o4 = (f32x4 *)out_p;
    const double cb = (double)n4 * 16;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, 96, "copy grid-stride U=1 blocks=%d", blocks); report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<0, 1>), dim3(blocks), dim3(256), 0, 0, a4, b4, o4, n4); }));
        snprintf(nm, 96, "copy grid-stride U=2 blocks=%d", blocks); report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<0, 2>), dim3(blocks), dim3(256), 0, 0, a4, b4, o4, n4); }));
        snprintf(nm, 96, "copy grid-stride U=4 blocks=%d", blocks); report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<0, 4>), dim3(blocks), dim3(256), 0, 0, a4, b4, o4, n4); }));
        snprintf(nm, 96, "copy grid-stride U=8 blocks=%d", blocks); report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<0, 8>), dim3(blocks), dim3(256), 0, 0, a4, b4, o4, n4); }));
        snprintf(nm, 96, "copy nt U=4 blocks=%d", blocks); report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<1, 4>), dim3(blocks), dim3(256), 0, 0, a4, b4, o4, n4); }));
    }
    for (int64_t per : {1024, 4096, 16384, 65536}) {   // float4s per workgroup slab: 16 KB .. 1 MB
        char nm[96];
        const int blocks = (int)((n4 + per - 1) / per);
        snprintf(nm, 96, "copy slab %lld KB/wg U=4 (%d wgs)", (long long)per * 16 / 1024, blocks);
        report(nm, 2 * cb, time_it(10, [&] { hipLaunchKernelGGL((k_copy_slab<4>), dim3(blocks), dim3(256), 0, 0, a4, o4, n4, per); }));
    }
    report("read only U=4 blocks=4096", cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<2, 4>), dim3(4096), dim3(256), 0, 0, a4, b4, o4, n4); }));
    report("read only U=8 blocks=8192", cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<2, 8>), dim3(8192), dim3(256), 0, 0, a4, b4, o4, n4); }));
    report("write only U=4 blocks=4096", cb, time_it(10, [&] { hipLaunchKernelGGL((k_stream<3, 4>), dim3(4096), dim3(256), 0, 0, a4, b4, o4, n4); }));
    {
        const int64_t m4 = B * T * F * 8 / 16;  // out_d as the second input
        report("a*b -> out (2R:1W) U=4 blocks=4096", 3.0 * m4 * 16, time_it(10, [&] { hipLaunchKernelGGL((k_stream<4, 4>), dim3(4096), dim3(256), 0, 0, a4, b4, o4, m4); }));
        report("a*b -> out (2R:1W) U=2 blocks=8192", 3.0 * m4 * 16, time_it(10, [&] { hipLaunchKernelGGL((k_stream<4, 2>), dim3(8192), dim3(256), 0, 0, a4, b4, o4, m4); }));
    }
    {   // hipMemcpyAsync device-to-device of the same size
        report("hipMemcpyDtoD", 2 * cb, time_it(10, [&] { CK(hipMemcpyAsync(out_p, spec_p, n4 * 16, hipMemcpyDeviceToDevice, 0)); }));
    }

    }
    if (argc > 2) {   // argv[2] = N: does a kernel run slower once N other streams (HW queues) of the process have been used?
        const int NS = atoi(argv[2]);
        std::vector<hipStream_t> st(NS);
        for (auto &q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        DfxDfrArgs R;
        const int O = 5, la = 2, Fs2 = 488;
        float *sp, *op; CK(hipMalloc(&sp, B * T * Fs2 * 8)); CK(hipMalloc(&op, B * T * Fs2 * 8)); CK(hipMemset(sp, 0, B * T * Fs2 * 8));
        R.spec = sp; R.coefs = coefs; R.gains = gains; R.bin2band = d_b2b; R.out = op;
        R.B = B; R.T = T; R.cs_b = (int64_t)O * T * nd; R.cs_n = T * nd; R.cs_t = nd;
        R.gT = T; R.out_T = T; R.out_toff = 0; R.Fs = Fs2; R.Fso = Fs2; R.F = F; R.nbdf = nd; R.lookahead = la; R.nb = E;
        R.pf_beta = 0.f; R.atten_lim = 0.f; R.t_begin = 0; R.t_end = T; R.rpw = 1; R.chunks = (int)T; R.zcols = 244;
        R.items = ((B + 7) / 8) * 8 * ((R.chunks + 3) / 4);
        const unsigned nblk = argc > 3 ? (unsigned)atoi(argv[3]) : (unsigned)R.items;
        const double alg = (double)(F * 8 + nd * O * 8 + E * 4 + F * 8) * B * T;
        hipStream_t main_s; CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
        auto timed = [&](const char *nm) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dfx_k_df_apply_rows<5, 4, false, 7>), dim3(nblk), dim3(256), 0, main_s, R);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, main_s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((dfx_k_df_apply_rows<5, 4, false, 7>), dim3(nblk), dim3(256), 0, main_s, R);
            CK(hipEventRecord(b, main_s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); report(nm, alg, ms / 10);
        };
        timed("df_apply, fresh process");
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (int rep = 0; rep < 20; ++rep)
            for (int i = 0; i < NS; ++i) {   // a chain of small kernels hopping over the streams, like the engine's event graph
                if (i) CK(hipStreamWaitEvent(st[i], ev, 0));
                hipLaunchKernelGGL((k_stream<0, 1>), dim3(64), dim3(256), 0, st[i], (const f32x4 *)spec_p, (const f32x4 *)out_d, (f32x4 *)out_p, (int64_t)1 << 16);
                CK(hipEventRecord(ev, st[i]));
            }
        CK(hipDeviceSynchronize());
        char nm[96]; snprintf(nm, 96, "df_apply after %d other streams were used", NS); timed(nm);
        // the same launches while the other streams hold PENDING packets: each waits for an event that is recorded behind the timed
        // launches (what the queues of the next enhance() call look like while the current call's last kernels run)
        for (int depth : {1, 8}) {
            hipEvent_t a, b, after; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreateWithFlags(&after, hipEventDisableTiming));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, (long long)300000);   // 3 ms: time to enqueue everything below
            CK(hipEventRecord(a, main_s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((dfx_k_df_apply_rows<5, 4, false, 7>), dim3(nblk), dim3(256), 0, main_s, R);
            CK(hipEventRecord(b, main_s));
            CK(hipEventRecord(after, main_s));
            for (int i = 0; i < NS; ++i) {
                CK(hipStreamWaitEvent(st[i], after, 0));
                for (int d = 0; d < depth; ++d) {
                    hipLaunchKernelGGL((k_stream<0, 1>), dim3(64), dim3(256), 0, st[i], (const f32x4 *)spec_p, (const f32x4 *)out_d, (f32x4 *)out_p, (int64_t)1 << 16);
                    CK(hipEventRecord(ev, st[i]));
                    CK(hipStreamWaitEvent(st[(i + 1) % NS], ev, 0));
                }
            }
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            snprintf(nm, 96, "df_apply, %d streams with pending packets (x%d)", NS, depth); report(nm, alg, ms / 10);
        }
        {   // the same dependency expressed with spinning wait kernels on the other streams (a flag in device memory set behind the timed launches)
            unsigned int *flag; CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, (long long)300000);
            CK(hipEventRecord(a, main_s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((dfx_k_df_apply_rows<5, 4, false, 7>), dim3(nblk), dim3(256), 0, main_s, R);
            CK(hipEventRecord(b, main_s));
            hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, main_s, flag, 1u);
            for (int i = 0; i < NS; ++i) {
                hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st[i], (const unsigned int *)flag, 1u);
                hipLaunchKernelGGL((k_stream<0, 1>), dim3(64), dim3(256), 0, st[i], (const f32x4 *)spec_p, (const f32x4 *)out_d, (f32x4 *)out_p, (int64_t)1 << 16);
            }
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            snprintf(nm, 96, "df_apply, %d streams held by spinning wait kernels", NS); report(nm, alg, ms / 10);
        }
        for (auto &q : st) CK(hipStreamDestroy(q));
        CK(hipDeviceSynchronize());
        timed("df_apply after those streams were destroyed");
        return 0;
    }
    // ---------------------------------------------------------------- deep-filter kernels
    for (int O : {5, 10}) {
        const int la = 2;
        const double alg = (double)(F * 8 + nd * O * 8 + E * 4 + F * 8) * B * T;
        {   // flat-stream kernel on dense rows (round 1)
            DfxDfaArgs A;
            A.spec = (const float2 *)spec_d; A.coefs = (const float2 *)coefs; A.gains = gains; A.bin2band = d_b2b; A.out = (float2 *)out_d;
            A.B = B; A.T = T; A.cs_b = (int64_t)O * T * nd; A.cs_n = T * nd; A.cs_t = nd; A.cs_f = 1;
            A.F = F; A.nbdf = nd; A.order = O; A.lookahead = la; A.nb = E; A.pf_beta = 0.f; A.atten_lim = 0.f;
            A.t_begin = 0; A.t_end = T; A.chunks = (T + 15) / 16; A.gT = T; A.out_T = T; A.out_toff = 0;
            auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
            const size_t smem = al((size_t)(16 + O - 1) * nd * 8) + al((size_t)16 * E * 4) + al((size_t)F);
            const unsigned nblk = (unsigned)(B * A.chunks);
            char nm[96]; snprintf(nm, 96, "df_apply flat (dense rows)  O=%d", O);
            report(nm, alg, time_it(20, [&] { hipLaunchKernelGGL(dfx_k_df_apply<16>, dim3(nblk), dim3(256), smem, 0, A); }));
        }
        for (int Fs : {482, 488, 496, 512}) {
            float *sp = spec_p, *op = out_p;
            if (Fs != 482) { CK(hipMalloc(&sp, B * T * Fs * 8)); CK(hipMalloc(&op, B * T * Fs * 8)); CK(hipMemset(sp, 0, B * T * Fs * 8)); }
            for (int rpw : {1, 2, 3, 4, 6, 8}) {
                DfxDfrArgs R;
                R.spec = sp; R.coefs = coefs; R.gains = gains; R.bin2band = d_b2b; R.out = op;
                R.B = B; R.T = T; R.cs_b = (int64_t)O * T * nd; R.cs_n = T * nd; R.cs_t = nd;
                R.gT = T; R.out_T = T; R.out_toff = 0; R.Fs = Fs; R.Fso = Fs; R.F = F; R.nbdf = nd; R.lookahead = la; R.nb = E;
                R.pf_beta = 0.f; R.atten_lim = 0.f; R.t_begin = 0; R.t_end = T; R.rpw = rpw; R.chunks = (int)((T + rpw - 1) / rpw);
                R.zcols = Fs == 482 ? 241 : 244;
                R.items = ((B + 7) / 8) * 8 * ((R.chunks + 3) / 4);
                const unsigned nblk = (unsigned)R.items;
                const double bytes = Fs == 496 ? alg + (double)2 * 56 * B * T : alg;   // + the 7 pad bins read and written per row
                char nm[96];
#define RUN(O_, NT_) do { snprintf(nm, 96, "df_apply rows Fs=%d rpw=%d O=%d nt=%d", Fs, rpw, O_, NT_); \
                    report(nm, alg, time_it(20, [&] { hipLaunchKernelGGL((dfx_k_df_apply_rows<O_, 4, false, NT_>), dim3(nblk), dim3(256), 0, 0, R); })); } while (0)
                (void)bytes;
                if (O == 5) { RUN(5, 2); RUN(5, 3); RUN(5, 6); RUN(5, 7); } else { RUN(10, 3); RUN(10, 7); }
            }
            if (Fs != 482) { CK(hipFree(sp)); CK(hipFree(op)); }
        }
    }
    return 0;
}
