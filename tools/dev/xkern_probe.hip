// Dev (round 6): which hardware path of a VICTIM kernel is disturbed while dfx_k_df_convp_h3 (no LDS, 352 registers, one wave per SIMD) runs on
// another stream?  Victims are self-describing: every output value encodes (iteration, lane), so a wrong value says where it came from.
//   lds   : a wave writes pattern(it, lane) into its own LDS row, reads it back through a lane permutation (wave-synchronous, like dfx_fft480_ip)
//   ldsb  : the same with __syncthreads() around the exchange (workgroup barrier instead of the wave-level assumption)
//   valu  : a chain of dependent FMAs in registers, no memory
//   gld   : a table in global memory read through L1 / L2 with a permutation
// Aggressors: convp (the real kernel on synthetic operands), mfma (a plain loop of v_mfma_f32_16x16x32_f16 chains), valu (FMA loop), none.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -Ideepfilternet_amd/csrc/env_hip -Ideepfilternet_amd/csrc tools/dev/xkern_probe.hip -o tools/dev/_build/xkern_probe
#include "dfx_nn_kernels.h"
#include "dfx.h"
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}

static __device__ __forceinline__ float pat(int it, int blk, int lane) { return (float)(it * 4096 + (blk & 15) * 256 + lane); }

template <int MODE>
__global__ void __launch_bounds__(256) victim(float *out, const float *table, int iters, int lds_pad_bytes) {
    extern __shared__ float2 sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2 *row = sm + wave * 64;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float v;
        const int src = (lane * 17 + 5 + it) & 63;
        if (MODE == 0 || MODE == 1) {
            if (MODE == 1) __syncthreads();
            row[lane] = make_float2(pat(it, blockIdx.x, lane), -pat(it, blockIdx.x, lane));
            if (MODE == 1) __syncthreads();
            else DFX_WAVE_SYNC();
            const float2 g = row[src];
            if (MODE == 0) DFX_WAVE_SYNC();
            v = g.x + 0.5f * (g.x + g.y);   // = g.x when the two halves belong together
        } else if (MODE == 2) {
            acc = pat(it, blockIdx.x, lane);
#pragma unroll 16
            for (int k = 0; k < 64; ++k) acc = fmaf(acc, 1.0000001f, 0.25f);
            v = acc;
        } else if (MODE == 4) {   // packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), as complex arithmetic compiles to
            f32x2 a = {pat(it, blockIdx.x, lane), 1.f + 0.001f * lane}, b = {1.0000001f, 0.9999999f}, c = {0.25f, -0.125f};
#pragma unroll 16
            for (int k = 0; k < 64; ++k) {
                a = a * b + c;
                a = a + f32x2{a[1], a[0]} * 1e-3f;
            }
            v = a[0] + a[1];
        } else if (MODE == 5 || MODE == 6 || MODE == 7) {   // one packed op only: fma / mul / add
            f32x2 a = {pat(it, blockIdx.x, lane), 1.f + 0.001f * lane};
            const f32x2 b = {1.0000001f, 0.9999999f}, c = {0.25f, -0.125f};
#pragma unroll 16
            for (int k = 0; k < 64; ++k) {
                if (MODE == 5) a = __builtin_elementwise_fma(a, b, c);
                else if (MODE == 6) a = a * b;
                else a = a + c;
            }
            v = a[0] + a[1];
        } else {
            v = table[(blockIdx.x & 63) * 64 + src];
        }
        out[((size_t)blockIdx.x * iters + it) * 256 + tid] = v;
    }
}

__global__ void __launch_bounds__(256) agg_mfma(float *sink, int iters) {
    dfx_h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i)), b[i] = (_Float16)(0.002f * (threadIdx.x - i));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = dfx_mfma_16x16x32_f16(a, b, c0);
        c1 = dfx_mfma_16x16x32_f16(a, b, c1);
        c2 = dfx_mfma_16x16x32_f16(a, b, c2);
        c3 = dfx_mfma_16x16x32_f16(a, b, c3);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
__global__ void __launch_bounds__(256) agg_mfma32(float *sink, int iters) {   // fp32 matrix ops (v_mfma_f32_16x16x4_f32)
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(256) agg_mfma_k(float *sink, int iters) {
    dfx_h8 a8, b8;
    h4v a4, b4;
    for (int i = 0; i < 8; ++i) a8[i] = (_Float16)(0.001f * (threadIdx.x + i)), b8[i] = (_Float16)(0.002f * (threadIdx.x - i));
    for (int i = 0; i < 4; ++i) a4[i] = a8[i], b4[i] = b8[i];
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 d[2] = {};
    float r = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {          // v_mfma_f32_16x16x16_f16 (the gfx90a form)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[j], 0, 0, 0);
        } else if constexpr (KIND == 1) {   // v_mfma_f32_32x32x8_f16
#pragma unroll
            for (int j = 0; j < 2; ++j) d[j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d[j], 0, 0, 0);
        } else if constexpr (KIND == 2) {   // v_mfma_f32_32x32x16_f16 (gfx950)
#pragma unroll
            for (int j = 0; j < 2; ++j) d[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[j], 0, 0, 0);
        } else if constexpr (KIND == 3) {   // v_mfma_f32_16x16x32_bf16 (gfx950)
            bf8v ab, bb;
            __builtin_memcpy(&ab, &a8, 16), __builtin_memcpy(&bb, &b8, 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c[j], 0, 0, 0);
        }
    }
    for (int j = 0; j < 4; ++j) r += c[j][0];
    r += d[0][0] + d[1][5];
    if (r == 12345.f) sink[0] = 1.f;
}
__global__ void __launch_bounds__(256) agg_valu(float *sink, int iters) {
    float x = threadIdx.x * 0.001f, y = x + 1.f, z = y + 1.f, w = z + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) x = fmaf(x, 1.0001f, y), y = fmaf(y, 0.9999f, z), z = fmaf(z, 1.0002f, w), w = fmaf(w, 0.9998f, x);
    }
    if (x + y + z + w == 12345.f) sink[0] = 1.f;
}

int main(int argc, char **argv) {
    const std::string vic = argc > 1 ? argv[1] : "lds", agg = argc > 2 ? argv[2] : "convp";
    const int trials = argc > 3 ? atoi(argv[3]) : 20;
    const int lds_extra = argc > 4 ? atoi(argv[4]) : 0;    // extra dynamic LDS of the victim (bytes): fewer victim workgroups per CU
    const int vblocks = 4096, viters = 64;
    const size_t nout = (size_t)vblocks * viters * 256;
    float *out, *table, *sink;
    CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&table, 64 * 64 * 4)); CK(hipMalloc(&sink, 64));
    {
        std::vector<float> t(64 * 64);
        for (int i = 0; i < 64 * 64; ++i) t[i] = (float)(i * 7 + 1);
        CK(hipMemcpy(table, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    // aggressor operands (dfx_k_df_convp_h3<64, 5> at the bench's shape: 256 clips x 202 frames x 96 bins)
    const int64_t B = 256, T = 202; const int Fd = 96, NO = 10;
    float *feat, *bias0, *bias, *cout; dfx_h8 *w0f, *wf;
    CK(hipMalloc(&feat, B * T * Fd * 8)); CK(hipMalloc(&bias0, 64 * 4)); CK(hipMalloc(&bias, 16 * 4)); CK(hipMalloc(&cout, B * (NO / 2) * T * Fd * 8));
    CK(hipMalloc(&w0f, 4 * 2 * 64 * 16)); CK(hipMalloc(&wf, 5 * 2 * 2 * 64 * 16));
    {
        std::vector<float> h(B * T * Fd * 2);
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(feat, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bias0, 0, 64 * 4)); CK(hipMemset(bias, 0, 16 * 4));
        std::vector<_Float16> w(5 * 2 * 2 * 64 * 8);
        for (auto &v : w) v = (_Float16)((float)rand() / RAND_MAX - 0.5f);
        CK(hipMemcpy(w0f, w.data(), 4 * 2 * 64 * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(wf, w.data(), 5 * 2 * 2 * 64 * 16, hipMemcpyHostToDevice));
    }
    DfxCphArgs A;
    A.feat = feat, A.w0f = w0f, A.bias0 = bias0, A.wf = wf, A.bias = bias, A.out = cout, A.B = B, A.T = T, A.Fd = Fd, A.NO = NO, A.L = 2;
    A.t_begin = 0, A.t_zero = 0, A.t_end = T, A.unscale0 = 1.f, A.unscale = 1.f, A.err = nullptr, A.nfb = 6, A.feat_T = 0;
    A.tseg = 205, A.nseg = 1;   // one segment per (clip, bin block): 1536 runs
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const size_t vsmem = 4 * 64 * 8 + (size_t)lds_extra;
    auto launch_victim = [&]() {
        if (vic == "lds") { CK(hipFuncSetAttribute((const void *)victim<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vsmem)); hipLaunchKernelGGL(victim<0>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0); }
        else if (vic == "ldsb") { CK(hipFuncSetAttribute((const void *)victim<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vsmem)); hipLaunchKernelGGL(victim<1>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0); }
        else if (vic == "valu") hipLaunchKernelGGL(victim<2>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
        else if (vic == "pk") hipLaunchKernelGGL(victim<4>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
        else if (vic == "pkfma") hipLaunchKernelGGL(victim<5>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
        else if (vic == "pkmul") hipLaunchKernelGGL(victim<6>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
        else if (vic == "pkadd") hipLaunchKernelGGL(victim<7>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
        else hipLaunchKernelGGL(victim<3>, dim3(vblocks), dim3(256), vsmem, sv, out, table, viters, 0);
    };
    auto launch_agg = [&]() {
        for (int r = 0; r < 4; ++r) {
            if (agg == "convp") hipLaunchKernelGGL((dfx_k_df_convp_h3<64, 5>), dim3(384), dim3(256), 0, sa, A);
            else if (agg == "mfma") hipLaunchKernelGGL(agg_mfma, dim3(1024), dim3(256), 0, sa, sink, 20000);
            else if (agg == "mfma32") hipLaunchKernelGGL(agg_mfma32, dim3(1024), dim3(256), 0, sa, sink, 20000);
            else if (agg == "m16x16x16") hipLaunchKernelGGL(agg_mfma_k<0>, dim3(1024), dim3(256), 0, sa, sink, 20000);
            else if (agg == "m32x32x8") hipLaunchKernelGGL(agg_mfma_k<1>, dim3(1024), dim3(256), 0, sa, sink, 10000);
            else if (agg == "m32x32x16") hipLaunchKernelGGL(agg_mfma_k<2>, dim3(1024), dim3(256), 0, sa, sink, 10000);
            else if (agg == "bf16x32") hipLaunchKernelGGL(agg_mfma_k<3>, dim3(1024), dim3(256), 0, sa, sink, 20000);
            else if (agg == "valu") hipLaunchKernelGGL(agg_valu, dim3(2048), dim3(256), 0, sa, sink, 4000);
        }
    };
    if (vic == "ana") {   // the library's STFT kernel through the C ABI (dfx_analysis) as the victim
        dfx_state *st = nullptr;
        if (dfx_state_create(48000, 960, 480, 32, 2, &st)) { printf("state\n"); return 1; }
        const int64_t Bv = 256, Tv = 96960, Tf = Tv / 480, F = 481;
        float *x, *spec;
        CK(hipMalloc(&x, Bv * Tv * 4)); CK(hipMalloc(&spec, Bv * Tf * F * 8));
        std::vector<float> hx(Bv * Tv);
        for (auto &v : hx) v = 0.1f * ((float)rand() / (float)RAND_MAX - 0.5f);
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        const size_t ns = (size_t)Bv * Tf * F * 2;
        std::vector<float> r0(ns), r1(ns);
        if (dfx_analysis(st, x, Bv, Tv, Tv, nullptr, nullptr, spec, sv)) { printf("analysis failed\n"); return 1; }
        CK(hipStreamSynchronize(sv)); CK(hipMemcpy(r0.data(), spec, ns * 4, hipMemcpyDeviceToHost));
        int bad_trials = 0; long long bad_vals = 0;
        for (int tr = 0; tr < trials; ++tr) {
            CK(hipMemsetAsync(spec, 0, ns * 4, sv)); CK(hipStreamSynchronize(sv));
            launch_agg();
            dfx_analysis(st, x, Bv, Tv, Tv, nullptr, nullptr, spec, sv);
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa)); CK(hipGetLastError());
            CK(hipMemcpy(r1.data(), spec, ns * 4, hipMemcpyDeviceToHost));
            long long nb = 0;
            for (size_t i = 0; i < ns; ++i)
                if (r1[i] != r0[i]) {
                    if (nb < 3 && bad_trials < 4) printf("  trial %d: float %zu (frame %zu bin %zu part %zu) got %g expected %g\n", tr, i, i / (F * 2), (i % (F * 2)) / 2, i % 2, r1[i], r0[i]);
                    ++nb;
                }
            if (nb) ++bad_trials, bad_vals += nb;
        }
        printf("SUMMARY victim=ana aggressor=%s: %d of %d trials wrong, %lld wrong values\n", agg.c_str(), bad_trials, trials, bad_vals);
        return 0;
    }
    std::vector<float> ref(nout), got(nout);
    launch_victim(); CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(ref.data(), out, nout * 4, hipMemcpyDeviceToHost));
    launch_victim(); CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost));
    printf("victim %s solo repeatable: %d\n", vic.c_str(), (int)(memcmp(ref.data(), got.data(), nout * 4) == 0));
    int bad_trials = 0; long long bad_vals = 0; int shown = 0;
    for (int tr = 0; tr < trials; ++tr) {
        CK(hipMemsetAsync(out, 0, nout * 4, sv)); CK(hipStreamSynchronize(sv));
        launch_agg();
        launch_victim();
        CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa)); CK(hipGetLastError());
        CK(hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost));
        long long nb = 0;
        for (size_t i = 0; i < nout; ++i)
            if (got[i] != ref[i]) {
                ++nb;
                if (shown < 40) {
                    const int tid = (int)(i % 256), it = (int)((i / 256) % viters), blk = (int)(i / 256 / viters);
                    // decode what was read: value = it' * 4096 + (blk' & 15) * 256 + lane'
                    const long long g = (long long)got[i];
                    printf("  trial %d block %d iter %d wave %d lane %d: got %.1f (= iter %lld blk&15 %lld lane %lld) expected %.1f\n", tr, blk, it, tid >> 6, tid & 63,
                           got[i], g / 4096, (g / 256) % 16, g % 256, ref[i]);
                    ++shown;
                }
            }
        if (nb) ++bad_trials, bad_vals += nb;
    }
    printf("SUMMARY victim=%s aggressor=%s lds_extra=%d: %d of %d trials wrong, %lld wrong values\n", vic.c_str(), agg.c_str(), lds_extra, bad_trials, trials, bad_vals);
    return 0;
}
