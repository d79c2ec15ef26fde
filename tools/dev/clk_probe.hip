// Dev: shader clock of a compute-only wave alone and beside a chip-wide streaming kernel (does the chip lower its clock under memory load?)
// One wave runs a fixed chain of dependent FMAs; s_memtime counts shader cycles, wall_clock64 a constant 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_spin(unsigned long long *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = __builtin_fmaf(a, b, 1e-7f);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    if (a == 123.f) out[0] = 0;
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
// the same with matrix ops: 8 independent accumulators, back to back (MODE 0), or LDS reads only (MODE 1), or barriers only (MODE 2)
template <int MODE> __global__ void __launch_bounds__(256, 1) k_spin2(unsigned long long *out, int iters) {
    __shared__ h8 sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = h8{1, 1, 1, 1, 1, 1, 1, 1};
    __syncthreads();
    h8 a = sm[threadIdx.x], b = sm[threadIdx.x + 256];
    f4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f4{0, 0, 0, 0};
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 7], 0, 0, 0);
        } else if (MODE == 3) {   // the same matrix ops in a loop body of ~16 KB of code (2048 ops unrolled)
#pragma unroll
            for (int j = 0; j < 2048; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 7], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { h8 v = sm[(threadIdx.x + 64 * j + i) & 2047]; acc[j & 7][0] += (float)v[0]; }
        } else {
            __syncthreads(); acc[0][0] += 1.f; __syncthreads(); acc[1][0] += 1.f;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    float t = 0; for (int j = 0; j < 8; ++j) t += acc[j][0];
    if (t == 123.f) out[0] = 0;
}
__global__ void __launch_bounds__(256) k_copy(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n, int reps, int ro) {
    float4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const float4 v = in[i];
            if (ro) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; } else out[i] = v;
        }
    if (ro && acc.x == 12345.678f) out[0] = acc;
}
int main() {
    const int64_t NS = (int64_t)1 << 26;
    float4 *in, *out; CK(hipMalloc(&in, NS * 16)); CK(hipMalloc(&out, NS * 16)); CK(hipMemset(in, 0, NS * 16));
    unsigned long long *d; CK(hipMalloc(&d, 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int kind = 0; kind < 5; ++kind)
    for (int mode = 0; mode < 3; ++mode) {   // 0 alone, 1 beside read-only stream, 2 beside copy
        for (int it = 0; it < 3; ++it) {
            CK(hipDeviceSynchronize());
            if (mode) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s1, in, out, NS, 24, mode == 1);
            if (kind == 0) hipLaunchKernelGGL(k_spin, dim3(80), dim3(256), 0, s2, d, 40000);
            if (kind == 1) hipLaunchKernelGGL(k_spin2<0>, dim3(80), dim3(256), 0, s2, d, 20000);
            if (kind == 2) hipLaunchKernelGGL(k_spin2<1>, dim3(80), dim3(256), 0, s2, d, 20000);
            if (kind == 3) hipLaunchKernelGGL(k_spin2<2>, dim3(80), dim3(256), 0, s2, d, 100000);
            if (kind == 4) hipLaunchKernelGGL(k_spin2<3>, dim3(80), dim3(256), 0, s2, d, 20000 / 64);
            CK(hipDeviceSynchronize());
            unsigned long long h[160]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            double mn = 1e9, mx = 0;
            for (int b = 0; b < 80; ++b) { const double mhz = (double)h[2 * b] / ((double)h[2 * b + 1] / 100.0); if (mhz < mn) mn = mhz; if (mhz > mx) mx = mhz; }
            if (it == 2) printf("%s, mode %d (%s): spin kernel %.3f ms, shader clock %.0f .. %.0f MHz over 80 workgroups\n", kind == 0 ? "fma chain" : kind == 1 ? "matrix ops" : kind == 2 ? "LDS reads" : kind == 3 ? "barriers" : "matrix ops, 16 KB loop body", mode, mode == 0 ? "alone" : mode == 1 ? "beside a read-only stream" : "beside a copy", h[1] / 1e5, mn, mx);
        }
    }
    return 0;
}
