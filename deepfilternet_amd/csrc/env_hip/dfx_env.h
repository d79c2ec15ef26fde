// Product environment for the dfx kernels: plain HIP on gfx950 (MI355X).  The only other file with this name is the
// CPU interpreter used by unit tests (tests/hipemu/dfx_env.h); it is never part of libdfx.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// dynamic LDS carve: base kept 16-byte aligned (cdna_hip_programming.md Guideline 17)
#define DFX_DYN_SMEM(T, name)                                                        \
    extern __shared__ __attribute__((aligned(16))) unsigned char dfx_dyn_smem_raw[]; \
    T *name = reinterpret_cast<T *>(dfx_dyn_smem_raw)

template <typename... KArgs, typename... Args>
static inline void dfx_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream,
                              Args &&...args) {
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, static_cast<KArgs>(args)...);
}

static inline int dfx_env_num_cus() {
    static int n = -1;
    if (n < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        else
            n = 256;
    }
    return n;
}
static inline bool dfx_env_is_emulator() { return false; }
static inline hipError_t dfx_env_set_max_dyn_smem(const void *func, size_t bytes) {
    return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// compiler fences used by the hand-scheduled kernels
#define DFX_OPAQUE(x) asm volatile("" : "+v"(x))
#define DFX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)

static __device__ __forceinline__ float dfx_fast_exp(float x) { return __expf(x); }
static __device__ __forceinline__ float dfx_fast_rcp(float x) { return __frcp_rn(x); }
