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
typedef float f32x2 __attribute__((ext_vector_type(2)));   // arithmetic on these compiles to the packed fp32 ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
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
// XCDs of the device (MI355X: 256 CUs in 8 XCDs; the runtime does not report the count: 32 CUs per XCD on gfx950)
static inline int dfx_env_num_xcds() { return dfx_env_num_cus() / 32; }
static inline bool dfx_env_is_emulator() { return false; }
// a value that is the same in every lane of a wave by construction (e.g. derived from threadIdx.x >> 6): tells the compiler so
// (address arithmetic on it stays in SGPRs and loads through it can be scalar loads)
static __device__ __forceinline__ int dfx_wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// non-temporal (streaming) global accesses for data that is touched exactly once: the lines are not kept in L2 / MALL at the
// expense of others (measured on a plain copy: 6.0 -> 6.7 TB/s, tools/dev/dfa_bench.hip)
#define DFX_NT_LOAD(p) __builtin_nontemporal_load(p)
#define DFX_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
// XCD (accelerator complex die) this wave runs on: HW_REG_XCC_ID[3:0].  The dispatcher deals the workgroups of a launch round-robin
// over the 8 XCDs but starts where the previous dispatch stopped (measured: tools/dev/xcd_probe.hip), so blockIdx % 8 identifies
// the XCD only up to a rotation that is constant within a launch; code that needs a *specific* XCD reads the register.
static __device__ __forceinline__ int dfx_xcc_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20); }
static inline hipError_t dfx_env_set_max_dyn_smem(const void *func, size_t bytes) {
    return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---- model error words.  They live in page-locked host memory that the device can write (fine-grained, uncached on the GPU): a kernel
// that finds a fault raises a word with ONE plain system-scope store (no read-modify-write: PCIe only carries add / swap / cas as atomics),
// the host reads the words with ordinary loads wherever it already waits for the device — no copy, no extra launch, nothing on the
// fault-free path.  (dev == host pointer value on this platform's unified addressing; both are kept.)
static inline hipError_t dfx_env_err_words_alloc(unsigned int **host, unsigned int **dev, size_t bytes) {
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    memset(h, 0, bytes);
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(h);
        return e;
    }
    *host = static_cast<unsigned int *>(h), *dev = static_cast<unsigned int *>(d);
    return hipSuccess;
}
static inline void dfx_env_err_words_free(unsigned int *host) {
    if (host) (void)hipHostFree(host);
}
static __device__ __forceinline__ void dfx_raise(unsigned int *word) { __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ---- fp16 split ("fp16x3") MFMA helpers: x = hi + lo with hi = f16(x), lo = f16(x - hi) carries 22 mantissa bits; the three
// products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (fp32 accumulate, f16 x f16 products are exact in fp32)
// reproduce an fp32 dot product to ~2^-21 relative at 16/3 x the fp32 MFMA rate.
typedef _Float16 dfx_h8 __attribute__((ext_vector_type(8)));
static __device__ __host__ __forceinline__ uint16_t dfx_f32_to_f16_bits(float x) {
    const _Float16 h = (_Float16)x;  // round to nearest even
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
static __device__ __host__ __forceinline__ float dfx_f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
// x = hi + lo, hi = f16(x), lo = f16(x - f32(hi)), both round-to-nearest.  Written on 2-vectors: gfx950 then converts two values per
// instruction (v_cvt_pk_f16_f32) and subtracts them packed (v_pk_add_f32) — 5 instructions per pair instead of 10 for the scalar form
// (convert, convert back, subtract, convert, pack, twice); the same operations on the same values, so the same bits.
typedef _Float16 dfx_h2v __attribute__((ext_vector_type(2)));
typedef float dfx_f2v __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void dfx_split8(const float *x, dfx_h8 &hi, dfx_h8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const dfx_f2v v = {x[i], x[i + 1]};
        const dfx_h2v h = __builtin_convertvector(v, dfx_h2v);
        const dfx_f2v r = v - __builtin_convertvector(h, dfx_f2v);
        const dfx_h2v l = __builtin_convertvector(r, dfx_h2v);
        hi[i] = h[0], hi[i + 1] = h[1];
        lo[i] = l[0], lo[i + 1] = l[1];
    }
}
// the same, also tracking the largest magnitude that went through the split (range guard of the fp16-split kernels: above 65504 the
// hi half is inf; the kernels report amax >= DFX_H3_LIMIT through the model's error word instead of handing back garbage)
#define DFX_H3_LIMIT 6.0e4f
static __device__ __forceinline__ void dfx_split8_g(const float *x, dfx_h8 &hi, dfx_h8 &lo, float &amax) {
#ifndef DFX_NO_H3_GUARD   /* dev: cost of the guard */
    amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))),
                             fmaxf(fmaxf(fabsf(x[4]), fabsf(x[5])), fmaxf(fabsf(x[6]), fabsf(x[7])))));
#endif
    dfx_split8(x, hi, lo);
}
// D[i][j] += sum_k A[i][k] B[k][j]; lane l: A row i = l&15, B col j = l&15, both hold k = 8*(l>>4) .. +7; D: col = l&15, row = 4*(l>>4)+r
// DFX_MFMA_K16 (build option, tools/dev/build_variant.sh k16 -DDFX_MFMA_K16=1): the same contraction as two v_mfma_f32_16x16x16_f16 (the gfx90a
// operation: k-slots 0..3 of every lane group, then 4..7).  Why one would want it: on the MI355X this was developed on the double-rate operation
// disturbs the packed fp32 arithmetic of OTHER kernels that run on the GPU at the same time (docs/measurements.md R6.1); the
// gfx90a operation does not.  Twice the matrix-op count; the results differ from the default build's in the last bits (order of the fp32 sums).
static __device__ __forceinline__ f32x4 dfx_mfma_16x16x32_f16(dfx_h8 a, dfx_h8 b, f32x4 c) {
#ifdef DFX_MFMA_K16
    typedef _Float16 dfx_h4v __attribute__((ext_vector_type(4)));
    const dfx_h4v a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]}, b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// lane gather with a ready-made byte address (4 * source lane, < 256): one ds_bpermute_b32, where __shfl() first masks and shifts the index
static __device__ __forceinline__ float dfx_lane_gather4(float v, unsigned byte_addr) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)byte_addr, __builtin_bit_cast(int, v)));
}
// small index products at the full VALU rate (v_mul_i32_i24; v_mul_lo_u32 issues at a quarter of it)
#define DFX_MUL24(a, b) __mul24((a), (b))
// compiler fences used by the hand-scheduled kernels
#define DFX_OPAQUE(x) asm volatile("" : "+v"(x))
#define DFX_ASSUME(c) __builtin_assume(c)   /* what an opaque value is known to be, e.g. a lane index: 0 <= x < 64 (lets the compiler drop clamps that never bind) */
#define DFX_PIN_AGPR(x) asm volatile("" : "+a"(x))   /* the value lives in an AGPR from here on (matrix-op operands of kernels with one wave per SIMD) */
#define DFX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Between the last matrix op of a group and the first vector instruction that reads its result when that reader may land in the NEXT basic
// block (an `if (valid)` epilogue: s_and_saveexec, then the read).  Round 5: dfx_fft480_mfma returned one wrong bin in ~10 % of its runs on the
// GPU (never on the interpreter) where the compiler had left 8 wait states between a chain of back-to-back DEPENDENT matrix ops and such a
// read; four more — or just keeping the scheduler from sinking the ops to the end of their block — and 450 of 450 runs were right
// (tools/dev/dbg_mf.py, tools/dev/scan_mfma_reads.py).  Chains are also kept three ops apart (term-major order), as everywhere else here.
#define DFX_MFMA_GUARD()                         \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        asm volatile("s_nop 7");                 \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
// same-XCD hand-overs (DfxXcd): every outstanding vector-memory operation of the wave has completed (stores: acknowledged by the L2) / this CU's
// L1 holds nothing stale (group-scope invalidate: the L2, shared by the XCD's CUs, is left alone)
#define DFX_VMEM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DFX_L1_INV() asm volatile("buffer_inv sc0" ::: "memory")
#define DFX_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)   /* ask for n instructions of a class next (0x008 MFMA, 0x002 VALU) */
// barrier + LDS visibility among the 64 lanes of ONE wave (LDS operations of a wave complete in order; the fences only pin the
// compiler's ordering) — costs nothing compared with s_barrier across the workgroup
#ifdef DFX_WAVE_SYNC_WAIT   /* dev (round 6): the wave's LDS operations have completed before it goes on */
#define DFX_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#else
#define DFX_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#endif

static __device__ __forceinline__ float dfx_fast_exp(float x) { return __expf(x); }
// v_rcp_f32: one instruction, 1 ulp.  (__frcp_rn — the correctly rounded reciprocal — is a ten-instruction sequence: v_div_scale, v_rcp,
// four v_fma, v_div_fmas, v_div_fixup; the GRU gates take three reciprocals per hidden unit and step, on the latency chain of the step.)
#ifndef DFX_RCP_RN
static __device__ __forceinline__ float dfx_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
static __device__ __forceinline__ float dfx_fast_rcp(float x) { return __frcp_rn(x); }
#endif
