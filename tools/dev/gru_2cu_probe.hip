// Dev (round 6, review item 3, deliverable 1): what does the per-step h exchange cost if a 16-clip group's recurrence is split over TWO CUs of one
// XCD (each holds half of W_hh resident, computes 128 of the 256 units, and needs the partner's 128 x 16 new h values before the next step)?
// Pairs of workgroups that found each other on one XCD (HW_REG_XCC_ID + a slot counter per XCD) run T steps of
//     matrix work (n_mfma chained v_mfma_f32_16x16x32_f16 per wave, stands for the half-size recurrence step)
//     -> store my 8 KB of h (fp32) -> drain -> barrier -> raise my step flag
//     -> wait for the partner's flag -> [invalidate L1] -> load its 8 KB
// with the SAME-XCD hand-over of R5.12 (s_waitcnt vmcnt(0), plain flag store, buffer_inv sc0) or the agent-scope release / acquire pair,
// alone or beside a streaming kernel on the other CUs.  usage: gru_2cu_probe <pairs> <steps> <n_mfma> <mode 0 light | 1 agent> <stream blocks, 0 = none>
#include "dfx_nn_kernels.h"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void dfx_set_error(const char *, ...) {}
bool dfx_prof_on(int) { return false; }
void dfx_prof_begin(int, hipStream_t) {}
void dfx_prof_end(int, hipStream_t) {}

struct PairArgs {
    float4 *buf;              // [pairs][2 sides][2 parities][512 float4] = 8 KB per side and parity
    unsigned int *flag;       // [pairs][2] (64-byte apart)
    unsigned int *slots;      // [8] per-XCD slot counters (zeroed before the launch)
    unsigned int *stat;       // [0] pairs whose sides ended on different XCDs (must be 0), [1] timeouts
    float *sink;
    int pairs, steps, n_mfma, mode;
};

__global__ void __launch_bounds__(256, 1) k_pair(PairArgs A) {
    extern __shared__ unsigned char whole_cu[];   // (160 KB of dynamic LDS: the workgroup owns its CU like a recurrence workgroup does)
    __shared__ int s_slot;
    bool dead = false;
    const int tid = threadIdx.x;
    const int xcd = dfx_xcc_id();
    if (tid == 0) s_slot = (int)atomicAdd(A.slots + xcd, 1u);
    __syncthreads();
    const int per = gridDim.x / 8;                 // workgroups per XCD (round-robin dispatch)
    const int slot = s_slot;
    if (slot >= per) {                             // the dispatcher gave this XCD more than its share: no partner here
        if (tid == 0) atomicAdd(A.stat, 1u);
        return;
    }
    const int pair = xcd * (per / 2) + slot / 2, side = slot & 1;
    float4 *mine = A.buf + ((size_t)(pair * 2 + side) * 2) * 512, *theirs = A.buf + ((size_t)(pair * 2 + (side ^ 1)) * 2) * 512;
    unsigned int *fmine = A.flag + (pair * 2 + side) * 16, *ftheirs = A.flag + (pair * 2 + (side ^ 1)) * 16;
    dfx_h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (tid + i)), b[i] = (_Float16)(0.002f * (tid - i));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float4 h0 = make_float4(tid, 1, 2, 3), h1 = h0;
    for (int t = 0; t < A.steps; ++t) {
        for (int i = 0; i < A.n_mfma; i += 4) {
            c0 = dfx_mfma_16x16x32_f16(a, b, c0);
            c1 = dfx_mfma_16x16x32_f16(a, b, c1);
            c2 = dfx_mfma_16x16x32_f16(a, b, c2);
            c3 = dfx_mfma_16x16x32_f16(a, b, c3);
        }
        h0.x += c0[0] + h1.x * 1e-9f, h1.y += c1[1] + c2[2] + c3[3];
        const int par = t & 1;
        mine[par * 512 + tid] = h0;
        mine[par * 512 + 256 + tid] = h1;
        if (A.mode == 0) {
            DFX_VMEM_DRAIN();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fmine, (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fmine, (unsigned int)(t + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0 && !dead) {
            int spins = 0;
            while ((int)(__hip_atomic_load(ftheirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned int)(t + 1)) < 0) {
                if (++spins > (1 << 22)) {
                    atomicAdd(A.stat + 1, 1u);
                    dead = true;   // (no second wait: a missing partner must not hold the GPU for steps x 2 s)
                    break;
                }
            }
        }
        __syncthreads();
        if (A.mode == 0) DFX_L1_INV();
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float4 g0 = theirs[par * 512 + tid], g1 = theirs[par * 512 + 256 + tid];
        h0.y = g0.x + g1.y, h1.x = g0.z + g1.w;
    }
    if (h0.x + h1.y == 12345.f) A.sink[0] = 1.f;
}

__global__ void __launch_bounds__(256) k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n, int reps) {
    __shared__ float pad_lds[256];   // (some LDS: cannot share a CU with a pair workgroup)
    pad_lds[threadIdx.x] = 0.f;
    for (int r = 0; r < reps; ++r)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

int main(int argc, char **argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 80, steps = argc > 2 ? atoi(argv[2]) : 1000, n_mfma = argc > 3 ? atoi(argv[3]) : 48, mode = argc > 4 ? atoi(argv[4]) : 0;
    const int sblocks = argc > 5 ? atoi(argv[5]) : 0;
    PairArgs A;
    CK(hipMalloc(&A.buf, (size_t)pairs * 2 * 2 * 512 * 16)); CK(hipMalloc(&A.flag, (size_t)pairs * 2 * 64)); CK(hipMalloc(&A.slots, 64)); CK(hipMalloc(&A.stat, 64)); CK(hipMalloc(&A.sink, 64));
    A.pairs = pairs, A.steps = steps, A.n_mfma = n_mfma, A.mode = mode;
    const int64_t NS = (int64_t)1 << 25;
    float4 *sin_, *sout; CK(hipMalloc(&sin_, NS * 16)); CK(hipMalloc(&sout, NS * 16)); CK(hipMemset(sin_, 0, NS * 16));
    hipStream_t sp, ss; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    float best = 1e9f; unsigned int st[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(A.flag, 0, (size_t)pairs * 2 * 64)); CK(hipMemset(A.slots, 0, 64)); CK(hipMemset(A.stat, 0, 64));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sp));
        hipLaunchKernelGGL(k_pair, dim3(2 * pairs), dim3(256), 160 * 1024 - 64, sp, A);
        CK(hipEventRecord(e1, sp));
        if (sblocks) hipLaunchKernelGGL(k_stream, dim3(sblocks), dim3(256), 0, ss, sin_, sout, NS, 64);   // (~2 GB per rep x 64: outlasts the pairs)
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st, A.stat, 8, hipMemcpyDeviceToHost));
        if (st[0] == 0 && st[1] == 0 && ms < best) best = ms;
    }
    printf("pairs %d steps %d n_mfma %d hand-over %s stream blocks %d: %.3f us per step (unpaired workgroups %u, timeouts %u in the last run)\n", pairs, steps, n_mfma,
           mode == 0 ? "same-XCD light" : "agent-scope", sblocks, best * 1e3f / steps, st[0], st[1]);
    return 0;
}
