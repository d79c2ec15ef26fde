// ---------------------------------------------------------------------------------------------------------------------
// The GRU recurrence on a PAIR of CUs (round 6): included by dfx_nn_kernels.h behind dfx_gru_h3_run, whose arithmetic it repeats bit for bit.
//
// dfx_gru_h3_run gives 16 clips to one CU; W_hh (768 x 256 weights as f16 hi / lo = 786 KB) does not fit there, 54 of a wave's 96 fragment pairs are
// streamed from the L2 every step (432 KB per step and CU), and the step runs at the CU's L2 fill rate: 5.1 us alone, 7.3-8.2 under the side traffic
// of the phase, against 1.9 us of matrix operations (docs/measurements.md R5.12, R6.2).  Here TWO workgroups of one XCD serve 32 clips together:
// half p holds the rows of W_hh for the units [128 p, 128 p + 128) of all three gates — 48 fragment pairs per wave, 33 in registers and 15 in LDS,
// nothing streamed —, computes those units for all 32 clips (two 16-column matrix-op tiles per fragment: 288 matrix operations per wave and step as
// before) and needs the partner's 128 x 32 new h values before the next step.  The exchange medium is y itself (each half stores its units of row t
// anyway): store -> drain -> barrier -> step flag -> wait for the partner's step flag -> L1 invalidate -> load its half of row t, with R5.12's same-XCD
// hand-over (no L2 write-back, no L2 invalidate) when both halves registered the same XCD and the agent-scope release / acquire pair otherwise
// (blocks b and b + 8 of a launch sit on one XCD under the round-robin dispatch; placement = speed only).  CUs: a pair per 32 clips = the same 80
// workgroups at 256 clips as one per 16.
//
// Bits: a lane's accumulators see the same products in the same order as in dfx_gru_h3_run (k-chunks ascending; lo*hi, hi*lo, hi*hi), the gate
// math is the same code, the f16 hi / lo copies of h are made by the same conversion from the same fp32 values: y is bit-identical
// (tools/dev/gru_p2_bench.hip compares; tests/test_enhance.py's variant test carries DFX_GRU_PAIR=0 / 1).
// Reference: DeepFilterNet/df/modules.py:702-738 (SqueezedGRU_S), :721 (nn.GRU).
// ---------------------------------------------------------------------------------------------------------------------
#pragma once

#define DFX_GP_NW 4
#define DFX_GP_THREADS (64 * DFX_GP_NW)
#define DFX_GP_NS 2                         /* 16-unit sub-tiles per gate and wave */
#define DFX_GP_TILES (3 * DFX_GP_NS)        /* accumulator tiles per wave and clip tile */
#define DFX_GP_NF (8 * DFX_GP_TILES)        /* fragment pairs per wave: all resident */
#ifndef DFX_GP_FL
#define DFX_GP_FL 15                        /* pairs per wave in LDS */
#endif
#define DFX_GP_FR (DFX_GP_NF - DFX_GP_FL)   /* pairs per wave in registers */
#ifndef DFX_GP_PIN
#define DFX_GP_PIN 30                       /* of those, pinned in the accumulation half of the register file (256 registers = 32 pairs at most) */
#endif
#ifndef DFX_GP_ABLATE
#define DFX_GP_ABLATE 0   /* dev (tools/dev/gru_p2_bench.hip): 1 no partner wait / load, 2 no gate math, 4 no matrix ops, 8 no drain */
#endif
#ifndef DFX_GP_TRACE
#define DFX_GP_TRACE 0    /* dev: thread 0 of every workgroup sums the shader-clock ticks between the phases of a step into DfxGpSync::ptrace[block][8] */
#endif
#if DFX_GP_TRACE
#define DFX_GP_TICK(i)                                                                  \
    do {                                                                                \
        unsigned long long now_;                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");    \
        tk[i] += now_ - tlast;                                                          \
        tlast = now_;                                                                   \
    } while (0)
#else
#define DFX_GP_TICK(i) do { } while (0)
#endif
#define DFX_GP_ROWS 32                      /* clips per pair */
#define DFX_GP_SMEM_W ((size_t)DFX_GP_FL * DFX_GP_NW * 2 * 64 * 16)
#define DFX_GP_SMEM_H ((size_t)2 * DFX_GP_ROWS * DFX_GH_HROW * 2)
#define DFX_GP_SMEM (DFX_GP_SMEM_W + DFX_GP_SMEM_H + 64)
#define DFX_GP_POS_KC(f) ((f) / DFX_GP_TILES)
#define DFX_GP_POS_GATE(f) (((f) % DFX_GP_TILES) / DFX_GP_NS)
#define DFX_GP_POS_S(f) (((f) % DFX_GP_TILES) % DFX_GP_NS)
#define DFX_GP_POS_TILE(f) ((f) % DFX_GP_TILES)

struct DfxGpSched {
    int cls[DFX_GP_NF];   // 0 = register, 1 = LDS
    int idx[DFX_GP_NF];   // index inside its class
};
static constexpr DfxGpSched dfx_gp_make_sched() {   // the LDS-resident pairs spread evenly among the register-resident ones
    DfxGpSched sc{};
    int nr = 0, nl = 0;
    for (int f = 0; f < DFX_GP_NF; ++f) {
        const bool lds = ((f + 1) * DFX_GP_FL / DFX_GP_NF) != (f * DFX_GP_FL / DFX_GP_NF);
        sc.cls[f] = lds ? 1 : 0;
        sc.idx[f] = lds ? nl++ : nr++;
    }
    return sc;
}

// Synchronisation of one half of a pair.  Step counters are pbase + steps, monotonic over the life of the model like DfxGhSync's.
struct DfxGpSync {
    unsigned int *pflag = nullptr;        // [2 halves] 16 words apart: steps whose y rows this half has stored (and drained / released)
    unsigned int *pxcd = nullptr;         // [2 halves]: tag | (xcd + 1)
    unsigned int pbase = 0, tag = 0;
    int spin_limit = 1 << 22;
    unsigned int *err = nullptr;
    unsigned int *stat = nullptr;         // dev aid: [1] counts pairs whose halves sit on different XCDs
    // chunk level (persistent form; as DfxGhSync)
    const unsigned int *ready = nullptr;  // chunks of gi available for this layer (layer 0: launches per chunk), ignored when giprog is set
    unsigned int *done = nullptr;         // this half's group word: chunks of y completed
    unsigned int base = 0;
    int K = 1;
    const int *tb = nullptr;
    unsigned long long *trace = nullptr;
    // followers (DfxGhSync::yprog / giprog): half p announces group 2 pg + p's rows — complete once the partner's step flag has been seen —
    // and waits for the followers of BOTH groups (it needs gi of all 32 clips)
    unsigned int *yprog = nullptr;
    const unsigned int *giprog[2] = {nullptr, nullptr};
    int sblk = 16, yblk = 16;
    unsigned int *xme = nullptr;                          // registration word of group 2 pg + p's recurrence (the followers' claims read it)
    const unsigned int *xprod[2] = {nullptr, nullptr};    // ... of the followers that feed the two groups
    const unsigned int *xcons = nullptr;                  // ... of the follower that consumes group 2 pg + p's rows
    unsigned long long *ptrace = nullptr;                 // dev (DFX_GP_TRACE)
    int far = 0;                                          // test hook (DFX_GRU_PAIR_FAR=1): treat every partner / follower as if it sat on another XCD (the agent-scope forms)
};

template <bool SEQ>
static __device__ __forceinline__ void dfx_gru_p2_run(const DfxGhArgs &A, int64_t pg, int half, const DfxGpSync &Y) {
    constexpr int H = 256, NF = DFX_GP_NF, HROW = DFX_GH_HROW, NW = DFX_GP_NW, NS = DFX_GP_NS, TILES = DFX_GP_TILES, ROWS = DFX_GP_ROWS;
    constexpr DfxGpSched SC = dfx_gp_make_sched();
    DFX_DYN_SMEM(unsigned char, smraw);
    dfx_h8 *wl = reinterpret_cast<dfx_h8 *>(smraw);                           // [FL][wave][hi,lo][lane]
    uint16_t *h16 = reinterpret_cast<uint16_t *>(smraw + DFX_GP_SMEM_W);      // [hi,lo][32][HROW]: ONE buffer (the exchange has the barriers a second one would save)
    volatile int *sflag = reinterpret_cast<volatile int *>(smraw + DFX_GP_SMEM_W + DFX_GP_SMEM_H);   // [0] the pair shares an L2, [1] so do both gi producers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int64_t b0 = pg * ROWS;
    bool valid[2];
    int64_t brow[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        valid[n] = b0 + 16 * n + jl < A.B;
        brow[n] = valid[n] ? b0 + 16 * n + jl : A.B - 1;
    }
    const int ubase = 128 * half + 32 * wave;   // this wave's first unit
    // fragment f = kc*TILES + gate*NS + s of this wave is global pair ((unit tile = 8 half + 2 wave + s)*8 + kc)*3 + gate  (DfxGhArgs::whf)
    const dfx_h8 *wg = A.whf + lane;
#define DFX_GP_GIDX(f) (((((size_t)(8 * half + NS * wave + DFX_GP_POS_S(f))) * 8 + DFX_GP_POS_KC(f)) * 3 + DFX_GP_POS_GATE(f)) * 2 * 64)
    dfx_h8 wr[DFX_GP_FR][2];
    dfx_static_for<0, NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        if constexpr (SC.cls[f] == 0) {
            wr[SC.idx[f]][0] = wg[DFX_GP_GIDX(f)];
            wr[SC.idx[f]][1] = wg[DFX_GP_GIDX(f) + 64];
            if constexpr (SC.idx[f] < DFX_GP_PIN) {
                DFX_PIN_AGPR(wr[SC.idx[f]][0]);
                DFX_PIN_AGPR(wr[SC.idx[f]][1]);
            }
        } else {
            wl[((SC.idx[f] * NW + wave) * 2 + 0) * 64 + lane] = wg[DFX_GP_GIDX(f)];
            wl[((SC.idx[f] * NW + wave) * 2 + 1) * 64 + lane] = wg[DFX_GP_GIDX(f) + 64];
        }
    });
    // ---- state: this lane owns clips jl and 16 + jl, units ubase + 16 s + 4 q + r
    float hp[2][NS][4];
    float4 bn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        bn[s] = *reinterpret_cast<const float4 *>(A.bhn + ubase + 16 * s + 4 * q);
#pragma unroll
        for (int n = 0; n < 2; ++n) hp[n][s][0] = hp[n][s][1] = hp[n][s][2] = hp[n][s][3] = 0.f;
    }
    auto put4 = [&](int row, int col, float v0, float v1, float v2, float v3) {   // f16 hi / lo of four consecutive units of one clip -> LDS (dfx_gru_h3_run's conversion)
        const float v[4] = {v0, v1, v2, v3};
        uint16_t hh[4], hl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hh[r] = dfx_f32_to_f16_bits(v[r]);
            hl[r] = dfx_f32_to_f16_bits(v[r] - dfx_f16_bits_to_f32(hh[r]));
        }
        *reinterpret_cast<uint2 *>(h16 + ((size_t)0 * ROWS + row) * HROW + col) = make_uint2((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16));
        *reinterpret_cast<uint2 *>(h16 + ((size_t)1 * ROWS + row) * HROW + col) = make_uint2((uint32_t)hl[0] | ((uint32_t)hl[1] << 16), (uint32_t)hl[2] | ((uint32_t)hl[3] << 16));
    };
    for (int i = tid; i < (int)(DFX_GP_SMEM_H / 4); i += DFX_GP_THREADS) reinterpret_cast<uint32_t *>(h16)[i] = 0u;   // h(0) = 0
    // ---- the pair finds out whether it shares an L2
    unsigned int *const fmine = Y.pflag + 16 * half;
    const unsigned int *const ftheirs = Y.pflag + 16 * (half ^ 1);
    bool dead = false;   // (thread 0) a wait of this workgroup has timed out: err[2] is raised, no further wait holds the device
    if (tid == 0) {
        const unsigned int me = Y.tag | (unsigned int)(dfx_xcc_id() + 1);
        __hip_atomic_store(Y.pxcd + half, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (Y.xme) __hip_atomic_store(Y.xme, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int v;
        int spins = 0;
        while ((((v = __hip_atomic_load(Y.pxcd + (half ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~15u) != Y.tag) || (v & 15u) == 0u) {
            if (++spins > Y.spin_limit) {
                dfx_raise(Y.err + 2);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sflag[0] = (!dead && v == me && !Y.far) ? 1 : 0;
        sflag[1] = 0;
        if (Y.stat && half == 0 && v != me) __hip_atomic_fetch_add(Y.stat + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool light = sflag[0] != 0;
    const float *gp[2];
    float *yp[2];
    const float *ypo[4];   // the partner's half of a row: this thread's four float4 pieces (row tid / 8, pieces tid % 8 + 8 i)
    const int prow = tid >> 3, pcol = 128 * (half ^ 1) + 4 * (tid & 7);
    {
        const int64_t r = b0 + prow < A.B ? b0 + prow : A.B - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) ypo[i] = A.y + r * A.T * H + pcol + 32 * i;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        gp[n] = A.gi + brow[n] * A.T * (3 * H) + ubase + 4 * q;
        yp[n] = A.y + brow[n] * A.T * H + ubase + 4 * q;
    }
    float4 gv[2][3][NS];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < NS; ++s) gv[n][g][s] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
    auto load_gi = [&](int64_t t) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int s = 0; s < NS; ++s) gv[n][g][s] = *reinterpret_cast<const float4 *>(gp[n] + t * (3 * H) + g * H + 16 * s);
    };
    // (thread 0) the followers of both groups have stored the gi rows of steps < upto
    auto poll_gi = [&](int64_t upto) {
        const unsigned int want = Y.pbase + (unsigned int)(upto < A.T ? upto : A.T);
        bool same = true;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!Y.giprog[i]) continue;
            int spins = 0;
            while (!dead && (int)(__hip_atomic_load(Y.giprog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > Y.spin_limit) {
                    dfx_raise(Y.err + 2);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            same = same && Y.xprod[i] && __hip_atomic_load(Y.xprod[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (Y.tag | (unsigned int)(dfx_xcc_id() + 1));
        }
        sflag[1] = same && !Y.far ? 1 : 0;
    };
#if DFX_GP_TRACE
    unsigned long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    const int nchunk = SEQ ? Y.K : 1;
    for (int ck = 0; ck < nchunk; ++ck) {
        const int64_t c0 = SEQ ? (int64_t)Y.tb[ck] : A.t0, c1 = SEQ ? (int64_t)Y.tb[ck + 1] : A.t1;
        if (SEQ) {   // the input projection of this chunk must exist
            if (tid == 0 && Y.trace) Y.trace[ck * 3 + 0] = wall_clock64();
            if (tid == 0 && !Y.giprog[0] && Y.ready) {
                const unsigned int want = Y.base + (unsigned int)ck + 1u;
                int spins = 0;
                while (!dead && (int)(__hip_atomic_load(Y.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                    if (++spins > Y.spin_limit) {
                        dfx_raise(Y.err + 2);
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (tid == 0 && Y.giprog[0] && ck == 0) poll_gi(Y.sblk);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid == 0 && Y.trace) Y.trace[ck * 3 + 1] = wall_clock64();
        }
        if (c1 > c0) load_gi(c0);
        for (int64_t t = c0; t < c1; ++t) {
            const unsigned char *hb = reinterpret_cast<const unsigned char *>(h16 + (size_t)jl * HROW + 8 * q);
            constexpr size_t HB_N = (size_t)16 * HROW * 2, HB_LO = (size_t)ROWS * HROW * 2, HB_KSTEP = 32 * 2;
            f32x4 acc[2][TILES];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < TILES; ++i) acc[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            dfx_h8 bh[2][2], bl[2][2];   // [k-chunk parity][clip tile]
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[0][n] = *reinterpret_cast<const dfx_h8 *>(hb + HB_N * n);
                bl[0][n] = *reinterpret_cast<const dfx_h8 *>(hb + HB_N * n + HB_LO);
            }
            dfx_h8 lhi[2][3], llo[2][3];   // LDS-resident fragments of the current / next group
            dfx_static_for<0, 3>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (SC.cls[i] == 1) {
                    lhi[0][i] = wl[((SC.idx[i] * NW + wave) * 2 + 0) * 64 + lane];
                    llo[0][i] = wl[((SC.idx[i] * NW + wave) * 2 + 1) * 64 + lane];
                }
            });
            // three fragments (three accumulator tiles of one k-chunk) x two clip tiles per group: 18 matrix ops, consecutive ones on different accumulators
            dfx_static_for<0, NF / 3>([&](auto gc) {
                constexpr int grp = decltype(gc)::value, f0 = 3 * grp;
                constexpr int kc = DFX_GP_POS_KC(f0);
                if constexpr (DFX_GP_POS_TILE(f0) == 0 && kc + 1 < 8) {   // next k-chunk of h, a chunk ahead
                    constexpr int kn = kc + 1;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        bh[kn & 1][n] = *reinterpret_cast<const dfx_h8 *>(hb + HB_N * n + HB_KSTEP * kn);
                        bl[kn & 1][n] = *reinterpret_cast<const dfx_h8 *>(hb + HB_N * n + HB_LO + HB_KSTEP * kn);
                    }
                }
                // the LDS-resident fragments of the NEXT group are requested now (one group = 18 matrix ops ahead of their use)
                if constexpr (grp + 1 < NF / 3) {
                    dfx_static_for<0, 3>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, f = f0 + 3 + i;
                        if constexpr (SC.cls[f] == 1) {
                            lhi[(grp + 1) & 1][i] = wl[((SC.idx[f] * NW + wave) * 2 + 0) * 64 + lane];
                            llo[(grp + 1) & 1][i] = wl[((SC.idx[f] * NW + wave) * 2 + 1) * 64 + lane];
                        }
                    });
                }
                dfx_h8 whi[3], wlo[3];
                dfx_static_for<0, 3>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, f = f0 + i;
                    if constexpr (SC.cls[f] == 0) {
                        whi[i] = wr[SC.idx[f]][0];
                        wlo[i] = wr[SC.idx[f]][1];
                    } else {
                        whi[i] = lhi[grp & 1][i];
                        wlo[i] = llo[grp & 1][i];
                    }
                });
                constexpr int ta = DFX_GP_POS_TILE(f0), tb_ = DFX_GP_POS_TILE(f0 + 1), tc = DFX_GP_POS_TILE(f0 + 2);
                if constexpr (DFX_GP_ABLATE & 4) {
                    acc[0][ta][0] += (float)whi[0][0] + (float)wlo[1][1] + (float)whi[2][0] + (float)bh[kc & 1][0][0] + (float)bl[kc & 1][1][0];
                    acc[1][tb_][0] += (float)wlo[0][0] + (float)whi[1][1] + (float)wlo[2][0] + (float)bh[kc & 1][1][0] + (float)bl[kc & 1][0][0];
                } else {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[n][ta] = dfx_mfma_16x16x32_f16(wlo[0], bh[kc & 1][n], acc[n][ta]);
                    acc[n][tb_] = dfx_mfma_16x16x32_f16(wlo[1], bh[kc & 1][n], acc[n][tb_]);
                    acc[n][tc] = dfx_mfma_16x16x32_f16(wlo[2], bh[kc & 1][n], acc[n][tc]);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[n][ta] = dfx_mfma_16x16x32_f16(whi[0], bl[kc & 1][n], acc[n][ta]);
                    acc[n][tb_] = dfx_mfma_16x16x32_f16(whi[1], bl[kc & 1][n], acc[n][tb_]);
                    acc[n][tc] = dfx_mfma_16x16x32_f16(whi[2], bl[kc & 1][n], acc[n][tc]);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[n][ta] = dfx_mfma_16x16x32_f16(whi[0], bh[kc & 1][n], acc[n][ta]);
                    acc[n][tb_] = dfx_mfma_16x16x32_f16(whi[1], bh[kc & 1][n], acc[n][tb_]);
                    acc[n][tc] = dfx_mfma_16x16x32_f16(whi[2], bh[kc & 1][n], acc[n][tc]);
                }
                }
                DFX_SCHED_BARRIER();
            });
            DFX_GP_TICK(0);   // matrix ops issued
            // ---- gates, new state (dfx_gru_h3_run's gate_unit), y
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gr = r == 0 ? gv[n][0][s].x : r == 1 ? gv[n][0][s].y : r == 2 ? gv[n][0][s].z : gv[n][0][s].w;
                        const float gz = r == 0 ? gv[n][1][s].x : r == 1 ? gv[n][1][s].y : r == 2 ? gv[n][1][s].z : gv[n][1][s].w;
                        const float gn = r == 0 ? gv[n][2][s].x : r == 1 ? gv[n][2][s].y : r == 2 ? gv[n][2][s].z : gv[n][2][s].w;
                        const float bb = r == 0 ? bn[s].x : r == 1 ? bn[s].y : r == 2 ? bn[s].z : bn[s].w;
                        if (DFX_GP_ABLATE & 2) {
                            hp[n][s][r] = 0.5f * hp[n][s][r] + 1e-3f * (gr + gz + gn + bb + acc[n][s][r] + acc[n][NS + s][r] + acc[n][2 * NS + s][r]);
                            continue;
                        }
                        const float rg = dfx_fast_rcp(1.f + dfx_fast_exp(-(gr + acc[n][0 * NS + s][r] * A.unscale)));
                        const float zg = dfx_fast_rcp(1.f + dfx_fast_exp(-(gz + acc[n][1 * NS + s][r] * A.unscale)));
                        const float pre = gn + rg * (acc[n][2 * NS + s][r] * A.unscale + bb);
                        const float ng = 2.f * dfx_fast_rcp(1.f + dfx_fast_exp(-2.f * pre)) - 1.f;
                        hp[n][s][r] = (1.f - zg) * ng + zg * hp[n][s][r];
                    }
                    if (valid[n]) *reinterpret_cast<float4 *>(yp[n] + t * H + 16 * s) = make_float4(hp[n][s][0], hp[n][s][1], hp[n][s][2], hp[n][s][3]);
                }
            // ---- the exchange: my units of row t are in the L2 (or released), everybody has read h(t) from the LDS
            DFX_GP_TICK(1);   // gates, y stores issued
            if (!(DFX_GP_ABLATE & 8)) DFX_VMEM_DRAIN();   // (s_barrier does not wait for the other waves' stores: each wave drains its own in front of it)
            DFX_GP_TICK(2);   // stores drained
            __syncthreads();
            DFX_GP_TICK(3);   // barrier B
            const bool gi_step = SEQ && Y.giprog[0] && ((t + 1) & (Y.sblk - 1)) == 0 && t + 1 < A.T;   // the next step requests the first row of the next block
            const unsigned int now = Y.pbase + (unsigned int)(t + 1);
            if (tid == 0) {
                if (light) __hip_atomic_store(fmine, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(fmine, now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int s = 0; s < NS; ++s) put4(16 * n + jl, ubase + 16 * s + 4 * q, hp[n][s][0], hp[n][s][1], hp[n][s][2], hp[n][s][3]);
            if (tid == 0) {
                int spins = 0;
                while (!(DFX_GP_ABLATE & 1) && !dead && (int)(__hip_atomic_load(ftheirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - now) < 0) {
                    if (++spins > Y.spin_limit) {
                        dfx_raise(Y.err + 2);
                        dead = true;
                        break;
                    }
                }
                if (gi_step) poll_gi(t + 1 + Y.sblk);
            }
            DFX_GP_TICK(4);   // own half in LDS, partner's flag seen
            __syncthreads();
            DFX_GP_TICK(5);   // barrier C
            if (light && !(gi_step && sflag[1] == 0)) DFX_L1_INV();
            else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // the partner's half first, the next step's gi behind it (vector loads return in order: the gi rows may come from HBM)
            const bool more = t + 1 < A.T && !(DFX_GP_ABLATE & 1);
            float4 o[4];
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = *reinterpret_cast<const float4 *>(ypo[i] + t * H);
            }
            DFX_SCHED_BARRIER();
            const int64_t tn = t + 1 < c1 ? t + 1 : t;
            load_gi(tn);
            DFX_SCHED_BARRIER();
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) put4(prow, pcol + 32 * i, o[i].x, o[i].y, o[i].z, o[i].w);
            }
            DFX_GP_TICK(6);   // partner's half loaded and in LDS
            if (SEQ && tid == 0 && Y.yprog && (((t + 1) & (Y.yblk - 1)) == 0 || t + 1 == A.T)) {   // a block of group 2 pg + p's rows is complete (both halves' units)
                const bool cons_same = !Y.far && Y.xcons && __hip_atomic_load(Y.xcons, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (Y.tag | (unsigned int)(dfx_xcc_id() + 1));
                if (cons_same) __hip_atomic_store(Y.yprog, Y.pbase + (unsigned int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(Y.yprog, Y.pbase + (unsigned int)(t + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            DFX_GP_TICK(7);   // barrier D
        }
        if (SEQ) {   // the pair's rows of chunk ck are complete (the last step's exchange): make them visible device-wide, then say so
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0 && Y.done) __hip_atomic_store(Y.done, Y.base + (unsigned int)ck + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0 && Y.trace) Y.trace[ck * 3 + 2] = wall_clock64();
        }
    }
#if DFX_GP_TRACE
    if (tid == 0 && Y.ptrace)
        for (int i = 0; i < 12; ++i) Y.ptrace[(size_t)blockIdx.x * 12 + i] = tk[i];
#endif
}
#undef DFX_GP_GIDX

// block -> (pair, half): blocks 16 j + x and 16 j + 8 + x are the halves of pair 8 j + x (one XCD under the round-robin dispatch)
static __device__ __forceinline__ void dfx_gp_block(unsigned int b, int &pair, int &half) {
    pair = (int)(b >> 4) * 8 + (int)(b & 7);
    half = (int)(b >> 3) & 1;
}
static inline unsigned int dfx_gp_grid(int pairs) { return (unsigned int)((pairs + 7) / 8) * 16u; }

// One layer, whole sequence, no chunk synchronisation (dev bench / tests): sync = [pairs][pflag: 32 words][pxcd: 2 words + pad] = 48 words per pair
struct DfxGpArgs {
    DfxGhArgs g;
    unsigned int *sync;
    unsigned int pbase, tag;
    unsigned int *err, *stat;
    int spin_limit;
    unsigned long long *ptrace = nullptr;
};
__global__ void __launch_bounds__(DFX_GP_THREADS, 1) dfx_k_gru_rec_p2(DfxGpArgs P) {
    int pair, half;
    dfx_gp_block(blockIdx.x, pair, half);
    if ((int64_t)pair * DFX_GP_ROWS >= P.g.B) return;
    DfxGpSync Y;
    Y.pflag = P.sync + (size_t)pair * 48, Y.pxcd = P.sync + (size_t)pair * 48 + 32;
    Y.pbase = P.pbase, Y.tag = P.tag, Y.spin_limit = P.spin_limit, Y.err = P.err, Y.stat = P.stat, Y.ptrace = P.ptrace;
    dfx_gru_p2_run<false>(P.g, pair, half, Y);
}

// All GRU layers of a forward pass in one persistent launch (dfx_k_gru_seq) on pairs: pair id = layer * P + pg with P = ceil(groups / 2) pairs per
// layer; half p of pair (l, pg) stands for group g = 2 pg + p towards the flag words of the phase (done, yprog, registration), which stay per
// 16-clip group — the followers and the host-launched consumers do not know about pairs.  A group index beyond `groups` (odd counts: the last
// pair's second half) has no words.  Grid: dfx_gp_grid(nlayers * P) blocks; needs S.psync and S.xtag unique per pass.
__global__ void __launch_bounds__(DFX_GP_THREADS, 1) dfx_k_gru_seq_p2(DfxGsArgs S) {
    int pair, half;
    dfx_gp_block(blockIdx.x, pair, half);
    const int P = (S.groups + 1) / 2;
    const int l = pair / P, pg = pair % P;
    if (l >= S.nlayers) return;
    const int g = 2 * pg + half;
    const bool has = g < S.groups;
    DfxGhArgs A;
    A.gi = S.gi[l];
    A.whf = S.whf[l];
    A.bhn = S.bhn[l];
    A.h_in = nullptr;
    A.h_out = nullptr;
    A.y = S.y[l];
    A.B = S.B;
    A.T = S.T;
    A.t0 = 0;
    A.t1 = S.T;
    A.unscale = S.unscale[l];
    A.xcd_mask = 0;
    DfxGpSync Y;
    Y.pflag = S.psync + (size_t)pair * 48, Y.pxcd = S.psync + (size_t)pair * 48 + 32;
    Y.pbase = S.pbase, Y.tag = S.xtag, Y.spin_limit = S.spin_limit, Y.err = S.err, Y.stat = S.xstat;
    Y.ready = S.ready + l, Y.done = has ? S.done + (size_t)l * S.done_stride + g : nullptr, Y.base = S.base, Y.K = S.K, Y.tb = S.tb;
    Y.trace = S.trace && has ? S.trace + ((size_t)l * S.groups + g) * S.K * 3 : nullptr;
    Y.yprog = S.yprog[l] && has ? S.yprog[l] + g : nullptr;
    Y.sblk = S.sblk, Y.yblk = S.yblk[l] > 0 ? S.yblk[l] : 16;
    Y.far = S.pair_far;
    for (int i = 0; i < 2; ++i) Y.giprog[i] = S.giprog[l] && 2 * pg + i < S.groups ? S.giprog[l] + 2 * pg + i : nullptr;
    if (S.xtab) {
        auto word = [&](int kind, int layer, int grp) { return S.xtab + ((size_t)kind * DFX_GS_MAX_LAYERS + layer) * S.xstride + grp; };
        Y.xme = has ? word(0, l, g) : nullptr;
        for (int i = 0; i < 2; ++i) Y.xprod[i] = Y.giprog[i] ? word(1, l, 2 * pg + i) : nullptr;
        Y.xcons = Y.yprog ? word(S.xcons_kind[l], S.xcons_layer[l], g) : nullptr;
    }
    dfx_gru_p2_run<true>(A, pg, half, Y);
}
