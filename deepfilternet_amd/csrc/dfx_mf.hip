// Multi-frame Wiener / MVDR filter ops (df/multiframe.py: MfWf :221-321, MfMvdr :324-413, _tik_reg :436-452) — the filter stage of
// the reference's DeepFilterNetMF model (deepfilternetmf.py:335-352), same data layout as the deep filter.
//
// Per (clip b, frame t, bin f < nb): an N x N complex matrix M (the network's estimate of Rxx^-1 / Rnn^-1, of Rxx / Rnn, or of a
// Cholesky factor of either) and an N-vector r (the speech inter-frame correlation):
//   cholesky:  (enforce: strict upper triangle := 0)  M := L L^H                                   (:297-302 / :388-393)
//   enforce && !inverse && !cholesky:  Im diag := 0, upper triangle := conj(lower)                  (:303-309 / :394-400)
//   !inverse:  M += (Re tr(M) * dload + eps) I;  u = M^-1 r   (LU with partial pivoting, like LAPACK cgesv behind torch.linalg.solve)
//    inverse:  u = M r
//   Wiener:    w = u                                   MVDR:  w = u * conj(r[N-1]) / (Re(r^H u) + eps)         (:406-409)
//   Y[b,t,f] = sum_n w[n] X[b, t + n - (N-1-lookahead), f]   (zero outside the clip);   bins >= nb pass through.
// HBM-bound: 8 N^2 + 8 N + 16 bytes and ~8 N^3/3 + 8 N^2 flops per bin (N = 5: 256 B, ~550 flop).  A wave (= a workgroup) owns 64 consecutive
// (t, f) items, whose matrices are one contiguous 64 * 8 N^2-byte run: staged through LDS with coalesced float4 loads, then every
// lane works on its own matrix in registers (rows padded to an odd word stride: conflict-free).
#include "dfx_common.h"

#define DFX_MF_THREADS 64   // one wave per workgroup: no cross-wave barrier, ~10 independent workgroups per CU overlap their load / solve / store phases
#define DFX_MF_MAXN 8

struct DfxMfArgs {
    const float2 *spec;  // [B, T, F]
    const float2 *ifc;   // [B, T, nb, N]
    const float2 *mat;   // [B, T, nb, N, N]
    float2 *out;         // [B, T, F]
    int64_t B, T;
    int F, nb, lookahead;
    int mvdr, cholesky, inverse, enforce;
    float eps, dload;
};

static __device__ __forceinline__ float2 mf_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
static __device__ __forceinline__ float2 mf_mulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
static __device__ __forceinline__ float2 mf_div(float2 a, float2 b) {
    const float d = 1.f / (b.x * b.x + b.y * b.y);
    return make_float2((a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d);
}

template <int N>
__global__ void __launch_bounds__(DFX_MF_THREADS) dfx_k_mf_filter(DfxMfArgs A) {
    constexpr int MW = 2 * N * N + 2 * N;      // words per item: matrix then vector
    constexpr int LD = MW | 1;                 // odd stride
    DFX_DYN_SMEM(float, sm);                   // [DFX_MF_THREADS][LD]
    const int64_t items = A.B * A.T * A.nb;
    const int64_t i0 = (int64_t)blockIdx.x * DFX_MF_THREADS;
    const int tid = threadIdx.x;
    const int cnt = (int)((items - i0) < DFX_MF_THREADS ? (items - i0) : DFX_MF_THREADS);
    // ---- stage: the block's matrices and vectors are two contiguous runs in HBM (16-byte aligned: 64 items per block), read as
    // float4.  Full blocks issue ALL their loads before the first LDS store (compile-time trip counts, values held in registers):
    // a load -> store loop would wait out one HBM latency per iteration.
    {
        const float *mg = reinterpret_cast<const float *>(A.mat) + i0 * (2 * N * N);
        const float *vg = reinterpret_cast<const float *>(A.ifc) + i0 * (2 * N);
        if (cnt == DFX_MF_THREADS) {
            constexpr int NQ = DFX_MF_THREADS * 2 * N * N / 4, IT = (NQ + DFX_MF_THREADS - 1) / DFX_MF_THREADS;
            constexpr int VQ = DFX_MF_THREADS * 2 * N / 4, VT = (VQ + DFX_MF_THREADS - 1) / DFX_MF_THREADS;
            const float4 *mg4 = reinterpret_cast<const float4 *>(mg);
            const float4 *vg4 = reinterpret_cast<const float4 *>(vg);
            float4 mv[IT], vv[VT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int q = tid + DFX_MF_THREADS * it;
                mv[it] = q < NQ ? mg4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < VT; ++it) {
                const int q = tid + DFX_MF_THREADS * it;
                vv[it] = q < VQ ? vg4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int q = tid + DFX_MF_THREADS * it;
                if (q < NQ) {
                    const float w[4] = {mv[it].x, mv[it].y, mv[it].z, mv[it].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * q + j;
                        sm[(e / (2 * N * N)) * LD + e % (2 * N * N)] = w[j];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < VT; ++it) {
                const int q = tid + DFX_MF_THREADS * it;
                if (q < VQ) {
                    const float w[4] = {vv[it].x, vv[it].y, vv[it].z, vv[it].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * q + j;
                        sm[(e / (2 * N)) * LD + 2 * N * N + e % (2 * N)] = w[j];
                    }
                }
            }
        } else {  // the last, partial block
            const int nm = cnt * 2 * N * N, nv = cnt * 2 * N;
            for (int e = tid; e < nm; e += DFX_MF_THREADS) sm[(e / (2 * N * N)) * LD + e % (2 * N * N)] = mg[e];
            for (int e = tid; e < nv; e += DFX_MF_THREADS) sm[(e / (2 * N)) * LD + 2 * N * N + e % (2 * N)] = vg[e];
        }
    }
    __syncthreads();
    // ---- pass-through bins (f >= nb) of the frames this block touches are copied by a grid-stride loop over all of them
    {
        const int hb = A.F - A.nb;
        const int64_t total = A.B * A.T * hb;
        for (int64_t e = (int64_t)blockIdx.x * DFX_MF_THREADS + tid; e < total; e += (int64_t)gridDim.x * DFX_MF_THREADS) {
            const int64_t r = e / hb;
            const int f = A.nb + (int)(e - r * hb);
            A.out[r * A.F + f] = A.spec[r * A.F + f];
        }
    }
    if (tid >= cnt) return;
    const int64_t it = i0 + tid;
    const int f = (int)(it % A.nb);
    const int64_t bt = it / A.nb, b = bt / A.T, t = bt - b * A.T;
    float2 M[N][N], r[N], u[N];
    const float *my = sm + tid * LD;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) M[i][j] = make_float2(my[2 * (i * N + j)], my[2 * (i * N + j) + 1]);
        r[i] = make_float2(my[2 * N * N + 2 * i], my[2 * N * N + 2 * i + 1]);
    }
    if (A.cholesky) {
        float2 L[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) L[i][j] = (A.enforce && j > i) ? make_float2(0.f, 0.f) : M[i][j];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const float2 p = mf_mulc(L[i][k], L[j][k]);
                    acc.x += p.x, acc.y += p.y;
                }
                M[i][j] = acc;
            }
    }
    if (A.enforce && !A.inverse && !A.cholesky) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            M[i][i].y = 0.f;
#pragma unroll
            for (int j = i + 1; j < N; ++j) M[i][j] = make_float2(M[j][i].x, -M[j][i].y);
        }
    }
    if (!A.inverse) {
        float tr = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) tr += M[i][i].x;
        const float e = tr * A.dload + A.eps;
#pragma unroll
        for (int i = 0; i < N; ++i) M[i][i].x += e;
        // Gaussian elimination with partial pivoting (pivot = largest |re| + |im| of the column, LAPACK's icamax rule); the row
        // exchange is done by value so that every index stays a compile-time constant (registers, no scratch)
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = r[i];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            int p = c;
            float best = fabsf(M[c][c].x) + fabsf(M[c][c].y);
#pragma unroll
            for (int i = c + 1; i < N; ++i) {
                const float v = fabsf(M[i][c].x) + fabsf(M[i][c].y);
                if (v > best) best = v, p = i;
            }
#pragma unroll
            for (int i = c + 1; i < N; ++i)
                if (p == i) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const float2 tmp = M[c][j];
                        M[c][j] = M[i][j];
                        M[i][j] = tmp;
                    }
                    const float2 tu = u[c];
                    u[c] = u[i];
                    u[i] = tu;
                }
            const float2 piv = M[c][c];
#pragma unroll
            for (int i = c + 1; i < N; ++i) {
                const float2 l = mf_div(M[i][c], piv);
#pragma unroll
                for (int j = c + 1; j < N; ++j) {
                    const float2 q = mf_mul(l, M[c][j]);
                    M[i][j].x -= q.x, M[i][j].y -= q.y;
                }
                const float2 q = mf_mul(l, u[c]);
                u[i].x -= q.x, u[i].y -= q.y;
            }
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            float2 acc = u[i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) {
                const float2 q = mf_mul(M[i][j], u[j]);
                acc.x -= q.x, acc.y -= q.y;
            }
            u[i] = mf_div(acc, M[i][i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float2 q = mf_mul(M[i][j], r[j]);
                acc.x += q.x, acc.y += q.y;
            }
            u[i] = acc;
        }
    }
    if (A.mvdr) {
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) den += r[i].x * u[i].x + r[i].y * u[i].y;   // Re(conj(r) u)
        const float inv = 1.f / (den + A.eps);
        const float2 sc = make_float2(r[N - 1].x, -r[N - 1].y);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float2 q = mf_mul(u[i], sc);
            u[i] = make_float2(q.x * inv, q.y * inv);
        }
    }
    float2 y = make_float2(0.f, 0.f);
    const float2 *xb = A.spec + b * A.T * A.F + f;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int64_t tt = t + n - (N - 1 - A.lookahead);
        if (tt >= 0 && tt < A.T) {
            const float2 q = mf_mul(xb[tt * A.F], u[n]);
            y.x += q.x, y.y += q.y;
        }
    }
    A.out[bt * A.F + f] = y;
}

template <int N>
static int mf_launch(const DfxMfArgs &A, hipStream_t s) {
    const int64_t items = A.B * A.T * A.nb;
    const int64_t nblk = dfx_ceil_div(items, DFX_MF_THREADS);
    if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_mf_filter: grid too large");
    const size_t smem = (size_t)DFX_MF_THREADS * ((2 * N * N + 2 * N) | 1) * sizeof(float);
    if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_mf_filter<N>, smem));
    dfx_launch(dfx_k_mf_filter<N>, dim3((unsigned)nblk), dim3(DFX_MF_THREADS), smem, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_mf_filter(const float *spec, const float *ifc, const float *mat, int op, int frame_size, int lookahead, int cholesky_decomp,
                             int inverse, int enforce_constraints, float eps, float dload, int64_t B, int64_t T, int F, int nb, float *out,
                             void *stream) {
    if (B < 0 || T < 0 || F <= 0 || nb <= 0 || nb > F || frame_size < 1 || lookahead < 0 || lookahead >= frame_size || (op != 0 && op != 1))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_mf_filter: bad arguments");
    if (frame_size > DFX_MF_MAXN) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_mf_filter: frame_size %d > %d", frame_size, DFX_MF_MAXN);
    if (int rc = dfx_require_device()) return rc;
    if (B == 0 || T == 0) return DFX_OK;
    if (!spec || !ifc || !mat || !out || out == spec) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_mf_filter: null buffer or out == spec (not in place)");
    DfxMfArgs A;
    A.spec = reinterpret_cast<const float2 *>(spec);
    A.ifc = reinterpret_cast<const float2 *>(ifc);
    A.mat = reinterpret_cast<const float2 *>(mat);
    A.out = reinterpret_cast<float2 *>(out);
    A.B = B, A.T = T, A.F = F, A.nb = nb, A.lookahead = lookahead;
    A.mvdr = op, A.cholesky = cholesky_decomp != 0, A.inverse = inverse != 0, A.enforce = enforce_constraints != 0;
    A.eps = eps, A.dload = dload;
    hipStream_t s = dfx_stream(stream);
    DfxKScope ks(DFX_K_MF, s);
    switch (frame_size) {
        case 1: return mf_launch<1>(A, s);
        case 2: return mf_launch<2>(A, s);
        case 3: return mf_launch<3>(A, s);
        case 4: return mf_launch<4>(A, s);
        case 5: return mf_launch<5>(A, s);
        case 6: return mf_launch<6>(A, s);
        case 7: return mf_launch<7>(A, s);
        default: return mf_launch<8>(A, s);
    }
}
