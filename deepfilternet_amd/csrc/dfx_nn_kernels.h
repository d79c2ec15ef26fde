// HIP kernels for the DNN half of DeepFilterNet3 (deepfilternet3.py:100-331) on MI355X / gfx950.
//
// Layout: every activation is channels-last, [rows = B*T][F][C] float32 — the layout the reference's own
// permute/flatten calls produce (deepfilternet3.py:178-181,248-249,328), so "emb = e3.permute(0,2,3,1).flatten(2)" and
// "df_convp(c0).permute(0,2,3,1)" are free here.
// Arithmetic: float32 everywhere (parity bar: 1e-4 RMS on the waveform against the reference's fp32 CPU path).  Dense
// contractions (pointwise 1x1 convs, grouped linears, GRU input projections, DF pathway conv) run on the matrix cores
// with v_mfma_f32_16x16x4_f32, which is bit-identical to an fmaf chain; everything else is VALU + LDS.
// BatchNorm (eval) is folded into the preceding convolution on the host (dfx_model.hip).
#pragma once

#include "dfx_common.h"

#include <type_traits>

#define DFX_ACT_NONE 0
#define DFX_ACT_RELU 1
#define DFX_ACT_TANH 2
#define DFX_ACT_SIGMOID 3

static __device__ __forceinline__ float dfx_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
static __device__ __forceinline__ float dfx_act(float v, int act) {
    if (act == DFX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DFX_ACT_TANH) return tanhf(v);   // (a 5-instruction exp / rcp form was measured: the grouped GEMMs are HBM-bound, no change)
    if (act == DFX_ACT_SIGMOID) return dfx_sigmoid(v);
    return v;
}

// compile-time loop (the fully unrolled bodies index register arrays with constants only)
template <int I, int N, typename F>
static __device__ __forceinline__ void dfx_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dfx_static_for<I + 1, N>(f);
    }
}

// Row map for time-chunked launches: logical row m of a chunk of Tk frames starting at frame t0 (all B clips) is physical
// row (m / Tk) * T + t0 + m % Tk of the [B*T, ...] activation arrays.  Tk == 0: identity.
struct DfxRowMap {
    int64_t T, Tk, t0;
};
static __device__ __forceinline__ int64_t dfx_row(const DfxRowMap &rm, int64_t m) {
    if (rm.Tk == 0) return m;
    // logical rows of a chunk launch are B * Tk <= B * T < 2^31 (forward_impl refuses more): a 32-bit division (~25 instructions; the 64-bit
    // one is ~130, and the staged kernels map a row per 16-byte piece they move: a dozen per item in the time-chunked pipeline)
    const uint32_t b = (uint32_t)m / (uint32_t)rm.Tk;
    return (int64_t)b * rm.T + rm.t0 + ((uint32_t)m - b * (uint32_t)rm.Tk);
}

// a / d for 0 <= a < 2^22 and d > 0 with inv = 1.0f / d: the float quotient is within one of the integer one (8 instructions; the
// integer division by a run-time value is ~20)
static __device__ __forceinline__ int dfx_div_small(int a, int d, float inv) {
    int q = (int)((float)a * inv);
    const int r = a - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
}
// The row map for the rows base + fr of ONE item (fr small): the division once per item (dfx_row_base), an add and a compare per row.
struct DfxRowBase {
    uint32_t b, r;   // base = b * Tk + r
};
static __device__ __forceinline__ DfxRowBase dfx_row_base(const DfxRowMap &rm, int64_t base) {
    DfxRowBase rb{0u, 0u};
    if (rm.Tk != 0) {
        rb.b = (uint32_t)base / (uint32_t)rm.Tk;
        rb.r = (uint32_t)base - rb.b * (uint32_t)rm.Tk;
    }
    return rb;
}
static __device__ __forceinline__ int64_t dfx_row_at(const DfxRowMap &rm, int64_t base, const DfxRowBase &rb, int fr) {
    if (rm.Tk == 0) return base + fr;
    uint32_t b = rb.b, r = rb.r + (uint32_t)fr;
    while (r >= (uint32_t)rm.Tk) r -= (uint32_t)rm.Tk, ++b;
    return (int64_t)b * rm.T + rm.t0 + r;
}

// ---------------------------------------------------------------------------------------------------------------------
// enc.erb_conv0: Conv2d(1 -> C, 3x3, causal in time, pad 1 in freq) + BN + ReLU   (deepfilternet3.py:106-108)
//   feat [B,T,E] -> out [B*T, E, C].  Lookahead L: tap kt of output frame t reads input frame t+L-2+kt, and is zero when
//   t-2+kt < 0 (causal pad applied AFTER the lookahead shift, deepfilternet3.py:357-361,409) or when the frame is >= T.
// One thread per (position, 4 channels): 16-byte coalesced stores (the kernel is bound by the 4*E*C bytes it writes per
// frame); a thread keeps its 9x4 weights in registers (its channel quad is fixed across the grid-stride loop); the 9
// inputs of a position are L1-broadcast loads shared by the C/4 threads of that position.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dfx_k_conv_in_erb(const float *feat, const float *w /*[3][3][C]*/,
                                                         const float *bias /*[C]*/, float *out, int64_t B, int64_t T, int E,
                                                         int C, int L) {
    const int C4 = C >> 2;
    const int64_t total = B * T * E * C4;
    const int64_t gid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = (int)(gid0 % C4);  // invariant: the grid stride is a multiple of C4 (C4 divides 256)
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = reinterpret_cast<const float4 *>(w)[k * C4 + c4];
    const float4 bv = reinterpret_cast<const float4 *>(bias)[c4];
    for (int64_t i = gid0; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C4;
        const int f = (int)(p % E);
        const int64_t r = p / E;
        const int64_t t = r % T, b = r / T;
        float4 acc = bv;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int64_t tau = t - 2 + kt, tin = tau + L;
            if (tau < 0 || tin >= T) continue;
            const float *row = feat + (b * T + tin) * E;
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int fin = f - 1 + kf;
                if (fin < 0 || fin >= E) continue;
                const float x = row[fin];
                const float4 ww = wv[kt * 3 + kf];
                acc.x += ww.x * x;
                acc.y += ww.y * x;
                acc.z += ww.z * x;
                acc.w += ww.w * x;
            }
        }
        reinterpret_cast<float4 *>(out)[i] = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused [depthwise 1x3 prologue] -> pointwise 1x1 (C x C, MFMA) -> +bias (folded BN) -> ReLU, entirely in registers.
// Covers Conv2dNormAct / ConvTranspose2dNormAct with separable=True (modules.py:18-126):
//   MODE_DW3   depthwise 1x3 conv over freq, stride s in {1,2}, pad 1                     (enc.erb_conv1-3, df_conv1, convt3)
//   MODE_DWT3  depthwise 1x3 transposed conv, stride 2, padding 1, output_padding 1        (erb_dec.convt2, convt1)
// Optional skip input for the decoder:  xin = relu(sk_a[c]*skip + sk_b[c]) + x   (conv{3,2,1}p pathway + Add,
// deepfilternet3.py:250-252; convNp is a per-channel scalar + BN + ReLU).
//
// A wave owns 16 output positions at a time; there is no LDS staging and no workgroup barrier in the loop.  The GEMM is
// computed transposed, out^T[n][pos] = sum_c W[n][c] * u[c][pos], on v_mfma_f32_16x16x4_f32 with
//   A = W   (lane l: row n = l&15, the persistent weight fragment, C*C/64 registers)
//   B = u   (lane l: column pos = l&15, k = l>>4)
// and the contraction index is enumerated so that k-step ks, k = q (q = l>>4) is channel (C/4)*q + ks: lane (pos, q) then
// needs exactly the C/4 CONSECUTIVE channels [(C/4)q, (C/4)(q+1)) of its position, which it loads straight from HBM as
// float4s, runs the 3-tap depthwise conv on in registers and feeds to the matrix core.  D comes back as 4 consecutive
// output channels per lane -> float4 stores.  Per 16 positions: 3*C/16 float4 loads per lane, (C/4)*(C/16) MFMAs.
// (Measured: giving the four q-lanes of a position 64 contiguous bytes per load instead is 35 % slower here.)
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_PW_MODE_DW3 0
#define DFX_PW_MODE_DWT3 1
#define DFX_PW_THREADS 256

struct DfxPwArgs {
    const float *x;      // [R, Fin, C]
    const float *skip;   // [R, Fin, C] or null
    const float *sk_a, *sk_b;  // [C]
    const float *dw;     // [3][C]
    const float *wt;     // [C][C]  wt[k][n] = W_pw[n][k] * bn_scale[n]
    const float *bias;   // [C]
    float *out;          // [R, Fout, C]
    int64_t R;          // logical rows (frames) of this launch
    int Fin, Fout, stride;
    DfxRowMap rm;       // logical row -> physical row of x, skip and out (time-chunked launches)
    const dfx_h8 *wt_h3 = nullptr;   // pointwise weights as pre-scaled f16 hi/lo fragments (dfx_k_pwconv_f<..., true>)
    float unscale = 1.f;
    unsigned int *err = nullptr;     // model error words ([1]: fp16-split range guard)
};

template <int C, int MODE, bool SKIP>
__global__ void __launch_bounds__(DFX_PW_THREADS) dfx_k_pwconv(DfxPwArgs A) {
    constexpr int CPL = C / 4;   // channels per lane == MFMA k-steps
    constexpr int NT = C / 16;   // 16-wide output channel tiles
    constexpr int V4 = CPL / 4;  // float4s per lane and tap
    __shared__ float4 dws[3 * C / 4];
    __shared__ float4 sks[2 * C / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    for (int i = tid; i < 3 * C / 4; i += DFX_PW_THREADS) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    if (SKIP)
        for (int i = tid; i < C / 4; i += DFX_PW_THREADS) {
            sks[i] = reinterpret_cast<const float4 *>(A.sk_a)[i];
            sks[C / 4 + i] = reinterpret_cast<const float4 *>(A.sk_b)[i];
        }
    float areg[NT][CPL];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks) areg[nt][ks] = A.wt[(CPL * q + ks) * C + 16 * nt + jl];
    float4 biasr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) biasr[nt] = reinterpret_cast<const float4 *>(A.bias)[4 * nt + q];
    __syncthreads();
    const int64_t total = A.R * A.Fout;
    const int64_t ntiles = (total + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t pos = tile * 16 + jl;
        const bool valid = pos < total;
        const int64_t rl = pos / A.Fout;
        const int fo = (int)(pos - rl * A.Fout);
        const int64_t r = dfx_row(A.rm, valid ? rl : 0);
        float u[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = 0.f;
        if constexpr (!SKIP) {
            // no pathway operand: the three taps' loads are all issued before the first one is used (one memory round trip per tile
            // instead of three; 2 x V4 more float4 live)
            float4 xt[3][V4];
            bool okj[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int fi;
                bool ok;
                if (MODE == DFX_PW_MODE_DW3) {
                    fi = fo * A.stride + j - 1;
                    ok = fi >= 0 && fi < A.Fin;
                } else {
                    const int num = fo + 1 - j;
                    fi = num >> 1;
                    ok = num >= 0 && (num & 1) == 0 && fi < A.Fin;
                }
                okj[j] = valid && ok;
                const float4 *xp = reinterpret_cast<const float4 *>(A.x + (r * A.Fin + (okj[j] ? fi : 0)) * C + CPL * q);
#pragma unroll
                for (int v = 0; v < V4; ++v) xt[j][v] = okj[j] ? xp[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (okj[j]) {
#pragma unroll
                    for (int v = 0; v < V4; ++v) {
                        const float4 xv = xt[j][v];
                        const float4 w = dws[j * (C / 4) + V4 * q + v];
                        u[4 * v + 0] += w.x * xv.x;
                        u[4 * v + 1] += w.y * xv.y;
                        u[4 * v + 2] += w.z * xv.z;
                        u[4 * v + 3] += w.w * xv.w;
                    }
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int fi;
            bool ok;
            if (MODE == DFX_PW_MODE_DW3) {
                fi = fo * A.stride + j - 1;
                ok = fi >= 0 && fi < A.Fin;
            } else {  // transposed: fo = 2*fi - 1 + j
                const int num = fo + 1 - j;
                fi = num >> 1;
                ok = num >= 0 && (num & 1) == 0 && fi < A.Fin;
            }
            if (valid && ok) {
                const int64_t off = (r * A.Fin + fi) * C + CPL * q;
                const float4 *xp = reinterpret_cast<const float4 *>(A.x + off);
                const float4 *sp = reinterpret_cast<const float4 *>(A.skip + (SKIP ? off : 0));
#pragma unroll
                for (int v = 0; v < V4; ++v) {
                    float4 xv = xp[v];
                    if (SKIP) {
                        const float4 sv = sp[v], a = sks[V4 * q + v], bb = sks[C / 4 + V4 * q + v];
                        xv.x += fmaxf(a.x * sv.x + bb.x, 0.f);
                        xv.y += fmaxf(a.y * sv.y + bb.y, 0.f);
                        xv.z += fmaxf(a.z * sv.z + bb.z, 0.f);
                        xv.w += fmaxf(a.w * sv.w + bb.w, 0.f);
                    }
                    const float4 w = dws[j * (C / 4) + V4 * q + v];
                    u[4 * v + 0] += w.x * xv.x;
                    u[4 * v + 1] += w.y * xv.y;
                    u[4 * v + 2] += w.z * xv.z;
                    u[4 * v + 3] += w.w * xv.w;
                }
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[nt][ks], u[ks], acc[nt], 0, 0, 0);
        if (valid) {
            float4 *op = reinterpret_cast<float4 *>(A.out + (r * A.Fout + fo) * C + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                op[4 * nt] = make_float4(fmaxf(acc[nt][0] + biasr[nt].x, 0.f), fmaxf(acc[nt][1] + biasr[nt].y, 0.f),
                                         fmaxf(acc[nt][2] + biasr[nt].z, 0.f), fmaxf(acc[nt][3] + biasr[nt].w, 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// enc.df_conv0: pad t(2,0) -> Conv2d(2 -> C, 3x3, groups=2) -> Conv2d(C -> C, 1x1) -> BN -> ReLU (deepfilternet3.py:115-117).
// There is no nonlinearity between the grouped 3x3 conv and the pointwise conv (modules.py:49-71), so the host folds them
// (and the BN scale) into ONE dense 3x3 conv 2 -> C:  W_eff[n][(kt,kf,ch)] = sum_{c in group ch} W_pw[n][c] * W_dw[c][kt][kf],
// i.e. a GEMM with K = 18 (padded to 20) instead of K = C.  out^T[n][pos] = sum_k W_eff[n][k] * im2col[k][pos] on the
// matrix core (same operand roles as dfx_k_pwconv); the B operand is gathered straight from feat_spec [B,T,Fin,2] with
// the lookahead shift L and the causal/border zeros.  The kernel is bound by its 4*Fin*C-byte-per-frame store.
// ---------------------------------------------------------------------------------------------------------------------
struct DfxCinArgs {
    const float *feat;  // [B, T, Fin, 2]
    const float *weff;  // [20][C]  weff[k][n], k = (kt*3 + kf)*2 + ch, rows 18..19 zero
    const float *bias;  // [C]
    float *out;         // [B*T, Fin, C]
    int64_t B, T;
    int Fin, L;
    // only frames [t_begin, T) of every clip are produced; frame t of clip b is stored at row b*out_T + t - t_begin + out_toff
    // (0, T, 0 for whole clips; the gated streaming runtime writes the newest frame into its per-stream c0 window)
    int64_t t_begin, out_T, out_toff;
};

template <int C>
__global__ void __launch_bounds__(DFX_PW_THREADS) dfx_k_conv_in_df(DfxCinArgs A) {
    constexpr int NT = C / 16, KS = 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    float areg[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) areg[nt][ks] = A.weff[(4 * ks + q) * C + 16 * nt + jl];
    float4 biasr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) biasr[nt] = reinterpret_cast<const float4 *>(A.bias)[4 * nt + q];
    const int64_t Tn = A.T - A.t_begin;
    const int64_t total = A.B * Tn * A.Fin;
    const int64_t ntiles = (total + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t pos = tile * 16 + jl;
        const bool valid = pos < total;
        const int64_t r = pos / A.Fin;
        const int fo = (int)(pos - r * A.Fin);
        const int64_t b = r / Tn, t = r - b * Tn + A.t_begin;
        float bv[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 4 * ks + q, tap = k >> 1, ch = k & 1, kt = tap / 3, kf = tap - 3 * kt;
            const int64_t tau = t - 2 + kt, tin = tau + A.L;
            const int fin = fo - 1 + kf;
            float v = 0.f;
            if (valid && k < 18 && tau >= 0 && tin < A.T && fin >= 0 && fin < A.Fin)
                v = A.feat[((b * A.T + tin) * A.Fin + fin) * 2 + ch];
            bv[ks] = v;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[nt][ks], bv[ks], acc[nt], 0, 0, 0);
        if (valid) {
            float4 *op = reinterpret_cast<float4 *>(A.out + ((b * A.out_T + t - A.t_begin + A.out_toff) * A.Fin + fo) * C + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                op[4 * nt] = make_float4(fmaxf(acc[nt][0] + biasr[nt].x, 0.f), fmaxf(acc[nt][1] + biasr[nt].y, 0.f),
                                         fmaxf(acc[nt][2] + biasr[nt].z, 0.f), fmaxf(acc[nt][3] + biasr[nt].w, 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// enc.df_conv0 computed on the fly (no c0 tensor in HBM).  c0 [B,T,Fd,C] is the largest activation of the network (4*Fd*C bytes
// per frame, written once and read by df_conv1 and df_convp), yet every element is only a K = 18 dot product of feat_spec, so its
// two consumers recompute the tiles they need on the matrix core instead:
//   dfx_c0_patch  gathers the im2col column of one (frame, bin) position as the MFMA B operand (k = 4*ks + q), exactly like
//                 dfx_k_conv_in_df;
//   dfx_c0_tile   runs the C/16 x 5 MFMAs and applies bias + ReLU.  The D fragment gives lane (pos, q) the channels
//                 {16*i + 4*q + r}: element [4*i + r] of the result.  That is directly a valid B operand for the next GEMM when the
//                 consumer enumerates its contraction index as k-step ks <-> channel 16*(ks>>2) + 4*q + (ks&3) (the weights are
//                 read in that order), so the chained GEMMs never leave the registers.
// ---------------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void dfx_c0_patch(const float *__restrict__ feat, int64_t b, int64_t t, int f, bool valid,
                                                    int64_t T, int Fin, int L, int q, float (&bv)[5]) {
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int k = 4 * ks + q, tap = k >> 1, ch = k & 1, kt = tap / 3, kf = tap - 3 * kt;
        const int64_t tau = t - 2 + kt, tin = tau + L;
        const int fin = f - 1 + kf;
        float v = 0.f;
        if (valid && k < 18 && tau >= 0 && tin < T && fin >= 0 && fin < Fin) v = feat[((b * T + tin) * Fin + fin) * 2 + ch];
        bv[ks] = v;
    }
}

template <int C>
static __device__ __forceinline__ void dfx_c0_tile(const float (&areg0)[C / 16][5], const float4 (&bias0)[C / 16],
                                                   const float (&bv)[5], bool keep, float (&dst)[C / 4]) {
    constexpr int NT = C / 16;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg0[nt][ks], bv[ks], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dst[4 * nt + 0] = keep ? fmaxf(acc[nt][0] + bias0[nt].x, 0.f) : 0.f;
        dst[4 * nt + 1] = keep ? fmaxf(acc[nt][1] + bias0[nt].y, 0.f) : 0.f;
        dst[4 * nt + 2] = keep ? fmaxf(acc[nt][2] + bias0[nt].z, 0.f) : 0.f;
        dst[4 * nt + 3] = keep ? fmaxf(acc[nt][3] + bias0[nt].w, 0.f) : 0.f;
    }
}

// enc.df_conv0 -> enc.df_conv1 fused (deepfilternet3.py:115-119,176-177): c1 = relu(bn(pw(dw_{1x3, fstride}(c0)))) with the three c0
// tiles a 16-position output tile needs (bins stride*fo - 1 + j) recomputed from feat_spec.  Per 16 outputs: 3*5*C/16 MFMAs for c0 +
// (C/4)*(C/16) for the pointwise conv; HBM traffic is the c1 store (4*Fout*C bytes per frame) and the (cached) feat_spec reads.
struct DfxC01Args {
    const float *feat;   // [B, T, Fin, 2]
    const float *weff0;  // [20][C]  folded df_conv0 (see dfx_k_conv_in_df)
    const float *bias0;  // [C]
    const float *dw;     // [3][C]
    const float *wt;     // [C][C]  wt[k][n] = W_pw[n][k] * bn_scale[n]
    const float *bias;   // [C]
    float *out;          // [B*T, Fout, C]
    int64_t B, T;
    int Fin, Fout, stride, L;
    int64_t t_begin;     // only frames [t_begin, t_end) of every clip are produced (t_end <= T; frames up to T may be read)
    int64_t t_end;
};

template <int C>
__global__ void __launch_bounds__(DFX_PW_THREADS) dfx_k_df_conv01(DfxC01Args A) {
    constexpr int CPL = C / 4, NT = C / 16;
    __shared__ float4 dws[3 * C / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    for (int i = tid; i < 3 * C / 4; i += DFX_PW_THREADS) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    float areg0[NT][5];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) areg0[nt][ks] = A.weff0[(4 * ks + q) * C + 16 * nt + jl];
    float areg[NT][CPL];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks) areg[nt][ks] = A.wt[(16 * (ks >> 2) + 4 * q + (ks & 3)) * C + 16 * nt + jl];
    float4 bias0[NT], biasr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias0[nt] = reinterpret_cast<const float4 *>(A.bias0)[4 * nt + q];
        biasr[nt] = reinterpret_cast<const float4 *>(A.bias)[4 * nt + q];
    }
    __syncthreads();
    const int64_t Tn = A.t_end - A.t_begin;
    const int64_t total = A.B * Tn * A.Fout;
    const int64_t ntiles = (total + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t lpos = tile * 16 + jl;
        const bool valid = lpos < total;
        const int64_t rl = lpos / A.Fout;
        const int fo = (int)(lpos - rl * A.Fout);
        const int64_t b = rl / Tn, t = A.t_begin + (rl - b * Tn);
        const int64_t pos = (b * A.T + t) * A.Fout + fo;  // physical output position
        float u[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int fi = fo * A.stride + j - 1;
            const bool ok = valid && fi >= 0 && fi < A.Fin;
            float bv[5], c0v[CPL];
            dfx_c0_patch(A.feat, b, t, fi, ok, A.T, A.Fin, A.L, q, bv);
            dfx_c0_tile<C>(areg0, bias0, bv, ok, c0v);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 w = dws[j * (C / 4) + 4 * nt + q];
                u[4 * nt + 0] += w.x * c0v[4 * nt + 0];
                u[4 * nt + 1] += w.y * c0v[4 * nt + 1];
                u[4 * nt + 2] += w.z * c0v[4 * nt + 2];
                u[4 * nt + 3] += w.w * c0v[4 * nt + 3];
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[nt][ks], u[ks], acc[nt], 0, 0, 0);
        if (valid) {
            float4 *op = reinterpret_cast<float4 *>(A.out + pos * C + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                op[4 * nt] = make_float4(fmaxf(acc[nt][0] + biasr[nt].x, 0.f), fmaxf(acc[nt][1] + biasr[nt].y, 0.f),
                                         fmaxf(acc[nt][2] + biasr[nt].z, 0.f), fmaxf(acc[nt][3] + biasr[nt].w, 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16-split ("fp16x3", see dfx_env.h) forms of the two fused DF-encoder kernels; these are the default, the fp32-MFMA forms
// above serve DFX_EXACT_FP32=1 and conv_ch < 32.  With fp32 MFMAs the fused kernels are matrix-pipe bound (the recomputation of c0
// costs 5*C/16 16x16x4 MFMAs per tile); on v_mfma_f32_16x16x32_f16 the whole K = 18 contraction is ONE k-chunk (3 MFMAs per 16
// channels) and the C-deep contractions take C/32 chunks.  Operands:
//   c0      B = patch, lane (pos, q) holds k = 8q..8q+7 = taps 4q..4q+3 x (re, im): four float2 loads from feat_spec
//           A = folded df_conv0 weights, fragment [nt][hi,lo] (k >= 18 zero), pre-scaled by a power of two
//   chained B = the 16 channels {16*i + 4*q + r} a lane holds after dfx_c0_tile (element e = 4*i + r): chunk kc takes elements
//           8*kc .. 8*kc+7, so k-index (kc, q, i) <-> channel 16*((8kc+i)>>2) + 4q + ((8kc+i)&3); the host packs the A fragments
//           of df_conv1's pointwise conv and of df_convp in that order (pack_h3 in dfx_model.hip).
// The feat_spec loads of the NEXT tile / frame are issued before the current one is computed (the only HBM/L2 latency in the loop).
// ---------------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void dfx_c0_patch_load(const float *__restrict__ feat, int64_t b, int64_t t, int f, bool valid,
                                                         int64_t T, int Fin, int L, int q, float2 (&raw)[4], int64_t Ts = 0) {
    if (Ts <= 0) Ts = T;   // frames per clip of the feature array (a window inside a longer buffer: Ts > T)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tap = 4 * q + i, kt = tap / 3, kf = tap - 3 * kt;
        const int64_t tau = t - 2 + kt, tin = tau + L;
        const int fin = f - 1 + kf;
        float2 v = make_float2(0.f, 0.f);
        if (valid && tap < 9 && tau >= 0 && tin < T && fin >= 0 && fin < Fin)
            v = *reinterpret_cast<const float2 *>(feat + ((b * Ts + tin) * Fin + fin) * 2);
        raw[i] = v;
    }
}

template <int C>
static __device__ __forceinline__ void dfx_c0_tile_h3(const dfx_h8 (&w0h)[C / 16], const dfx_h8 (&w0l)[C / 16],
                                                      const float4 (&bias0)[C / 16], float unscale0, const float2 (&raw)[4],
                                                      bool keep, float (&dst)[C / 4], float &amax) {
    constexpr int NT = C / 16;
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[2 * i] = raw[i].x, x[2 * i + 1] = raw[i].y;
    dfx_h8 ph, pl;
    dfx_split8_g(x, ph, pl, amax);
    f32x4 acc[NT];  // the NT chains are independent: term-major order keeps dependent MFMAs NT issues apart
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0l[nt], ph, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], pl, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], ph, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dst[4 * nt + 0] = keep ? fmaxf(acc[nt][0] * unscale0 + bias0[nt].x, 0.f) : 0.f;
        dst[4 * nt + 1] = keep ? fmaxf(acc[nt][1] * unscale0 + bias0[nt].y, 0.f) : 0.f;
        dst[4 * nt + 2] = keep ? fmaxf(acc[nt][2] * unscale0 + bias0[nt].z, 0.f) : 0.f;
        dst[4 * nt + 3] = keep ? fmaxf(acc[nt][3] * unscale0 + bias0[nt].w, 0.f) : 0.f;
    }
}

struct DfxC01hArgs {
    const float *feat;   // [B, T, Fin, 2]
    const dfx_h8 *w0f;   // [C/16][hi,lo][64]        folded df_conv0
    const float *bias0;  // [C]
    const float *dw;     // [3][C]
    const dfx_h8 *wpf;   // [C/16][C/32][hi,lo][64]  df_conv1 pointwise (BN-scaled)
    const float *bias;   // [C]
    float *out;          // [B*T, Fout, C]
    int64_t B, T;
    int Fin, Fout, stride, L;
    float unscale0, unscale;
    int64_t t_begin;     // only frames [t_begin, t_end) of every clip are produced (t_end <= T; frames up to T may be read)
    int64_t t_end;
    unsigned int *err;   // model error words: bit 0 of err[1] = a value >= DFX_H3_LIMIT reached an f16 split (results invalid)
    int64_t feat_T = 0;  // > 0: frames per clip of feat (the T frames are a window inside a longer buffer: the streaming runtime's linear form)
};

// Register budget: two waves per SIMD (<= 256 registers) so that one wave's LDS / global latencies hide behind the other's matrix
// ops — with everything in registers the kernel needs ~430 and runs 1.6x slower.  Only the df_conv0 fragments (used 3x per tile)
// stay in registers; the pointwise fragments (16 KB) and the biases are read from LDS where they are used, and the feat_spec
// patch of the next tile is requested as soon as the current one has been split.
template <int C>
#ifndef DFX_C01_MINB
#define DFX_C01_MINB 3   /* three waves per SIMD (168 registers): 2.2 -> 1.6 ms at config 2; the range guard pushed the two-wave build to 178 */
#endif
#ifndef DFX_C01_PIN
#define DFX_C01_PIN 0         /* dev: bias / tap reads pinned ahead of the `keep` branches (the compiler sinks each read into the branch of its
                                 use: 36 dependent LDS round trips per tile) — costs registers the three-wave build does not have: measured below */
#endif
#ifndef DFX_C01_EPI_BATCH
#define DFX_C01_EPI_BATCH 2   /* channel groups whose bias / tap are read together (DFX_C01_PIN) */
#endif
__global__ void __launch_bounds__(DFX_PW_THREADS, DFX_C01_MINB) dfx_k_df_conv01_h3(DfxC01hArgs A) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C >= 32 ? C / 32 : 1, C4 = C / 4;
    static_assert(C % 32 == 0, "one k-chunk is 32 channels");
    __shared__ float4 dws[3 * C4];
    __shared__ float4 b0s[C4], b1s[C4];
    __shared__ dfx_h8 wps[NT * KC * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    for (int i = tid; i < 3 * C4; i += DFX_PW_THREADS) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    for (int i = tid; i < C4; i += DFX_PW_THREADS) {
        b0s[i] = reinterpret_cast<const float4 *>(A.bias0)[i];
        b1s[i] = reinterpret_cast<const float4 *>(A.bias)[i];
    }
    for (int i = tid; i < NT * KC * 2 * 64; i += DFX_PW_THREADS) wps[i] = A.wpf[i];
    dfx_h8 w0h[NT], w0l[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        w0h[nt] = A.w0f[(nt * 2 + 0) * 64 + lane];
        w0l[nt] = A.w0f[(nt * 2 + 1) * 64 + lane];
    }
    __syncthreads();
    // Work decomposition: tile = (frame, one of its ceil(Fout/16) groups of 16 output bins); a wave takes tiles wave index + i * waves of
    // the grid.  Clip, frame and group are WAVE-UNIFORM (scalar registers: the (frame, group) pair advances incrementally, one 32-bit
    // division per tile finds the clip) and a lane only adds its bin: the flat (clip, frame, bin) decomposition of every lane and tap
    // in 64-bit arithmetic cost 780 integer VALU + 880 scalar instructions per tile against 540 floating-point / matrix ones (now 560 +
    // 670; the kernel's time did not move: it is bound by the floating-point VALU work of the splits and epilogues, see profiles/).
    const unsigned Tn = (unsigned)(A.t_end - A.t_begin);
    const unsigned NF = (unsigned)A.B * Tn;                 // frames to produce (B * T < 2^31: checked by the host)
    const int TPF = (A.Fout + 15) >> 4;                     // tiles per frame
    const unsigned nwaves = gridDim.x * 4;
    const int T32 = (int)A.T, Fin = A.Fin, Lk = A.L;
    // the four taps (of the 3x3 window over (t, f), padded to 16) this lane feeds into the k index of the matrix op: loop invariant
    int tdt[4], tdf[4];
    bool tok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tap = 4 * q + i, kt = tap / 3;
        tok[i] = tap < 9;
        tdt[i] = kt - 2 + Lk;            // input frame = t + tdt (tau = t - 2 + kt must be >= 0, tau + L < T)
        tdf[i] = tap - 3 * kt - 1;
    }
    float2 raw[3][4];
    bool okj[3];
    // next position (uniform): frame nfr = (clip nb, frame nt_), tile nft; patch j of that tile is requested with issue(j)
    const unsigned wid = (unsigned)blockIdx.x * 4 + (unsigned)dfx_wave_uniform(wave);
    const unsigned step_fr = nwaves / (unsigned)TPF;
    const int step_ft = (int)(nwaves - step_fr * (unsigned)TPF);
    unsigned nfr = wid / (unsigned)TPF;
    int nft = (int)(wid - nfr * (unsigned)TPF), nb = 0, nt_ = 0;
    bool nlive = nfr < NF;
    auto settle = [&]() {   // (nb, nt_) of frame nfr
        if (nlive) {
            nb = (int)(nfr / Tn);
            nt_ = (int)A.t_begin + (int)(nfr - (unsigned)nb * Tn);
        }
    };
    settle();
    auto issue = [&](int j) {
        const int fo = nft * 16 + jl;
        const int fi = fo * A.stride + j - 1;
        okj[j] = nlive && fo < A.Fout && fi >= 0 && fi < Fin;
        const float2 *clip = reinterpret_cast<const float2 *>(A.feat) + (int64_t)nb * (A.feat_T > 0 ? A.feat_T : (int64_t)T32) * Fin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tin = nt_ + tdt[i], fin = fi + tdf[i];
            float2 v = make_float2(0.f, 0.f);
            if (okj[j] && tok[i] && tin - Lk >= 0 && tin < T32 && fin >= 0 && fin < Fin) v = clip[tin * Fin + fin];
            raw[j][i] = v;
        }
    };
    float amax = 0.f;   // largest magnitude that went through an f16 split (range guard)
#pragma unroll
    for (int j = 0; j < 3; ++j) issue(j);
    while (nlive) {
        // the tile whose patches are in raw[]
        const int fo_c = nft * 16 + jl;
        const bool valid = fo_c < A.Fout;
        const int64_t pos = ((int64_t)nb * T32 + nt_) * A.Fout + fo_c;    // physical output position
        // advance to the next tile (its patches are requested below, one by one, as raw[j] becomes free)
        nft += step_ft;
        nfr += step_fr;
        if (nft >= TPF) nft -= TPF, ++nfr;
        nlive = nfr < NF;
        settle();
        float u[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float x[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[2 * i] = raw[j][i].x, x[2 * i + 1] = raw[j][i].y;
            const bool keep = okj[j];
            dfx_h8 ph, pl;
            dfx_split8_g(x, ph, pl, amax);
            issue(j);  // raw[j] is free again: fetch the same patch of the next tile
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0l[nt], ph, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], pl, acc[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], ph, acc[nt]);
#if DFX_C01_PIN
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += DFX_C01_EPI_BATCH) {
                float4 bz[DFX_C01_EPI_BATCH], w[DFX_C01_EPI_BATCH];
#pragma unroll
                for (int d = 0; d < DFX_C01_EPI_BATCH; ++d) {
                    bz[d] = b0s[4 * (n0 + d) + q], w[d] = dws[j * C4 + 4 * (n0 + d) + q];
                    // pinned here: the compiler otherwise sinks each read into the `keep` branch of its use (read -> wait -> four operations,
                    // 36 dependent LDS round trips per tile)
                    DFX_OPAQUE(bz[d].x);
                    DFX_OPAQUE(bz[d].y);
                    DFX_OPAQUE(bz[d].z);
                    DFX_OPAQUE(bz[d].w);
                    DFX_OPAQUE(w[d].x);
                    DFX_OPAQUE(w[d].y);
                    DFX_OPAQUE(w[d].z);
                    DFX_OPAQUE(w[d].w);
                }
#pragma unroll
                for (int d = 0; d < DFX_C01_EPI_BATCH; ++d) {
                    const int nt = n0 + d;
                    u[4 * nt + 0] += w[d].x * (keep ? fmaxf(acc[nt][0] * A.unscale0 + bz[d].x, 0.f) : 0.f);
                    u[4 * nt + 1] += w[d].y * (keep ? fmaxf(acc[nt][1] * A.unscale0 + bz[d].y, 0.f) : 0.f);
                    u[4 * nt + 2] += w[d].z * (keep ? fmaxf(acc[nt][2] * A.unscale0 + bz[d].z, 0.f) : 0.f);
                    u[4 * nt + 3] += w[d].w * (keep ? fmaxf(acc[nt][3] * A.unscale0 + bz[d].w, 0.f) : 0.f);
                }
            }
#else
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bz = b0s[4 * nt + q], w = dws[j * C4 + 4 * nt + q];
                u[4 * nt + 0] += w.x * (keep ? fmaxf(acc[nt][0] * A.unscale0 + bz.x, 0.f) : 0.f);
                u[4 * nt + 1] += w.y * (keep ? fmaxf(acc[nt][1] * A.unscale0 + bz.y, 0.f) : 0.f);
                u[4 * nt + 2] += w.z * (keep ? fmaxf(acc[nt][2] * A.unscale0 + bz.z, 0.f) : 0.f);
                u[4 * nt + 3] += w.w * (keep ? fmaxf(acc[nt][3] * A.unscale0 + bz.w, 0.f) : 0.f);
            }
#endif
        }
        dfx_h8 uh[KC], ul[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) dfx_split8_g(u + 8 * kc, uh[kc], ul[kc], amax);
        float4 *op = reinterpret_cast<float4 *>(A.out + pos * C + 4 * q);
        int zoff = 0;
        DFX_OPAQUE(zoff);  // the fragment reads are loop invariant: keep the compiler from hoisting them into 64 registers
        const dfx_h8 *wpl_ = wps + lane + zoff;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 aa = f32x4{0.f, 0.f, 0.f, 0.f}, ab = aa, ac = aa;  // independent chains per product term
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const dfx_h8 wh = wpl_[((nt * KC + kc) * 2 + 0) * 64], wl = wpl_[((nt * KC + kc) * 2 + 1) * 64];
                aa = dfx_mfma_16x16x32_f16(wl, uh[kc], aa);
                ab = dfx_mfma_16x16x32_f16(wh, ul[kc], ab);
                ac = dfx_mfma_16x16x32_f16(wh, uh[kc], ac);
            }
            if (valid) {
                const float4 bz = b1s[4 * nt + q];
                op[4 * nt] = make_float4(fmaxf(((aa[0] + ab[0]) + ac[0]) * A.unscale + bz.x, 0.f),
                                         fmaxf(((aa[1] + ab[1]) + ac[1]) * A.unscale + bz.y, 0.f),
                                         fmaxf(((aa[2] + ab[2]) + ac[2]) * A.unscale + bz.z, 0.f),
                                         fmaxf(((aa[3] + ab[3]) + ac[3]) * A.unscale + bz.w, 0.f));
            }
        }
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// The DF branch of the encoder in ONE kernel (round 4): df_conv0 -> df_conv1 -> df_fc_emb (+ e3) -> linear_in of the encoder GRU
// (deepfilternet3.py:115-123,176-182, modules.py:702-738,741-780).  c1 = df_conv1's output (12 KB per frame: written by dfx_k_df_conv01_h3,
// read back by dfx_k_enc_fan = 6.3 GB per pass at config 2) never exists: the 16 lanes of a matrix-op column are 16 consecutive FRAMES of
// one clip at ONE output bin instead of 16 bins of one frame, so the D fragments of df_conv1 — lane (frame, q): channels 16 nt + 4 q + r of
// that bin — are, after a split, the B operand of df_fc_emb's contraction over (bin, channel) with the frames as columns, exactly like the
// chained products inside dfx_k_df_conv01_h3.  df_fc_emb's groups are 96 = 3 x 32 consecutive inputs of the flattened [bin][channel]
// vector: a k-chunk (32 channels of one bin) lies in exactly one group at offset 0 / 32 / 64, so walking the bins in order finishes a
// group every 1.5 bins; two finished groups (32 features of emb_in = relu(fc) + e3) are one group of linear_in.  Per bin and 16 frames:
// 36 (c0, three taps) + 24 (df_conv1 pointwise) + 6 (fc) + 1 (linear_in) matrix ops against 60 + the c1 store + dfx_k_enc_fan before.
// The tile body up to c1 is dfx_k_df_conv01_h3's, expression for expression (same c1 bits); the feat_spec patches are gathered per frame
// (rows 768 B apart: L1 / L2 hits, feat_spec is 197 MB per pass), the fc fragments stream from L2 (196 KB per tile and wave).
// Needs Kg(fc) % 32 == 0, C % 32 == 0, Ng(fc) == 16 and linear_in in groups of 32 -> 16 (the released shapes; else the two kernels run).
// ---------------------------------------------------------------------------------------------------------------------
struct DfxDfEncArgs {
    const float *feat;   // [B, T, Fin, 2]
    const dfx_h8 *w0f;   // [C/16][hi,lo][64]        folded df_conv0
    const float *bias0;  // [C]
    const float *dw;     // [3][C]
    const dfx_h8 *wpf;   // [C/16][C/32][hi,lo][64]  df_conv1 pointwise (BN-scaled)
    const float *bias;   // [C]
    const dfx_h8 *wfc;   // [Fout * C/32 chunks][hi,lo][64]: chunk ci = fo * C/32 + kc of df_fc_emb (group (32 ci) / Kg, offset (32 ci) % Kg)
    const dfx_h8 *win;   // [emb/32][hi,lo][64]      linear_in, group j = features [32 j, 32 j + 32)
    const float *e3;     // [B*T, emb]
    float *emb_in;       // [B*T, emb] or null (only a skip connection around the encoder GRU reads it)
    float *xa;           // [B*T, emb/2]: relu(linear_in(emb_in))
    int64_t B, T;
    int Fin, Fout, stride, L;
    int cpg;             // k-chunks per fc group (Kg / 32)
    int nsplit = 1;      // the bins of a tile's frames are dealt to nsplit waves, Fout / nsplit consecutive bins each — whole pairs of fc groups (=
                         // whole linear_in groups), which are independent of each other: a pass of few frames (a streaming hop: 4096 frames = 256
                         // tiles) then runs on nsplit x as many waves
    float unscale0, unscale, unscale_fc, unscale_in;
    int64_t t_begin, t_end;
    unsigned int *err;
    int64_t feat_T = 0;
};

template <int C>
__global__ void __launch_bounds__(DFX_PW_THREADS, 3) dfx_k_df_enc_h3(DfxDfEncArgs A) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C / 32, C4 = C / 4;
    static_assert(C % 32 == 0, "one k-chunk is 32 channels");
    __shared__ float4 dws[3 * C4];
    __shared__ float4 b0s[C4], b1s[C4];
    __shared__ dfx_h8 wps[NT * KC * 2 * 64];
    __shared__ dfx_h8 w0s[NT * 2 * 64];   // df_conv0's fragments: read where they are used (in registers they are 32 of the 168 a wave may have)
    __shared__ __attribute__((aligned(16))) int tis[4 * DFX_PW_THREADS];
    __shared__ float4 uns[NT * DFX_PW_THREADS];   // lane-private: the carried first term of the next bin's depthwise sum (below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    for (int i = tid; i < 3 * C4; i += DFX_PW_THREADS) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    for (int i = tid; i < C4; i += DFX_PW_THREADS) {
        b0s[i] = reinterpret_cast<const float4 *>(A.bias0)[i];
        b1s[i] = reinterpret_cast<const float4 *>(A.bias)[i];
    }
    for (int i = tid; i < NT * KC * 2 * 64; i += DFX_PW_THREADS) wps[i] = A.wpf[i];
    for (int i = tid; i < NT * 2 * 64; i += DFX_PW_THREADS) w0s[i] = A.w0f[i];
    __syncthreads();
    // tile = 16 consecutive (clip, frame) pairs of the B * Tn frames to produce, one per lane of a matrix-op column: 16 consecutive frames of
    // a clip in a batch pass, the newest frame of 16 consecutive streams in a frame-by-frame pass (Tn = 1)
    const unsigned Tn = (unsigned)(A.t_end - A.t_begin), NFR = (unsigned)A.B * Tn, nsp = (unsigned)A.nsplit, ntiles = ((NFR + 15) >> 4) * nsp;
    const int bpp = A.Fout / A.nsplit;   // bins per part
    const unsigned nwaves = gridDim.x * 4;
    const int T32 = (int)A.T, Fin = A.Fin, Lk = A.L, Fout = A.Fout, emb = Fout * C / A.cpg / 2;   // emb = groups * 16 = Fout * KC / cpg * 16
    const int64_t fT = A.feat_T > 0 ? A.feat_T : (int64_t)T32;
    // the four taps (of the 3x3 window over (t, f), padded to 16) this lane feeds into the k index of the matrix op, packed (dt + 64 |
    // (df + 1) << 8 | ok << 16) and parked in LDS: the kernel sits at the 168-register edge of three waves per SIMD
    {
        int ti[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tap = 4 * q + i, kt = tap / 3;
            ti[i] = (kt - 2 + Lk + 64) | ((tap - 3 * kt) << 8) | ((tap < 9 ? 1 : 0) << 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tis[4 * tid + i] = ti[i];
    }
    float amax = 0.f;
    for (unsigned tile = (unsigned)blockIdx.x * 4 + (unsigned)dfx_wave_uniform(wave); tile < ntiles; tile += nwaves) {
        const unsigned ftile = tile / nsp, part = tile - ftile * nsp;
        const int fo0 = (int)part * bpp, fo1 = fo0 + bpp;
        const unsigned rl = (ftile << 4) + (unsigned)jl, b = rl / Tn;
        const int t = (int)A.t_begin + (int)(rl - b * Tn);   // this lane's frame
        const bool live = rl < NFR;
        // 32-bit element offsets from the (scalar) array bases: B * feat_T * Fin < 2^29 and B * T * emb < 2^31 are checked by the host
        const float2 *feat2 = reinterpret_cast<const float2 *>(A.feat);
        const unsigned cbase = b * (unsigned)fT * (unsigned)Fin;
        const unsigned row = b * (unsigned)T32 + (unsigned)t;
        float2 raw[2][4];   // taps 1 and 2 (and, before the first bin, tap 0 in raw[0])
        auto issue_to = [&](float2 (&dst)[4], int j, int fo) {   // patch j (input bin fo * stride + j - 1) of this lane's frame
            const int fi = fo * A.stride + j - 1;
            const bool ok = live && fo < fo1 && fi >= 0 && fi < Fin;
            int zt = tid;
            DFX_OPAQUE(zt);   // (a read per use, not a value carried in registers)
            const int tinfo[4] = {tis[4 * zt], tis[4 * zt + 1], tis[4 * zt + 2], tis[4 * zt + 3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tin = t + (tinfo[i] & 0xff) - 64, fin = fi + ((tinfo[i] >> 8) & 0xff) - 1;
                float2 v = make_float2(0.f, 0.f);
                if (ok && (tinfo[i] >> 16) && tin - Lk >= 0 && tin < T32 && fin >= 0 && fin < Fin) v = feat2[cbase + (unsigned)(tin * Fin + fin)];
                dst[i] = v;
            }
        };
        // One c0 tile (input bin fo * stride + j - 1 of this lane's frame): its patch -> split -> three products per 16 channels -> ReLU, added
        // to the depthwise sum u with tap j's weight (ADD) and / or parked with tap 0's weight as the first term of the NEXT bin's sum (NEXT).
        // With stride 2 (the host launches nothing else) the last tap of output bin fo and the first tap of fo + 1 read the SAME c0 tile (input bin
        // 2 fo + 1): it is computed once — two tiles per bin instead of three (24 instead of 36 of the bin's 67 matrix ops, a third of the patch loads, splits and
        // epilogues).  The parked term is w0 * v = what `0 + w0 * v` is in the three-tile form: the same bits.  It waits in LDS, lane-private
        // (16 registers the kernel does not have at three waves per SIMD).
        auto c0_tile = [&](auto jc, auto addc, auto nextc, int fo, float2 (&rw)[4], float (&u)[CPL], bool refill) {
            constexpr int j = decltype(jc)::value;
            constexpr bool ADD = decltype(addc)::value, NEXT = decltype(nextc)::value;
            float x[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[2 * i] = rw[i].x, x[2 * i + 1] = rw[i].y;
            const int fi = fo * A.stride + j - 1;
            const bool keep = live && fi >= 0 && fi < Fin;
            dfx_h8 ph, pl;
            dfx_split8_g(x, ph, pl, amax);
            if (refill) issue_to(rw, j, fo + 1);   // the patch registers are free again: the same patch of the next bin (beyond the last: zeros, no loads)
            int z0 = 0;
            DFX_OPAQUE(z0);     // (loop-invariant LDS reads: not to be hoisted back into registers)
            dfx_h8 w0h[NT], w0l[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                w0h[nt] = w0s[(nt * 2 + 0) * 64 + lane + z0];
                w0l[nt] = w0s[(nt * 2 + 1) * 64 + lane + z0];
            }
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0l[nt], ph, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], pl, acc[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = dfx_mfma_16x16x32_f16(w0h[nt], ph, acc[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bz = b0s[4 * nt + q];
                const float v0 = keep ? fmaxf(acc[nt][0] * A.unscale0 + bz.x, 0.f) : 0.f, v1 = keep ? fmaxf(acc[nt][1] * A.unscale0 + bz.y, 0.f) : 0.f;
                const float v2 = keep ? fmaxf(acc[nt][2] * A.unscale0 + bz.z, 0.f) : 0.f, v3 = keep ? fmaxf(acc[nt][3] * A.unscale0 + bz.w, 0.f) : 0.f;
                if constexpr (ADD) {
                    const float4 w = dws[j * C4 + 4 * nt + q];
                    u[4 * nt + 0] += w.x * v0;
                    u[4 * nt + 1] += w.y * v1;
                    u[4 * nt + 2] += w.z * v2;
                    u[4 * nt + 3] += w.w * v3;
                }
                if constexpr (NEXT) {
                    const float4 wn = dws[4 * nt + q];
                    uns[nt * DFX_PW_THREADS + tid] = make_float4(0.f + wn.x * v0, 0.f + wn.y * v1, 0.f + wn.z * v2, 0.f + wn.w * v3);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        f32x4 accg = f32x4{0.f, 0.f, 0.f, 0.f};   // the fc group in progress: lane (frame, q) holds its outputs 4 q + r
        float ev[8];                              // emb_in features 16 (g & 1) + 4 q + r of the two groups of a linear_in group
        int ci = fo0 * KC;                        // k-chunk of the flattened (bin, channel) vector (wave-uniform)
        {   // tap 0 of the part's first bin: the one tile nothing before it has computed
            float none[CPL];
            issue_to(raw[0], 0, fo0);
            c0_tile(I0{}, std::false_type{}, std::true_type{}, fo0, raw[0], none, false);
            issue_to(raw[0], 1, fo0);   // raw[0] / raw[1] carry taps 1 / 2 from here on
            issue_to(raw[1], 2, fo0);
        }
        for (int fo = fo0; fo < fo1; ++fo) {
            float u[CPL];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 c = uns[nt * DFX_PW_THREADS + tid];
                u[4 * nt + 0] = c.x, u[4 * nt + 1] = c.y, u[4 * nt + 2] = c.z, u[4 * nt + 3] = c.w;
            }
            c0_tile(I1{}, std::true_type{}, std::false_type{}, fo, raw[0], u, true);
            c0_tile(I2{}, std::true_type{}, std::true_type{}, fo, raw[1], u, true);
            dfx_h8 uh[KC], ul[KC];
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) dfx_split8_g(u + 8 * kc, uh[kc], ul[kc], amax);
            int zoff = 0;
            DFX_OPAQUE(zoff);  // the fragment reads are loop invariant: keep the compiler from hoisting them into 64 registers
            const dfx_h8 *wpl_ = wps + lane + zoff;
            // df_conv1's output of this bin, 32 channels (two 16-channel tiles = one k-chunk of df_fc_emb) at a time: element 4 (nt & 1) + r of
            // c1v = channel 16 nt + 4 q + r of this lane's frame
#pragma unroll
            for (int kc = 0; kc < KC; ++kc, ++ci) {
                // (this chunk's fc fragments are requested before the twelve matrix ops of the pointwise conv that produce its operand)
                int lz = lane;
                DFX_OPAQUE(lz);   // (the lane's part of the address per chunk, not a pair of 64-bit lane pointers held across the tile loop)
                const dfx_h8 *wfl_ = A.wfc + lz;
                const dfx_h8 fh = wfl_[((size_t)ci * 2 + 0) * 64], fl = wfl_[((size_t)ci * 2 + 1) * 64];
                float c1v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nt = 2 * kc + h;
                    f32x4 aa = f32x4{0.f, 0.f, 0.f, 0.f}, ab = aa, ac = aa;  // independent chains per product term
#pragma unroll
                    for (int k2 = 0; k2 < KC; ++k2) {
                        const dfx_h8 wh = wpl_[((nt * KC + k2) * 2 + 0) * 64], wl = wpl_[((nt * KC + k2) * 2 + 1) * 64];
                        aa = dfx_mfma_16x16x32_f16(wl, uh[k2], aa);
                        ab = dfx_mfma_16x16x32_f16(wh, ul[k2], ab);
                        ac = dfx_mfma_16x16x32_f16(wh, uh[k2], ac);
                    }
                    const float4 bz = b1s[4 * nt + q];
                    c1v[4 * h + 0] = fmaxf(((aa[0] + ab[0]) + ac[0]) * A.unscale + bz.x, 0.f);
                    c1v[4 * h + 1] = fmaxf(((aa[1] + ab[1]) + ac[1]) * A.unscale + bz.y, 0.f);
                    c1v[4 * h + 2] = fmaxf(((aa[2] + ab[2]) + ac[2]) * A.unscale + bz.z, 0.f);
                    c1v[4 * h + 3] = fmaxf(((aa[3] + ab[3]) + ac[3]) * A.unscale + bz.w, 0.f);
                }
                // ---- df_fc_emb over these 32 channels
                dfx_h8 ch, cl;
                dfx_split8_g(c1v, ch, cl, amax);
                accg = dfx_mfma_16x16x32_f16(fl, ch, accg);
                accg = dfx_mfma_16x16x32_f16(fh, cl, accg);
                accg = dfx_mfma_16x16x32_f16(fh, ch, accg);
                if ((ci + 1) % A.cpg == 0) {   // group g = ci / cpg is complete (wave-uniform)
                    const int g = ci / A.cpg;
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (live) e = *reinterpret_cast<const float4 *>(A.e3 + (row * (unsigned)emb + (unsigned)(16 * g + 4 * q)));
                    const float4 v = make_float4(fmaxf(accg[0] * A.unscale_fc, 0.f) + e.x, fmaxf(accg[1] * A.unscale_fc, 0.f) + e.y,
                                                 fmaxf(accg[2] * A.unscale_fc, 0.f) + e.z, fmaxf(accg[3] * A.unscale_fc, 0.f) + e.w);
                    if (A.emb_in && live) *reinterpret_cast<float4 *>(A.emb_in + (row * (unsigned)emb + (unsigned)(16 * g + 4 * q))) = v;
                    accg = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (g & 1) {   // with the previous group: one group of linear_in, k-index (q, i) <-> feature 16 (i >> 2) + 4 q + (i & 3)
                        ev[4] = v.x, ev[5] = v.y, ev[6] = v.z, ev[7] = v.w;
                        const int jg = g >> 1;
                        const dfx_h8 *wil_ = A.win + lane + zoff;
                        const dfx_h8 ih = wil_[((size_t)jg * 2 + 0) * 64], il = wil_[((size_t)jg * 2 + 1) * 64];
                        dfx_h8 eh, el;
                        dfx_split8_g(ev, eh, el, amax);
                        f32x4 o = dfx_mfma_16x16x32_f16(il, eh, f32x4{0.f, 0.f, 0.f, 0.f});
                        o = dfx_mfma_16x16x32_f16(ih, el, o);
                        o = dfx_mfma_16x16x32_f16(ih, eh, o);
                        if (live)
                            *reinterpret_cast<float4 *>(A.xa + (row * (unsigned)(emb / 2) + (unsigned)(16 * jg + 4 * q))) =
                                make_float4(fmaxf(o[0] * A.unscale_in, 0.f), fmaxf(o[1] * A.unscale_in, 0.f), fmaxf(o[2] * A.unscale_in, 0.f),
                                            fmaxf(o[3] * A.unscale_in, 0.f));
                    } else {
                        ev[0] = v.x, ev[1] = v.y, ev[2] = v.z, ev[3] = v.w;
                    }
                }
            }
        }
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// df_dec.df_convp with df_conv0 recomputed, fp16-split form of dfx_k_df_convp2<C, KT, true> (same run decomposition: a wave walks a
// segment of frames for 16 bins of one clip).  The window holds the hi/lo halves of the last KT c0 frames (the B operands), the
// A fragments [KT][C/32][hi,lo] are persistent: 3*KT*C/32 + 3*C/16 MFMAs per 16 bins and frame.
struct DfxCphArgs {
    const float *feat;   // [B, T, Fd, 2]
    const dfx_h8 *w0f;   // [C/16][hi,lo][64]       folded df_conv0
    const float *bias0;  // [C]
    const dfx_h8 *wf;    // [KT][C/32][hi,lo][64]   folded df_convp, n = 2*O outputs padded to 16
    const float *bias;   // [16]
    float *out;          // [B, NO/2, T, Fd, 2]  (tap-major, DFX_COEF_BOTF)
    int64_t B, T;
    int Fd, NO, nfb, nseg, tseg, L;
    float unscale0, unscale;
    int64_t t_begin, t_zero, t_end;  // as in DfxCp2Args
    unsigned int *err;        // as in DfxC01hArgs
    int64_t feat_T = 0;       // as in DfxC01hArgs
};

// (Two 16-bin blocks per wave side by side — two independent tiles in one instruction stream, df_conv0's fragments in LDS to make room for the
// second window — was built and measured the same: 2.07 ms alone, 14.13 / 14.07 vs 14.10 / 14.17 ms per step.)
// (Round 4: the fragments in LDS instead — 28 KB per workgroup, read where they are used — so that the kernel fits two waves per SIMD and one
// wave's VALU work, ~310 instructions per tile, runs under the other's 42 matrix ops, was built again on the current tree and measured:
// 256 registers + 94 spilled, 2.19 instead of 2.01 ms alone, the step +0.7 ms: profiles/r04_df_out_and_convp_lds.log.  Everything in registers,
// one wave per SIMD, stays.)
// (Round 5: a form with PENDING SUMS instead of the window of c0 frames — a frame's c0 tile, split once, times all KT taps into the sums of the
// outputs t .. t + KT - 1, as dfx_k_df_convp_step does — with df_conv0's and the lo fragments in LDS fits two waves per SIMD and is faster ALONE,
// 1.56 instead of 1.90 ms, one wave's matrix ops under the other's vector work; but beside the GRU phase or beside the front the step got 0.7-1.4 ms
// SLOWER (14.0-14.8 vs 13.3 ms, every placement): two 256-register waves per SIMD leave no room for another kernel's wave, where this kernel's
// single 352-register wave leaves 160 registers per SIMD to the kernels it runs beside.  Removed; profiles/r05_convp_pending_sums.log.)
template <int C, int KT>
__global__ void __launch_bounds__(256, 1) dfx_k_df_convp_h3(DfxCphArgs A) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C >= 32 ? C / 32 : 1;
    static_assert(C % 32 == 0, "one k-chunk is 32 channels");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    float4 bias0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias0[nt] = reinterpret_cast<const float4 *>(A.bias0)[4 * nt + q];
    dfx_h8 w0h[NT], w0l[NT], wh[KT][KC], wl[KT][KC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        w0h[nt] = A.w0f[(nt * 2 + 0) * 64 + lane];
        w0l[nt] = A.w0f[(nt * 2 + 1) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            wh[k][kc] = A.wf[((k * KC + kc) * 2 + 0) * 64 + lane];
            wl[k][kc] = A.wf[((k * KC + kc) * 2 + 1) * 64 + lane];
            // The 80 registers of df_convp's fragments are pinned in the accumulation half of the register file, where the matrix ops read them
            // directly.  Left to the allocator they moved between the halves: 468 v_accvgpr copies among the 2605 vector instructions of the
            // unrolled frame loop (32 of 2176 with the pin; the step 14.16-14.20 -> 13.93-14.06 ms).
            DFX_PIN_AGPR(wh[k][kc]);
            DFX_PIN_AGPR(wl[k][kc]);
        }
    float biasr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasr[r] = A.bias[4 * q + r];
    const int64_t nruns = A.B * A.nfb * A.nseg;
    float amax = 0.f;   // range guard of the f16 splits
    // Round 4: the run decomposition is WAVE-UNIFORM (clip, bin block, segment and every frame index live in scalar registers; the wave index
    // goes through readfirstlane) and a lane adds its bin and its taps as 32-bit element offsets from the scalar base of its clip's frame — the
    // per-lane 64-bit (clip, frame, bin) arithmetic and the division of the tap index by 3 at every patch load were ~40 % of the kernel's
    // non-matrix instructions.  B * feat_T * Fd < 2^29 (checked by the host: 32-bit offsets).
    const int Fd = A.Fd, T32 = (int)A.T, Lk = A.L;
    const int fT = (int)(A.feat_T > 0 ? A.feat_T : A.T);
    int toff[4], tdt[4], tdf[4];   // per lane and tap i: element offset (dt * Fd + df) from (frame, bin), dt, df; taps >= 9 are the padding of K = 16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tap = 4 * q + i, kt = tap / 3;
        tdt[i] = tap < 9 ? kt - 2 + Lk : (1 << 20);   // (padding taps: a frame index that is never inside the clip)
        tdf[i] = tap - 3 * kt - 1;
        toff[i] = (kt - 2 + Lk) * Fd + tdf[i];
    }
    const float2 *feat2 = reinterpret_cast<const float2 *>(A.feat);
    for (int64_t run = (int64_t)blockIdx.x * 4 + dfx_wave_uniform(wave); run < nruns; run += (int64_t)gridDim.x * 4) {
        const int seg = (int)(run % A.nseg);
        const int64_t rest = run / A.nseg;
        const int fb = (int)(rest % A.nfb);
        const int64_t b = rest / A.nfb;
        const int f = fb * 16 + jl;
        const bool fvalid = f < Fd;
        const int64_t t0 = A.t_begin + (int64_t)seg * A.tseg;
        const int64_t t1 = (t0 + A.tseg < A.t_end) ? t0 + A.tseg : A.t_end;
        const unsigned cbase = (unsigned)b * (unsigned)fT * (unsigned)Fd;   // scalar
        // the patch of frame t (scalar) and this lane's bin: four taps, zero outside the clip / the bins / the causal padding
        auto patch = [&](int t, bool ok, float2 (&rw)[4]) {
            const unsigned pbase = cbase + (unsigned)(t * Fd + f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tin = t + tdt[i], fin = f + tdf[i];
                // branch-free: a tap outside reads the clip's first element (always there) and is zeroed by a select — an exec-masked load is a
                // save / branch / restore sequence per tap
                const bool in = ok && tin - Lk >= 0 && tin < T32 && fin >= 0 && fin < Fd;
                const float2 v = feat2[in ? pbase + (unsigned)toff[i] : cbase];
                rw[i] = make_float2(in ? v.x : 0.f, in ? v.y : 0.f);
            }
        };
        dfx_h8 xh[KT][KC], xl[KT][KC];  // frame tau lives in slot (tau - t0) mod KT
        float2 raw[4];
        auto make_frame = [&](dfx_h8 (&dh)[KC], dfx_h8 (&dl)[KC], int64_t tau, const float2 (&rw)[4]) {
            float c0v[CPL];
            // frames before the clip are the zero padding of c0 itself (wave-uniform test)
            dfx_c0_tile_h3<C>(w0h, w0l, bias0, A.unscale0, rw, fvalid && tau >= A.t_zero, c0v, amax);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) dfx_split8_g(c0v + 8 * kc, dh[kc], dl[kc], amax);
        };
        dfx_static_for<1, KT>([&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            const int64_t tau = t0 - KT + sl;
            patch((int)tau, fvalid && tau >= 0, raw);
            make_frame(xh[sl], xl[sl], tau, raw);
        });
        patch((int)t0, fvalid, raw);
        for (int64_t tb = t0; tb < t1; tb += KT) {
            dfx_static_for<0, KT>([&](auto pc) {
                constexpr int ph = decltype(pc)::value;
                const int64_t t = tb + ph;
                if (t < t1) {  // wave-uniform
                    float2 cur[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur[i] = raw[i];
                    patch((int)t + 1, fvalid && t + 1 < t1, raw);
                    make_frame(xh[ph], xl[ph], t, cur);
                    // three independent accumulation chains (one per product term), summed small-to-large at the end
                    f32x4 aa = f32x4{0.f, 0.f, 0.f, 0.f}, ab = aa, ac = aa;
                    dfx_static_for<0, KT>([&](auto kcn) {
                        constexpr int k = decltype(kcn)::value;
                        constexpr int sl = (ph + 1 + k) % KT;  // tap k reads frame t - (KT-1) + k
#pragma unroll
                        for (int kc = 0; kc < KC; ++kc) {
                            const dfx_h8 whk = wh[k][kc], wlk = wl[k][kc];
                            aa = dfx_mfma_16x16x32_f16(wlk, xh[sl][kc], aa);
                            ab = dfx_mfma_16x16x32_f16(whk, xl[sl][kc], ab);
                            ac = dfx_mfma_16x16x32_f16(whk, xh[sl][kc], ac);
                        }
                    });
                    f32x4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = (aa[r] + ab[r]) + ac[r];
                    if (fvalid) {
                        float *op = A.out + ((b * (A.NO / 2) * A.T + t) * A.Fd + f) * 2;
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (4 * q + 2 * h < A.NO)
                                *reinterpret_cast<float2 *>(op + (int64_t)(2 * q + h) * A.T * A.Fd * 2) =
                                    make_float2(fmaxf(acc[2 * h] * A.unscale + biasr[2 * h], 0.f),
                                                fmaxf(acc[2 * h + 1] * A.unscale + biasr[2 * h + 1], 0.f));
                    }
                }
            });
        }
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// dfx_k_df_convp_step: df_convp for ONE new frame per stream (the frame-by-frame runtime).  dfx_k_df_convp_h3 recomputes the kt - 1
// frames of c0 in front of a segment as warm-up — for a segment of one frame four of its five c0 tiles per output (334 us at 4096
// streams, longer than the whole ERB branch of the hop).  The convolution is linear in its time taps, so the handle keeps, per stream
// and 16-bin block, the PENDING SUMS of the next kt - 1 outputs instead of any history of c0: out[t + j] so far = the taps of the frames
// <= t (fp32 accumulator fragments as the matrix op leaves them, in the weights' scale: pend[b][slot][fb][lane], slot = frame % (kt - 1),
// 1 KB each).  A wave owns (stream, block): it computes the new frame's c0 tile once, multiplies it with all kt taps — tap kt-1 completes
// out[t] from its pending sum, tap k adds to the sum of out[t + kt-1-k], tap 0 starts the sum of out[t + kt-1] in the slot out[t] frees
// (read first, written last, by the same lane).  Per block and hop 4 KB are read and 4 KB written; the first form of this kernel kept
// the split c0 tiles of the last kt - 1 frames (16 KB read, 4 KB written: 490 MB per hop at 4096 streams, 150 us beside which the
// encoder's GRU step took 95 us instead of 40).  REBUILD: the sums are not current (after a reset they are all zeros = the causal
// padding and need none; after calls of several hops or gated passes they do): the kt - 1 older frames are recomputed from the feature
// window like dfx_k_df_convp_h3 does, and their taps summed up.
// Gated handles (par / cnt non-null): the delay line in front of df_convp only moves for the streams whose DF decoder runs on this hop,
// and that is decided later in the pass.  The sums then live twice per stream, pend[b][parity]: the kernel reads the half par[b] names
// (slot of the new frame: cnt[b] % (kt - 1), the frames that stream's decoder has consumed) and writes the updated sums into the OTHER half;
// dfx_k_gate_pend_commit flips par[b] and counts the frame where the decoder ran — elsewhere the new frame is simply dropped, as the
// reference's pulsed model drops it (tract.rs: a sub-model that is not run does not advance).
template <int C, int KT, bool REBUILD>
__global__ void __launch_bounds__(256, REBUILD ? 1 : 2) dfx_k_df_convp_step(DfxCphArgs A, f32x4 *pend, int slot_new, const unsigned char *par = nullptr,
                                                             const int *cnt = nullptr) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C / 32, NS = KT - 1;
    static_assert(C % 32 == 0 && KT >= 2, "one k-chunk is 32 channels; kt = 1 has no history");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    dfx_h8 w0h[NT], w0l[NT];
    float4 bias0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        w0h[nt] = A.w0f[(nt * 2 + 0) * 64 + lane];
        w0l[nt] = A.w0f[(nt * 2 + 1) * 64 + lane];
        bias0[nt] = reinterpret_cast<const float4 *>(A.bias0)[4 * nt + q];
    }
    float biasr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasr[r] = A.bias[4 * q + r];
    const int64_t nruns = A.B * A.nfb;
    const int64_t t = A.T - 1;   // the new frame (local index in the feature window)
    float amax = 0.f;
    for (int64_t run = (int64_t)blockIdx.x * 4 + wave; run < nruns; run += (int64_t)gridDim.x * 4) {
        const int fb = (int)(run % A.nfb);
        const int64_t b = run / A.nfb;
        const int f = fb * 16 + jl;
        const bool fvalid = f < A.Fd;
        const int half_in = par ? (int)(par[b] & 1) : 0, half_out = par ? half_in ^ 1 : 0, halves = par ? 2 : 1;
        const int slot0 = cnt ? cnt[b] % NS : slot_new;
        auto slot_ptr = [&](int j, int half) { return pend + ((((size_t)b * halves + half) * NS + (slot0 + j) % NS) * A.nfb + fb) * 64 + lane; };   // sum of out[t + j]
        // sums[o]: out[t + o] — o = 0 is completed here, 1 .. NS go back to the handle; per sum three chains (one per product term, added
        // small-to-large at the end: the order of dfx_k_df_convp_h3 within a tap)
        f32x4 sa[KT], sb[KT], sc[KT];
#pragma unroll
        for (int o = 0; o < KT; ++o) sa[o] = sb[o] = f32x4{0.f, 0.f, 0.f, 0.f}, sc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!REBUILD) {
#pragma unroll
            for (int o = 0; o < NS; ++o) sc[o] = *slot_ptr(o, half_in);   // (out[t + NS] has no taps yet)
        }
        dfx_static_for<(REBUILD ? 0 : KT - 1), KT>([&](auto dc) {
            constexpr int d = decltype(dc)::value;   // frame t - (KT-1) + d; its tap k belongs to out[t + d - k]
            const int64_t tau = t - (KT - 1) + d;
            float2 raw[4];
            dfx_c0_patch_load(A.feat, b, tau, f, fvalid && tau >= 0, A.T, A.Fd, A.L, q, raw, A.feat_T);
            float c0v[CPL];
            dfx_c0_tile_h3<C>(w0h, w0l, bias0, A.unscale0, raw, fvalid && tau >= A.t_zero, c0v, amax);
            dfx_h8 xh[KC], xl[KC];
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) dfx_split8_g(c0v + 8 * kc, xh[kc], xl[kc], amax);
            // consecutive matrix ops go to different sums (an op that waits for its predecessor's accumulator stalls for that op's latency)
            int zoff = 0;
            DFX_OPAQUE(zoff);   // the fragment loads are loop invariant: hoisted out of the run loop they are 80 registers (28 bytes of scratch at two waves per SIMD)
            const dfx_h8 *wfl = A.wf + lane + zoff;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                dfx_h8 wl[KT], wh[KT];
                dfx_static_for<0, d + 1>([&](auto kcn) {
                    constexpr int k = decltype(kcn)::value;
                    wl[k] = wfl[((k * KC + kc) * 2 + 1) * 64], wh[k] = wfl[((k * KC + kc) * 2 + 0) * 64];
                });
                dfx_static_for<0, d + 1>([&](auto kcn) { constexpr int k = decltype(kcn)::value; sa[d - k] = dfx_mfma_16x16x32_f16(wl[k], xh[kc], sa[d - k]); });
                dfx_static_for<0, d + 1>([&](auto kcn) { constexpr int k = decltype(kcn)::value; sb[d - k] = dfx_mfma_16x16x32_f16(wh[k], xl[kc], sb[d - k]); });
                dfx_static_for<0, d + 1>([&](auto kcn) { constexpr int k = decltype(kcn)::value; sc[d - k] = dfx_mfma_16x16x32_f16(wh[k], xh[kc], sc[d - k]); });
            }
        });
        f32x4 sum[KT];
#pragma unroll
        for (int o = 0; o < KT; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[o][r] = (sa[o][r] + sb[o][r]) + sc[o][r];
#pragma unroll
        for (int o = 1; o < KT; ++o) *slot_ptr(o, half_out) = sum[o];   // (slot of out[t + NS] = the one out[t] was read from)
        if (fvalid) {
            float *op = A.out + ((b * (A.NO / 2) * A.T + t) * A.Fd + f) * 2;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (4 * q + 2 * h < A.NO)
                    *reinterpret_cast<float2 *>(op + (int64_t)(2 * q + h) * A.T * A.Fd * 2) =
                        make_float2(fmaxf(sum[0][2 * h] * A.unscale + biasr[2 * h], 0.f), fmaxf(sum[0][2 * h + 1] * A.unscale + biasr[2 * h + 1], 0.f));
        }
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Frame-resident conv chains.  Apart from their first layer every conv of the ERB encoder / decoder is per frame (1x3 over
// frequency), so a whole chain can run on one frame pair without its intermediates leaving the CU: a wave owns DFX_CH_NF = 2
// consecutive rows (frames) of the [B*T] axis, keeps each stage's output in a private LDS strip [positions][C + 4] and feeds the
// next stage's depthwise taps from there.  HBM then only sees what the network really needs (the four encoder outputs, which are
// the decoder's skips) instead of every intermediate twice.  Two frames make the 8-position stages a full 16-wide MFMA tile.
// dfx_chain_stage is dfx_k_pwconv's tile body with the B operand read from LDS (same operand roles, same k order -> the same
// bits); only wave-level synchronisation is needed because a strip is private to its wave.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_CH_NF 2
template <int C, int MODE, typename Epi>
static __device__ __forceinline__ void dfx_chain_stage(const float *in, int Fin, int Fout, int stride, int npos,
                                                       const float4 *dws, const float (&areg)[C / 16][C / 4],
                                                       const float4 (&biasr)[C / 16], int lane, Epi &&epi) {
    constexpr int CPL = C / 4, NT = C / 16, V4 = CPL / 4, LD = C + 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int p0 = 0; p0 < npos; p0 += 16) {
        const int p = p0 + jl;
        const bool valid = p < npos;
        const int fr = p / Fout, fo = p - fr * Fout;
        float u[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int fi;
            bool ok;
            if (MODE == DFX_PW_MODE_DW3) {
                fi = fo * stride + j - 1;
                ok = fi >= 0 && fi < Fin;
            } else {  // transposed: fo = 2*fi - 1 + j
                const int num = fo + 1 - j;
                fi = num >> 1;
                ok = num >= 0 && (num & 1) == 0 && fi < Fin;
            }
            if (valid && ok) {
                const float4 *xp = reinterpret_cast<const float4 *>(in + (fr * Fin + fi) * LD + CPL * q);
#pragma unroll
                for (int v = 0; v < V4; ++v) {
                    const float4 xv = xp[v];
                    const float4 w = dws[j * (C / 4) + V4 * q + v];
                    u[4 * v + 0] += w.x * xv.x;
                    u[4 * v + 1] += w.y * xv.y;
                    u[4 * v + 2] += w.z * xv.z;
                    u[4 * v + 3] += w.w * xv.w;
                }
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[nt][ks], u[ks], acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)  // lane (p, q) holds output channels 16*nt + 4*q .. +3
            epi(p, valid, nt,
                make_float4(fmaxf(acc[nt][0] + biasr[nt].x, 0.f), fmaxf(acc[nt][1] + biasr[nt].y, 0.f),
                            fmaxf(acc[nt][2] + biasr[nt].z, 0.f), fmaxf(acc[nt][3] + biasr[nt].w, 0.f)));
    }
}

template <int C>
static __device__ __forceinline__ void dfx_chain_load_w(const float *wt, const float *bias, int lane, float (&areg)[C / 16][C / 4],
                                                        float4 (&biasr)[C / 16]) {
    constexpr int CPL = C / 4, NT = C / 16;
    const int q = lane >> 4, jl = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks) areg[nt][ks] = wt[(CPL * q + ks) * C + 16 * nt + jl];
        biasr[nt] = reinterpret_cast<const float4 *>(bias)[4 * nt + q];
    }
}

// The same stage on the fp16-split matrix path (docs/measurements.md §5a): the pointwise C x C contraction runs as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_f16 (3 * C/32 matrix ops of 16 cycles per 16 output channels instead of C/4 fp32 ops of 32 cycles: the fp32
// form keeps the matrix pipe busy for more than half of these kernels' run time).  The depthwise taps stay fp32 on the VALU; u is
// split on the fly; lane (pos, q) feeds channels (C/4)q + 8kc + i as element i of k-chunk kc, the host packs W accordingly (pack_pw_h3).
template <int C>
static __device__ __forceinline__ void dfx_chain_load_w_h3(const dfx_h8 *frags, const float *bias, int lane, dfx_h8 (&ahi)[C / 16][C / 32],
                                                           dfx_h8 (&alo)[C / 16][C / 32], float4 (&biasr)[C / 16]) {
    constexpr int NT = C / 16, KC = C / 32;
    const int q = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            ahi[nt][kc] = frags[((size_t)(nt * KC + kc) * 2 + 0) * 64 + lane];
            alo[nt][kc] = frags[((size_t)(nt * KC + kc) * 2 + 1) * 64 + lane];
        }
        biasr[nt] = reinterpret_cast<const float4 *>(bias)[4 * nt + q];
    }
}
template <int C, int MODE, typename Epi>
static __device__ __forceinline__ void dfx_chain_stage_h3(const float *in, int Fin, int Fout, int stride, int npos, const float4 *dws,
                                                          const dfx_h8 (&ahi)[C / 16][C / 32], const dfx_h8 (&alo)[C / 16][C / 32],
                                                          const float4 (&biasr)[C / 16], float unscale, float &amax, int lane, Epi &&epi) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C / 32, V4 = CPL / 4, LD = C + 4;
    const int q = lane >> 4, jl = lane & 15;
    for (int p0 = 0; p0 < npos; p0 += 16) {
        const int p = p0 + jl;
        const bool valid = p < npos;
        const int fr = p / Fout, fo = p - fr * Fout;
        float u[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int fi;
            bool ok;
            if (MODE == DFX_PW_MODE_DW3) {
                fi = fo * stride + j - 1;
                ok = fi >= 0 && fi < Fin;
            } else {  // transposed: fo = 2*fi - 1 + j
                const int num = fo + 1 - j;
                fi = num >> 1;
                ok = num >= 0 && (num & 1) == 0 && fi < Fin;
            }
            if (valid && ok) {
                const float4 *xp = reinterpret_cast<const float4 *>(in + (fr * Fin + fi) * LD + CPL * q);
#pragma unroll
                for (int v = 0; v < V4; ++v) {
                    const float4 xv = xp[v];
                    const float4 w = dws[j * (C / 4) + V4 * q + v];
                    u[4 * v + 0] += w.x * xv.x;
                    u[4 * v + 1] += w.y * xv.y;
                    u[4 * v + 2] += w.z * xv.z;
                    u[4 * v + 3] += w.w * xv.w;
                }
            }
        }
        dfx_h8 bhi[KC], blo[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) dfx_split8_g(u + 8 * kc, bhi[kc], blo[kc], amax);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = dfx_mfma_16x16x32_f16(alo[nt][kc], bhi[kc], acc[nt]);
                acc[nt] = dfx_mfma_16x16x32_f16(ahi[nt][kc], blo[kc], acc[nt]);
                acc[nt] = dfx_mfma_16x16x32_f16(ahi[nt][kc], bhi[kc], acc[nt]);
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            epi(p, valid, nt,
                make_float4(fmaxf(acc[nt][0] * unscale + biasr[nt].x, 0.f), fmaxf(acc[nt][1] * unscale + biasr[nt].y, 0.f),
                            fmaxf(acc[nt][2] * unscale + biasr[nt].z, 0.f), fmaxf(acc[nt][3] * unscale + biasr[nt].w, 0.f)));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_pwconv_f: dfx_k_pwconv with whole frames staged through LDS.  dfx_k_pwconv hands every lane the 64 bytes of ITS position and
// channel quarter straight from HBM: a 16-byte load instruction of a wave then touches 16 positions x 4 quarters = 64 separate
// 64-byte segments, and the four stores of a tile do the same — the vector cache serves such an instruction segment by segment
// and the kernel runs at ~1.8 TB/s.  Here a wave owns G consecutive frames (G * Fout = one or more 16-position tiles): it reads
// their G * Fin * C input floats as ONE contiguous run per frame with fully coalesced float4 loads (the pathway operand of the
// decoder is folded in on the way: x + relu(a * skip + b)), parks them in a wave-private strip [rows][C + 4], runs
// dfx_chain_stage on the strip (same operand roles and k order as dfx_k_pwconv: same bits), collects the D fragments in a second
// strip and writes the G * Fout * C outputs back as contiguous float4 runs.  The next item's loads are issued before the current
// item's matrix work; only wave-level synchronisation.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_PWF_MAXV 8   /* float4s per lane and item, in and out */
static __host__ __device__ __forceinline__ int dfx_pwf_group(int C, int Fin, int Fout) {
    int g = Fout < 16 && 16 % Fout == 0 ? 16 / Fout : 1;   // whole 16-position tiles
    while (2 * g * Fin * (C / 4) <= 64 * 4 && 2 * g * Fout * (C / 4) <= 64 * DFX_PWF_MAXV) g *= 2;   // fill four loads per lane
    return g;
}
static __host__ __device__ __forceinline__ size_t dfx_pwf_wave_floats(int C, int Fin, int Fout) {
    return (size_t)dfx_pwf_group(C, Fin, Fout) * (size_t)(Fin + Fout) * (size_t)(C + 4);
}
static __host__ __device__ __forceinline__ size_t dfx_pwf_smem(int C, int Fin, int Fout) {
    return (size_t)(5 * C / 4) * 16 + 4 * dfx_pwf_wave_floats(C, Fin, Fout) * sizeof(float);
}
static __host__ __device__ __forceinline__ int dfx_pwf_nvi(int C, int Fin, int Fout) {
    return dfx_pwf_group(C, Fin, Fout) * Fin * (C / 4) <= 64 * 4 ? 4 : DFX_PWF_MAXV;
}
static __host__ __device__ __forceinline__ bool dfx_pwf_ok(int C, int Fin, int Fout) {
    const int G = dfx_pwf_group(C, Fin, Fout);
    return G * Fin * (C / 4) <= 64 * DFX_PWF_MAXV && G * Fout * (C / 4) <= 64 * DFX_PWF_MAXV && dfx_pwf_smem(C, Fin, Fout) <= 64 * 1024;
}

template <int C, int MODE, bool SKIP, int NVI /* input float4s per lane and item: 4 or DFX_PWF_MAXV */, bool H3 = false>
__global__ void __launch_bounds__(DFX_PW_THREADS, (SKIP && NVI == DFX_PWF_MAXV) ? 1 : 2) dfx_k_pwconv_f(DfxPwArgs A) {   // (pathway + 8 loads in flight: 44 registers over the budget of two waves per SIMD; only the unfused decoder tail, DFX_FUSE_TAIL=0, runs that form)
    constexpr int NT = C / 16, CPL = C / 4, LD = C + 4, C4 = C / 4, KC = H3 ? C / 32 : 1;
    DFX_DYN_SMEM(float4, dfx_pwf_smem4);
    float4 *dws = dfx_pwf_smem4, *sks = dfx_pwf_smem4 + 3 * C4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    const int G = dfx_pwf_group(C, A.Fin, A.Fout);
    const int nrow_in = G * A.Fin, npos = G * A.Fout;
    float *sin = reinterpret_cast<float *>(dfx_pwf_smem4 + 5 * C4) + (size_t)wave * dfx_pwf_wave_floats(C, A.Fin, A.Fout);
    float *sout = sin + (size_t)nrow_in * LD;
    for (int i = tid; i < 3 * C4; i += DFX_PW_THREADS) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    for (int i = tid; i < 2 * C4; i += DFX_PW_THREADS)
        sks[i] = SKIP ? (i < C4 ? reinterpret_cast<const float4 *>(A.sk_a)[i] : reinterpret_cast<const float4 *>(A.sk_b)[i - C4])
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    float areg[H3 ? 1 : NT][H3 ? 1 : CPL];
    dfx_h8 ahi[NT][KC], alo[NT][KC];
    float4 biasr[NT];
    float amax = 0.f;
    if constexpr (H3) dfx_chain_load_w_h3<C>(A.wt_h3, A.bias, lane, ahi, alo, biasr);
    else dfx_chain_load_w<C>(A.wt, A.bias, lane, areg, biasr);
    __syncthreads();
    const int64_t nitems = (A.R + G - 1) / G;
    const int nin4 = nrow_in * C4, nout4 = npos * C4, fin4 = A.Fin * C4, fout4 = A.Fout * C4;
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const float4 *s4 = reinterpret_cast<const float4 *>(A.skip);
    float4 *o4 = reinterpret_cast<float4 *>(A.out);
    float4 xr[NVI], sr[SKIP ? NVI : 1];
    // frame of a 16-byte piece and its row: a float-reciprocal quotient and an incremental row map (one division per item) — as integer
    // divisions per piece these were two dozen per item, a third of the kernel's vector instructions in the time-chunked pipeline
    const float inv_fin4 = 1.0f / (float)fin4, inv_fout = 1.0f / (float)A.Fout;
    auto issue = [&](int64_t item) {
        const int64_t base = item * G;
        const DfxRowBase rb = dfx_row_base(A.rm, base);
        // (the per-piece parts of an address — frame of the piece, its offset inside the frame — are loop invariant: hoisted as 64-bit values
        // per piece they left the widest form of the kernel one register pair short: a scratch reload and a full vmcnt(0) in the middle of
        // every item's loads.  Recomputed per item from an opaque copy of the lane index: a dozen instructions.)
        int ln = lane;
        DFX_OPAQUE(ln);
        const float4 *xq = x4, *sq = s4;
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            const int idx = ln + 64 * i;
            xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (SKIP) sr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < nin4) {
                const int fr = dfx_div_small(idx, fin4, inv_fin4);
                const int64_t rl = base + fr;
                if (rl < A.R) {
                    const int64_t off = dfx_row_at(A.rm, base, rb, fr) * fin4 + (idx - fr * fin4);
                    xr[i] = xq[off];
                    if (SKIP) sr[i] = sq[off];
                }
            }
        }
    };
    int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item < nitems) issue(item);
    // the pathway's scale / shift of this lane's channel quad: (lane + 64 i) % C4 does not depend on i (C4 divides 64), and read from LDS
    // between the strip stores below each pair was a round trip of its own (an LDS read cannot be moved across an LDS store)
    static_assert(64 % C4 == 0, "channel quad of a lane must not depend on the load index");
    const float4 ska = sks[lane % C4], skb = sks[C4 + lane % C4];
    for (; item < nitems; item += (int64_t)gridDim.x * 4) {
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nin4) {
                const int row = idx / C4, c4 = idx - row * C4;
                float4 v = xr[i];
                if (SKIP) {
                    const float4 sv = sr[i], a = ska, bb = skb;
                    v.x += fmaxf(a.x * sv.x + bb.x, 0.f);
                    v.y += fmaxf(a.y * sv.y + bb.y, 0.f);
                    v.z += fmaxf(a.z * sv.z + bb.z, 0.f);
                    v.w += fmaxf(a.w * sv.w + bb.w, 0.f);
                }
                *reinterpret_cast<float4 *>(sin + row * LD + 4 * c4) = v;
            }
        }
        const int64_t next = item + (int64_t)gridDim.x * 4;
        if (next < nitems) issue(next);   // in flight during this item's matrix work
        DFX_WAVE_SYNC();
        auto epi = [&](int p, bool valid, int nt, float4 v) {
            if (valid) *reinterpret_cast<float4 *>(sout + p * LD + 16 * nt + 4 * q) = v;
        };
        if constexpr (H3) dfx_chain_stage_h3<C, MODE>(sin, A.Fin, A.Fout, A.stride, npos, dws, ahi, alo, biasr, A.unscale, amax, lane, epi);
        else dfx_chain_stage<C, MODE>(sin, A.Fin, A.Fout, A.stride, npos, dws, areg, biasr, lane, epi);
        DFX_WAVE_SYNC();
        const int64_t obase = item * G;
        const DfxRowBase orb = dfx_row_base(A.rm, obase);
#pragma unroll
        for (int i = 0; i < DFX_PWF_MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nout4) {
                const int p = idx / C4, c4 = idx - p * C4;
                const int fr = dfx_div_small(p, A.Fout, inv_fout);
                const int64_t rl = obase + fr;
                if (rl < A.R) o4[dfx_row_at(A.rm, obase, orb, fr) * fout4 + (idx - fr * fout4)] = *reinterpret_cast<const float4 *>(sout + p * LD + 4 * c4);
            }
        }
    }
    if (H3 && amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);   // a value left the f16 range of the split: reported, not hidden
}

// ERB encoder head, fused: erb_conv0 (3x3 from one channel, VALU, same arithmetic as dfx_k_conv_in_erb) -> erb_conv1 (stride 2)
// (deepfilternet3.py:106-109,168-169).  A wave owns one frame: its E positions of e0 are one LDS strip and its E/2 positions of e1
// one or two MFMA tiles.  e0 (the largest ERB activation) is written once and never read back by the encoder; the three feat_erb
// rows of the NEXT frame are fetched while the current one is computed.  (Fusing erb_conv2/3 as well was measured slower: their
// weights push the kernel to one wave per SIMD, where its dependent LDS -> VALU -> MFMA phases cannot overlap.)
struct DfxEncArgs {
    const float *feat;  // [B, T, E]
    const float *w0, *b0;            // erb_conv0 [3][3][C], [C]
    const float *dw, *wt, *bias;     // erb_conv1: [3][C], [C][C], [C]
    float *e0, *e1;                  // [B*T, E, C], [B*T, E/2, C]
    int64_t B, T;
    int E, L;
    int64_t t_begin, t_end;          // only frames [t_begin, t_end) of every clip are produced (streaming: the frames before are history)
    const dfx_h8 *wt_h3 = nullptr;   // erb_conv1's pointwise weights as f16 hi/lo fragments (H3 form)
    float unscale = 1.f;
    unsigned int *err = nullptr;
    int64_t feat_T = 0;              // > 0: frames per clip of feat (see DfxC01hArgs)
};
#define DFX_ENC_FS(E) (3 * ((E) + 2))                                   /* zero-bordered feat rows of one frame */
#define DFX_ENC_WAVE_FLOATS(C, E) ((E) * ((C) + 4) + DFX_ENC_FS(E) + 2) /* + pad to a multiple of 4 floats below */
#define DFX_ENC_SMEM(C, E) ((size_t)(3 * (C) / 4) * 16 + (size_t)4 * ((DFX_ENC_WAVE_FLOATS(C, E) + 3) / 4 * 4) * 4)

template <int C, bool H3 = false>
__global__ void __launch_bounds__(256, 2) dfx_k_erb_enc(DfxEncArgs A) {
    constexpr int NT = C / 16, CPL = C / 4, LD = C + 4, C4 = C / 4, KC = H3 ? C / 32 : 1;
    DFX_DYN_SMEM(float4, sm4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    const int E = A.E, E1 = E / 2, EP = E + 2;
    float4 *dws = sm4;  // [3][C/4]
    float *s0 = reinterpret_cast<float *>(sm4 + 3 * C4) + (size_t)wave * ((DFX_ENC_WAVE_FLOATS(C, E) + 3) / 4 * 4);  // e0 strip [E][LD]
    float *fs = s0 + E * LD;                                                                                         // [3][E + 2]
    for (int i = tid; i < 3 * C4; i += 256) dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
    float a1[H3 ? 1 : NT][H3 ? 1 : CPL];
    dfx_h8 ahi[NT][KC], alo[NT][KC];
    float4 b1[NT];
    float amax = 0.f;
    if constexpr (H3) dfx_chain_load_w_h3<C>(A.wt_h3, A.bias, lane, ahi, alo, b1);
    else dfx_chain_load_w<C>(A.wt, A.bias, lane, a1, b1);
    // erb_conv0: lane -> (position slot lane / C4, channel quad lane % C4); C4 <= 16 divides 64
    const int c4 = lane % C4, pslot = lane / C4, pstep = 64 / C4;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = reinterpret_cast<const float4 *>(A.w0)[k * C4 + c4];
    const float4 bv = reinterpret_cast<const float4 *>(A.b0)[c4];
    __syncthreads();
    const int64_t Tn = A.t_end - A.t_begin, R = A.B * Tn;  // logical rows: (clip, produced frame)
    const int64_t rstep = (int64_t)gridDim.x * 4;
    // element i of the zero-bordered tap rows [3][E+2] of frame r (zero: border, causal pad after the lookahead shift, beyond T)
    float fx[3];
    auto fetch = [&](int64_t rl) {
        const int64_t b = rl / Tn, t = A.t_begin + (rl - b * Tn);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + 64 * k, kt = i / EP, fp = i - kt * EP;
            const int64_t tau = t - 2 + kt, tin = tau + A.L;
            float v = 0.f;
            if (rl < R && i < 3 * EP && fp >= 1 && fp <= E && tau >= 0 && tin < A.T) v = A.feat[(b * (A.feat_T > 0 ? A.feat_T : A.T) + tin) * E + fp - 1];
            fx[k] = v;
        }
    };
    int64_t rl = (int64_t)blockIdx.x * 4 + wave;
    fetch(rl);
    for (; rl < R; rl += rstep) {
        const int64_t r = (rl / Tn) * A.T + A.t_begin + rl % Tn;  // physical row
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane + 64 * k < 3 * EP) fs[lane + 64 * k] = fx[k];
        fetch(rl + rstep);
        DFX_WAVE_SYNC();
        for (int p = pslot; p < E; p += pstep) {
            float4 acc = bv;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) {
                    const float x = fs[kt * EP + p + kf];  // bin p - 1 + kf, border included
                    const float4 ww = wv[kt * 3 + kf];
                    acc.x += ww.x * x;
                    acc.y += ww.y * x;
                    acc.z += ww.z * x;
                    acc.w += ww.w * x;
                }
            const float4 o = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
            if (A.e0) reinterpret_cast<float4 *>(A.e0 + (r * E + p) * C)[c4] = o;   // (null: the decoder tail recomputes e0 from the features, dfx_k_erb_tail)
            *reinterpret_cast<float4 *>(s0 + p * LD + 4 * c4) = o;
        }
        DFX_WAVE_SYNC();
        auto epi = [&](int p, bool valid, int nt, float4 o) {
            if (valid) *reinterpret_cast<float4 *>(A.e1 + (r * E1 + p) * C + 16 * nt + 4 * q) = o;
        };
        if constexpr (H3) dfx_chain_stage_h3<C, DFX_PW_MODE_DW3>(s0, E, E1, 2, E1, dws, ahi, alo, b1, A.unscale, amax, lane, epi);
        else dfx_chain_stage<C, DFX_PW_MODE_DW3>(s0, E, E1, 2, E1, dws, a1, b1, lane, epi);
        DFX_WAVE_SYNC();  // the strips are rewritten by the next frame
    }
    if (H3 && amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// erb_dec.conv0_out: Conv2d(C -> 1, 1x3) + BN(1) + Sigmoid on  xin = relu(a*e0 + b) + d1   (deepfilternet3.py:241-243,253)
//   m[r, f] = sigmoid(bias + sum_j sum_c w[j][c] * xin[r, f+j-1, c])
// A tile is a whole number of frames (E positions each); xin is staged in LDS, threads (pos, j) form the three per-
// position dot products, then one thread per position combines the neighbours.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_CO_THREADS 256
template <int C>
__global__ void __launch_bounds__(DFX_CO_THREADS) dfx_k_conv_out(const float *x, const float *skip, const float *sk_a,
                                                                 const float *sk_b, const float *w /*[3][C]*/, float bias,
                                                                 float *out, int64_t R, int E, int frames_per_tile, DfxRowMap rm) {
    constexpr int LDA = C + 1;
    DFX_DYN_SMEM(float, sm);
    const int MT = frames_per_tile * E;
    float *As = sm;                 // [MT][LDA]
    float *V = sm + MT * LDA;       // [MT][3]
    float *W = V + MT * 3;          // [3][C]
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * C; i += DFX_CO_THREADS) W[i] = w[i];
    const int64_t ntiles = (R + frames_per_tile - 1) / frames_per_tile;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * frames_per_tile;
        const int64_t npos = ((R - r0) < frames_per_tile ? (R - r0) : frames_per_tile) * E;
        for (int i = tid; i < MT * C; i += DFX_CO_THREADS) {
            const int p = i / C, c = i - p * C;
            float v = 0.f;
            if (p < npos) {
                const int64_t g = (dfx_row(rm, r0 + p / E) * E + p % E) * C + c;
                v = x[g] + fmaxf(sk_a[c] * skip[g] + sk_b[c], 0.f);
            }
            As[p * LDA + c] = v;
        }
        __syncthreads();
        for (int i = tid; i < MT * 3; i += DFX_CO_THREADS) {
            const int p = i / 3, j = i - p * 3;
            float acc = 0.f;
            for (int c = 0; c < C; ++c) acc += W[j * C + c] * As[p * LDA + c];
            V[i] = acc;
        }
        __syncthreads();
        for (int p = tid; p < npos; p += DFX_CO_THREADS) {
            const int f = p % E;
            float acc = bias + V[p * 3 + 1];
            if (f > 0) acc += V[(p - 1) * 3 + 0];
            if (f < E - 1) acc += V[(p + 1) * 3 + 2];
            out[dfx_row(rm, r0 + p / E) * E + f] = dfx_sigmoid(acc);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ERB decoder tail, fused: erb_dec.convt1 (transposed 1x3, stride 2, with the conv1p pathway) -> erb_dec.conv0_out (with the conv0p
// pathway) + sigmoid   (deepfilternet3.py:252-253).  d1 [B*T, E, C], the widest decoder activation, never reaches HBM: a wave owns
// one frame, the convt1 tiles (dfx_k_pwconv's DWT3 + SKIP body, operands from HBM) put  xin = d1 + relu(a0*e0 + b0)  into an LDS
// strip [E][C + 4], and the 3-tap C -> 1 conv reads it back exactly like dfx_k_conv_out (same per-(position, tap) dot products, same
// combination order -> the same bits).  HBM per frame: reads d2, e1 (E/2 positions) and e0, writes E mask values.
// ---------------------------------------------------------------------------------------------------------------------
struct DfxDec10Args {
    const float *x;      // d2 [R, E/2, C]
    const float *skip1;  // e1 [R, E/2, C]
    const float *sk1_a, *sk1_b;  // conv1p pathway [C]
    const float *dw;     // convt1 depthwise [3][C]
    const float *wt;     // convt1 pointwise [C][C] (BN-scaled)
    const float *bias;   // [C]
    const float *skip0;  // e0 [R, E, C]
    const float *sk0_a, *sk0_b;  // conv0p pathway [C]
    const float *wo;     // conv0_out [3][C]
    float bias_o;
    float *out;          // mask [R, E]
    int64_t R;           // logical rows of this launch
    int E;
    DfxRowMap rm;
};
#define DFX_DEC10_WAVE_FLOATS(C, E) ((E) * ((C) + 4) + 4 * (E))
#define DFX_DEC10_SMEM(C, E) ((size_t)(3 * (C) / 4 + 4 * (C) / 4 + 3 * (C) / 4) * 16 + (size_t)4 * DFX_DEC10_WAVE_FLOATS(C, E) * 4)

template <int C>
__global__ void __launch_bounds__(256, 2) dfx_k_erb_dec10(DfxDec10Args A) {
    constexpr int NT = C / 16, CPL = C / 4, V4 = CPL / 4, LD = C + 4, C4 = C / 4;
    DFX_DYN_SMEM(float4, sm4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int E = A.E, E1 = E / 2;
    float4 *dws = sm4;            // [3][C/4]
    float4 *sks = dws + 3 * C4;   // [4][C/4]: a1, b1, a0, b0
    float4 *wos = sks + 4 * C4;   // [3][C/4]
    float *strip = reinterpret_cast<float *>(wos + 3 * C4) + (size_t)wave * DFX_DEC10_WAVE_FLOATS(C, E);  // xin [E][LD]
    float *V = strip + E * LD;                                                                            // [E][3] (+ pad)
    for (int i = tid; i < 3 * C4; i += 256) {
        dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
        wos[i] = reinterpret_cast<const float4 *>(A.wo)[i];
    }
    for (int i = tid; i < C4; i += 256) {
        sks[i] = reinterpret_cast<const float4 *>(A.sk1_a)[i];
        sks[C4 + i] = reinterpret_cast<const float4 *>(A.sk1_b)[i];
        sks[2 * C4 + i] = reinterpret_cast<const float4 *>(A.sk0_a)[i];
        sks[3 * C4 + i] = reinterpret_cast<const float4 *>(A.sk0_b)[i];
    }
    float areg[NT][CPL];
    float4 biasr[NT];
    dfx_chain_load_w<C>(A.wt, A.bias, lane, areg, biasr);
    __syncthreads();
    for (int64_t rl = (int64_t)blockIdx.x * 4 + wave; rl < A.R; rl += (int64_t)gridDim.x * 4) {
        const int64_t r = dfx_row(A.rm, rl);
        for (int p0 = 0; p0 < E; p0 += 16) {
            const int fo = p0 + jl;
            const bool valid = fo < E;
            // conv0p pathway operand of this lane's outputs, fetched ahead of the matrix ops
            float4 s0v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                s0v[nt] = valid ? *reinterpret_cast<const float4 *>(A.skip0 + (r * E + fo) * C + 16 * nt + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            float u[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {  // transposed: fo = 2*fi - 1 + j
                const int num = fo + 1 - j, fi = num >> 1;
                if (valid && num >= 0 && (num & 1) == 0 && fi < E1) {
                    const int64_t off = (r * E1 + fi) * C + CPL * q;
                    const float4 *xp = reinterpret_cast<const float4 *>(A.x + off);
                    const float4 *sp = reinterpret_cast<const float4 *>(A.skip1 + off);
#pragma unroll
                    for (int v = 0; v < V4; ++v) {
                        float4 xv = xp[v];
                        const float4 sv = sp[v], a = sks[V4 * q + v], bb = sks[C4 + V4 * q + v];
                        xv.x += fmaxf(a.x * sv.x + bb.x, 0.f);
                        xv.y += fmaxf(a.y * sv.y + bb.y, 0.f);
                        xv.z += fmaxf(a.z * sv.z + bb.z, 0.f);
                        xv.w += fmaxf(a.w * sv.w + bb.w, 0.f);
                        const float4 w = dws[j * C4 + V4 * q + v];
                        u[4 * v + 0] += w.x * xv.x;
                        u[4 * v + 1] += w.y * xv.y;
                        u[4 * v + 2] += w.z * xv.z;
                        u[4 * v + 3] += w.w * xv.w;
                    }
                }
            }
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < CPL; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[nt][ks], u[ks], acc[nt], 0, 0, 0);
            if (valid) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 a = sks[2 * C4 + 4 * nt + q], bb = sks[3 * C4 + 4 * nt + q];
                    float4 o;  // xin = d1 + relu(a0 * e0 + b0), d1 = relu(acc + bias)
                    o.x = fmaxf(acc[nt][0] + biasr[nt].x, 0.f) + fmaxf(a.x * s0v[nt].x + bb.x, 0.f);
                    o.y = fmaxf(acc[nt][1] + biasr[nt].y, 0.f) + fmaxf(a.y * s0v[nt].y + bb.y, 0.f);
                    o.z = fmaxf(acc[nt][2] + biasr[nt].z, 0.f) + fmaxf(a.z * s0v[nt].z + bb.z, 0.f);
                    o.w = fmaxf(acc[nt][3] + biasr[nt].w, 0.f) + fmaxf(a.w * s0v[nt].w + bb.w, 0.f);
                    *reinterpret_cast<float4 *>(strip + fo * LD + 16 * nt + 4 * q) = o;
                }
            }
        }
        DFX_WAVE_SYNC();
        for (int i = lane; i < 3 * E; i += 64) {  // V[p][j] = sum_c wo[j][c] * xin[p][c]
            const int p = i / 3, j = i - 3 * p;
            const float4 *xr = reinterpret_cast<const float4 *>(strip + p * LD);
            float acc = 0.f;
#pragma unroll 4
            for (int c = 0; c < C4; ++c) {
                const float4 x = xr[c], w = wos[j * C4 + c];
                acc += w.x * x.x;
                acc += w.y * x.y;
                acc += w.z * x.z;
                acc += w.w * x.w;
            }
            V[i] = acc;
        }
        DFX_WAVE_SYNC();
        for (int f = lane; f < E; f += 64) {
            float acc = A.bias_o + V[f * 3 + 1];
            if (f > 0) acc += V[(f - 1) * 3 + 0];
            if (f < E - 1) acc += V[(f + 1) * 3 + 2];
            A.out[r * E + f] = dfx_sigmoid(acc);
        }
        DFX_WAVE_SYNC();  // strip and V are rewritten by the next frame
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_erb_dec10_f: the same fused decoder tail with every HBM operand streamed as whole contiguous frames (see dfx_k_pwconv_f):
// d2 + the conv1p pathway of e1 -> in-strip [E/2][C+4]; the conv0p pathway of e0 -> xin strip [E][C+4]; dfx_chain_stage (fp32 or
// fp16-split) adds d1 = relu(convt1) into the xin strip; the C -> 1 conv reads it back as before.  The three operand runs of the NEXT
// frame are requested before the current frame's matrix work.  Needs E/2 * C/4 <= 256 and E * C/4 <= 512 float4s per frame.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_DEC10F_WAVE_FLOATS(C, E) (((E) / 2 + (E)) * ((C) + 4) + 4 * (E))
#define DFX_DEC10F_SMEM(C, E) ((size_t)(3 * (C) / 4 + 4 * (C) / 4 + 3 * (C) / 4) * 16 + (size_t)4 * DFX_DEC10F_WAVE_FLOATS(C, E) * 4)
static __host__ __device__ __forceinline__ bool dfx_dec10f_ok(int C, int E) {
    return E % 2 == 0 && (E / 2) * (C / 4) <= 64 * 4 && E * (C / 4) <= 64 * 8 && DFX_DEC10F_SMEM(C, E) <= 64 * 1024;
}
struct DfxDec10fArgs {
    DfxDec10Args a;
    const dfx_h8 *wt_h3 = nullptr;
    float unscale = 1.f;
    unsigned int *err = nullptr;
};

template <int C, bool H3>
__global__ void __launch_bounds__(256, 2) dfx_k_erb_dec10_f(DfxDec10fArgs AA) {
    const DfxDec10Args &A = AA.a;
    constexpr int NT = C / 16, CPL = C / 4, LD = C + 4, C4 = C / 4, KC = H3 ? C / 32 : 1;
    DFX_DYN_SMEM(float4, sm4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    const int E = A.E, E1 = E / 2;
    float4 *dws = sm4;            // [3][C/4]
    float4 *sks = dws + 3 * C4;   // [4][C/4]: a1, b1, a0, b0
    float4 *wos = sks + 4 * C4;   // [3][C/4]
    float *sin = reinterpret_cast<float *>(wos + 3 * C4) + (size_t)wave * DFX_DEC10F_WAVE_FLOATS(C, E);  // convt1 input [E/2][LD]
    float *strip = sin + E1 * LD;                                                                         // xin [E][LD]
    float *V = strip + E * LD;                                                                            // [E][3] (+ pad)
    for (int i = tid; i < 3 * C4; i += 256) {
        dws[i] = reinterpret_cast<const float4 *>(A.dw)[i];
        wos[i] = reinterpret_cast<const float4 *>(A.wo)[i];
    }
    for (int i = tid; i < C4; i += 256) {
        sks[i] = reinterpret_cast<const float4 *>(A.sk1_a)[i];
        sks[C4 + i] = reinterpret_cast<const float4 *>(A.sk1_b)[i];
        sks[2 * C4 + i] = reinterpret_cast<const float4 *>(A.sk0_a)[i];
        sks[3 * C4 + i] = reinterpret_cast<const float4 *>(A.sk0_b)[i];
    }
    float areg[H3 ? 1 : NT][H3 ? 1 : CPL];
    dfx_h8 ahi[NT][KC], alo[NT][KC];
    float4 biasr[NT];
    float amax = 0.f;
    if constexpr (H3) dfx_chain_load_w_h3<C>(AA.wt_h3, A.bias, lane, ahi, alo, biasr);
    else dfx_chain_load_w<C>(A.wt, A.bias, lane, areg, biasr);
    __syncthreads();
    const int n1 = E1 * C4, n0 = E * C4;   // float4s per frame of d2 / e1 and of e0
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x), *s14 = reinterpret_cast<const float4 *>(A.skip1),
                 *s04 = reinterpret_cast<const float4 *>(A.skip0);
    float4 xr[4], s1r[4], s0r[8];
    auto issue = [&](int64_t rl) {
        const int64_t r = dfx_row(A.rm, rl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            xr[i] = s1r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < n1) xr[i] = x4[r * n1 + idx], s1r[i] = s14[r * n1 + idx];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + 64 * i;
            s0r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < n0) s0r[i] = s04[r * n0 + idx];
        }
    };
    int64_t rl = (int64_t)blockIdx.x * 4 + wave;
    if (rl < A.R) issue(rl);
    static_assert(64 % C4 == 0, "channel quad of a lane must not depend on the load index");
    for (; rl < A.R; rl += (int64_t)gridDim.x * 4) {
        const int64_t r = dfx_row(A.rm, rl);
        // scale / shift of the two pathways for this lane's channel quad: read once per frame, before the first strip store (see
        // dfx_k_pwconv_f; held across the frame loop they would not fit the 256 registers of two waves per SIMD)
        int lq = lane % C4;
        DFX_OPAQUE(lq);
        const float4 sk1a = sks[lq], sk1b = sks[C4 + lq], sk0a = sks[2 * C4 + lq], sk0b = sks[3 * C4 + lq];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            if (idx < n1) {
                const int row = idx / C4, c4 = idx - row * C4;
                float4 v = xr[i];
                const float4 sv = s1r[i], a = sk1a, bb = sk1b;
                v.x += fmaxf(a.x * sv.x + bb.x, 0.f);
                v.y += fmaxf(a.y * sv.y + bb.y, 0.f);
                v.z += fmaxf(a.z * sv.z + bb.z, 0.f);
                v.w += fmaxf(a.w * sv.w + bb.w, 0.f);
                *reinterpret_cast<float4 *>(sin + row * LD + 4 * c4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + 64 * i;
            if (idx < n0) {
                const int row = idx / C4, c4 = idx - row * C4;
                const float4 sv = s0r[i], a = sk0a, bb = sk0b;
                *reinterpret_cast<float4 *>(strip + row * LD + 4 * c4) =
                    make_float4(fmaxf(a.x * sv.x + bb.x, 0.f), fmaxf(a.y * sv.y + bb.y, 0.f), fmaxf(a.z * sv.z + bb.z, 0.f), fmaxf(a.w * sv.w + bb.w, 0.f));
            }
        }
        const int64_t next = rl + (int64_t)gridDim.x * 4;
        if (next < A.R) issue(next);
        DFX_WAVE_SYNC();
        auto epi = [&](int p, bool valid, int nt, float4 d1) {   // xin = d1 + relu(a0 * e0 + b0)
            if (valid) {
                float4 *dst = reinterpret_cast<float4 *>(strip + p * LD + 16 * nt + 4 * q);
                const float4 e = *dst;
                *dst = make_float4(d1.x + e.x, d1.y + e.y, d1.z + e.z, d1.w + e.w);
            }
        };
        if constexpr (H3) dfx_chain_stage_h3<C, DFX_PW_MODE_DWT3>(sin, E1, E, 2, E, dws, ahi, alo, biasr, AA.unscale, amax, lane, epi);
        else dfx_chain_stage<C, DFX_PW_MODE_DWT3>(sin, E1, E, 2, E, dws, areg, biasr, lane, epi);
        DFX_WAVE_SYNC();
        for (int i = lane; i < 3 * E; i += 64) {  // V[p][j] = sum_c wo[j][c] * xin[p][c]
            const int p = i / 3, j = i - 3 * p;
            const float4 *xrow = reinterpret_cast<const float4 *>(strip + p * LD);
            float acc = 0.f;
#pragma unroll 4
            for (int c = 0; c < C4; ++c) {
                const float4 x = xrow[c], w = wos[j * C4 + c];
                acc += w.x * x.x;
                acc += w.y * x.y;
                acc += w.z * x.z;
                acc += w.w * x.w;
            }
            V[i] = acc;
        }
        DFX_WAVE_SYNC();
        for (int f = lane; f < E; f += 64) {
            float acc = A.bias_o + V[f * 3 + 1];
            if (f > 0) acc += V[(f - 1) * 3 + 0];
            if (f < E - 1) acc += V[(f + 1) * 3 + 2];
            A.out[r * E + f] = dfx_sigmoid(acc);
        }
        DFX_WAVE_SYNC();  // the strips and V are rewritten by the next frame
    }
    if (H3 && amax >= DFX_H3_LIMIT && AA.err) dfx_raise(AA.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_erb_tail: the whole convolutional half of the ERB decoder for one frame in one kernel (deepfilternet3.py:250-253):
//     d3 = convt3(demb + conv3p(e3))   [E/4] -> [E/4]        d2 = convt2(d3 + conv2p(e2))   [E/4] -> [E/2]
//     d1 = convt1(d2 + conv1p(e1))     [E/2] -> [E]          mask = sigmoid(conv0_out(d1 + conv0p(e0)))
// As three launches (dfx_k_pwconv_f x 2, dfx_k_erb_dec10_f) d3 and d2 made a round trip through HBM each — 12 KB per frame written
// and read back BESIDE the GRU chain, where every byte costs the chain time; here they live in LDS strips.
// A wave owns a frame (E = 32: 8 / 8 / 16 / 32 positions).  Strips [16 rows][C + 4], wave-private (wave-level synchronisation only):
//     X: rows [0, 8) = convt3's input, rows [8, 16) = convt2's input (conv2p(e2), d3 added by convt3's epilogue); during convt1 it holds
//        one 16-position tile of conv0_out's input at a time (conv0p(e0) stored, d1 added by the epilogue, the C -> 1 taps taken, next tile)
//     Y: convt1's input (conv1p(e1), d2 added by convt2's epilogue)
// The pad columns of the strips' rows carry the per-position partial sums of conv0_out across the two tiles.
// What bounds these per-frame chains is latency (LDS -> taps -> split -> matrix ops -> LDS, four times per frame), i.e. waves per SIMD, and
// what bounds the waves is LDS: the first version (strips for a whole frame, 8 waves per CU, a layer's fragments held in 64 registers)
// ran at 2.19 ms for the 1.5 ms of the three launches it replaced.  Here the strips are 8.5 KB per wave and the three layers' fp16-split
// fragments (48 KB at C = 64) are read from LDS one k-chunk at a time (32 registers), so that a workgroup of 12 waves = 3 per SIMD fits
// one CU (156 KB of LDS, <= 168 registers).
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_TAIL_WAVES 12
#define DFX_TAIL_WAVE_FLOATS(C) (32 * ((C) + 4))
#define DFX_TAIL_TAB4(C) (3 * 3 * (C) / 4 + 4 * 2 * (C) / 4 + 3 * (C) / 4)   /* float4s: dw x3, pathway a/b x4, bias x3 */
#define DFX_TAIL_WFRAG(C) ((size_t)((C) / 16) * ((C) / 32) * 2 * 64)                     /* dfx_h8 per layer */
#define DFX_TAIL_W0_4(C) (((C) / 16) * 2 * 32 + ((C) / 32) * 2 * 16)   /* dfx_h8 (16 bytes each): erb_conv0's fragments [C/16][hi, lo], the 32 lanes that carry k-slots < 16 (e0 recomputed in the kernel); conv0_out's [C/32][hi, lo], the lanes of rows 0..3 */
#define DFX_TAIL_SMEM(C) ((size_t)(DFX_TAIL_TAB4(C) + DFX_TAIL_W0_4(C)) * 16 + 3 * DFX_TAIL_WFRAG(C) * 16 + (size_t)DFX_TAIL_WAVES * DFX_TAIL_WAVE_FLOATS(C) * 4)
static __host__ __device__ __forceinline__ bool dfx_tail_ok(int C, int E) {
    return (C == 32 || C == 64) && E == 32 && DFX_TAIL_SMEM(C) <= (size_t)160 * 1024;
}
struct DfxTailArgs {
    const float *demb, *e3, *e2, *e1, *e0;   // [R, E/4, C] x3, [R, E/2, C], [R, E, C]
    const float *dw[3], *bias[3];            // ct3, ct2, ct1: depthwise [3][C], BN shift [C]
    const dfx_h8 *wh3[3];                    // their pointwise fragments (pack_pw_h3)
    float unscale[3];
    const float *ska[4], *skb[4];            // pathway scale / shift of conv3p, conv2p, conv1p, conv0p
    const float *wo;                         // conv0_out [3][C]
    const dfx_h8 *woh3;                      // the same as fp16-split fragments [C/32][hi, lo][64] (rows 0..2 of a 16-row tile)
    const dfx_h8 *w0h3;                      // erb_conv0 (+ bias) as fragments [C/16][hi, lo][64] (e0 recomputed)
    float unscale_wo, unscale_w0;
    float bias_o;
    float *out;                              // mask [R, E]
    int64_t R;
    int E;
    DfxRowMap rm;
    unsigned int *err;
    // e0 == null: e0 = relu(erb_conv0(feat_erb)) is recomputed per frame from the three feature rows it depends on (384 bytes instead of
    // E * C * 4 = 8 KB per frame read beside the GRU chain — and the encoder does not write it): deepfilternet3.py:106,168, the
    // arithmetic of dfx_k_erb_enc in the same order
    const float *feat = nullptr;             // [B, feat_T or T, E]
    const float *w0 = nullptr, *b0 = nullptr;   // erb_conv0 folded [3][3][C], [C]
    int64_t T = 0, feat_T = 0;               // frames per clip of the row space / of feat (0: T)
    int L = 0;                               // conv lookahead
};
// one 16-position tile of a separable stage on the fp16-split path, fragments and bias read from LDS (dfx_chain_stage_h3's tile body: same
// operand roles, same k order, same bits); positions [p0, p0 + 16) of npos, input rows in `in`
// FB: channel tiles whose fragments are read per batch (NT: one k-chunk at a time, 32 registers at C = 64; NT / 2: half of that, for a caller
// whose epilogue needs the registers)
template <int C, int MODE, int FB = C / 16, typename Epi>
static __device__ __forceinline__ void dfx_chain_tile_h3_lds(const float *in, int Fin, int stride, int p0, int npos, const float4 *dws,
                                                             const dfx_h8 *wfr, const float4 *bias4, float unscale, float &amax, int lane, Epi &&epi) {
    constexpr int CPL = C / 4, NT = C / 16, KC = C / 32, V4 = CPL / 4, LD = C + 4;
    const int q = lane >> 4, jl = lane & 15;
    const int fo = p0 + jl;
    const bool valid = fo < npos;
    float u[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) u[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int fi;
        bool ok;
        if (MODE == DFX_PW_MODE_DW3) {
            fi = fo * stride + j - 1;
            ok = fi >= 0 && fi < Fin;
        } else {  // transposed: fo = 2*fi - 1 + j
            const int num = fo + 1 - j;
            fi = num >> 1;
            ok = num >= 0 && (num & 1) == 0 && fi < Fin;
        }
        if (valid && ok) {
            const float4 *xp = reinterpret_cast<const float4 *>(in + fi * LD + CPL * q);
#pragma unroll
            for (int v = 0; v < V4; ++v) {
                const float4 xv = xp[v];
                const float4 w = dws[j * (C / 4) + V4 * q + v];
                u[4 * v + 0] += w.x * xv.x;
                u[4 * v + 1] += w.y * xv.y;
                u[4 * v + 2] += w.z * xv.z;
                u[4 * v + 3] += w.w * xv.w;
            }
        }
    }
    dfx_h8 bhi[KC], blo[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) dfx_split8_g(u + 8 * kc, bhi[kc], blo[kc], amax);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_assert(NT % FB == 0, "whole batches of channel tiles");
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += FB) {
            dfx_h8 ahi[FB], alo[FB];   // one batch of a k-chunk's fragments at a time (per accumulator the same order of terms for any FB: same bits)
#pragma unroll
            for (int d = 0; d < FB; ++d) {
                ahi[d] = wfr[((size_t)((n0 + d) * KC + kc) * 2 + 0) * 64 + lane];
                alo[d] = wfr[((size_t)((n0 + d) * KC + kc) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int d = 0; d < FB; ++d) {
                acc[n0 + d] = dfx_mfma_16x16x32_f16(alo[d], bhi[kc], acc[n0 + d]);
                acc[n0 + d] = dfx_mfma_16x16x32_f16(ahi[d], blo[kc], acc[n0 + d]);
                acc[n0 + d] = dfx_mfma_16x16x32_f16(ahi[d], bhi[kc], acc[n0 + d]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4 bz = bias4[4 * nt + q];
        epi(fo, valid, nt,
            make_float4(fmaxf(acc[nt][0] * unscale + bz.x, 0.f), fmaxf(acc[nt][1] * unscale + bz.y, 0.f), fmaxf(acc[nt][2] * unscale + bz.z, 0.f),
                        fmaxf(acc[nt][3] * unscale + bz.w, 0.f)));
    }
}
template <int C, bool RE0 = false>
__global__ void __launch_bounds__(64 * DFX_TAIL_WAVES, 1) dfx_k_erb_tail(DfxTailArgs A) {
    constexpr int LD = C + 4, C4 = C / 4, NTH = 64 * DFX_TAIL_WAVES;
    constexpr int E = 32, E1 = 16, E4 = 8;
    constexpr int N3 = E4 * C4, N1 = E1 * C4, N0T = 16 * C4;   // float4s per frame of demb / e3 / e2, of e1, and per 16-position tile of e0
    constexpr int NV3 = (N3 + 63) / 64, NV1 = (N1 + 63) / 64, NV0 = (N0T + 63) / 64;
    DFX_DYN_SMEM(float4, sm4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    constexpr int NT = C / 16, KC = C / 32;
    float4 *dws = sm4;                  // [3 layers][3][C4]
    float4 *sks = dws + 9 * C4;         // [4 pathways][a, b][C4]
    float4 *bis = sks + 8 * C4;         // [3 layers][C4]
    dfx_h8 *w0f = reinterpret_cast<dfx_h8 *>(bis + 3 * C4);   // [NT][hi, lo][32]: erb_conv0's fragments, the lanes of k-slots 0..15 (e0 recomputed)
    dfx_h8 *wof = w0f + NT * 2 * 32;                           // [KC][hi, lo][4 q][4 rows]: conv0_out's fragments, rows 0..3 (only rows 0..2 of the tile are read back)
    dfx_h8 *wfr = w0f + DFX_TAIL_W0_4(C);                      // [3 layers][NT * KC * 2 * 64]
    float *X = reinterpret_cast<float *>(wfr + 3 * DFX_TAIL_WFRAG(C)) + (size_t)wave * DFX_TAIL_WAVE_FLOATS(C);
    float *Y = X + 16 * LD;
#pragma unroll
    for (int l = 0; l < 3; ++l) {   // (constant indices into the argument arrays: a run-time index would move the struct to scratch)
        for (int i = tid; i < 3 * C4; i += NTH) dws[l * 3 * C4 + i] = reinterpret_cast<const float4 *>(A.dw[l])[i];
        for (int i = tid; i < C4; i += NTH) bis[l * C4 + i] = reinterpret_cast<const float4 *>(A.bias[l])[i];
        for (int i = tid; i < (int)DFX_TAIL_WFRAG(C); i += NTH) wfr[l * DFX_TAIL_WFRAG(C) + i] = A.wh3[l][i];
    }
#pragma unroll
    for (int pth = 0; pth < 4; ++pth)
        for (int i = tid; i < C4; i += NTH) {
            sks[pth * 2 * C4 + i] = reinterpret_cast<const float4 *>(A.ska[pth])[i];
            sks[pth * 2 * C4 + C4 + i] = reinterpret_cast<const float4 *>(A.skb[pth])[i];
        }
    constexpr bool re0 = RE0;   // (a template parameter: the e0 loads and their registers do not exist in this form)
    if (re0) {
        for (int i = tid; i < NT * 2 * 32; i += NTH) w0f[i] = A.w0h3[(i >> 5) * 64 + (i & 31)];
    }
    for (int i = tid; i < KC * 2 * 16; i += NTH) wof[i] = A.woh3[(i >> 4) * 64 + 16 * ((i & 15) >> 2) + (i & 3)];
    __syncthreads();
    const float4 *pd = reinterpret_cast<const float4 *>(A.demb), *p3 = reinterpret_cast<const float4 *>(A.e3), *p2 = reinterpret_cast<const float4 *>(A.e2),
                 *p1 = reinterpret_cast<const float4 *>(A.e1), *p0 = reinterpret_cast<const float4 *>(A.e0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rd[NV3], r3[NV3], r2[NV3];
    // demb / e3 / e2 of a frame are requested one frame ahead; e1 and e0 inside the frame, each one stage before it is needed
    auto issue = [&](int64_t r) {
#pragma unroll
        for (int i = 0; i < NV3; ++i) {
            const int idx = lane + 64 * i;
            rd[i] = r3[i] = r2[i] = z4;
            if (idx < N3) rd[i] = pd[r * N3 + idx], r3[i] = p3[r * N3 + idx], r2[i] = p2[r * N3 + idx];
        }
    };
    auto path = [](float4 sv, float4 a, float4 b) {
        return make_float4(fmaxf(a.x * sv.x + b.x, 0.f), fmaxf(a.y * sv.y + b.y, 0.f), fmaxf(a.z * sv.z + b.z, 0.f), fmaxf(a.w * sv.w + b.w, 0.f));
    };
    auto add_into = [&](float *strip, int prow0) {
        return [=](int p, bool valid, int nt, float4 d) {
            if (valid) {
                float4 *dst = reinterpret_cast<float4 *>(strip + (p - prow0) * LD + 16 * nt + 4 * q);
                const float4 e = *dst;
                *dst = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
            }
        };
    };
    static_assert(64 % C4 == 0, "channel quad of a lane must not depend on the load index");
    float amax = 0.f;
    // (the frame index is wave-uniform: said so, its row-map divisions run on the scalar unit and their reciprocals live in scalar registers — as
    // vector values hoisted out of the frame loop they were the three registers this kernel does not have: 12 bytes of scratch per lane once the
    // library was built without packed fp32 operations)
    int64_t rl = (int64_t)blockIdx.x * DFX_TAIL_WAVES + dfx_wave_uniform(wave);
    if (rl < A.R) issue(dfx_row(A.rm, rl));
    for (; rl < A.R; rl += (int64_t)gridDim.x * DFX_TAIL_WAVES) {
        const int64_t r = dfx_row(A.rm, rl);
        int lq = lane % C4;
        DFX_OPAQUE(lq);
        float4 r1[NV1];
        int l1 = lane;
        DFX_OPAQUE(l1);   // (the lane's share of the address per frame, not a 64-bit lane pointer held — spilled — across the frame loop)
#pragma unroll
        for (int i = 0; i < NV1; ++i) {
            const int idx = l1 + 64 * i;
            r1[i] = idx < N1 ? p1[r * N1 + idx] : z4;
        }
        // e0 recomputed: lane fp of fxr[kt] = element fp of the zero-bordered tap row kt of this frame, [E + 2] (dfx_k_erb_enc's `fs`: zero =
        // border, causal pad after the lookahead shift, beyond T); requested here, used after convt2
        float fxr[3] = {0.f, 0.f, 0.f};
        if (re0) {
            const uint32_t cb = (uint32_t)r / (uint32_t)A.T;
            const int64_t ct = r - (int64_t)cb * A.T, fT = A.feat_T > 0 ? A.feat_T : A.T;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const int64_t tau = ct - 2 + kt, tin = tau + A.L;
                if (lane >= 1 && lane <= E && tau >= 0 && tin < A.T) fxr[kt] = A.feat[((int64_t)cb * fT + tin) * E + lane - 1];
            }
        }
        {   // convt3's input and conv2p(e2) into X
            const float4 a3 = sks[lq], b3 = sks[C4 + lq], a2 = sks[2 * C4 + lq], b2 = sks[3 * C4 + lq];
#pragma unroll
            for (int i = 0; i < NV3; ++i) {
                const int idx = lane + 64 * i;
                if (idx < N3) {
                    const int row = idx / C4, c4 = idx - row * C4;
                    const float4 pv = path(r3[i], a3, b3);
                    *reinterpret_cast<float4 *>(X + row * LD + 4 * c4) = make_float4(rd[i].x + pv.x, rd[i].y + pv.y, rd[i].z + pv.z, rd[i].w + pv.w);
                    *reinterpret_cast<float4 *>(X + (E4 + row) * LD + 4 * c4) = path(r2[i], a2, b2);
                }
            }
        }
        DFX_WAVE_SYNC();
        // ---- convt3: X[0, 8) -> += into X[8, 16)
        dfx_chain_tile_h3_lds<C, DFX_PW_MODE_DW3>(X, E4, 1, 0, E4, dws, wfr, bis, A.unscale[0], amax, lane, add_into(X + E4 * LD, 0));
        {   // conv1p(e1) into Y
            const float4 a1 = sks[4 * C4 + lq], b1 = sks[5 * C4 + lq];
#pragma unroll
            for (int i = 0; i < NV1; ++i) {
                const int idx = lane + 64 * i;
                if (idx < N1) {
                    const int row = idx / C4, c4 = idx - row * C4;
                    *reinterpret_cast<float4 *>(Y + row * LD + 4 * c4) = path(r1[i], a1, b1);
                }
            }
        }
        DFX_WAVE_SYNC();
        // ---- convt2: X[8, 16) -> += into Y[0, 16)
        dfx_chain_tile_h3_lds<C, DFX_PW_MODE_DWT3>(X + E4 * LD, E4, 2, 0, E1, dws + 3 * C4, wfr + DFX_TAIL_WFRAG(C), bis + C4, A.unscale[1], amax, lane,
                                                   add_into(Y, 0));
        DFX_WAVE_SYNC();
        // ---- convt1 + conv0_out, one 16-position tile at a time through X
        const float4 a0 = sks[6 * C4 + lq], b0 = sks[7 * C4 + lq];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (re0) {
                // e0 of the tile's 16 positions on the matrix pipe (round 5; on the VALU this stage was 72 ds_bpermute + 72 weight reads per frame, a
                // third of the LDS time of an LDS-bound kernel): D[channel][position] = W0[channel][k] P[k][position], k-slot 8 q + i = tap
                // 3 kt + kf for k < 9, a constant 1 against the bias at k = 9.  Lane (position jl, q = 0) gathers its eight taps from the three
                // feature rows (lane fp of fxr[kt] = element fp of the zero-bordered row), q = 1 the ninth; the fragments' lanes 32..63 multiply
                // zeros and re-read lanes 0..31.
                int jl = lane & 15;
                DFX_OPAQUE(jl);   // (addresses / shuffle indices recomputed per tile instead of held across the frame loop)
                float xk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xk[i] = __shfl(fxr[i / 3], 16 * t + jl + i % 3);
                const float x8 = __shfl(fxr[2], 16 * t + jl + 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) xk[i] = q == 0 ? xk[i] : 0.f;
                if (q == 1) xk[0] = x8, xk[1] = 1.f;
                dfx_h8 ph, pl;
                dfx_split8_g(xk, ph, pl, amax);
                int l32 = lane & 31;
                DFX_OPAQUE(l32);
                // (term-major over the NT accumulators: dependent matrix ops NT issues apart; a guard before the epilogue reads them — DFX_MFMA_GUARD)
                // (the hi fragments are read once per term: the kernel has LDS reads to spare here, registers it has not)
                f32x4 acc0[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc0[nt] = dfx_mfma_16x16x32_f16(w0f[(nt * 2 + 1) * 32 + l32], ph, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc0[nt] = dfx_mfma_16x16x32_f16(w0f[(nt * 2 + 0) * 32 + l32], pl, acc0[nt]);
                DFX_OPAQUE(l32);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc0[nt] = dfx_mfma_16x16x32_f16(w0f[(nt * 2 + 0) * 32 + l32], ph, acc0[nt]);
                DFX_MFMA_GUARD();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {   // lane (position jl, q): channels 16 nt + 4 q + r = channel quad 4 nt + q
                    const float4 a0q = sks[6 * C4 + 4 * nt + q], b0q = sks[7 * C4 + 4 * nt + q];
                    const float4 ev = make_float4(fmaxf(acc0[nt][0] * A.unscale_w0, 0.f), fmaxf(acc0[nt][1] * A.unscale_w0, 0.f),
                                                  fmaxf(acc0[nt][2] * A.unscale_w0, 0.f), fmaxf(acc0[nt][3] * A.unscale_w0, 0.f));
                    *reinterpret_cast<float4 *>(X + jl * LD + 16 * nt + 4 * q) = path(ev, a0q, b0q);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV0; ++i) {
                    const int idx = lane + 64 * i;
                    if (idx < N0T) {
                        const int row = idx / C4, c4 = idx - row * C4;
                        // (e0 from HBM — DFX_E0_RECOMPUTE=0 or an encoder that is not fused — is read where it is used: requested a stage
                        // ahead, its 16 registers were 48 bytes of scratch per lane at this kernel's three waves per SIMD)
                        *reinterpret_cast<float4 *>(X + row * LD + 4 * c4) = path(p0[(r * 2 + t) * N0T + idx], a0, b0);
                    }
                }
            }
            if (t == 1) {   // the next frame's first operands: in flight during this frame's last tile
                const int64_t next = rl + (int64_t)gridDim.x * DFX_TAIL_WAVES;
                if (next < A.R) issue(dfx_row(A.rm, next));
            }
            DFX_WAVE_SYNC();
            // convt1's tile; its epilogue adds the pathway term parked in X and hands the sums — conv0_out's operand — to the matrix pipe two channel
            // tiles (= one k-chunk) at a time: element 4 (nt & 1) + r of sv = channel 16 nt + 4 q + r of this lane's position, the k order conv0_out's
            // fragments were packed in.  V[j][p] = sum_c wo[j][c] * xin[p][c] for the tile's 16 positions (round 5; on the VALU: 32 row / weight reads
            // of 16 bytes per tile by 48 lanes): rows j = 0..2 of one tile, the lanes q = 0 hold them.  The fragments' rows 3..15 multiply into rows of
            // the result nobody reads: those lanes re-read row 3 (1 KB of LDS instead of 4).
            float sv[8];
            f32x4 va = f32x4{0.f, 0.f, 0.f, 0.f}, vb = va, vc = va;   // one chain per product term (dependent ops three issues apart)
            int lw = 4 * q + ((lane & 15) < 3 ? (lane & 15) : 3);
            DFX_OPAQUE(lw);
            dfx_chain_tile_h3_lds<C, DFX_PW_MODE_DWT3, (C >= 64 ? C / 32 : C / 16)>(Y, E1, 2, 16 * t, E, dws + 6 * C4, wfr + 2 * DFX_TAIL_WFRAG(C), bis + 2 * C4, A.unscale[2], amax, lane,
                                                       [&](int p, bool, int nt, float4 d) {
                                                           const float4 e = *reinterpret_cast<const float4 *>(X + (p - 16 * t) * LD + 16 * nt + 4 * q);
                                                           const int o = 4 * (nt & 1);
                                                           sv[o + 0] = d.x + e.x, sv[o + 1] = d.y + e.y, sv[o + 2] = d.z + e.z, sv[o + 3] = d.w + e.w;
                                                           if (nt & 1) {
                                                               const int kc = nt >> 1;
                                                               dfx_h8 sh, sl;
                                                               dfx_split8_g(sv, sh, sl, amax);
                                                               const dfx_h8 woh = wof[(kc * 2 + 0) * 16 + lw], wol = wof[(kc * 2 + 1) * 16 + lw];
                                                               va = dfx_mfma_16x16x32_f16(wol, sh, va);
                                                               vb = dfx_mfma_16x16x32_f16(woh, sl, vb);
                                                               vc = dfx_mfma_16x16x32_f16(woh, sh, vc);
                                                           }
                                                       });
            DFX_MFMA_GUARD();
            const f32x4 vj = (va + vb) + vc;
            DFX_WAVE_SYNC();   // (the reads of X above are done before a pad column of X is written — and before the next tile's e0 rows are)
            if (lane < 16) {
                float *Vp = t == 0 ? Y : X;
                int lv = lane;
                DFX_OPAQUE(lv);   // (six loop-invariant strip addresses otherwise: recomputed per tile instead of held — or spilled — across the frame loop)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int i = lv * 3 + j;
                    Vp[(i >> 2) * LD + C + (i & 3)] = vj[j] * A.unscale_wo;
                }
            }
        }
        DFX_WAVE_SYNC();
        if (lane < E) {
            auto V = [&](int p, int j) -> float {
                const int i = (p & 15) * 3 + j;
                return (p < 16 ? Y : X)[(i >> 2) * LD + C + (i & 3)];
            };
            int f = lane;
            DFX_OPAQUE(f);   // (the three strip addresses below are loop invariant: recomputed per frame instead of held — or spilled — across the frame loop)
            float acc = A.bias_o + V(f, 1);
            if (f > 0) acc += V(f - 1, 0);
            if (f < E - 1) acc += V(f + 1, 2);
            A.out[r * E + f] = dfx_sigmoid(acc);
        }
        DFX_WAVE_SYNC();  // the strips are rewritten by the next frame
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// df_dec.df_convp: Conv2d(C -> 2*O, (kt,1), groups = gcd(C, 2*O)) [+ 1x1 (2O x 2O) when kt > 1] + BN + ReLU
// (deepfilternet3.py:293-295, modules.py:49-71).  in c0 [R, Fd, C] -> out [R, Fd, 2*O].
// Per group: out1[pos][o] = sum_k sum_ci c0[t-kt+1+k, f, g*CG+ci] * W1[g][k][ci][o]   (causal: zero for frames < 0 of the clip)
// as MFMA 16x16x4 with N padded to 16; then the small 2O x 2O pointwise + bias + ReLU on the VALU.
// Tile: DFX_CP_TT frames x 16 bins of one clip; the kt-1 halo frames are staged with it.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_CP_TT 8
#define DFX_CP_FB 16
#define DFX_CP_THREADS 256
struct DfxCpArgs {
    const float *c0;   // [B*T, Fd, C]
    const float *w1;   // [G][kt][CG][16]   (o padded to 16 with zeros; BN-scaled when there is no pointwise conv)
    const float *w2;   // [NO][NO] w2[n][o] (BN-scaled; identity when there is no pointwise conv)
    const float *bias; // [NO]
    float *out;        // [B, NO/2, T, Fd, 2]  (tap-major, DFX_COEF_BOTF)
    int64_t B, T;
    int Fd, kt, G, NO; // NO = 2*O outputs, G groups, OG = NO/G outputs per group
    int tchunks, fchunks;
};

template <int C>
__global__ void __launch_bounds__(DFX_CP_THREADS) dfx_k_df_convp(DfxCpArgs A) {
    constexpr int LDC = C + 2;
    DFX_DYN_SMEM(float, sm);
    const int halo = DFX_CP_TT + A.kt - 1;
    const int CG = C / A.G, OG = A.NO / A.G;
    float *tile = sm;                                   // [halo][FB][LDC]
    float *w1s = tile + halo * DFX_CP_FB * LDC;         // [G][kt][CG][16]
    float *s1 = w1s + A.G * A.kt * CG * 16;             // [TT*FB][NO] stage-1 results
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int blk = blockIdx.x;
    const int fc = blk % A.fchunks;
    blk /= A.fchunks;
    const int tc = blk % A.tchunks;
    const int64_t b = blk / A.tchunks;
    const int64_t t0 = (int64_t)tc * DFX_CP_TT;
    const int f0 = fc * DFX_CP_FB;
    const int nt_valid = (int)((A.T - t0) < DFX_CP_TT ? (A.T - t0) : DFX_CP_TT);
    const int nf_valid = (A.Fd - f0) < DFX_CP_FB ? (A.Fd - f0) : DFX_CP_FB;
    for (int i = tid; i < A.G * A.kt * CG * 16; i += DFX_CP_THREADS) w1s[i] = A.w1[i];
    for (int i = tid; i < halo * DFX_CP_FB * C; i += DFX_CP_THREADS) {
        const int c = i % C;
        const int hf = i / C;
        const int f = hf % DFX_CP_FB, h = hf / DFX_CP_FB;
        const int64_t t = t0 - (A.kt - 1) + h;
        float v = 0.f;
        if (t >= 0 && t < A.T && f < nf_valid) v = A.c0[((b * A.T + t) * A.Fd + f0 + f) * C + c];
        tile[(h * DFX_CP_FB + f) * LDC + c] = v;
    }
    __syncthreads();
    // units: (t_local, g); M-tile = the 16 bins of frame t_local
    const int nunits = DFX_CP_TT * A.G;
    for (int u = wave; u < nunits; u += DFX_CP_THREADS / 64) {
        const int tl = u / A.G, g = u - tl * A.G;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < A.kt; ++k) {
            const float *arow = tile + ((tl + k) * DFX_CP_FB + (lane & 15)) * LDC + g * CG + (lane >> 4);
            const float *brow = w1s + ((g * A.kt + k) * CG + (lane >> 4)) * 16 + (lane & 15);
            for (int ks = 0; ks < CG / 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks], brow[4 * ks * 16], acc, 0, 0, 0);
        }
        // D[row = bin 4*(lane>>4)+r][col = o]
        const int o = lane & 15;
        if (o < OG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * (lane >> 4) + r;
                s1[(tl * DFX_CP_FB + f) * A.NO + g * OG + o] = acc[r];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < DFX_CP_TT * DFX_CP_FB * A.NO; i += DFX_CP_THREADS) {
        const int n = i % A.NO;
        const int p = i / A.NO;
        const int f = p % DFX_CP_FB, tl = p / DFX_CP_FB;
        if (tl >= nt_valid || f >= nf_valid) continue;
        float acc = A.bias[n];
        for (int o = 0; o < A.NO; ++o) acc += A.w2[n * A.NO + o] * s1[p * A.NO + o];
        A.out[(((b * (A.NO / 2) + (n >> 1)) * A.T + t0 + tl) * A.Fd + f0 + f) * 2 + (n & 1)] = fmaxf(acc, 0.f);  // BOTF
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// df_dec.df_convp, sliding-window form (used for kt <= 5).  The grouped (kt,1) conv, the optional 2O x 2O pointwise conv and
// the BN scale are folded on the host into one dense causal conv  out[pos][n] = relu(b[n] + sum_{k<kt} sum_c W_eff[k][c][n] *
// c0[t-kt+1+k][f][c])  (no nonlinearity sits between them, modules.py:49-71): a GEMM with K = kt*C, N = 2O padded to 16.
// A wave owns 16 frequency bins of one clip and walks a segment of frames: the kt-frame window of c0 lives in registers as
// the MFMA B operand (lane (bin, q) holds channels {16i + 4q .. +3} of each frame), so every c0 element is read from
// HBM exactly once (plus kt-1 halo frames per segment), and W_eff is the persistent A fragment (kt*C/4 registers).
// ---------------------------------------------------------------------------------------------------------------------
struct DfxCp2Args {
    const float *c0;    // [B*T, Fd, C]   (FUSE_C0: unused)
    const float *feat;  // FUSE_C0: feat_spec [B, T, Fd, 2], folded df_conv0 weights [20][C] and bias [C], lookahead L
    const float *weff0, *bias0;
    int L;
    int64_t t_begin;    // only frames [t_begin, t_end) are produced (the segments start there)
    int64_t t_end;
    int64_t t_zero;     // FUSE_C0: c0 of frames < t_zero is the zero padding (0 for whole clips; streaming: frames before the stream began)
    const float *weff;  // [kt][C][16]  weff[(k*C + c)*16 + n], n >= NO zero
    const float *bias;  // [16]
    float *out;         // [B, NO/2, T, Fd, 2]  (tap-major, DFX_COEF_BOTF)
    int64_t B, T;
    int Fd, NO, nfb, nseg, tseg;  // nfb = ceil(Fd/16) bin blocks, nseg segments of tseg frames (tseg % kt == 0)
};

// FUSE_C0: the window frames are not loaded but recomputed from feat_spec (dfx_c0_tile above): no c0 tensor exists and the kernel's
// HBM traffic is its 8*O*Fd-byte-per-frame store; it then is matrix-core bound (5*C/16 + KT*C/4 MFMAs per 16 bins and frame).
template <int C, int KT, bool FUSE_C0>
__global__ void __launch_bounds__(256) dfx_k_df_convp2(DfxCp2Args A) {
    constexpr int CPL = C / 4, V4 = CPL / 4, NT = C / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    float areg0[NT][5];
    float4 bias0[NT];
    if (FUSE_C0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) areg0[nt][ks] = A.weff0[(4 * ks + q) * C + 16 * nt + jl];
            bias0[nt] = reinterpret_cast<const float4 *>(A.bias0)[4 * nt + q];
        }
    }
    float areg[KT][CPL];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int ks = 0; ks < CPL; ++ks) areg[k][ks] = A.weff[((size_t)(k * C + 16 * (ks >> 2) + 4 * q + (ks & 3))) * 16 + jl];
    float biasr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasr[r] = A.bias[4 * q + r];
    const int64_t nruns = A.B * A.nfb * A.nseg;
    float amax = 0.f;   // range guard of the f16 splits
    for (int64_t run = (int64_t)blockIdx.x * 4 + wave; run < nruns; run += (int64_t)gridDim.x * 4) {
        const int seg = (int)(run % A.nseg);
        const int64_t rest = run / A.nseg;
        const int fb = (int)(rest % A.nfb);
        const int64_t b = rest / A.nfb;
        const int f = fb * 16 + jl;
        const bool fvalid = f < A.Fd;
        const int64_t t0 = A.t_begin + (int64_t)seg * A.tseg;
        const int64_t t1 = (t0 + A.tseg < A.t_end) ? t0 + A.tseg : A.t_end;
        float win[KT][CPL];
        auto load_frame = [&](float (&dst)[CPL], int64_t tau) {
            if (FUSE_C0) {
                if (tau >= A.t_zero) {  // wave-uniform; frames before the clip are the zero padding of c0 itself
                    float bv[5];
                    dfx_c0_patch(A.feat, b, tau, f, fvalid, A.T, A.Fd, A.L, q, bv);
                    dfx_c0_tile<C>(areg0, bias0, bv, fvalid, dst);
                } else {
#pragma unroll
                    for (int i = 0; i < CPL; ++i) dst[i] = 0.f;
                }
            } else if (fvalid && tau >= A.t_zero) {
                const float4 *p = reinterpret_cast<const float4 *>(A.c0 + ((b * A.T + tau) * A.Fd + f) * C + 4 * q);
#pragma unroll
                for (int v = 0; v < V4; ++v) {  // float4 v of this lane = channels 16*v + 4*q .. +3 (k-steps 4v .. 4v+3)
                    const float4 x = p[4 * v];
                    dst[4 * v + 0] = x.x;
                    dst[4 * v + 1] = x.y;
                    dst[4 * v + 2] = x.z;
                    dst[4 * v + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) dst[i] = 0.f;
            }
        };
        // frame tau lives in slot (tau - t0) mod KT; preload the kt-1 frames before the segment
        dfx_static_for<1, KT>([&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            load_frame(win[sl], t0 - KT + sl);
        });
        for (int64_t tb = t0; tb < t1; tb += KT) {
            dfx_static_for<0, KT>([&](auto pc) {
                constexpr int ph = decltype(pc)::value;
                const int64_t t = tb + ph;
                if (t < t1) {  // wave-uniform
                    load_frame(win[ph], t);
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    dfx_static_for<0, KT>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        constexpr int sl = (ph + 1 + k) % KT;  // tap k reads frame t - (KT-1) + k
#pragma unroll
                        for (int ks = 0; ks < CPL; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[k][ks], win[sl][ks], acc, 0, 0, 0);
                    });
                    if (fvalid) {
                        // output channel 4q + r = 2*tap + {re,im}; stored tap-major: [b][tap][t][f][2] (DFX_COEF_BOTF)
                        float *op = A.out + ((b * (A.NO / 2) * A.T + t) * A.Fd + f) * 2;
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (4 * q + 2 * h < A.NO)
                                *reinterpret_cast<float2 *>(op + (int64_t)(2 * q + h) * A.T * A.Fd * 2) =
                                    make_float2(fmaxf(acc[2 * h] + biasr[2 * h], 0.f), fmaxf(acc[2 * h + 1] + biasr[2 * h + 1], 0.f));
                    }
                }
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped GEMM on the matrix cores:  out[m, g*Ng + n] = act( sum_k A[m, g*Kg + k] * W[g][k][n] + bias[g*Ng+n] ) + res[m, ...]
// Covers GroupedLinearEinsum (modules.py:741-780; weight layout [G, I/G, H/G] used as is), the GRU input projections
// (G = 1, W = W_ih^T) and nn.Linear-shaped ops.  Kg % 4 == 0 and Ng % 4 == 0 are required (checked on the host).
// Block: 64 rows x BN columns of one group, K tiled by 32 through LDS; 4 waves, wave w owns rows [16w, 16w+16).
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_GG_BM 64
#define DFX_GG_KT 32
#define DFX_GG_THREADS 256
struct DfxGgArgs {
    const float *a;     // [M, lda]
    const float *a2;    // [M, lda] or null: the operand is a + a2 (an operand that is the sum of two activations, without materialising it)
    const float *w;     // [G][Kg][Ng]
    const float *bias;  // [G*Ng] or null
    const float *res;   // [M, ldo] or null (added after the activation)
    float *out;         // [M, ldo]
    int64_t M;
    int lda, ldo, G, Kg, Ng, act, ntn /* N tiles per group */;
    int perm_inner, perm_F;  // > 0: output column j = f*inner + i of row m = b*perm_T + t is stored tap-major, [B][inner/2][T][F][2]
    int64_t perm_T;          //      (DFX_COEF_BOTF, the reference's DfOutputReshapeMF layout); out/res are then addressed without ldo
    DfxRowMap rm;            // logical row -> physical row of a, out, res (time-chunked launches)
};

template <int BN>
__global__ void __launch_bounds__(DFX_GG_THREADS) dfx_k_ggemm(DfxGgArgs A) {
    constexpr int LDA = DFX_GG_KT + 2;
    constexpr int LDB = (BN == 64) ? 80 : 48;
    constexpr int NT = BN / 16;
    __shared__ float As[DFX_GG_BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[DFX_GG_KT * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD-aware: the G*ntn column tiles of one row tile are consecutive blocks of ONE XCD (id % 8), so the row
    // tile of A is served by that L2 after its first read and partially written output lines are merged there
    const int ncol = A.G * A.ntn;
    const int64_t jj = (int64_t)blockIdx.x >> 3;
    const int64_t mt = (jj / ncol) * 8 + (blockIdx.x & 7);
    const int yt = (int)(jj % ncol);
    const int64_t m0 = mt * DFX_GG_BM;
    if (m0 >= A.M) return;
    const int g = yt / A.ntn, n0 = (yt - g * A.ntn) * BN;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wg = A.w + (size_t)g * A.Kg * A.Ng;
    // Tiles go HBM -> registers -> LDS: all loads of a k-tile are issued together (a load -> LDS-store loop waits out one memory
    // latency per iteration), and the next k-tile is requested before the matrix ops of the current one (its latency hides behind
    // them and the barriers).  A tile: 64 rows x 32 k, float4 along k (2 per thread); B tile: 32 k x BN n, float4 along n.
    constexpr int AI = DFX_GG_BM * (DFX_GG_KT / 4) / DFX_GG_THREADS;                                    // 2
    constexpr int BI = (DFX_GG_KT * (BN / 4) + DFX_GG_THREADS - 1) / DFX_GG_THREADS;                    // 1 (BN <= 32) or 2
    float4 ra[AI], ra2[AI], rb[BI];
#pragma unroll
    for (int u = 0; u < AI; ++u) ra2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    // a thread stages the same rows of every k-tile: their offsets once (the row map of a time-chunked launch is a 64-bit division,
    // ~130 instructions; inside request() it ran twice per k-tile and thread, against 8 * NT matrix ops per tile and wave)
    int64_t aoff[AI];
#pragma unroll
    for (int u = 0; u < AI; ++u) {
        const int64_t m = m0 + (tid + u * DFX_GG_THREADS) / (DFX_GG_KT / 4);
        aoff[u] = m < A.M ? dfx_row(A.rm, m) * A.lda + g * A.Kg : -1;
    }
    auto request = [&](int k0) {
#pragma unroll
        for (int u = 0; u < AI; ++u) {
            const int i = tid + u * DFX_GG_THREADS;
            const int row = i / (DFX_GG_KT / 4), kq = i - row * (DFX_GG_KT / 4);
            const int k = k0 + 4 * kq;
            ra[u] = ra2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (aoff[u] >= 0 && k < A.Kg) {
                const int64_t off = aoff[u] + k;
                ra[u] = *reinterpret_cast<const float4 *>(A.a + off);
                if (A.a2) ra2[u] = *reinterpret_cast<const float4 *>(A.a2 + off);   // (added when the tile is stored: adding here would wait for both loads)
            }
        }
#pragma unroll
        for (int u = 0; u < BI; ++u) {
            const int i = tid + u * DFX_GG_THREADS;
            const int kk = i / (BN / 4), nq = i - kk * (BN / 4);
            const int k = k0 + kk, n = n0 + 4 * nq;
            rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < DFX_GG_KT * (BN / 4) && k < A.Kg && n < A.Ng) rb[u] = *reinterpret_cast<const float4 *>(wg + (size_t)k * A.Ng + n);
        }
    };
    request(0);
    for (int k0 = 0; k0 < A.Kg; k0 += DFX_GG_KT) {
#pragma unroll
        for (int u = 0; u < AI; ++u) {
            const int i = tid + u * DFX_GG_THREADS;
            const int row = i / (DFX_GG_KT / 4), kq = i - row * (DFX_GG_KT / 4);
            float *d = As + row * LDA + 4 * kq;
            if (A.a2) ra[u] = make_float4(ra[u].x + ra2[u].x, ra[u].y + ra2[u].y, ra[u].z + ra2[u].z, ra[u].w + ra2[u].w);
            d[0] = ra[u].x;
            d[1] = ra[u].y;
            d[2] = ra[u].z;
            d[3] = ra[u].w;
        }
#pragma unroll
        for (int u = 0; u < BI; ++u) {
            const int i = tid + u * DFX_GG_THREADS;
            const int kk = i / (BN / 4), nq = i - kk * (BN / 4);
            if (i < DFX_GG_KT * (BN / 4)) *reinterpret_cast<float4 *>(Bs + kk * LDB + 4 * nq) = rb[u];
        }
        __syncthreads();
        if (k0 + DFX_GG_KT < A.Kg) request(k0 + DFX_GG_KT);
        const float *arow = As + (16 * wave + (lane & 15)) * LDA + (lane >> 4);
        const float *brow = Bs + (lane >> 4) * LDB + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < DFX_GG_KT / 4; ++ks) {
            const float a = arow[4 * ks];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, brow[4 * ks * LDB + 16 * nt], acc[nt], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t ml = m0 + 16 * wave + 4 * (lane >> 4) + r;
        if (ml >= A.M) continue;
        const int64_t m = dfx_row(A.rm, ml);
        // (clip / frame of the row once per row and in 32 bits — rows are < 2^31, see dfx_row — instead of a 64-bit division per output value)
        uint32_t pb = 0, pt = 0;
        if (A.perm_inner > 0) {
            pb = (uint32_t)m / (uint32_t)A.perm_T;
            pt = (uint32_t)m - pb * (uint32_t)A.perm_T;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 16 * nt + (lane & 15);
            if (n >= A.Ng) continue;
            int col = g * A.Ng + n;
            float v = acc[nt][r];
            if (A.bias) v += A.bias[col];
            v = dfx_act(v, A.act);
            int64_t idx = m * A.ldo + col;
            if (A.perm_inner > 0) {
                const int f = col / A.perm_inner, i = col - f * A.perm_inner;
                const int64_t b = pb, t = pt;
                idx = (((b * (A.perm_inner >> 1) + (i >> 1)) * A.perm_T + t) * A.perm_F + f) * 2 + (i & 1);
            }
            if (A.res) v += A.res[idx];
            A.out[idx] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// df_dec.df_out as a kernel of its own (round 4): coefs = tanh(df_out(y [+ skip])).view(b, t, F', 2 O) + c0p, stored tap-major
// [B, O, T, F'][2] (deepfilternet3.py:326-330, modules.py:741-780).  The grouped linear is 15 k MACs per frame against 9.7 KB of traffic:
// HBM-bound — as dfx_k_ggemm (exact fp32 matrix ops, K through LDS, a tap-major permutation per output value) it was instruction-bound,
// 1.23 ms per pass at config 2.  Here a workgroup owns 16 frames: the frames are the columns of fp16-split matrix ops (A = the group's
// weights, 16 outputs x K = Kg <= 32 per fragment, B = the frames' 16 inputs of the group), tanh on the D fragments, which are parked in an
// LDS image of the 16 frames' output rows [O][16 frames][F' x 2] (row stride + 4 floats: the 16 lanes of a column write 16 different bank
// pairs); then every row leaves as coalesced 16-byte stores with c0p added on the way.  Wave w computes groups w, w + 4, ...
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_DFO_THREADS 256
#define DFX_DFO_NPT 16   /* most float4s of the tile's output rows a thread moves with the c0p values prefetched (more: loaded in the write-out loop) */
struct DfxDfOutArgs {
    const float *a, *a2;   // [R, G * Kg] operand (+ second addend or null)
    const dfx_h8 *wf;      // [G][ceil(Ng / 16)][hi,lo][64]: fragment (g, u): lane (o = l & 15, q = l >> 4), element i = W[g][8 q + i][16 u + o], zero beyond Kg / Ng
    const float *c0p;      // [B, O, T, Fd][2]
    float *out;            // [B, O, T, Fd][2]
    int64_t R, T;          // logical rows of this launch, frames per clip
    int G, Kg, Ng, NO, Fd; // NO = 2 * O values per bin
    float unscale;
    DfxRowMap rm;
    unsigned int *err;
};
#define DFX_DFO_SMEM(NO, Fd) ((size_t)((NO) / 2) * 16 * (2 * (Fd) + 4) * 4 + 16 * 8)

__global__ void __launch_bounds__(DFX_DFO_THREADS, 2) dfx_k_df_out_h3(DfxDfOutArgs A) {
    DFX_DYN_SMEM(float, img);   // [O][16][RS] then the 16 rows' (clip, frame)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int O = A.NO / 2, RS = 2 * A.Fd + 4, NU = (A.Ng + 15) >> 4, K = A.G * A.Kg;
    int *rows = reinterpret_cast<int *>(img + (size_t)O * 16 * RS);   // [16][2]: clip, frame (-1: beyond R)
    float amax = 0.f;
    const int64_t ntiles = (A.R + 15) >> 4;
    const int F4 = A.Fd / 2, NP4 = O * 16 * F4;   // float4s per output row, float4s of a tile's output rows
    const bool pre = NP4 <= DFX_DFO_NPT * DFX_DFO_THREADS;   // the c0p values of a tile are requested before its compute phase
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * 16 + jl;
        const bool live = m < A.R;
        const int64_t r = live ? dfx_row(A.rm, m) : 0;
        if (tid < 16) {
            const uint32_t cb = (uint32_t)r / (uint32_t)A.T;
            rows[2 * tid] = live ? (int)cb : -1;
            rows[2 * tid + 1] = (int)((uint32_t)r - cb * (uint32_t)A.T);
        }
        __syncthreads();
        // ---- this thread's pieces of the write-out: element offsets (B * O * T * 2 Fd < 2^31: host) and, ahead of the compute phase, c0p
        float4 cv[DFX_DFO_NPT];
        unsigned off[DFX_DFO_NPT];
#pragma unroll
        for (int k = 0; k < DFX_DFO_NPT; ++k) {
            const int idx = tid + DFX_DFO_THREADS * k;
            off[k] = 0xffffffffu;
            cv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pre && idx < NP4) {
                const int row = idx / F4, c4 = idx - row * F4, n = row >> 4, i = row & 15;
                const int cb = rows[2 * i];
                if (cb >= 0) {
                    off[k] = (((unsigned)cb * (unsigned)O + (unsigned)n) * (unsigned)A.T + (unsigned)rows[2 * i + 1]) * (unsigned)(2 * A.Fd) + 4u * (unsigned)c4;
                    cv[k] = *reinterpret_cast<const float4 *>(A.c0p + off[k]);
                }
            }
        }
        // ---- compute: groups wave, wave + 4, ...; lane (frame jl, q) feeds k = 8 q .. 8 q + 7 of the group's Kg inputs
        for (int g = wave; g < A.G; g += DFX_DFO_THREADS / 64) {
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 0.f;
            if (live && 8 * q < A.Kg) {   // (Kg % 8 == 0: host)
                const float4 *p = reinterpret_cast<const float4 *>(A.a + r * K + g * A.Kg + 8 * q);
                float4 v0 = p[0], v1 = p[1];
                if (A.a2) {
                    const float4 *p2 = reinterpret_cast<const float4 *>(A.a2 + r * K + g * A.Kg + 8 * q);
                    const float4 w0 = p2[0], w1 = p2[1];
                    v0 = make_float4(v0.x + w0.x, v0.y + w0.y, v0.z + w0.z, v0.w + w0.w);
                    v1 = make_float4(v1.x + w1.x, v1.y + w1.y, v1.z + w1.z, v1.w + w1.w);
                }
                x[0] = v0.x, x[1] = v0.y, x[2] = v0.z, x[3] = v0.w, x[4] = v1.x, x[5] = v1.y, x[6] = v1.z, x[7] = v1.w;
            }
            dfx_h8 xh, xl;
            dfx_split8_g(x, xh, xl, amax);
            dfx_h8 wh = A.wf[(((size_t)g * NU) * 2 + 0) * 64 + lane], wl = A.wf[(((size_t)g * NU) * 2 + 1) * 64 + lane];
            for (int u = 0; u < NU; ++u) {
                const dfx_h8 ch = wh, cl = wl;
                if (u + 1 < NU) {   // the next tile's fragments are requested before this one's matrix ops
                    wh = A.wf[(((size_t)g * NU + u + 1) * 2 + 0) * 64 + lane];
                    wl = A.wf[(((size_t)g * NU + u + 1) * 2 + 1) * 64 + lane];
                }
                f32x4 d = dfx_mfma_16x16x32_f16(cl, xh, f32x4{0.f, 0.f, 0.f, 0.f});
                d = dfx_mfma_16x16x32_f16(ch, xl, d);
                d = dfx_mfma_16x16x32_f16(ch, xh, d);
                // D: lane (frame jl, q) holds outputs 16 u + 4 q + r of the group = flat index o = g Ng + 16 u + 4 q + r -> (bin o / NO, value o % NO)
#pragma unroll
                for (int r2 = 0; r2 < 4; r2 += 2) {
                    const int ol = 16 * u + 4 * q + r2;
                    if (ol < A.Ng) {   // (Ng and NO even: a pair (re, im) never straddles)
                        const int o = g * A.Ng + ol, f = o / A.NO, i = o - f * A.NO;
                        *reinterpret_cast<float2 *>(img + ((size_t)(i >> 1) * 16 + jl) * RS + 2 * f) =
                            make_float2(dfx_act(d[r2] * A.unscale, DFX_ACT_TANH), dfx_act(d[r2 + 1] * A.unscale, DFX_ACT_TANH));
                    }
                }
            }
        }
        __syncthreads();
        // ---- write-out: row (tap n, frame i) = 2 Fd floats, + c0p, coalesced 16-byte accesses
        if (pre) {
#pragma unroll
            for (int k = 0; k < DFX_DFO_NPT; ++k) {
                if (off[k] != 0xffffffffu) {
                    const int idx = tid + DFX_DFO_THREADS * k, row = idx / F4, c4 = idx - row * F4;
                    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)row * RS + 4 * c4);
                    *reinterpret_cast<float4 *>(A.out + off[k]) = make_float4(v.x + cv[k].x, v.y + cv[k].y, v.z + cv[k].z, v.w + cv[k].w);
                }
            }
        } else {
            for (int idx = tid; idx < NP4; idx += DFX_DFO_THREADS) {
                const int row = idx / F4, c4 = idx - row * F4, n = row >> 4, i = row & 15;
                const int cb = rows[2 * i];
                if (cb < 0) continue;
                const int64_t o64 = (((int64_t)cb * O + n) * A.T + rows[2 * i + 1]) * (2 * A.Fd) + 4 * c4;
                const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)row * RS + 4 * c4);
                const float4 c = *reinterpret_cast<const float4 *>(A.c0p + o64);
                *reinterpret_cast<float4 *>(A.out + o64) = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
            }
        }
        __syncthreads();   // the image and the row table are rewritten by the next tile
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// dfx_k_df_out_h3r<NU> (round 6): the same kernel with the wave's weight fragments RESIDENT in registers (<= 4 groups per wave x NU tiles x hi / lo =
// 128 registers at NU = 4: two waves per SIMD have 256 each) and a tile's operand rows requested in one batch.  dfx_k_df_out_h3 asks for a group's rows when
// it reaches the group and for every tile's fragments one tile ahead: per 16-frame tile a wave waited out four row latencies and sixteen L2 latencies — ~24 us
// per tile for ~2 us of arithmetic, 3.6 of the 8 TB/s.  Here the operand rows of all four groups are in flight at once, the prior c0p values are requested in one batch
// behind the arithmetic (into the registers the rows have left), and the loop over the tile's fragments reads registers only.  Same expressions, same order: same bits.
template <int NU>
__global__ void __launch_bounds__(DFX_DFO_THREADS, 2) dfx_k_df_out_h3r(DfxDfOutArgs A) {
    constexpr int GW = 4;   // groups per wave (G <= 4 GW: host)
    DFX_DYN_SMEM(float, img);   // [O][16][RS] then the 16 rows' (clip, frame)
    const int tid = threadIdx.x, lane = tid & 63, wave = dfx_wave_uniform(tid >> 6), q = lane >> 4, jl = lane & 15;
    const int O = A.NO / 2, RS = 2 * A.Fd + 4, K = A.G * A.Kg;
    int *rows = reinterpret_cast<int *>(img + (size_t)O * 16 * RS);   // [16][2]: clip, frame (-1: beyond R)
    float amax = 0.f;
    const int64_t ntiles = (A.R + 15) >> 4;
    const int F4 = A.Fd / 2, NP4 = O * 16 * F4;   // float4s per output row, float4s of a tile's output rows (<= DFX_DFO_NPT * DFX_DFO_THREADS: host)
    dfx_h8 wr[GW][NU][2];
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
        const int g = wave + 4 * gi, gc = g < A.G ? g : 0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            wr[gi][u][0] = A.wf[(((size_t)gc * NU + u) * 2 + 0) * 64 + lane];
            wr[gi][u][1] = A.wf[(((size_t)gc * NU + u) * 2 + 1) * 64 + lane];
        }
    }
    // this thread's pieces of a tile's write-out, idx = tid + 256 k -> (output row idx / F4, float4 column idx % F4): the first one by division, once; piece
    // k + 1 from piece k by adding 256 = da * F4 + db (the divisions by the run-time row length were a quarter of the kernel's vector instructions when every
    // piece of every tile repeated them: the kernel is VALU-bound, ~800 vector instructions per frame)
    const int row0 = tid / F4, col0 = tid - row0 * F4, da = DFX_DFO_THREADS / F4, db = DFX_DFO_THREADS - da * F4;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t m = tile * 16 + jl;
        const bool live = m < A.R;
        const int64_t r = live ? dfx_row(A.rm, m) : 0;
        // ---- the operand rows of the wave's groups: lane (frame jl, q) feeds k = 8 q .. 8 q + 7 of a group's Kg inputs (Kg % 8 == 0: host)
        float4 xa[GW][2], xb[GW][2];
        const bool feeds = live && 8 * q < A.Kg;
#pragma unroll
        for (int gi = 0; gi < GW; ++gi) {
            const int g = wave + 4 * gi;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            xa[gi][0] = xa[gi][1] = xb[gi][0] = xb[gi][1] = z;
            if (feeds && g < A.G) {
                const float4 *p = reinterpret_cast<const float4 *>(A.a + r * K + g * A.Kg + 8 * q);
                xa[gi][0] = p[0], xa[gi][1] = p[1];
                if (A.a2) {
                    const float4 *p2 = reinterpret_cast<const float4 *>(A.a2 + r * K + g * A.Kg + 8 * q);
                    xb[gi][0] = p2[0], xb[gi][1] = p2[1];
                }
            }
        }
        if (tid < 16) {
            const uint32_t cb = (uint32_t)r / (uint32_t)A.T;
            rows[2 * tid] = live ? (int)cb : -1;
            rows[2 * tid + 1] = (int)((uint32_t)r - cb * (uint32_t)A.T);
        }
        __syncthreads();
        // this thread's pieces of the write-out: element offsets (B * O * T * 2 Fd < 2^31: host)
        int qq = q, jq = jl;   // (opaque per tile: the ~50 tile-invariant LDS / row addresses derived from them are recomputed, not held across the tile loop — no scratch)
        DFX_OPAQUE(qq);
        DFX_OPAQUE(jq);
        auto piece_off = [&](int row, int c4) -> unsigned {   // element offset of (tap n = row / 16, frame i = row % 16, column c4) in c0p / out
            const int n = row >> 4, i = row & 15;
            const int cb = rows[2 * i];
            if (row >= 16 * O || cb < 0) return 0xffffffffu;
            return (((unsigned)cb * (unsigned)O + (unsigned)n) * (unsigned)A.T + (unsigned)rows[2 * i + 1]) * (unsigned)(2 * A.Fd) + 4u * (unsigned)c4;
        };
        auto piece_next = [&](int &row, int &c4) {
            row += da, c4 += db;
            if (c4 >= F4) c4 -= F4, row += 1;
        };
#pragma unroll
        for (int gi = 0; gi < GW; ++gi) {
            const int g = wave + 4 * gi;
            if (g < A.G) {
                float x[8];
                float4 v0 = xa[gi][0], v1 = xa[gi][1];
                if (A.a2) {
                    const float4 w0 = xb[gi][0], w1 = xb[gi][1];
                    v0 = make_float4(v0.x + w0.x, v0.y + w0.y, v0.z + w0.z, v0.w + w0.w);
                    v1 = make_float4(v1.x + w1.x, v1.y + w1.y, v1.z + w1.z, v1.w + w1.w);
                }
                x[0] = v0.x, x[1] = v0.y, x[2] = v0.z, x[3] = v0.w, x[4] = v1.x, x[5] = v1.y, x[6] = v1.z, x[7] = v1.w;
                dfx_h8 xh, xl;
                dfx_split8_g(x, xh, xl, amax);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const dfx_h8 ch = wr[gi][u][0], cl = wr[gi][u][1];
                    f32x4 d = dfx_mfma_16x16x32_f16(cl, xh, f32x4{0.f, 0.f, 0.f, 0.f});
                    d = dfx_mfma_16x16x32_f16(ch, xl, d);
                    d = dfx_mfma_16x16x32_f16(ch, xh, d);
                    // D: lane (frame jl, q) holds outputs 16 u + 4 q + r of the group = flat index o = g Ng + 16 u + 4 q + r -> (bin o / NO, value o % NO)
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2 += 2) {
                        const int ol = 16 * u + 4 * qq + r2;
                        if (ol < A.Ng) {   // (Ng and NO even: a pair (re, im) never straddles)
                            const int o = g * A.Ng + ol, f = o / A.NO, i = o - f * A.NO;
                            *reinterpret_cast<float2 *>(img + ((size_t)(i >> 1) * 16 + jq) * RS + 2 * f) =
                                make_float2(dfx_act(d[r2] * A.unscale, DFX_ACT_TANH), dfx_act(d[r2 + 1] * A.unscale, DFX_ACT_TANH));
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- write-out: row (tap n, frame i) = 2 Fd floats, + c0p, coalesced 16-byte accesses; c0p in two batches of DFX_DFO_NPT / 2 loads behind the
        // tile's arithmetic (a whole tile's values in flight beside the operand rows and the resident fragments do not fit 256 registers)
        int prow = row0, pcol = col0;
        DFX_OPAQUE(prow);
        DFX_OPAQUE(pcol);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 cv[DFX_DFO_NPT / 2];
            unsigned off[DFX_DFO_NPT / 2];
            int lrow[DFX_DFO_NPT / 2];   // (row << 8 | column: where the piece sits in the LDS image)
#pragma unroll
            for (int k = 0; k < DFX_DFO_NPT / 2; ++k) {
                off[k] = piece_off(prow, pcol);
                lrow[k] = (prow << 8) | pcol;
                piece_next(prow, pcol);
                cv[k] = *reinterpret_cast<const float4 *>(A.c0p + (off[k] != 0xffffffffu ? off[k] : 0u));
            }
#pragma unroll
            for (int k = 0; k < DFX_DFO_NPT / 2; ++k) {
                if (off[k] != 0xffffffffu) {
                    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)(lrow[k] >> 8) * RS + 4 * (lrow[k] & 255));
                    *reinterpret_cast<float4 *>(A.out + off[k]) = make_float4(v.x + cv[k].x, v.y + cv[k].y, v.z + cv[k].z, v.w + cv[k].w);
                }
            }
        }
        __syncthreads();   // the image and the row table are rewritten by the next tile
    }
    if (amax >= DFX_H3_LIMIT && A.err) dfx_raise(A.err + 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense projection with a weight-stationary LDS tile:  out[m, n] = sum_k a[m, k] * w[k, n] + bias[n]   (K = 256).
// Used for the GRU input projections W_ih x + b (5 launches of [B*T, 256] x [256, 768] — 60 % of all GEMM flops).
// A persistent workgroup keeps its 256 x 128 column tile of W in LDS for its whole life (147 KB, 1 workgroup per CU) and
// streams row tiles past it: no per-tile barrier, activations never touch LDS.  Computed transposed like dfx_k_pwconv:
//   A operand = W^T fragment read from LDS (lane (n, q): W[64q + ks][n], bank-conflict free with a 144-float row stride),
//   B operand = x: lane (row, q) holds the 64 contiguous K values [64q, 64q+64) of its row, loaded as float4s from HBM,
//   D = 4 consecutive output columns per lane -> float4 stores.  The next row tile is prefetched during the 512 MFMAs.
// Grid: (N/128) column tiles x row groups, the column tiles of one row group on one XCD (its rows are L2 hits 5 of 6 times).
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_PJ_K 256
#define DFX_PJ_BN 128
#define DFX_PJ_LDW (DFX_PJ_BN + 16)
#define DFX_PJ_SMEM ((size_t)DFX_PJ_K * DFX_PJ_LDW * 4)
struct DfxPjArgs {
    const float *a;     // [M, 256]
    const float *w;     // [256][N]
    const float *bias;  // [N]
    float *out;         // [M, N]
    int64_t M;
    int N, ncol, rgroups;  // ncol = N/128 column tiles, rgroups = row groups (grid = 8 * ceil(rgroups/8) * ncol)
    DfxRowMap rm = DfxRowMap{0, 0, 0};   // logical row -> physical row of a and out (time-chunked launches of the layer-pipelined phase)
};

// MODE (dev ablations, tools/dev/proj_bench.hip): 0 = product; 1 = no activation loads; 2 = no LDS fragment re-reads
#define DFX_PJ_THREADS 512  // 8 waves: two per SIMD (a single wave cannot keep the fp32 matrix pipe busy, measured 38-50 %)
template <int MODE>
__global__ void __launch_bounds__(DFX_PJ_THREADS, 2) dfx_k_proj256(DfxPjArgs A) {
    constexpr int K = DFX_PJ_K, BN = DFX_PJ_BN, LDW = DFX_PJ_LDW, NT = BN / 16;
    DFX_DYN_SMEM(float, wl);  // [(ks*4 + q)][LDW]: row (ks, q) holds W[64q + ks][n0 .. n0+128)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int64_t jj = (int64_t)blockIdx.x >> 3;
    const int64_t rg = (jj / A.ncol) * 8 + (blockIdx.x & 7);
    const int ct = (int)(jj % A.ncol);
    if (rg >= A.rgroups) return;
    const int n0 = ct * BN;
    for (int i = tid; i < K * (BN / 4); i += DFX_PJ_THREADS) {
        const int k = i / (BN / 4), n4 = i - k * (BN / 4);
        const float4 v = *reinterpret_cast<const float4 *>(A.w + (size_t)k * A.N + n0 + 4 * n4);
        *reinterpret_cast<float4 *>(wl + ((k & 63) * 4 + (k >> 6)) * LDW + 4 * n4) = v;
    }
    __syncthreads();
    constexpr int NW = DFX_PJ_THREADS / 64;
    const int64_t ntiles = (A.M + 15) / 16;  // 16-row tiles; a wave takes tiles rg*NW + wave, + rgroups*NW, ...
    const int64_t tstride = (int64_t)A.rgroups * NW;
    const float *wfrag = wl + q * LDW + jl;  // + (ks*4)*LDW + 16*nt
    const float4 *bias4 = reinterpret_cast<const float4 *>(A.bias + n0) + q;
    // half h of a tile's rows (8 float4 per lane): the second half of the k range and the first are refilled separately (below)
    auto load_half = [&](float4 *dst, int64_t tl, int h) {
        const int64_t m = tl * 16 + jl;
        if (MODE != 1 && tl < ntiles && m < A.M) {
            const float4 *p = reinterpret_cast<const float4 *>(A.a + dfx_row(A.rm, m) * K + 64 * q) + 8 * h;
#pragma unroll
            for (int v = 0; v < 8; ++v) dst[8 * h + v] = p[v];
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) dst[8 * h + v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // One register buffer, refilled in halves while the matrix core works: the next tile's first 32 k values go into xv[0..8) as soon as
    // k-step 31 has read them (8192 matrix cycles before they are needed), its last 32 after k-step 63.  (Two whole buffers — 128 registers —
    // left the kernel 44 registers short of its two waves per SIMD: 176 bytes of scratch per lane.)
    float4 xv[16];
    int64_t tile = rg * NW + wave;
    load_half(xv, tile, 0);
    load_half(xv, tile, 1);
    // one 16-row tile: 64 k-steps x NT MFMAs; the W^T fragments of k-step ks+1 are read from LDS while k-step ks computes
    while (tile < ntiles) {
        const int64_t next = tile + tstride;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float fa[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fa[0][nt] = wfrag[16 * nt];
        dfx_static_for<0, 64>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks + 1 < 64 && MODE != 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fa[(ks + 1) & 1][nt] = wfrag[(ks + 1) * 4 * LDW + 16 * nt];
            }
            const float4 x4 = xv[ks >> 2];
            const float xk = (ks & 3) == 0 ? x4.x : (ks & 3) == 1 ? x4.y : (ks & 3) == 2 ? x4.z : x4.w;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[MODE == 2 ? 0 : (ks & 1)][nt], xk, acc[nt], 0, 0, 0);
            DFX_SCHED_BARRIER();
            if constexpr (ks == 31) load_half(xv, next, 0);
        });
        load_half(xv, next, 1);
        const int64_t m = tile * 16 + jl;
        if (m < A.M) {
            float4 *op = reinterpret_cast<float4 *>(A.out + dfx_row(A.rm, m) * A.N + n0 + 4 * q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bz = bias4[4 * nt];
                op[4 * nt] = make_float4(acc[nt][0] + bz.x, acc[nt][1] + bz.y, acc[nt][2] + bz.z, acc[nt][3] + bz.w);
            }
        }
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense projection on the fp16-split matrix path ("fp16x3"):  out[m, n] = sum_k a[m, k] * w[k, n] + bias[n], K = 256.
// Every fp32 operand is split exactly-enough into two halves (hi = f16(x), lo = f16(x - hi): 22 mantissa bits; the weights
// are pre-scaled by a power of two so that lo stays out of the f16 subnormals) and the product is formed as
// lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation: ~2^-21 relative to the fp32 result (the waveform
// parity bar is 1e-4 RMS; measured end to end: unchanged to 1e-7) at 16/3 x the fp32 MFMA rate, which turns the GRU input
// projections from MFMA-bound (50 GMAC each) into HBM-bound (1.05 GB each).
// Workgroup = 128 rows (8 waves x 16 rows) x all N columns: `a` is read from HBM once; W streams from L2 through a
// double-buffered 64 KB LDS stage per 64-column chunk, already in fragment order (one 16-byte ds_read per fragment and lane).
// Transposed roles as in dfx_k_pwconv: A operand = W^T fragment, B operand = activations, D = 4 consecutive columns per lane.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_PH_THREADS 512
#define DFX_PH_BM 128
#define DFX_PH_NC 64
#define DFX_PH_CHUNK_H8 (8 * 4 * 2 * 64)                 /* dfx_h8 per column chunk: [kc][ct][hi,lo][lane] */
#define DFX_PH_SMEM ((size_t)2 * DFX_PH_CHUNK_H8 * 16)
// Same-XCD hand-overs of the persistent GRU phase's followers (round 5, see DfxGhSync::x).
struct DfxXcd {
    unsigned int *me = nullptr;            // this party's registration word: tag | (xcd + 1)
    const unsigned int *prod = nullptr;    // the registration word of the party whose output this one consumes, or null
    const unsigned int *cons = nullptr;    // ... of the party that consumes this one's output, or null
    const unsigned int *cons2 = nullptr;   // a second consumer (the emb follower feeds two projection followers), or null
    unsigned int tag = 0;                  // of this pass (low four bits zero)
    unsigned int *stat = nullptr;          // dev aid: counts light releases
};
static __device__ __forceinline__ void dfx_xcd_register(const DfxXcd &X) {
    if (X.me && threadIdx.x == 0) __hip_atomic_store(X.me, X.tag | (unsigned int)(dfx_xcc_id() + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ bool dfx_xcd_same(const DfxXcd &X, const unsigned int *other) {
    return other && __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (X.tag | (unsigned int)(dfx_xcc_id() + 1));
}
// the consumer's side of a block hand-over, behind the barrier that follows the poll (every thread)
static __device__ __forceinline__ void dfx_xcd_acquire(const DfxXcd &X) {
    if (dfx_xcd_same(X, X.prod)) DFX_L1_INV();
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// the producer's side: thread 0, behind a barrier in front of which EVERY wave has drained its stores (DFX_VMEM_DRAIN)
static __device__ __forceinline__ void dfx_xcd_release(const DfxXcd &X, unsigned int *flag, unsigned int value) {
    if (dfx_xcd_same(X, X.cons) && (!X.cons2 || dfx_xcd_same(X, X.cons2))) {
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (X.stat) __hip_atomic_fetch_add(X.stat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// A producer kernel of the persistent GRU phase announces its own completion (round 5): every workgroup, when its rows are stored, adds itself to a
// counter; the last one resets the counter and raises the flag the consumers poll — the dfx_k_flag_set launch behind the producer (and the ~8 us of
// host enqueue and dispatch it costs per layer and chunk) is gone.  cnt == nullptr: nothing (every launch outside that phase).
// A follower workgroup picks the 16-clip group it serves: one whose RECURRENCE runs on this workgroup's XCD (workgroups go round-robin over the 8
// XCDs with a rotation that differs from launch to launch, so block index g of the follower launch is usually NOT next to group g's recurrence).
// rec[g] = registration words of the recurrences of one layer (all layers of a launch share the mapping), claim[8] = slots handed out per XCD
// (zeroed by the host before the launch).  Own XCD first, then the others in turn: as many workgroups as groups, so everyone finds exactly one —
// also if the dispatch was not an exact round robin.  Thread 0 decides, the result travels through *slot (LDS).  rec == nullptr: group = fallback.
static __device__ __forceinline__ int dfx_xcd_claim(const unsigned int *rec, unsigned int tag, int groups, unsigned int *claim, int fallback, int spin_limit,
                                                    unsigned int *err, int *slot) {
    if (!rec || groups > 64) return fallback;
    if (threadIdx.x == 0) {
        auto xcd_of = [&](int g) { return (int)(__hip_atomic_load(rec + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 15u) - 1; };   // (re-read: no array in scratch)
        bool ok = true;
        for (int g = 0; g < groups && ok; ++g) {   // every recurrence of this pass has registered (they start first)
            unsigned int v;
            int spins = 0;
            while (((v = __hip_atomic_load(rec + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~15u) != tag || (v & 15u) == 0u) {
                if (++spins > spin_limit) {
                    dfx_raise(err + 2);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        int got = -1;
        const int x0 = dfx_xcc_id();
        for (int d = 0; d < 8 && got < 0 && ok; ++d) {
            const int x = (x0 + d) & 7;
            int cnt = 0;
            for (int g = 0; g < groups; ++g) cnt += xcd_of(g) == x;
            if (!cnt) continue;
            const unsigned int k = __hip_atomic_fetch_add(claim + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)k >= cnt) continue;
            for (int g = 0, j = 0; g < groups; ++g)
                if (xcd_of(g) == x && j++ == (int)k) got = g;
        }
        *slot = got < 0 ? fallback : got;   // (got < 0: registrations timed out; err is raised)
    }
    __syncthreads();
    const int r = *slot;
    __syncthreads();
    return r;
}
struct DfxPublish {
    unsigned int *cnt = nullptr;   // completion counter of this producer (one word per stream: launches on a stream do not overlap)
    unsigned int *flag = nullptr;
    unsigned int value = 0, nblocks = 0;
};
static __device__ __forceinline__ void dfx_publish(const DfxPublish &P) {
    if (!P.cnt) return;
    DFX_VMEM_DRAIN();  // (s_barrier does not wait for the other waves' stores: each wave drains its own in front of it)
    __syncthreads();   // the workgroup's stores happen before thread 0's release below (one L2 write-back per workgroup; a release fence in
                       // every thread in front of the barrier is one per WAVE and costs the step 1.6 ms, measurements R5.10)
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(P.cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == P.nblocks) {
            __hip_atomic_store(P.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.flag, P.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
struct DfxPhArgs {
    const float *a;      // [M, 256]
    const dfx_h8 *wf;    // [N/64][8][4][2][64] fragment-ordered, pre-scaled f16 hi/lo of W[k][n]
    const float *bias;   // [N]
    float *out;          // [M, N]
    int64_t M;
    int N;
    float unscale;       // 1 / weight scale (a power of two)
    DfxRowMap rm;        // logical row -> physical row of a and out
    int parts = 1;       // dfx_k_proj256_h3 only: the column chunks of a row block are dealt to this many workgroups (divides N / 64).  A launch of
                         // few rows (a streaming hop: 4096 rows = 32 row blocks) then runs on parts x 32 CUs and each workgroup streams 1 / parts of W
    DfxPublish pub;      // persistent GRU phase: the kernel raises its consumers' flag itself
};

__global__ void __launch_bounds__(DFX_PH_THREADS, 2) dfx_k_proj256_h3(DfxPhArgs A) {
    DFX_DYN_SMEM(dfx_h8, ws);  // [2][DFX_PH_CHUNK_H8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int nparts = A.parts > 0 ? A.parts : 1;
    const int cper = A.N / DFX_PH_NC / nparts;
    const int cfirst = (int)(blockIdx.x % nparts) * cper, nchunks = cfirst + cper;
    const int64_t ml = (int64_t)(blockIdx.x / nparts) * DFX_PH_BM + 16 * wave + jl;
    const bool ok = ml < A.M;
    const int64_t m = ok ? dfx_row(A.rm, ml) : 0;
    constexpr int PER_T = DFX_PH_CHUNK_H8 / DFX_PH_THREADS;  // 8 x 16 bytes per thread and chunk
    // stage the first chunk
#pragma unroll
    for (int i = 0; i < PER_T; ++i) ws[(size_t)(cfirst & 1) * DFX_PH_CHUNK_H8 + i * DFX_PH_THREADS + tid] = A.wf[(size_t)cfirst * DFX_PH_CHUNK_H8 + i * DFX_PH_THREADS + tid];
    // this lane's B operands: row m, k = 32*kc + 8*q .. +7.  The row is scaled by a power of two (exact) so that its largest magnitude
    // sits just below 2^14 before the f16 split: any finite row then keeps ~22 bits relative to ITS OWN scale — rows of tiny values
    // would otherwise lose their lo halves to the f16 subnormals (|x| < 6e-5) and values above 65504 would turn into inf.
    dfx_h8 xh[8], xl[8];
    float row_unscale = 1.f;
    {
        const float4 *p = reinterpret_cast<const float4 *>(A.a + m * 256 + 8 * q);
        float4 xu[8], xv[8];
        float mx = 0.f;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            xu[kc] = ok ? p[8 * kc] : make_float4(0.f, 0.f, 0.f, 0.f);
            xv[kc] = ok ? p[8 * kc + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xu[kc].x), fabsf(xu[kc].y)), fmaxf(fabsf(xu[kc].z), fabsf(xu[kc].w))));
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xv[kc].x), fabsf(xv[kc].y)), fmaxf(fabsf(xv[kc].z), fabsf(xv[kc].w))));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));   // the four lanes (q) that share row jl
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);            // mx = f * 2^ex, f in [0.5, 1)
            e = 14 - ex;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = ldexpf(1.f, e);
        row_unscale = ldexpf(1.f, -e);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            float x[8];
            x[0] = xu[kc].x * sc, x[1] = xu[kc].y * sc, x[2] = xu[kc].z * sc, x[3] = xu[kc].w * sc;
            x[4] = xv[kc].x * sc, x[5] = xv[kc].y * sc, x[6] = xv[kc].z * sc, x[7] = xv[kc].w * sc;
            dfx_split8(x, xh[kc], xl[kc]);
        }
    }
    const float unscale = A.unscale * row_unscale;   // D leaves lane (jl, q) with row jl: the row's own scale
    __syncthreads();
    for (int c = cfirst; c < nchunks; ++c) {
        const dfx_h8 *wc = ws + (size_t)(c & 1) * DFX_PH_CHUNK_H8;
        dfx_h8 pre[PER_T];
        if (c + 1 < nchunks) {
            const dfx_h8 *src = A.wf + (size_t)(c + 1) * DFX_PH_CHUNK_H8;
#pragma unroll
            for (int i = 0; i < PER_T; ++i) pre[i] = src[i * DFX_PH_THREADS + tid];
        }
        // the four column tiles of the chunk advance together: consecutive matrix ops go to DIFFERENT accumulators (an op that waits for
        // the previous op's accumulator stalls for the whole latency of that op; per accumulator the order of the terms is unchanged)
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            dfx_h8 whi[4], wlo[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                whi[ct] = wc[((kc * 4 + ct) * 2 + 0) * 64 + lane];
                wlo[ct] = wc[((kc * 4 + ct) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(wlo[ct], xh[kc], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(whi[ct], xl[kc], acc[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(whi[ct], xh[kc], acc[ct]);
        }
        if (ok) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = c * DFX_PH_NC + 16 * ct + 4 * q;
                const float4 bz = *reinterpret_cast<const float4 *>(A.bias + n);
                *reinterpret_cast<float4 *>(A.out + m * A.N + n) =
                    make_float4(acc[ct][0] * unscale + bz.x, acc[ct][1] * unscale + bz.y, acc[ct][2] * unscale + bz.z, acc[ct][3] * unscale + bz.w);
            }
        }
        if (c + 1 < nchunks) {
            dfx_h8 *dst = ws + (size_t)((c + 1) & 1) * DFX_PH_CHUNK_H8;
#pragma unroll
            for (int i = 0; i < PER_T; ++i) dst[i * DFX_PH_THREADS + tid] = pre[i];
        }
        __syncthreads();
    }
    dfx_publish(A.pub);
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_gru_step_h3: ONE time step of a GRU layer for many streams — input projection, recurrent product, gates and the new state in one
// launch (the frame-by-frame streaming runtime: 4096 streams x 1 hop).  As dfx_k_proj256_h3 + dfx_k_gru_rec_h3 a step cost two dependent
// launches (29 + 37 us at 4096 streams, three layers deep per hop), the second of which made each of its 256 workgroups stream all of
// W_hh for 16 rows.  Here a workgroup owns 128 rows and ONE block of 64 hidden units: it streams the three gates' 64-column chunks of
// W_ih and of W_hh (6 x 64 KB, double-buffered through LDS like the projection kernel), keeps gi in accumulators, and finishes its
// units.  The old state is read from h_in and the new one goes to h_out (another buffer: other workgroups still read all of h_in).
//   gates as in dfx_k_gru_rec_h3: r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * (gh_n + b_hn)), h' = (1 - z) n + z h
// ---------------------------------------------------------------------------------------------------------------------
struct DfxGstArgs {
    const float *x;        // layer input, row of stream b = dfx_row(xrm, b), 256 values
    const float *h_in;     // [B, 256]
    float *h_out;          // [B, 256]
    float *y;              // the layer's output for the following kernels (row dfx_row(yrm, b)) or null
    const dfx_h8 *wif, *whf;   // W_ih / W_hh as dfx_k_proj256_h3's fragments [12][8][4][2][64]
    const float *bias_i;   // [768]: b_ih (+ b_hr, b_hz)
    const float *bhn;      // [256]
    float unscale_i, unscale_h;
    int64_t B;
    DfxRowMap xrm, yrm;
};
// CT: 16-unit tiles per workgroup.  CT = 4 (64 units, 64-KB chunks): B / 128 x 4 workgroups — 128 at 4096 streams, half the chip, two waves per
// SIMD sharing its matrix pipe (576 ops each: 7.7 us of pipe time).  CT = 2 (32 units, 32-KB chunks: the halves of the same fragment
// chunks): 256 workgroups, one per CU.  Both operands (the layer input and the old state) are loaded and split before the first chunk —
// the recurrent operand used to be fetched between chunks 2 and 3, a full HBM latency with nothing to cover it — and the six chunk
// phases are unrolled, so that the accumulators of all six stay in registers without a selection chain.
#ifndef DFX_GST_ABLATE
#define DFX_GST_ABLATE 0   /* dev ablations (tools/dev/gru_step_bench.hip): 1 no matrix ops, 2 no chunk loads after the first, 4 no operand rows, 8 no gate math */
#endif
template <int CT>
__global__ void __launch_bounds__(DFX_PH_THREADS, 2) dfx_k_gru_step_h3(DfxGstArgs A) {
    static_assert(CT == 2 || CT == 4, "32 or 64 hidden units per workgroup");
    constexpr int CHUNK = 8 * CT * 2 * 64;   // dfx_h8 per chunk of this workgroup: [kc][ct][hi,lo][lane]
    constexpr int PER_T = CHUNK / DFX_PH_THREADS;
    constexpr int NU = 16 / CT;              // unit blocks per layer
    DFX_DYN_SMEM(dfx_h8, ws);  // [2][CHUNK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    // The NU workgroups that share a block of 128 rows are dealt to ONE XCD (workgroups go round-robin over the 8 XCDs: block i runs on XCD
    // i % 8 up to a rotation that is constant within a launch): the operand rows then come from HBM into one L2 instead of all eight —
    // with row-major block order the step spent half its time fetching 8 x 8 MB of rows (grid: row blocks padded to a multiple of 8)
    const int xcd = (int)(blockIdx.x & 7), kk = (int)(blockIdx.x >> 3);
    const int u = kk % NU;   // block of 16 * CT hidden units
    const int64_t rb = xcd + 8 * (int64_t)(kk / NU);
    if (rb * DFX_PH_BM >= A.B) return;
    const int64_t b = rb * DFX_PH_BM + 16 * wave + jl;
    const bool ok = b < A.B;
    // chunk i of this workgroup: i < 3: W_ih, gate i; else W_hh, gate i - 3: columns 256 g + 16 CT u .. of the [12][8][4][2][64] fragments,
    // i.e. the tiles ct0 .. ct0 + CT - 1 of the 64-column chunk 4 g + (CT u) / 4
    const int ct0 = (CT * u) & 3;
    auto chunk_src = [&](int i) { return (i < 3 ? A.wif : A.whf) + (size_t)(4 * (i % 3) + (CT * u) / 4) * DFX_PH_CHUNK_H8; };
    auto src_index = [&](int e) {   // element e of this workgroup's chunk inside the 64-column chunk
        const int kc = e / (CT * 128), rem = e - kc * (CT * 128);
        return (kc * 4 + ct0) * 128 + rem;
    };
    // both operand rows are requested before anything waits for data (they are the kernel's only HBM reads: a step without them was
    // measured 12 us instead of 23), the first chunk goes to LDS meanwhile
    float4 xraw[16], hraw[16];
    const int64_t bl = ok ? b : A.B - 1;
    const int64_t xr = dfx_row(A.xrm, bl);
    auto fetch_row = [&](const float *row, float4 (&raw)[16]) {
        const float4 *p = reinterpret_cast<const float4 *>(row + 8 * q);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            // (rows past the end read the last row: a clamped index is one instruction, a predicated load five and a branch — tools/dev/scan_isa.py)
            raw[2 * kc] = !(DFX_GST_ABLATE & 4) ? p[8 * kc] : make_float4(0.f, 0.f, 0.f, 0.f);
            raw[2 * kc + 1] = !(DFX_GST_ABLATE & 4) ? p[8 * kc + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch_row(A.x + xr * 256, xraw);
    fetch_row(A.h_in + bl * 256, hraw);
    {
        const dfx_h8 *src = chunk_src(0);
#pragma unroll
        for (int i = 0; i < PER_T; ++i) ws[i * DFX_PH_THREADS + tid] = src[src_index(i * DFX_PH_THREADS + tid)];
    }
    // a row's 256 values as B operands, scaled by the row's own power of two before the f16 split (see dfx_k_proj256_h3)
    auto split_row = [&](const float4 (&raw)[16], dfx_h8 (&vh)[8], dfx_h8 (&vl)[8]) -> float {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(raw[i].x), fabsf(raw[i].y)), fmaxf(fabsf(raw[i].z), fabsf(raw[i].w))));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            e = 14 - ex;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = ldexpf(1.f, e);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            const float4 xu = raw[2 * kc], xv = raw[2 * kc + 1];
            float x[8];
            x[0] = xu.x * sc, x[1] = xu.y * sc, x[2] = xu.z * sc, x[3] = xu.w * sc;
            x[4] = xv.x * sc, x[5] = xv.y * sc, x[6] = xv.z * sc, x[7] = xv.w * sc;
            dfx_split8(x, vh[kc], vl[kc]);
        }
        return ldexpf(1.f, -e);
    };
    dfx_h8 xh[8], xl[8], hh[8], hl[8];
    const float us_x = A.unscale_i * split_row(xraw, xh, xl);
    float us_h = 0.f;   // (the recurrent operand is split under the matrix ops of chunk 1: it is first used by chunk 3)
    f32x4 g[6][CT];   // gi (r, z, n) then gh (r, z, n)
    __syncthreads();
    dfx_static_for<0, 6>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const dfx_h8 *wc = ws + (size_t)(c & 1) * CHUNK;
        dfx_h8 pre[PER_T];
        if constexpr (c + 1 < 6 && !(DFX_GST_ABLATE & 2)) {
            const dfx_h8 *src = chunk_src(c + 1);
#pragma unroll
            for (int i = 0; i < PER_T; ++i) pre[i] = src[src_index(i * DFX_PH_THREADS + tid)];
            DFX_SCHED_BARRIER();   // the loads are issued HERE: left alone, the compiler sinks them below the matrix ops, next to the LDS stores that use them
        }
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (c == 1) us_h = A.unscale_h * split_row(hraw, hh, hl);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            dfx_h8 whi[CT], wlo[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                whi[ct] = wc[((kc * CT + ct) * 2 + 0) * 64 + lane];
                wlo[ct] = wc[((kc * CT + ct) * 2 + 1) * 64 + lane];
            }
            const dfx_h8 vh = c < 3 ? xh[kc] : hh[kc], vl = c < 3 ? xl[kc] : hl[kc];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(wlo[ct], vh, acc[ct]);
            if (DFX_GST_ABLATE & 1) continue;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(whi[ct], vl, acc[ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = dfx_mfma_16x16x32_f16(whi[ct], vh, acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) g[c][ct] = acc[ct] * (c < 3 ? us_x : us_h);
        if constexpr (c + 1 < 6) {
            dfx_h8 *dst = ws + (size_t)((c + 1) & 1) * CHUNK;
#pragma unroll
            for (int i = 0; i < PER_T; ++i) dst[i * DFX_PH_THREADS + tid] = pre[i];
            __syncthreads();
        }
    });
    if (ok) {
        const int64_t yr = A.y ? dfx_row(A.yrm, b) : 0;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n0 = 16 * CT * u + 16 * ct + 4 * q;   // D leaves lane (row jl, q) with the units n0 .. n0 + 3
            const float4 br = *reinterpret_cast<const float4 *>(A.bias_i + n0), bz = *reinterpret_cast<const float4 *>(A.bias_i + 256 + n0),
                         bn = *reinterpret_cast<const float4 *>(A.bias_i + 512 + n0), bh = *reinterpret_cast<const float4 *>(A.bhn + n0),
                         hp = *reinterpret_cast<const float4 *>(A.h_in + b * 256 + n0);
            const float brr[4] = {br.x, br.y, br.z, br.w}, bzz[4] = {bz.x, bz.y, bz.z, bz.w}, bnn[4] = {bn.x, bn.y, bn.z, bn.w},
                        bhh[4] = {bh.x, bh.y, bh.z, bh.w}, hpp[4] = {hp.x, hp.y, hp.z, hp.w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (DFX_GST_ABLATE & 8) {
                    o[r] = g[0][ct][r] + g[1][ct][r] + g[2][ct][r] + g[3][ct][r] + g[4][ct][r] + g[5][ct][r] + brr[r] + bzz[r] + bnn[r] + bhh[r] + hpp[r];
                    continue;
                }
                const float rg = dfx_fast_rcp(1.f + dfx_fast_exp(-((g[0][ct][r] + brr[r]) + g[3][ct][r])));
                const float zg = dfx_fast_rcp(1.f + dfx_fast_exp(-((g[1][ct][r] + bzz[r]) + g[4][ct][r])));
                const float pre = (g[2][ct][r] + bnn[r]) + rg * (g[5][ct][r] + bhh[r]);
                const float ng = 2.f * dfx_fast_rcp(1.f + dfx_fast_exp(-2.f * pre)) - 1.f;
                o[r] = (1.f - zg) * ng + zg * hpp[r];
            }
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4 *>(A.h_out + b * 256 + n0) = ov;
            if (A.y) *reinterpret_cast<float4 *>(A.y + yr * 256 + n0) = ov;
        }
    }
}

// The same projection with TWO row tiles per wave: every W fragment read from LDS feeds 6 matrix ops instead of 3.  dfx_k_proj256_h3 is
// bound by its LDS fragment reads (per 64-column chunk and CU: 8 waves x 64 ds_read_b128 = 512 KB against 3072 matrix-pipe cycles per
// SIMD, and ds_read_b128 runs at half the LDS rate); here a workgroup is NW waves x 32 rows: NW = 8 (256 rows, two waves per SIMD, ~220
// registers) halves the fragment reads per row; NW = 4 (128 rows, one wave per SIMD) was measured slower than the one-tile kernel
// (0.61 vs 0.43 ms for 256 k rows: a lone wave per SIMD cannot hide its own LDS / barrier latencies).
#ifndef DFX_PH_ABLATE
#define DFX_PH_ABLATE 0   /* dev ablations (tools/dev/proj_h3_bench.hip): 1 no output stores, 2 no LDS refill of the next chunk, 4 no matrix ops, 8 no prefetch loads */
#endif
// CT: 16-column tiles per LDS chunk.  CT = 4 (64 columns, 2 x 64 KB of LDS: one workgroup per CU) or CT = 2 (32 columns, 2 x 32 KB: TWO
// workgroups of 4 waves per CU).  What the ablations of tools/dev/proj_h3_ablate.sh show for <8, 4> (0.364 ms per 256 k rows): without the
// output stores 0.262, without the matrix ops 0.308, without the LDS refill 0.33 — the phases of a chunk (matrix ops | 64 KB of output
// stores at the HBM write rate | refill | barrier) run one after the other, in all eight waves at once.  Two independent workgroups per
// CU do not share a barrier: one stores while the other computes.
template <int NW, int CT>
__global__ void __launch_bounds__(64 * NW, CT == 2 ? 2 : 1) dfx_k_proj256_h3x2(DfxPhArgs A) {
    constexpr int DFX_PH2_THREADS = 64 * NW, BM2 = 32 * NW;
    constexpr int CH8 = 8 * CT * 2 * 64;   // dfx_h8 per chunk: [kc][ct][hi,lo][lane]
    DFX_DYN_SMEM(dfx_h8, ws);  // [2][CH8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int nchunks = A.N / (16 * CT);
    constexpr int PER_T = CH8 / DFX_PH2_THREADS;  // 16-byte pieces per thread and chunk
    // chunk c of CT tiles inside the host's 64-column blocks [N/64][8][4][hi,lo][64]: piece i -> (kc, ct, half, lane)
    auto src_of = [&](int c, int i) -> const dfx_h8 * {
        const int e = i * DFX_PH2_THREADS + tid;              // element inside the chunk: ((kc * CT + ct) * 2 + half) * 64 + lane
        const int kc = e / (CT * 128), r = e - kc * (CT * 128);   // r = (ct * 2 + half) * 64 + lane
        const int blk = (c * CT) / 4, ct0 = (c * CT) % 4;
        return A.wf + (size_t)blk * DFX_PH_CHUNK_H8 + (size_t)(kc * 4 + ct0) * 128 + r;
    };
#pragma unroll
    for (int i = 0; i < PER_T; ++i) ws[i * DFX_PH2_THREADS + tid] = *src_of(0, i);
    dfx_h8 xh[2][8], xl[2][8];
    float unscale[2];
    int64_t mrow[2];
    bool okr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t ml = (int64_t)blockIdx.x * BM2 + 32 * wave + 16 * t + jl;
        okr[t] = ml < A.M;
        mrow[t] = okr[t] ? dfx_row(A.rm, ml) : 0;
        const float4 *p = reinterpret_cast<const float4 *>(A.a + mrow[t] * 256 + 8 * q);
        float4 xu[8], xv[8];
        float mx = 0.f;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            xu[kc] = okr[t] ? p[8 * kc] : make_float4(0.f, 0.f, 0.f, 0.f);
            xv[kc] = okr[t] ? p[8 * kc + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xu[kc].x), fabsf(xu[kc].y)), fmaxf(fabsf(xu[kc].z), fabsf(xu[kc].w))));
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xv[kc].x), fabsf(xv[kc].y)), fmaxf(fabsf(xv[kc].z), fabsf(xv[kc].w))));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));   // the four lanes (q) that share row jl
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {       // per-row power-of-two scale, exactly as in dfx_k_proj256_h3
            int ex;
            (void)frexpf(mx, &ex);
            e = 14 - ex;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = ldexpf(1.f, e);
        unscale[t] = A.unscale * ldexpf(1.f, -e);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            float x[8];
            x[0] = xu[kc].x * sc, x[1] = xu[kc].y * sc, x[2] = xu[kc].z * sc, x[3] = xu[kc].w * sc;
            x[4] = xv[kc].x * sc, x[5] = xv[kc].y * sc, x[6] = xv[kc].z * sc, x[7] = xv[kc].w * sc;
            dfx_split8(x, xh[t][kc], xl[t][kc]);
        }
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const dfx_h8 *wc = ws + (size_t)(c & 1) * CH8;
        dfx_h8 pre[PER_T];
        if (c + 1 < nchunks && !(DFX_PH_ABLATE & 8)) {
#pragma unroll
            for (int i = 0; i < PER_T; ++i) pre[i] = *src_of(c + 1, i);
        }
        f32x4 acc[2][CT];  // 2 * CT independent accumulators: no matrix op waits for its predecessor
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            dfx_h8 whi[CT], wlo[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                whi[ct] = wc[((kc * CT + ct) * 2 + 0) * 64 + lane];
                wlo[ct] = wc[((kc * CT + ct) * 2 + 1) * 64 + lane];
            }
            // per accumulator the same order of the three terms as in the one-tile kernel: same bits
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][ct] = (DFX_PH_ABLATE & 4) ? acc[t][ct] + f32x4{(float)wlo[ct][0], 0.f, 0.f, (float)xh[t][kc][0]} : dfx_mfma_16x16x32_f16(wlo[ct], xh[t][kc], acc[t][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][ct] = (DFX_PH_ABLATE & 4) ? acc[t][ct] : dfx_mfma_16x16x32_f16(whi[ct], xl[t][kc], acc[t][ct]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][ct] = (DFX_PH_ABLATE & 4) ? acc[t][ct] : dfx_mfma_16x16x32_f16(whi[ct], xh[t][kc], acc[t][ct]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n = c * (16 * CT) + 16 * ct + 4 * q;
            const float4 bz = *reinterpret_cast<const float4 *>(A.bias + n);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (okr[t] && (!(DFX_PH_ABLATE & 1) || acc[t][ct][0] == 1.2345e-30f))
                    *reinterpret_cast<float4 *>(A.out + mrow[t] * A.N + n) =
                        make_float4(acc[t][ct][0] * unscale[t] + bz.x, acc[t][ct][1] * unscale[t] + bz.y, acc[t][ct][2] * unscale[t] + bz.z,
                                    acc[t][ct][3] * unscale[t] + bz.w);
        }
        if (c + 1 < nchunks && !(DFX_PH_ABLATE & 2)) {
            dfx_h8 *dst = ws + (size_t)((c + 1) & 1) * CH8;
#pragma unroll
            for (int i = 0; i < PER_T; ++i) dst[i * DFX_PH2_THREADS + tid] = pre[i];
        }
        __syncthreads();
    }
    dfx_publish(A.pub);
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_emb_fan: everything that hangs off the encoder GRU's output, in ONE pass over its rows (deepfilternet3.py:149-158 linear_out of
// enc.emb_gru [+ skip], :163-165 lsnr_fc, :245-249 erb_dec.emb_gru.linear_in, :323-326 df_dec.df_gru.linear_in and df_skip):
//     emb   = relu(y . W_out) [+ res]                      y [R,256] -> emb [R,512]      (16 groups of 16 -> 32)
//     out_c = act_c(emb . W_c)       c = dec_in, df_skip: 16 groups of 32 -> 16 ("narrow"); dfg_in: 8 groups of 64 -> 32 ("wide")
//     lsnr  = sigmoid(emb . w_l + b) * scale + offset
// As separate grouped GEMMs these moved emb (2 KB per frame) five times beside the GRU chain (written once, read by four consumers:
// 16 KB per frame in all); here emb only exists in registers: 1 KB in, <= 3 KB out per frame.
// The grouped linears are block diagonal and their blocks nest: columns [32J, 32J+32) of y -> features [64J, 64J+64) of emb ->
// columns [32J, 32J+32) of every consumer.  A wave owns 16 * RT rows and walks the super-chunks J; per super-chunk and row tile the work
// is a chain of exact fp32 matrix ops (v_mfma_f32_16x16x4_f32) that never leaves the registers, with the roles transposed like
// dfx_chain_stage: D[feature][row] = W^T (A) x y^T (B) leaves lane (row, q) with features 16t + 4q + r of its row, which IS a valid B
// operand of the next product once its contraction index is enumerated in that order (the host packs W_c accordingly, pack_fan).
// No LDS: the kernel can share a CU with anything.  KIND of a consumer: 0 absent, 1 narrow, 2 wide.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_FAN_NC 3       /* consumers: dec_in, dfg_in, df_skip */
#define DFX_FAN_WPJ 28     /* float4 fragments per lane and super-chunk: 4 (stage 1) + 8 per consumer (a narrow one uses 4) */
struct DfxFanArgs {
    const float *y;             // [R, 32 * nj]
    const float4 *wfrag;        // [nj][DFX_FAN_WPJ][64] (pack_fan)
    const float *res;           // [R, 64 * nj] or null: added to emb after the ReLU (identity / grouped-linear skip of the encoder GRU)
    float *emb_out;             // [R, 64 * nj] or null (only written when something else still reads emb)
    float *out[DFX_FAN_NC];     // [R, 32 * nj] each
    int act[DFX_FAN_NC];
    const float *lsnr_w;        // [64 * nj] or null
    float lsnr_b, lsnr_scale, lsnr_off;
    float *lsnr;                // [R]
    int64_t R;
    int nj;                     // super-chunks: hidden / 32
    int parts;                  // a row tile's super-chunks are dealt to this many waves (divides nj; > 1 only without lsnr: few rows, a streaming hop)
    DfxRowMap rm;
    DfxPublish pub;             // persistent GRU phase: the kernel raises its consumers' flag itself
};
// one item of dfx_k_emb_fan: row tile `tile` (16 * RT logical rows), super-chunks [J0, J1)
template <int RT, int K0, int K1, int K2>
static __device__ __forceinline__ void dfx_emb_fan_item(const DfxFanArgs &A, int64_t tile, int J0, int J1) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int H = 32 * A.nj, EMB = 64 * A.nj;
    int64_t prow[RT];
    bool ok[RT];
    float ls[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t lr = tile * (16 * RT) + 16 * rt + n;
        ok[rt] = lr < A.R;
        prow[rt] = ok[rt] ? dfx_row(A.rm, lr) : 0;
        ls[rt] = 0.f;
    }
    // stage-1 fragments and the y columns of the NEXT super-chunk are requested before the current one's matrix ops; the consumers'
    // fragments of a super-chunk are requested at its top and first used after its stage 1
    float4 w1n[4], yn[RT][2];
    auto request = [&](int J) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w1n[i] = A.wfrag[((size_t)J * DFX_FAN_WPJ + i) * 64 + lane];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                yn[rt][ch] = ok[rt] ? *reinterpret_cast<const float4 *>(A.y + prow[rt] * H + 32 * J + 16 * ch + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    request(J0);
    for (int J = J0; J < J1; ++J) {
        float4 w1[4], yc[RT][2], wc[DFX_FAN_NC][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) w1[i] = w1n[i];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) yc[rt][0] = yn[rt][0], yc[rt][1] = yn[rt][1];
        dfx_static_for<0, DFX_FAN_NC>([&](auto CI) {
            constexpr int c = decltype(CI)::value;
            constexpr int kind = c == 0 ? K0 : (c == 1 ? K1 : K2);
#pragma unroll
            for (int i = 0; i < (kind == 2 ? 8 : (kind == 1 ? 4 : 0)); ++i) wc[c][i] = A.wfrag[((size_t)J * DFX_FAN_WPJ + 4 + 8 * c + i) * 64 + lane];
        });
        if (J + 1 < J1) request(J + 1);
        float4 lw[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            lw[tt] = A.lsnr_w ? *reinterpret_cast<const float4 *>(A.lsnr_w + 64 * J + 16 * tt + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float e[4][4];   // [feature tile tt = 2 ch + t][r]: feature 64 J + 16 tt + 4 q + r of row n
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const float ys[4] = {yc[rt][ch].x, yc[rt][ch].y, yc[rt][ch].z, yc[rt][ch].w};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tt = 2 * ch + t;
                    const float ws[4] = {w1[tt].x, w1[tt].y, w1[tt].z, w1[tt].w};
                    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) d = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[k], ys[k], d, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[tt][r] = fmaxf(d[r], 0.f);
                    const int64_t eoff = prow[rt] * EMB + 64 * J + 16 * tt + 4 * q;
                    if (A.res && ok[rt]) {
                        const float4 rv = *reinterpret_cast<const float4 *>(A.res + eoff);
                        e[tt][0] += rv.x, e[tt][1] += rv.y, e[tt][2] += rv.z, e[tt][3] += rv.w;
                    }
                    if (A.emb_out && ok[rt]) *reinterpret_cast<float4 *>(A.emb_out + eoff) = make_float4(e[tt][0], e[tt][1], e[tt][2], e[tt][3]);
                    ls[rt] += e[tt][0] * lw[tt].x;
                    ls[rt] += e[tt][1] * lw[tt].y;
                    ls[rt] += e[tt][2] * lw[tt].z;
                    ls[rt] += e[tt][3] * lw[tt].w;
                }
            }
            dfx_static_for<0, DFX_FAN_NC>([&](auto CI) {
                constexpr int c = decltype(CI)::value;
                constexpr int kind = c == 0 ? K0 : (c == 1 ? K1 : K2);
                if constexpr (kind != 0) {
                    constexpr int NTT = kind == 2 ? 4 : 2;   // feature tiles an output tile contracts over
#pragma unroll
                    for (int u = 0; u < 2; ++u) {   // output columns 32 J + 16 u + 4 q + r
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < NTT; ++i) {
                            const int tt = kind == 2 ? i : 2 * u + i;
                            const float4 wv = wc[c][NTT * u + i];
                            const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[r], e[tt][r], d, 0, 0, 0);
                        }
                        if (ok[rt])
                            *reinterpret_cast<float4 *>(A.out[c] + prow[rt] * H + 32 * J + 16 * u + 4 * q) =
                                make_float4(dfx_act(d[0], A.act[c]), dfx_act(d[1], A.act[c]), dfx_act(d[2], A.act[c]), dfx_act(d[3], A.act[c]));
                    }
                }
            });
        }
    }
    if (A.lsnr) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float v = ls[rt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0 && ok[rt]) A.lsnr[prow[rt]] = dfx_sigmoid(v + A.lsnr_b) * A.lsnr_scale + A.lsnr_off;
        }
    }
}
template <int RT, int K0, int K1, int K2>
__global__ void __launch_bounds__(256, 2) dfx_k_emb_fan(DfxFanArgs A) {
    const int64_t ntile = (A.R + 16 * RT - 1) / (16 * RT);
    const int jper = A.nj / A.parts;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < ntile * A.parts; item += (int64_t)gridDim.x * 4) {
        const int64_t tile = item / A.parts;
        const int J0 = (int)(item - tile * A.parts) * jper;
        dfx_emb_fan_item<RT, K0, K1, K2>(A, tile, J0, J0 + jper);
    }
    dfx_publish(A.pub);
}
// The same arithmetic as a persistent FOLLOWER of the encoder GRU's recurrence (round 5, see dfx_k_proj_follow): workgroup g serves the 16
// clips of group g; per block of DFX_EF_STEPS steps it waits for the recurrence's yprog word, runs one item per wave (16 rows, all
// super-chunks: RT = 1) and raises embprog.  A's pointers are those of the whole pass; rm is ignored.
#define DFX_EF_STEPS 8
struct DfxFollowSync {
    const unsigned int *src;   // [groups]: steps the producer has completed (pbase + steps)
    unsigned int *dst;         // [groups]: steps this follower has completed
    unsigned int pbase;
    unsigned int *err;
    int spin_limit;
    int64_t B, T;
    DfxXcd x;   // same-XCD hand-overs: the words of group 0 (the kernel adds its group)
    unsigned int *xclaim = nullptr;   // [8], zeroed before the launch (dfx_xcd_claim over x.prod: the encoder layer's recurrences), or null
    int groups = 0;
};
template <int K0, int K1, int K2>
__global__ void __launch_bounds__(512, 1) dfx_k_emb_follow(DfxFanArgs A, DfxFollowSync Y) {
    __shared__ int gslot;
    const int g = dfx_xcd_claim(Y.xclaim ? Y.x.prod : nullptr, Y.x.tag, Y.groups, Y.xclaim, (int)blockIdx.x, Y.spin_limit, Y.err, &gslot);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t b0 = (int64_t)g * 16;
    const int nclip = (int)(Y.B - b0 < 16 ? Y.B - b0 : 16);
    if (nclip <= 0) return;
    const int H = 32 * A.nj, EMB = 64 * A.nj;
    // this group's rows: every array starts at clip b0 (a logical row of a block is (clip, step) through rm)
    A.y += b0 * Y.T * H;
    if (A.res) A.res += b0 * Y.T * EMB;
    if (A.emb_out) A.emb_out += b0 * Y.T * EMB;
#pragma unroll
    for (int c = 0; c < DFX_FAN_NC; ++c)
        if (A.out[c]) A.out[c] += b0 * Y.T * H;
    if (A.lsnr) A.lsnr += b0 * Y.T;
    A.parts = 1;
    DfxXcd X = Y.x;
    if (X.me) X.me += g, X.prod = X.prod ? X.prod + g : nullptr, X.cons = X.cons ? X.cons + g : nullptr, X.cons2 = X.cons2 ? X.cons2 + g : nullptr;
    dfx_xcd_register(X);
    for (int64_t t0 = 0; t0 < Y.T; t0 += DFX_EF_STEPS) {
        const int64_t t1 = t0 + DFX_EF_STEPS < Y.T ? t0 + DFX_EF_STEPS : Y.T;
        if (tid == 0) {
            const unsigned int want = Y.pbase + (unsigned int)t1;
            int spins = 0;
            while ((int)(__hip_atomic_load(Y.src + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > Y.spin_limit) {
                    dfx_raise(Y.err + 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
        dfx_xcd_acquire(X);
        A.rm = DfxRowMap{Y.T, t1 - t0, t0};
        A.R = (int64_t)nclip * (t1 - t0);
        const int ntile = (int)((A.R + 15) / 16);
        for (int tile = wave; tile < ntile; tile += 8) dfx_emb_fan_item<1, K0, K1, K2>(A, tile, 0, A.nj);
        DFX_VMEM_DRAIN();
        __syncthreads();   // the block's rows are stored (every wave's stores acknowledged: the barrier alone does not wait for them)
        // ALWAYS the agent-scope release here: dec_in / dfg_in only go to this XCD's projection followers, but df_skip, emb and lsnr are read by
        // kernels anywhere on the chip (the DF tail, the decoder tails) while this kernel is still running — they must leave this XCD's L2.
        // (Those readers are ordered behind a recurrence's chunk flag, whose release writes back the RECURRENCE's L2: the same one only as long as
        // nothing else disturbs the round-robin dispatch — two handles at once gave wrong samples with the light form here.)
        if (tid == 0) __hip_atomic_store(Y.dst + g, Y.pbase + (unsigned int)t1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_enc_fan: the two grouped linears at the end of the encoder front in one pass over c1 (deepfilternet3.py:179-182, modules.py:702-738):
//     emb_in = relu(df_fc_emb(c1.flatten)) + e3.flatten          c1 [R, 3072] -> [R, 512]    (32 groups of 96 -> 16)
//     xa     = relu(linear_in(emb_in))                            -> [R, 256]                 (16 groups of 32 -> 16): the encoder GRU's input
// Same register-chained exact fp32 matrix ops as dfx_k_emb_fan (the D fragment of the first product is the B operand of the second):
// emb_in (2 KB per frame written and read back on the front's critical path) only exists in registers unless a skip connection of the
// encoder GRU needs it, and the front loses a launch.  A wave owns 16 * RT rows and walks the 32 groups of df_fc_emb; the 96 inputs of a
// group are one 384-byte run per row (six float4 per lane quad).  The next group's operands are requested before the current one's
// matrix ops.
// ---------------------------------------------------------------------------------------------------------------------
struct DfxEncFanArgs {
    const float *c1;            // [R, 96 * ng]
    const float4 *w1;           // [ng][6][64] (pack_encfan)
    const float4 *w2;           // [ng / 2][2][64]
    const float *e3;            // [R, 16 * ng]: added after the ReLU
    float *emb_out;             // [R, 16 * ng] or null
    float *out;                 // [R, 8 * ng]
    int64_t R;
    int ng;                     // groups of df_fc_emb (even)
    int parts;                  // a row tile's groups are dealt to this many waves (divides ng / 2): few rows (a streaming hop) then still fill the chip
    DfxRowMap rm;
};
template <int RT>
__global__ void __launch_bounds__(256, 2) dfx_k_enc_fan(DfxEncFanArgs A) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int K = 96 * A.ng, EMB = 16 * A.ng, H = 8 * A.ng;
    const int64_t ntile = (A.R + 16 * RT - 1) / (16 * RT);
    const int gper = A.ng / A.parts;   // (even)
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < ntile * A.parts; item += (int64_t)gridDim.x * 4) {
        const int64_t tile = item / A.parts;
        const int g0 = (int)(item - tile * A.parts) * gper, g1 = g0 + gper;
        int64_t prow[RT];
        bool ok[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t lr = tile * (16 * RT) + 16 * rt + n;
            ok[rt] = lr < A.R;
            prow[rt] = ok[rt] ? dfx_row(A.rm, lr) : 0;
        }
        float4 wn[6], yn[RT][6];
        auto request = [&](int g) {
#pragma unroll
            for (int i = 0; i < 6; ++i) wn[i] = A.w1[((size_t)g * 6 + i) * 64 + lane];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    yn[rt][i] = ok[rt] ? *reinterpret_cast<const float4 *>(A.c1 + prow[rt] * K + 96 * g + 16 * i + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        request(g0);
        float e[RT][2][4];   // emb_in features 16 g + 4 q + r of the two groups of the current pair
        for (int g = g0; g < g1; ++g) {
            float4 w[6], y[RT][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = wn[i];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 6; ++i) y[rt][i] = yn[rt][i];
            const int t = g & 1;
            float4 w2[2];
            if (t) {
                w2[0] = A.w2[((size_t)(g >> 1) * 2 + 0) * 64 + lane];
                w2[1] = A.w2[((size_t)(g >> 1) * 2 + 1) * 64 + lane];
            }
            if (g + 1 < g1) request(g + 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float ws[4] = {w[i].x, w[i].y, w[i].z, w[i].w}, ys[4] = {y[rt][i].x, y[rt][i].y, y[rt][i].z, y[rt][i].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) d = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[k], ys[k], d, 0, 0, 0);
                }
                const int64_t eoff = prow[rt] * EMB + 16 * g + 4 * q;
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[rt]) rv = *reinterpret_cast<const float4 *>(A.e3 + eoff);
                float ev[4] = {fmaxf(d[0], 0.f) + rv.x, fmaxf(d[1], 0.f) + rv.y, fmaxf(d[2], 0.f) + rv.z, fmaxf(d[3], 0.f) + rv.w};
                if (A.emb_out && ok[rt]) *reinterpret_cast<float4 *>(A.emb_out + eoff) = make_float4(ev[0], ev[1], ev[2], ev[3]);
                if (t == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[rt][0][r] = ev[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[rt][1][r] = ev[r];
                    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const float ws[4] = {w2[tt].x, w2[tt].y, w2[tt].z, w2[tt].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[r], e[rt][tt][r], o, 0, 0, 0);
                    }
                    if (ok[rt])
                        *reinterpret_cast<float4 *>(A.out + prow[rt] * H + 16 * (g >> 1) + 4 * q) =
                            make_float4(fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f));
                }
            }
        }
    }
}

// enc.lsnr_fc: Linear(emb -> 1) + Sigmoid, scaled to [lsnr_min, lsnr_max] (deepfilternet3.py:163-165,184).  One wave per row.
__global__ void dfx_k_lsnr(const float *emb, const float *w, float bias, float scale, float offset, float *out, int64_t R,
                           int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= R) return;  // whole waves exit together (blockDim is a multiple of 64)
    float acc = 0.f;
    const float *e = emb + row * D;
    if ((D & 255) == 0 && ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        // a row of 256 values is one 16-byte load per lane (four 4-byte loads per lane moved the 260 MB of emb at 1.5 TB/s)
        for (int i = 4 * lane; i < D; i += 256) {
            const float4 ev = *reinterpret_cast<const float4 *>(e + i), wv = *reinterpret_cast<const float4 *>(w + i);
            acc += ev.x * wv.x;
            acc += ev.y * wv.y;
            acc += ev.z * wv.z;
            acc += ev.w * wv.w;
        }
    } else {
        for (int i = lane; i < D; i += 64) acc += e[i] * w[i];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = dfx_sigmoid(acc + bias) * scale + offset;
}

// the same for the logical rows of a row-mapped launch (the new frames of a streaming window)
__global__ void dfx_k_lsnr_rows(const float *emb, const float *w, float bias, float scale, float offset, float *out, int64_t R, int D, DfxRowMap rm) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lrow >= R) return;
    const int64_t row = dfx_row(rm, lrow);
    float acc = 0.f;
    const float *e = emb + row * D;
    for (int i = 4 * lane; i < D; i += 256) {
        const float4 ev = *reinterpret_cast<const float4 *>(e + i), wv = *reinterpret_cast<const float4 *>(w + i);
        acc += ev.x * wv.x;
        acc += ev.y * wv.y;
        acc += ev.z * wv.z;
        acc += ev.w * wv.w;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = dfx_sigmoid(acc + bias) * scale + offset;
}

__global__ void dfx_k_add(const float *a, const float *b, float *out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// GRU recurrence (torch.nn.GRU semantics, gate order r,z,n; modules.py:721), hidden size 256.
//   gi [B, T, 768] already holds W_ih x + b_ih (+ b_hr, b_hz folded in);   per step:
//   r = s(gi_r + W_hr h), z = s(gi_z + W_hz h), n = tanh(gi_n + r*(W_hn h + b_hn)), h' = (1-z)*n + z*h
// Clips are independent, so a workgroup owns DFX_GRU_ROWS = 2 clips for all T steps: no inter-workgroup synchronisation,
// B/2 workgroups (128 CUs at batch 256).  The step is a latency chain of T*layers = 5010 iterations, so the design goal is
// the shortest possible step.  W_hh (768 KB fp32) exceeds what one CU can hold (512 KB of VGPRs + 160 KB LDS), so it is
// split three ways per thread and only the remainder is re-read from L2 every step:
//   512 threads; thread (j = tid>>1, kh = tid&1) owns hidden unit j for the 32 k4-blocks {2*i + kh} (a k4-block = 4
//   consecutive k of the three gates = 12 floats).  Blocks [0,KR) live in VGPRs for the whole kernel (168 registers),
//   blocks [KR,KR+KL) in LDS (144 KB), the last KS blocks are streamed from L2 through a two-deep register ring that is
//   refilled across the step boundary (the weights do not depend on t).  h lives in LDS (double buffered, one barrier per
//   step) and is read as wave-broadcast float4s; the two k-halves of a unit are adjacent lanes and are combined with
//   one DPP shuffle per accumulator; lane kh then finishes row kh (gates, h', store).
// Per step and CU: 288 KB from L2, 1536 FMA wave-instructions per SIMD, 1 barrier.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_GRU_H 256
#define DFX_GRU_ROWS 2
#define DFX_GRU_THREADS 512
#define DFX_GRU_KR 14
#define DFX_GRU_KL 6
#define DFX_GRU_KS (32 - DFX_GRU_KR - DFX_GRU_KL)
#define DFX_GRU_SMEM ((size_t)DFX_GRU_KL * 3 * DFX_GRU_THREADS * 16 + (size_t)2 * DFX_GRU_ROWS * DFX_GRU_H * 4)

__global__ void __launch_bounds__(DFX_GRU_THREADS, 2) dfx_k_gru_rec(const float *gi, const float4 *__restrict__ whh4,
                                                                    const float *bhn, const float *h_in, float *h_out,
                                                                    float *y, int64_t B, int64_t T) {
    constexpr int H = DFX_GRU_H, MR = DFX_GRU_ROWS, KR = DFX_GRU_KR, KL = DFX_GRU_KL, KS = DFX_GRU_KS, NT = DFX_GRU_THREADS;
    static_assert(KS >= 2 && KS % 2 == 0 && MR == 2, "ring of two single-block buffers, lane pair = row pair");
    DFX_DYN_SMEM(unsigned char, smraw);
    float4 *wl = reinterpret_cast<float4 *>(smraw);                            // [KL][3][NT]
    float *hs = reinterpret_cast<float *>(smraw + (size_t)KL * 3 * NT * 16);   // [2][MR][H]
    const int tid = threadIdx.x, j = tid >> 1, kh = tid & 1;
    const int64_t b0 = (int64_t)blockIdx.x * MR;
    // float4 index of this thread's k4-block i, gate g inside whh4 ([k4][gate][j] float4, k4 = 2*i + kh)
#define DFX_GRU_WIDX(i, g) (((2 * (i) + kh) * 3 + (g)) * H + j)
    float4 wr[KR][3];
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int g = 0; g < 3; ++g) wr[k][g] = whh4[DFX_GRU_WIDX(k, g)];
    for (int k = 0; k < KL; ++k)
        for (int g = 0; g < 3; ++g) wl[(k * 3 + g) * NT + tid] = whh4[DFX_GRU_WIDX(KR + k, g)];
    const float bn = bhn[j];
    const bool valid = b0 + kh < B;
    const int64_t brow = valid ? b0 + kh : B - 1;
    hs[kh * H + j] = h_in ? h_in[brow * H + j] : 0.f;
    __syncthreads();
    int cur = 0;
    const float4 *ws = whh4 + DFX_GRU_WIDX(KR + KL, 0);  // streamed block s, gate g: ws[(s*6 + g)*H]
    const float *gp = gi + brow * T * (3 * H) + j;
    float *yp = y + brow * T * H + j;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (T > 0) {
        gr = gp[0];
        gz = gp[H];
        gn = gp[2 * H];
    }
    float4 sA[3], sB[3];
#define DFX_GRU_ISSUE(BUF, S) \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) BUF[g] = wst[((S) * 6 + g) * H];
#define DFX_GRU_BLOCK(W0, W1, W2, I)                                                                       \
    _Pragma("unroll") for (int r = 0; r < MR; ++r) {                                                       \
        const float4 hv = *reinterpret_cast<const float4 *>(hc + r * H + 4 * (2 * (I) + kh));            \
        ar[r] = fmaf(W0.w, hv.w, fmaf(W0.z, hv.z, fmaf(W0.y, hv.y, fmaf(W0.x, hv.x, ar[r]))));           \
        az[r] = fmaf(W1.w, hv.w, fmaf(W1.z, hv.z, fmaf(W1.y, hv.y, fmaf(W1.x, hv.x, az[r]))));           \
        an[r] = fmaf(W2.w, hv.w, fmaf(W2.z, hv.z, fmaf(W2.y, hv.y, fmaf(W2.x, hv.x, an[r]))));           \
    }
    {
        const float4 *wst = ws;
        DFX_GRU_ISSUE(sA, 0)
    }
    for (int64_t t = 0; t < T; ++t) {
        float ar[MR], az[MR], an[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) ar[r] = az[r] = an[r] = 0.f;
        const float *hc = hs + cur * MR * H;
        int zoff = 0;
        DFX_OPAQUE(zoff);  // an opaque zero: keeps the streamed weight loads inside the time loop (they are loop invariant)
        const float4 *wst = ws + zoff;
        // next step's input projection (independent of the recurrence)
        const int64_t tn = t + 1 < T ? t + 1 : t;
        const float ngr = gp[tn * 3 * H], ngz = gp[tn * 3 * H + H], ngn = gp[tn * 3 * H + 2 * H];
        DFX_GRU_ISSUE(sB, 1)
        DFX_SCHED_BARRIER();
#pragma unroll
        for (int k = 0; k < KR; ++k) { DFX_GRU_BLOCK(wr[k][0], wr[k][1], wr[k][2], k) }
        constexpr int per = (KL + KS - 1) / KS;  // LDS-resident blocks interleaved per streamed block
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            DFX_SCHED_BARRIER();
            DFX_GRU_BLOCK(sA[0], sA[1], sA[2], KR + KL + s)
            DFX_SCHED_BARRIER();
            if (s + 2 < KS) { DFX_GRU_ISSUE(sA, s + 2) } else { DFX_GRU_ISSUE(sA, 0) }  // wraps into the next step
#pragma unroll
            for (int k = s * per; k < (s + 1) * per && k < KL; ++k) {
                const float4 w0 = wl[(k * 3 + 0) * NT + tid], w1 = wl[(k * 3 + 1) * NT + tid], w2 = wl[(k * 3 + 2) * NT + tid];
                DFX_GRU_BLOCK(w0, w1, w2, KR + k)
            }
            DFX_SCHED_BARRIER();
            DFX_GRU_BLOCK(sB[0], sB[1], sB[2], KR + KL + s + 1)
            DFX_SCHED_BARRIER();
            if (s + 3 < KS) { DFX_GRU_ISSUE(sB, s + 3) }
#pragma unroll
            for (int k = (s + 1) * per; k < (s + 2) * per && k < KL; ++k) {
                const float4 w0 = wl[(k * 3 + 0) * NT + tid], w1 = wl[(k * 3 + 1) * NT + tid], w2 = wl[(k * 3 + 2) * NT + tid];
                DFX_GRU_BLOCK(w0, w1, w2, KR + k)
            }
        }
        // combine the two k-halves of each unit (adjacent lanes), then lane kh finishes row kh
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            ar[r] += __shfl_xor(ar[r], 1);
            az[r] += __shfl_xor(az[r], 1);
            an[r] += __shfl_xor(an[r], 1);
        }
        const float sr = kh ? ar[1] : ar[0], sz = kh ? az[1] : az[0], sn = kh ? an[1] : an[0];
        const float rg = dfx_sigmoid(gr + sr);
        const float zg = dfx_sigmoid(gz + sz);
        const float ng = tanhf(gn + rg * (sn + bn));
        const float hn = (1.f - zg) * ng + zg * hc[kh * H + j];
        hs[(cur ^ 1) * MR * H + kh * H + j] = hn;
        if (valid) yp[t * H] = hn;
        gr = ngr;
        gz = ngz;
        gn = ngn;
        __syncthreads();
        cur ^= 1;
    }
    if (h_out && valid) h_out[brow * H + j] = hs[cur * MR * H + kh * H + j];
#undef DFX_GRU_WIDX
#undef DFX_GRU_ISSUE
#undef DFX_GRU_BLOCK
}

// ---------------------------------------------------------------------------------------------------------------------
// GRU recurrence on the fp16-split matrix path (default; DFX_EXACT_FP32=1 selects dfx_k_gru_rec above).
// The per-step product h[16 rows, 256] x W_hh^T[256, 768] runs on v_mfma_f32_16x16x32_f16 as lo*hi + hi*lo + hi*hi
// (see dfx_k_proj256_h3 for the numerics), which makes 16 clips per workgroup as cheap per step as 2 were on the VALU:
// a layer needs B/16 workgroups (16 CUs at batch 256 instead of 128) and the chain is bound by the W_hh bytes that do
// not fit on the CU, not by arithmetic.
//   256 threads = 4 waves; wave w owns hidden units [64w, 64w+64) for all three gates = 12 MFMA tiles (gate g, sub-tile s).
//   Transposed roles: A operand = W_hh fragment (lane (unit, kgrp): 8 consecutive k), B operand = h fragment (lane (row,
//   kgrp)) read from an f16 hi/lo copy of h in LDS (row stride 528 B: conflict-free ds_read_b128); D[unit][row] leaves
//   each lane with ONE clip and 4 consecutive units per tile -> float4 gi loads / y stores, and the lane keeps its 16
//   h values in fp32 registers across steps (no fp32 h in LDS).
//   The 96 fragment pairs (hi, lo: 2 KB per pair and wave) a wave consumes per step, in kc-major order, live in three
//   places fixed at compile time: FR pairs in VGPRs, FL pairs in LDS, the rest streamed from L2 through a D-slot register
//   ring that is refilled right after use and wraps into the next step (the weights do not depend on t).
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_GH_ROWS 16
#define DFX_GS_MAX_LAYERS 8   /* layers one persistent dfx_k_gru_seq launch can carry */
#define DFX_GS_MAX_CHUNKS 96  /* time chunks of the persistent GRU phase (the default is 12; without followers 16: finer cuts lose to the hand-overs, profiles/r05_gru_chain.log) */
#ifndef DFX_GH_ABLATE
#define DFX_GH_ABLATE 0  /* dev ablations (tools/dev/gru_h3_bench.hip): 1 no stream refill, 2 no gi loads, 4 no matrix ops, 8 no gate math, 16 no y stores */
#endif
#ifndef DFX_GH_PIN
#define DFX_GH_PIN 1
#endif
#ifndef DFX_GH_NW
#define DFX_GH_NW 4      /* waves per workgroup: 4 (one per SIMD, 512 registers each) or 8 */
#endif
#define DFX_GH_THREADS (64 * DFX_GH_NW)
#define DFX_GH_NS (16 / DFX_GH_NW)          /* 16-unit sub-tiles per gate and wave */
#define DFX_GH_TILES (3 * DFX_GH_NS)        /* accumulator tiles per wave */
#define DFX_GH_NF (8 * DFX_GH_TILES)        /* fragment pairs a wave consumes per step (8 k-chunks) */
#ifndef DFX_GH_FR
#define DFX_GH_FR 27     /* pairs per wave resident in registers (33 with a 4-slot ring measured 0.2 ms slower per step under load) */
#endif
#ifndef DFX_GH_FL
#define DFX_GH_FL 15     /* pairs per wave resident in LDS */
#endif
#define DFX_GH_FS (DFX_GH_NF - DFX_GH_FR - DFX_GH_FL)
#ifndef DFX_GH_D
#define DFX_GH_D 6       /* ring slots for the streamed pairs */
#endif
#define DFX_GH_HROW 264  /* halves per row of the f16 copy of h (256 + 8 pad) */
#define DFX_GH_HROW32 260 /* floats per row of the fp32 copy of h of the exact form (X32): 260 = 4 mod 64 words, so the 16 rows a ds_read_b128 pass touches hit 64 different banks */
// Consumption order of the fragments: position f -> (k-chunk, gate, sub-tile), kc-major over all 3*NS tiles.  (Walking the sub-tiles in two
// halves with the first half's gate math issued between the second half's matrix ops — one matrix op : three VALU through
// sched_group_barrier — was built and measured +0.2 ms per step: one wave per SIMD does not overlap the two pipes that way.)
#define DFX_GH_POS_KC(f) ((f) / DFX_GH_TILES)
#define DFX_GH_POS_TT(f) ((f) % DFX_GH_TILES)
#define DFX_GH_POS_GATE(f) (((f) % DFX_GH_TILES) / DFX_GH_NS)
#define DFX_GH_POS_S(f) (((f) % DFX_GH_TILES) % DFX_GH_NS)
#define DFX_GH_POS_TILE(f) (DFX_GH_POS_GATE(f) * DFX_GH_NS + DFX_GH_POS_S(f))
#define DFX_GH_SMEM_W ((size_t)DFX_GH_FL * DFX_GH_NW * 2 * 64 * 16)
#define DFX_GH_SMEM (DFX_GH_SMEM_W + (size_t)2 * 2 * DFX_GH_ROWS * DFX_GH_HROW * 2)

struct DfxGhSched {
    int cls[DFX_GH_NF];   // 0 = register, 1 = LDS, 2 = streamed
    int idx[DFX_GH_NF];   // index inside its class
    int spos[DFX_GH_FS];  // fragment position of streamed fragment i
};
// uniform interleave: the resident fragments consumed between two streamed ones cover the stream's latency.
// Round 5, three forms aimed at the recurrence's first-touch gi reads (7.4 us per step under load, 5.7 when they hit in L2: profiles/r04_gi_warm.log),
// built, measured and not kept (profiles/r05_gru_chain.log): (1) a step that OPENS with a run of 12 / 20 / 28 resident fragments, so that the ring
// refills queued behind the gi loads are not needed before those have returned (vector loads return in order): 5.15 -> 5.6-5.8 us per step alone,
// 7.4 -> 7.7-7.9 under load — the streamed part gets denser than the CU's vector-memory path delivers; (2) the gi rows of TWO steps requested
// together every other step (a second register set: 256 + 254 registers, no scratch): 5.5 alone, 7.6-7.7 under load; (3) non-temporal gi loads /
// y stores here and non-temporal gi stores / x loads in the input projection: 6.0 alone, 7.8 under load.  The cost of those reads is not an
// in-order bubble a schedule can hide: it is the L2 capacity they and the side traffic take from the streamed W_hh ring.
static constexpr DfxGhSched dfx_gh_make_sched() {
    DfxGhSched sc{};
    constexpr int R = DFX_GH_FR + DFX_GH_FL, FS = DFX_GH_FS;
    int p = 0, nres = 0;
    for (int st = 0; st < FS; ++st) {
        const int hi = (st + 1) * R / FS;
        for (; nres < hi; ++nres, ++p) {
            // spread the LDS-resident fragments evenly among the register-resident ones
            const bool lds = ((nres + 1) * DFX_GH_FL / R) != (nres * DFX_GH_FL / R);
            sc.cls[p] = lds ? 1 : 0;
            sc.idx[p] = lds ? nres * DFX_GH_FL / R : nres - (nres * DFX_GH_FL / R + (lds ? 0 : 0));
        }
        sc.cls[p] = 2;
        sc.idx[p] = st;
        sc.spos[st] = p;
        ++p;
    }
    // renumber the register class densely
    int nr = 0;
    for (int i = 0; i < DFX_GH_NF; ++i)
        if (sc.cls[i] == 0) sc.idx[i] = nr++;
    return sc;
}

struct DfxGhArgs {
    const float *gi;      // [B, T, 768]
    const dfx_h8 *whf;    // [16 unit tiles][8 k-chunks][3 gates][hi, lo][64 lanes], pre-scaled (layout independent of NW)
    const float *bhn;     // [256]
    const float *h_in;    // [B, 256] or null
    float *h_out;         // [B, 256] or null
    float *y;             // [B, T, 256]
    int64_t B, T;
    int64_t t0, t1;       // steps [t0, t1) of the T frames (time-chunked launches carry h through h_in / h_out)
    float unscale;
    int xcd_mask;         // != 0: the grid is oversized 8x / popcount and only blocks that find themselves on an XCD of the mask work
                          // (the k-th block of an XCD is blockIdx >> 3 under the round-robin dispatch): a layer's workgroups then
                          // share few L2s, and L2s that other kernels can be kept away from (placement = speed only)
};

// Chunk-level synchronisation of the persistent form (dfx_k_gru_seq): the time axis is cut into K chunks; chunk k of a layer may run once
// the layer's input projection of that chunk exists (ready >= base + k + 1, written by a flag kernel behind the projection on its
// stream) and is announced to the consumers when it is complete (done[group] = base + k + 1).  Flags are monotonic over the life of the
// model (no resets), compared as signed differences; every spin is bounded (a timeout sets err[2] and lets the kernel run on).
struct DfxGhSync {
    const unsigned int *ready;   // one word: chunks of gi available for this layer
    unsigned int *done;          // [groups]: chunks of y completed by each workgroup of this layer
    unsigned int base;
    int K;
    const int *tb;               // [K + 1] chunk boundaries in frames (short chunks at both ends fill and drain the layer pipeline quickly)
    unsigned int *err;
    unsigned long long *trace;   // dev aid (DFX_SEQ_TRACE=1): [K][3] wall-clock ticks of this workgroup: wait begin, compute begin, chunk end
    int spin_limit;              // polls before a wait gives up and raises err[2]
    // block-granular hand-over to / from a follower workgroup (dfx_k_proj_follow), null = none.  Both words count STEPS of this 16-clip group
    // (pbase + steps): yprog is raised by this workgroup every `sblk` steps (and at the end of the sequence) for the follower of the layer above;
    // giprog is raised by this layer's own follower — this workgroup then does not wait for `ready` (no host-launched projection feeds it).
    unsigned int *yprog = nullptr;
    const unsigned int *giprog = nullptr;
    unsigned int pbase = 0;
    int sblk = 16;               // steps per block of this layer's follower (a power of two)
    int yblk = 16;               // steps between two yprog announcements (= the block of the consumer's follower)
    // Same-XCD hand-overs (DfxXcd): a producer and a consumer that share an L2 need neither the L2 write-back of an agent-scope release nor the L2
    // invalidate of an agent-scope acquire — the invalidate alone throws the XCD's share of the streamed W_hh out of the L2 every 16 steps of every
    // recurrence on it (measured with both left out: 7.9 -> 7.3 us per step).  Every party registers the XCD it runs on; a hand-over whose partner
    // is registered on the same XCD drains its stores (s_waitcnt vmcnt(0) in every wave in front of the barrier), raises the flag with a plain
    // device-scope store, and the consumer invalidates its L1 only.  Anything else — partner elsewhere, or not registered yet — takes the full form.
    DfxXcd x;
};
#define DFX_SYNC_SPIN_LIMIT (1 << 22)   /* default bound of every flag wait: polls with s_sleep, ~2 s (dfx_model::spin_limit, DFX_SYNC_SPIN_LIMIT) */

// X32 (DFX_EXACT_FP32=1): the same kernel on exact fp32 matrix ops.  A fragment "pair" holds the same 32 bytes per lane — the eight weights
// W[unit][32 kc + 8 q + 0..7] as fp32 instead of their f16 hi / lo halves — and feeds eight v_mfma_f32_16x16x4_f32 (op j contracts the
// k-set {32 kc + 8 q + j}: A = the lane's j-th weight, B = the j-th of the eight consecutive h values the lane reads from an fp32 copy of h
// in LDS); same residency plan, same ring, same accumulator layout.  768 matrix ops of 32 cycles per step and wave: 10.2 us per step —
// twice the VALU kernel dfx_k_gru_rec's 5.0, but on 16 CUs per layer instead of 128, so all layers fit the persistent, layer-pipelined phase.
template <bool SEQ, bool X32 = false>
static __device__ __forceinline__ void dfx_gru_h3_run(const DfxGhArgs &A, int64_t grp, const DfxGhSync &Y) {
    constexpr int H = 256, FR = DFX_GH_FR, FS = DFX_GH_FS, NF = DFX_GH_NF, D = DFX_GH_D, HROW = DFX_GH_HROW;
    constexpr int NW = DFX_GH_NW, NS = DFX_GH_NS, TILES = DFX_GH_TILES, UW = 16 * NS;  // UW = units per wave
    static_assert(FS % D == 0 && FS >= D, "ring slots must line up across the step boundary");
    constexpr DfxGhSched SC = dfx_gh_make_sched();
    DFX_DYN_SMEM(unsigned char, smraw);
    dfx_h8 *wl = reinterpret_cast<dfx_h8 *>(smraw);                           // [FL][wave][hi,lo][lane]
    uint16_t *h16 = reinterpret_cast<uint16_t *>(smraw + DFX_GH_SMEM_W);      // [buf][hi,lo][16][HROW]
    float *h32 = reinterpret_cast<float *>(smraw + DFX_GH_SMEM_W);            // X32: [buf][16][HROW32]
    static_assert((size_t)2 * DFX_GH_ROWS * DFX_GH_HROW32 * 4 <= (size_t)2 * 2 * DFX_GH_ROWS * DFX_GH_HROW * 2, "the fp32 copy of h fits where the f16 copies live");
    // byte geometry of the B-operand reads: a k-chunk's 16 bytes for the `hi` slot, the partner 16 bytes for the `lo` slot
    constexpr size_t HB_KSTEP = X32 ? 32 * 4 : 32 * 2, HB_LO = X32 ? 16 : (size_t)DFX_GH_ROWS * HROW * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    const int64_t b0 = grp * DFX_GH_ROWS;
    const bool valid = b0 + jl < A.B;
    const int64_t brow = valid ? b0 + jl : A.B - 1;
    // fragment f = kc*TILES + gate*NS + s of this wave is global pair ((unit tile = wave*NS + s)*8 + kc)*3 + gate
    const dfx_h8 *wg = A.whf + lane;
#define DFX_GH_GIDX(f) (((((size_t)wave * NS + DFX_GH_POS_S(f)) * 8 + DFX_GH_POS_KC(f)) * 3 + DFX_GH_POS_GATE(f)) * 2 * 64)
    // ---- resident fragments
    dfx_h8 wr[FR][2];
    dfx_static_for<0, NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        if constexpr (SC.cls[f] == 0) {
            wr[SC.idx[f]][0] = wg[DFX_GH_GIDX(f)];
            wr[SC.idx[f]][1] = wg[DFX_GH_GIDX(f) + 64];
            // The resident fragments live in the accumulation half of the register file, where the matrix ops read them directly: left to the
            // allocator they were parked there anyway and copied back before every use — 190 v_accvgpr_read among the 619 vector instructions
            // of a step (0 of 428 with the pin).  (Not the exact form: its matrix ops take single registers of a fragment and the pin spills.)
            if constexpr (DFX_GH_PIN && !X32) {
                DFX_PIN_AGPR(wr[SC.idx[f]][0]);
                DFX_PIN_AGPR(wr[SC.idx[f]][1]);
            }
        } else if constexpr (SC.cls[f] == 1) {
            wl[((SC.idx[f] * NW + wave) * 2 + 0) * 64 + lane] = wg[DFX_GH_GIDX(f)];
            wl[((SC.idx[f] * NW + wave) * 2 + 1) * 64 + lane] = wg[DFX_GH_GIDX(f) + 64];
        }
    });
    // ---- state: this lane owns clip jl, units 64*wave + 16*s + 4*q + r
    float hp[NS][4];
    float4 bn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int u0 = UW * wave + 16 * s + 4 * q;
        bn[s] = *reinterpret_cast<const float4 *>(A.bhn + u0);
        float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (A.h_in) h0 = *reinterpret_cast<const float4 *>(A.h_in + brow * H + u0);
        hp[s][0] = h0.x, hp[s][1] = h0.y, hp[s][2] = h0.z, hp[s][3] = h0.w;
    }
    auto put_h16 = [&](int buf, int s) {  // f16 hi/lo of hp[s][0..3] -> LDS (8 bytes each); X32: the four fp32 values
        if constexpr (X32) {
            *reinterpret_cast<float4 *>(h32 + ((size_t)buf * DFX_GH_ROWS + jl) * DFX_GH_HROW32 + UW * wave + 16 * s + 4 * q) =
                make_float4(hp[s][0], hp[s][1], hp[s][2], hp[s][3]);
            return;
        }
        uint16_t hh[4], hl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hh[r] = dfx_f32_to_f16_bits(hp[s][r]);
            hl[r] = dfx_f32_to_f16_bits(hp[s][r] - dfx_f16_bits_to_f32(hh[r]));
        }
        const int col = UW * wave + 16 * s + 4 * q;
        uint16_t *ph = h16 + ((size_t)(buf * 2 + 0) * DFX_GH_ROWS + jl) * HROW + col;
        uint16_t *pl = h16 + ((size_t)(buf * 2 + 1) * DFX_GH_ROWS + jl) * HROW + col;
        *reinterpret_cast<uint2 *>(ph) = make_uint2((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16));
        *reinterpret_cast<uint2 *>(pl) = make_uint2((uint32_t)hl[0] | ((uint32_t)hl[1] << 16), (uint32_t)hl[2] | ((uint32_t)hl[3] << 16));
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) put_h16(0, s);
    __syncthreads();
    // ---- streamed ring
    dfx_h8 ring[D][2];
    dfx_static_for<0, D>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        ring[d][0] = wg[DFX_GH_GIDX(SC.spos[d])];
        ring[d][1] = wg[DFX_GH_GIDX(SC.spos[d]) + 64];
    });
    const float *gp = A.gi + brow * A.T * (3 * H) + UW * wave + 4 * q;
    float *yp = A.y + brow * A.T * H + UW * wave + 4 * q;
    int cur = 0;
    // the input projection of a step (3 gates x 4 sub-tiles per lane) is loaded one step ahead: each sub-tile's registers are
    // refilled for step t+1 as soon as the gate math of step t has consumed them, so the loads land during the matrix phase
    float4 gv[3][NS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < NS; ++s) gv[g][s] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
    // (follower-fed layer) the gi rows of steps < upto must exist before they are requested
    auto wait_gi = [&](int64_t upto) {
        if (tid == 0) {
            const unsigned int want = Y.pbase + (unsigned int)(upto < A.T ? upto : A.T);
            int spins = 0;
            while ((int)(__hip_atomic_load(Y.giprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > Y.spin_limit) {
                    dfx_raise(Y.err + 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        dfx_xcd_acquire(Y.x);
    };
    if (SEQ) dfx_xcd_register(Y.x);
    const int nchunk = SEQ ? Y.K : 1;
    const bool dead = false;
    for (int ck = 0; ck < nchunk; ++ck) {
    const int64_t c0 = SEQ ? (int64_t)Y.tb[ck] : A.t0, c1 = SEQ ? (int64_t)Y.tb[ck + 1] : A.t1;
    if (SEQ) {   // the input projection of this chunk must exist
        if (tid == 0 && Y.trace) Y.trace[ck * 3 + 0] = wall_clock64();
        if (tid == 0 && !dead && !Y.giprog) {
            const unsigned int want = Y.base + (unsigned int)ck + 1u;
            int spins = 0;
            while ((int)(__hip_atomic_load(Y.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > Y.spin_limit) {
                    dfx_raise(Y.err + 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (Y.giprog && ck == 0) wait_gi(Y.sblk);   // (later blocks: one step before their first row is requested, below)
        if (tid == 0 && Y.trace) Y.trace[ck * 3 + 1] = wall_clock64();
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (c1 > c0 && !(DFX_GH_ABLATE & 2)) gv[g][s] = *reinterpret_cast<const float4 *>(gp + c0 * (3 * H) + g * H + 16 * s);
    for (int64_t t = c0; t < c1; ++t) {
        if (SEQ && Y.giprog && ((t + 1) & (Y.sblk - 1)) == 0 && t + 1 < A.T) wait_gi(t + 1 + Y.sblk);   // this step requests the first row of the next block
        int zoff = 0;
        DFX_OPAQUE(zoff);  // keeps the (loop-invariant) streamed weight loads inside the time loop
        const dfx_h8 *wst = wg + zoff;
        const int64_t tn = t + 1 < c1 ? t + 1 : t;
        const unsigned char *hb = X32 ? reinterpret_cast<const unsigned char *>(h32 + ((size_t)cur * DFX_GH_ROWS + jl) * DFX_GH_HROW32 + 8 * q)
                                      : reinterpret_cast<const unsigned char *>(h16 + (size_t)(cur * 2) * DFX_GH_ROWS * HROW + (size_t)jl * HROW + 8 * q);
        f32x4 acc[TILES];
#pragma unroll
        for (int i = 0; i < TILES; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dfx_h8 bh[2], bl[2];
        bh[0] = *reinterpret_cast<const dfx_h8 *>(hb);
        bl[0] = *reinterpret_cast<const dfx_h8 *>(hb + HB_LO);
        // ---- gates, new state (lane: clip jl, units UW*w + 16s + 4q + r)
        auto gate_unit = [&](int s, int r) {
            const float gr = r == 0 ? gv[0][s].x : r == 1 ? gv[0][s].y : r == 2 ? gv[0][s].z : gv[0][s].w;
            const float gz = r == 0 ? gv[1][s].x : r == 1 ? gv[1][s].y : r == 2 ? gv[1][s].z : gv[1][s].w;
            const float gn = r == 0 ? gv[2][s].x : r == 1 ? gv[2][s].y : r == 2 ? gv[2][s].z : gv[2][s].w;
            const float bb = r == 0 ? bn[s].x : r == 1 ? bn[s].y : r == 2 ? bn[s].z : bn[s].w;
            if (DFX_GH_ABLATE & 8) {
                hp[s][r] = 0.5f * hp[s][r] + 1e-3f * (gr + gz + gn + bb + acc[s][r] + acc[NS + s][r] + acc[2 * NS + s][r]);
                return;
            }
            const float rg = dfx_fast_rcp(1.f + dfx_fast_exp(-(gr + acc[0 * NS + s][r] * A.unscale)));
            const float zg = dfx_fast_rcp(1.f + dfx_fast_exp(-(gz + acc[1 * NS + s][r] * A.unscale)));
            const float pre = gn + rg * (acc[2 * NS + s][r] * A.unscale + bb);
            const float ng = 2.f * dfx_fast_rcp(1.f + dfx_fast_exp(-2.f * pre)) - 1.f;
            hp[s][r] = (1.f - zg) * ng + zg * hp[s][r];
        };
        auto gate_finish = [&](int s) {   // the sub-tile's four units are done: next step's gi, y, the f16 copy of h for the next step
            if (!(DFX_GH_ABLATE & 2)) {
#pragma unroll
                for (int g = 0; g < 3; ++g) gv[g][s] = *reinterpret_cast<const float4 *>(gp + tn * (3 * H) + g * H + 16 * s);
            }
            if (valid && !(DFX_GH_ABLATE & 16)) *reinterpret_cast<float4 *>(yp + t * H + 16 * s) = make_float4(hp[s][0], hp[s][1], hp[s][2], hp[s][3]);
            put_h16(cur ^ 1, s);
        };
        // three fragments (= three different accumulator tiles of one k-chunk) per group: the 9 MFMAs are issued so that
        // consecutive ones never touch the same accumulator (a dependent 16x16x32 MFMA would wait for its predecessor)
        dfx_static_for<0, NF / 3>([&](auto gc) {
            constexpr int grp = decltype(gc)::value, f0 = 3 * grp;
            constexpr int kc = DFX_GH_POS_KC(f0), tt0 = DFX_GH_POS_TT(f0);
            // next k-chunk of h, half a chunk ahead
            if constexpr (tt0 == (DFX_GH_TILES / 6) * 3 && kc + 1 < 8) {
                constexpr int kn = kc + 1;
                bh[kn & 1] = *reinterpret_cast<const dfx_h8 *>(hb + HB_KSTEP * kn);
                bl[kn & 1] = *reinterpret_cast<const dfx_h8 *>(hb + HB_LO + HB_KSTEP * kn);
            }
            dfx_h8 whi[3], wlo[3];
            dfx_static_for<0, 3>([&](auto ic) {
                constexpr int i = decltype(ic)::value, f = f0 + i;
                if constexpr (SC.cls[f] == 0) {
                    whi[i] = wr[SC.idx[f]][0];
                    wlo[i] = wr[SC.idx[f]][1];
                } else if constexpr (SC.cls[f] == 1) {
                    whi[i] = wl[((SC.idx[f] * NW + wave) * 2 + 0) * 64 + lane];
                    wlo[i] = wl[((SC.idx[f] * NW + wave) * 2 + 1) * 64 + lane];
                } else {
                    whi[i] = ring[SC.idx[f] % D][0];
                    wlo[i] = ring[SC.idx[f] % D][1];
                }
            });
            constexpr int ta = DFX_GH_POS_TILE(f0), tb_ = DFX_GH_POS_TILE(f0 + 1), tc = DFX_GH_POS_TILE(f0 + 2);
            if constexpr (X32) {
                const f32x4 b0 = __builtin_bit_cast(f32x4, bh[kc & 1]), b1 = __builtin_bit_cast(f32x4, bl[kc & 1]);
                const f32x4 a0[3] = {__builtin_bit_cast(f32x4, whi[0]), __builtin_bit_cast(f32x4, whi[1]), __builtin_bit_cast(f32x4, whi[2])};
                const f32x4 a1[3] = {__builtin_bit_cast(f32x4, wlo[0]), __builtin_bit_cast(f32x4, wlo[1]), __builtin_bit_cast(f32x4, wlo[2])};
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // k ascending inside the chunk; the three accumulators alternate
                    acc[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][j], b0[j], acc[ta], 0, 0, 0);
                    acc[tb_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][j], b0[j], acc[tb_], 0, 0, 0);
                    acc[tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2][j], b0[j], acc[tc], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0][j], b1[j], acc[ta], 0, 0, 0);
                    acc[tb_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1][j], b1[j], acc[tb_], 0, 0, 0);
                    acc[tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[2][j], b1[j], acc[tc], 0, 0, 0);
                }
            } else if (!(DFX_GH_ABLATE & 4)) {
                acc[ta] = dfx_mfma_16x16x32_f16(wlo[0], bh[kc & 1], acc[ta]);
                acc[tb_] = dfx_mfma_16x16x32_f16(wlo[1], bh[kc & 1], acc[tb_]);
                acc[tc] = dfx_mfma_16x16x32_f16(wlo[2], bh[kc & 1], acc[tc]);
                acc[ta] = dfx_mfma_16x16x32_f16(whi[0], bl[kc & 1], acc[ta]);
                acc[tb_] = dfx_mfma_16x16x32_f16(whi[1], bl[kc & 1], acc[tb_]);
                acc[tc] = dfx_mfma_16x16x32_f16(whi[2], bl[kc & 1], acc[tc]);
                acc[ta] = dfx_mfma_16x16x32_f16(whi[0], bh[kc & 1], acc[ta]);
                acc[tb_] = dfx_mfma_16x16x32_f16(whi[1], bh[kc & 1], acc[tb_]);
                acc[tc] = dfx_mfma_16x16x32_f16(whi[2], bh[kc & 1], acc[tc]);
            } else {
                acc[ta][0] += (float)whi[0][0] + (float)wlo[0][1] + (float)bh[kc & 1][0] + (float)bl[kc & 1][0];
                acc[tb_][0] += (float)whi[1][0] + (float)wlo[1][1] + (float)bh[kc & 1][0] + (float)bl[kc & 1][0];
                acc[tc][0] += (float)whi[2][0] + (float)wlo[2][1] + (float)bh[kc & 1][0] + (float)bl[kc & 1][0];
            }
            DFX_SCHED_BARRIER();
            dfx_static_for<0, 3>([&](auto ic) {
                constexpr int f = f0 + decltype(ic)::value;
                if constexpr (SC.cls[f] == 2 && !(DFX_GH_ABLATE & 1)) {
                    constexpr int nxt = SC.spos[(SC.idx[f] + D) % FS];  // wraps into the next step
                    ring[SC.idx[f] % D][0] = wst[DFX_GH_GIDX(nxt)];
                    ring[SC.idx[f] % D][1] = wst[DFX_GH_GIDX(nxt) + 64];
                }
            });
            DFX_SCHED_BARRIER();
        });
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gate_unit(s, r);
            gate_finish(s);
        }
        const bool ypub = SEQ && Y.yprog && (((t + 1) & (Y.yblk - 1)) == 0 || t + 1 == A.T);   // a block of y rows is complete with this step
        if (ypub) DFX_VMEM_DRAIN();
        __syncthreads();
        cur ^= 1;
        if (ypub && tid == 0) dfx_xcd_release(Y.x, Y.yprog, Y.pbase + (unsigned int)(t + 1));
    }
    if (SEQ) {   // this workgroup's rows of chunk ck are complete: make them visible device-wide, then say so
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(Y.done + grp, Y.base + (unsigned int)ck + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0 && Y.trace) Y.trace[ck * 3 + 2] = wall_clock64();
    }
    }
    if (A.h_out && valid) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            *reinterpret_cast<float4 *>(A.h_out + brow * H + UW * wave + 16 * s + 4 * q) = make_float4(hp[s][0], hp[s][1], hp[s][2], hp[s][3]);
    }
}
#undef DFX_GH_GIDX

template <bool X32>
static __device__ __forceinline__ void dfx_gru_rec_body(const DfxGhArgs &A) {
    int64_t grp = blockIdx.x;
    if (A.xcd_mask) {
        const int x = dfx_xcc_id();   // the XCD itself, not blockIdx % 8 (equal only up to a per-launch rotation)
        if (!((A.xcd_mask >> x) & 1)) return;
        grp = (int64_t)(blockIdx.x >> 3) * __builtin_popcount(A.xcd_mask) + __builtin_popcount(A.xcd_mask & ((1 << x) - 1));
        if (grp * DFX_GH_ROWS >= A.B) return;
    }
    dfx_gru_h3_run<false, X32>(A, grp, DfxGhSync{nullptr, nullptr, 0u, 1, nullptr, nullptr, nullptr, 0});
}
__global__ void __launch_bounds__(DFX_GH_THREADS, DFX_GH_NW / 4) dfx_k_gru_rec_h3(DfxGhArgs A) { dfx_gru_rec_body<false>(A); }
// whf: fp32 fragments [16 unit tiles][8 k-chunks][3 gates][2 halves][64 lanes] float4 (pack_whh_x32), unscale = 1
__global__ void __launch_bounds__(DFX_GH_THREADS, DFX_GH_NW / 4) dfx_k_gru_rec_x32(DfxGhArgs A) { dfx_gru_rec_body<true>(A); }

// All GRU layers of a forward pass in ONE persistent launch: workgroup (layer l, group g) keeps its share of W_hh on the CU for the
// whole sequence and walks the K time chunks, synchronised with the kernels around it through DfxGhSync flags instead of kernel
// boundaries.  What this removes compared with one launch per (layer, chunk): the launch gaps between chunks (45-90 us each), the wait
// for a completely free CU at every relaunch while background kernels occupy the chip, and the pending cross-queue barrier packets
// that were measured to slow every running kernel (tools/dev/dfa_bench.hip: -22 % with 16 waiting queues).
// Needs all nlayers * groups workgroups co-resident (each owns a CU): the host only uses it when they fit.
struct DfxGsArgs {
    const float *gi[DFX_GS_MAX_LAYERS];
    float *y[DFX_GS_MAX_LAYERS];
    const dfx_h8 *whf[DFX_GS_MAX_LAYERS];
    const float *bhn[DFX_GS_MAX_LAYERS];
    float unscale[DFX_GS_MAX_LAYERS];
    int64_t B, T;
    int nlayers, groups, K;
    int tb[DFX_GS_MAX_CHUNKS + 1];   // chunk boundaries (frames)
    unsigned int *ready;   // [DFX_GS_MAX_LAYERS]
    unsigned int *done;    // [DFX_GS_MAX_LAYERS][done_stride]
    int done_stride;
    unsigned int base;
    unsigned int *err;
    unsigned long long *trace;   // dev aid: [layers][groups][K][3] or null
    int spin_limit;
    unsigned int *yprog[DFX_GS_MAX_LAYERS] = {};         // [groups] each or null (DfxGhSync)
    unsigned int *giprog[DFX_GS_MAX_LAYERS] = {};        // [groups] each or null
    unsigned int pbase = 0;
    int sblk = 16;
    int yblk[DFX_GS_MAX_LAYERS] = {};
    // same-XCD hand-overs: the registration words [kind][layer][xstride]: kind 0 recurrences, 1 projection followers, 2 the emb follower (row 0)
    unsigned int *xtab = nullptr;
    int xstride = 0;
    int xcons_kind[DFX_GS_MAX_LAYERS] = {}, xcons_layer[DFX_GS_MAX_LAYERS] = {};   // who consumes layer l's yprog blocks
    unsigned int xtag = 0;
    unsigned int *xstat = nullptr;
    int pair_far = 0;                // test hook of the pair form: agent-scope hand-overs everywhere
    unsigned int *psync = nullptr;   // pair form (dfx_k_gru_seq_p2, dfx_gru_pair.h): [nlayers * pairs per layer][48] step flags and XCD words of the pairs' halves
};
template <bool X32>
static __device__ __forceinline__ void dfx_gru_seq_body(const DfxGsArgs &S) {
    // block -> (layer, group): consecutive blocks of a layer are dealt round-robin over the XCDs, so every L2 holds a part of every
    // layer's streamed weights (16 groups of a layer = 2 per XCD; confining layers to XCD subsets measured the same: 17.96 vs 18.02 ms)
    const int l = (int)(blockIdx.x / (unsigned)S.groups), g = (int)(blockIdx.x % (unsigned)S.groups);
    if (l >= S.nlayers) return;
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
    DfxGhSync Y{S.ready + l, S.done + (size_t)l * S.done_stride, S.base, S.K, S.tb, S.err,
                S.trace ? S.trace + ((size_t)l * S.groups + g) * S.K * 3 : nullptr, S.spin_limit};
    Y.yprog = S.yprog[l] ? S.yprog[l] + g : nullptr;
    Y.giprog = S.giprog[l] ? S.giprog[l] + g : nullptr;
    Y.pbase = S.pbase, Y.sblk = S.sblk, Y.yblk = S.yblk[l] > 0 ? S.yblk[l] : 16;
    if (S.xtab) {   // (every layer registers: the followers' claims read the encoder layer's row)
        auto word = [&](int kind, int layer) { return S.xtab + ((size_t)kind * DFX_GS_MAX_LAYERS + layer) * S.xstride + g; };
        Y.x.me = word(0, l);
        Y.x.prod = Y.giprog ? word(1, l) : nullptr;
        Y.x.cons = Y.yprog ? word(S.xcons_kind[l], S.xcons_layer[l]) : nullptr;
        Y.x.tag = S.xtag, Y.x.stat = S.xstat;
    }
    dfx_gru_h3_run<true, X32>(A, g, Y);
}
__global__ void __launch_bounds__(DFX_GH_THREADS, DFX_GH_NW / 4) dfx_k_gru_seq(DfxGsArgs S) { dfx_gru_seq_body<false>(S); }
__global__ void __launch_bounds__(DFX_GH_THREADS, DFX_GH_NW / 4) dfx_k_gru_seq_x32(DfxGsArgs S) { dfx_gru_seq_body<true>(S); }

#include "dfx_gru_pair.h"   // the recurrence on a pair of CUs (W_hh fully resident, h exchanged per step through the XCD's L2)

// ---------------------------------------------------------------------------------------------------------------------
// dfx_k_proj_follow: the input projection of a decoder GRU layer as persistent FOLLOWER workgroups of the recurrences instead of one launch
// per time chunk (round 5; the default, DFX_SEQ_FOLLOW).  The layer's input is the output of the layer below (the stacks' second layers) or what the
// emb follower wrote behind the encoder GRU (their first layers: dfx_k_emb_follow).  Workgroup (f, g) serves 16 clips of layer f — the group whose
// recurrence runs on ITS XCD (dfx_xcd_claim), with L2-local hand-overs (DfxXcd) — : it waits until its producer has completed the next block of 16
// steps of those clips (a step counter: yprog / embprog), runs dfx_k_proj256_h3x2<8, 4>'s arithmetic on the block's 256 rows (wave w, tile t: step 2 w + t of the
// block; lane: clip — the same bits, a row's result does not depend on the tiling) and raises the layer's giprog word.  The layer above
// therefore starts 16 steps + one block (~45 us) behind the layer below instead of one time chunk (63-84 steps) + a wait kernel + a
// projection launch (~0.6 ms), and its projections no longer travel through the side streams.  W_ih is streamed from L2 once per block
// through LDS (2 x 64 KB), like the launch form.
// ---------------------------------------------------------------------------------------------------------------------
#define DFX_PF_MAX 4
struct DfxPfArgs {
    const float *x[DFX_PF_MAX];        // [B, T, 256]: y of the layer below
    float *gi[DFX_PF_MAX];             // [B, T, 768]
    const dfx_h8 *wf[DFX_PF_MAX];      // as DfxPhArgs::wf
    const float *bias[DFX_PF_MAX];
    float unscale[DFX_PF_MAX];
    const unsigned int *yprog[DFX_PF_MAX];   // [groups]: steps the layer below has completed (pbase + steps)
    unsigned int *giprog[DFX_PF_MAX];        // [groups]: steps of gi this follower has completed
    int64_t B, T;
    int nf, groups;
    unsigned int pbase;
    unsigned int *err;
    int spin_limit;
    // same-XCD hand-overs (DfxXcd): registration words of this follower's producer / itself / its consumer, [groups] each, or null
    const unsigned int *xprod[DFX_PF_MAX] = {};
    unsigned int *xme[DFX_PF_MAX] = {};
    const unsigned int *xcons[DFX_PF_MAX] = {};
    unsigned int xtag = 0;
    unsigned int *xstat = nullptr;
    const unsigned int *xrec = nullptr;   // registration words of one layer's recurrences [groups] (dfx_xcd_claim) or null: group = block index
    unsigned int *xclaim = nullptr;       // [DFX_PF_MAX][8], zeroed before the launch
};
static __device__ __forceinline__ DfxXcd dfx_pf_xcd(const DfxPfArgs &A, int f, int g) {
    DfxXcd X;
    if (A.xme[f]) X.me = A.xme[f] + g, X.prod = A.xprod[f] ? A.xprod[f] + g : nullptr, X.cons = A.xcons[f] ? A.xcons[f] + g : nullptr, X.tag = A.xtag, X.stat = A.xstat;
    return X;
}
__global__ void __launch_bounds__(512, 1) dfx_k_proj_follow(DfxPfArgs A) {
    constexpr int NW = 8, CT = 4, NTH = 64 * NW, SB = 2 * NW, N = 768;   // SB steps per block
    constexpr int CH8 = 8 * CT * 2 * 64;
    constexpr int PER_T = CH8 / NTH;
    DFX_DYN_SMEM(dfx_h8, ws);  // [2][CH8]
    const int f = (int)(blockIdx.x / (unsigned)A.groups);
    if (f >= A.nf) return;
    const int g = dfx_xcd_claim(A.xrec ? (A.xcons[f] ? A.xcons[f] : A.xrec) : nullptr, A.xtag, A.groups,   // (the registrations of the layer this follower FEEDS: with groups % 8 != 0 the layers sit on different XCDs)
                                A.xclaim ? A.xclaim + 8 * f : nullptr, (int)(blockIdx.x % (unsigned)A.groups), A.spin_limit, A.err,
                                reinterpret_cast<int *>(ws));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    constexpr int nchunks = N / (16 * CT);
    const int T = (int)A.T;
    dfx_xcd_register(dfx_pf_xcd(A, f, g));
    for (int t0 = 0; t0 < T; t0 += SB) {
        const int t1 = t0 + SB < T ? t0 + SB : T;
        int zoff = 0, jz = jl;
        DFX_OPAQUE(zoff);   // (keeps the block-invariant addresses out of registers held across the block loop: no scratch)
        DFX_OPAQUE(jz);
        const dfx_h8 *wf = A.wf[f] + zoff;
        const float *bias = A.bias[f] + zoff;
        const int64_t clip = (int64_t)g * 16 + jz;
        const bool okc = clip < A.B;
        const float *xrow = A.x[f] + (okc ? clip : 0) * A.T * 256 + 8 * q;
        float *orow = A.gi[f] + (okc ? clip : 0) * A.T * N;
        // the first chunk of W does not depend on the producer: stage it in front of the wait (buffer 0 is free: nchunks is even)
#pragma unroll
        for (int i = 0; i < PER_T; ++i) ws[i * NTH + tid] = wf[i * NTH + tid];
        if (tid == 0) {
            const unsigned int want = A.pbase + (unsigned int)t1;
            int spins = 0;
            while ((int)(__hip_atomic_load(A.yprog[f] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > A.spin_limit) {
                    dfx_raise(A.err + 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
        dfx_xcd_acquire(dfx_pf_xcd(A, f, g));
        dfx_h8 xh[2][8], xl[2][8];
        float unscale[2];
        int trow[2];
        bool okr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            trow[t] = t0 + 2 * wave + t;
            okr[t] = okc && trow[t] < t1;
            if (!okr[t]) trow[t] = t0;
            const float4 *p = reinterpret_cast<const float4 *>(xrow + trow[t] * 256);
            float4 xu[8], xv[8];
            float mx = 0.f;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                xu[kc] = okr[t] ? p[8 * kc] : make_float4(0.f, 0.f, 0.f, 0.f);
                xv[kc] = okr[t] ? p[8 * kc + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xu[kc].x), fabsf(xu[kc].y)), fmaxf(fabsf(xu[kc].z), fabsf(xu[kc].w))));
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xv[kc].x), fabsf(xv[kc].y)), fmaxf(fabsf(xv[kc].z), fabsf(xv[kc].w))));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));   // the four lanes (q) that share a row
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            int e = 0;
            if (mx > 0.f && mx < 3.0e38f) {       // per-row power-of-two scale, exactly as in dfx_k_proj256_h3
                int ex;
                (void)frexpf(mx, &ex);
                e = 14 - ex;
                e = e > 100 ? 100 : (e < -100 ? -100 : e);
            }
            const float sc = ldexpf(1.f, e);
            unscale[t] = A.unscale[f] * ldexpf(1.f, -e);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                float x[8];
                x[0] = xu[kc].x * sc, x[1] = xu[kc].y * sc, x[2] = xu[kc].z * sc, x[3] = xu[kc].w * sc;
                x[4] = xv[kc].x * sc, x[5] = xv[kc].y * sc, x[6] = xv[kc].z * sc, x[7] = xv[kc].w * sc;
                dfx_split8(x, xh[t][kc], xl[t][kc]);
            }
        }
        for (int c = 0; c < nchunks; ++c) {
            const dfx_h8 *wc = ws + (size_t)(c & 1) * CH8;
            // the next chunk travels to the other LDS buffer in two halves (16 registers in flight instead of 32: the kernel's 256 are full)
            constexpr int HP = PER_T / 2;
            dfx_h8 pre[HP];
            const dfx_h8 *src = wf + (size_t)(c + 1) * CH8;   // (CT = 4: a chunk is one of the host's 64-column blocks)
            dfx_h8 *dst = ws + (size_t)((c + 1) & 1) * CH8;
            if (c + 1 < nchunks) {
#pragma unroll
                for (int i = 0; i < HP; ++i) pre[i] = src[i * NTH + tid];
            }
            f32x4 acc[2][CT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                dfx_h8 whi[CT], wlo[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    whi[ct] = wc[((kc * CT + ct) * 2 + 0) * 64 + lane];
                    wlo[ct] = wc[((kc * CT + ct) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t][ct] = dfx_mfma_16x16x32_f16(wlo[ct], xh[t][kc], acc[t][ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t][ct] = dfx_mfma_16x16x32_f16(whi[ct], xl[t][kc], acc[t][ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t][ct] = dfx_mfma_16x16x32_f16(whi[ct], xh[t][kc], acc[t][ct]);
                if (kc == 3 && c + 1 < nchunks) {
#pragma unroll
                    for (int i = 0; i < HP; ++i) dst[i * NTH + tid] = pre[i];
#pragma unroll
                    for (int i = 0; i < HP; ++i) pre[i] = src[(HP + i) * NTH + tid];
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int n = c * (16 * CT) + 16 * ct + 4 * q;
                const float4 bz = *reinterpret_cast<const float4 *>(bias + n);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (okr[t])
                        *reinterpret_cast<float4 *>(orow + trow[t] * N + n) =
                            make_float4(acc[t][ct][0] * unscale[t] + bz.x, acc[t][ct][1] * unscale[t] + bz.y, acc[t][ct][2] * unscale[t] + bz.z,
                                        acc[t][ct][3] * unscale[t] + bz.w);
            }
            if (c + 1 < nchunks) {
#pragma unroll
                for (int i = 0; i < HP; ++i) dst[(HP + i) * NTH + tid] = pre[i];
            } else {
                DFX_VMEM_DRAIN();   // the block's gi rows have reached the L2 (same-XCD hand-over)
            }
            __syncthreads();
        }
        // the block's gi rows are stored (in front of the last barrier): say so
        if (tid == 0) dfx_xcd_release(dfx_pf_xcd(A, f, g), A.giprog[f] + g, A.pbase + (unsigned int)t1);
    }
}

// The exact form (DFX_EXACT_FP32=1): the same follower on v_mfma_f32_16x16x4_f32.  A.wf is W_ih^T as plain fp32 [256][768] (GruW::wih_t); a 64-column
// chunk of it (64 KB) is staged per chunk in the k order of dfx_k_proj256 — LDS row 4 ks + q holds W[64 q + ks][.], so that matrix op ks contracts
// the k-set {64 q + ks} and a lane's B operands are the 64 CONTIGUOUS values [64 q, 64 q + 64) of its row (float4 loads) — with bit 4 of the
// column index flipped in odd rows: the 2 x 16 lanes a ds_read_b32 serves per cycle (q = 0 / 1, then 2 / 3) then hit 32 different banks.
// 6144 matrix ops of 32 cycles per wave and block, two waves per SIMD: ~165 us per 16 steps of a recurrence that takes 14 us per step.
__global__ void __launch_bounds__(512, 1) dfx_k_proj_follow_x32(DfxPfArgs A) {
    constexpr int NW = 8, CT = 4, NTH = 64 * NW, SB = 2 * NW, N = 768, K = 256;
    constexpr int CHF = K * 16 * CT;          // floats per chunk
    constexpr int PER_T = CHF / 4 / NTH;      // float4 pieces per thread and chunk (8)
    constexpr int HP = PER_T / 2;
    DFX_DYN_SMEM(float, wsf);  // [2][CHF]
    const int f = (int)(blockIdx.x / (unsigned)A.groups);
    if (f >= A.nf) return;
    const int g = dfx_xcd_claim(A.xrec ? (A.xcons[f] ? A.xcons[f] : A.xrec) : nullptr, A.xtag, A.groups,   // (the registrations of the layer this follower FEEDS: with groups % 8 != 0 the layers sit on different XCDs)
                                A.xclaim ? A.xclaim + 8 * f : nullptr, (int)(blockIdx.x % (unsigned)A.groups), A.spin_limit, A.err,
                                reinterpret_cast<int *>(wsf));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, jl = lane & 15;
    constexpr int nchunks = N / (16 * CT);
    const int T = (int)A.T;
    dfx_xcd_register(dfx_pf_xcd(A, f, g));
    // piece i of a chunk (i compile-time): k = 32 i + kt, kt = tid / 16, columns 4 n4 .. + 3 of the chunk, n4 = tid % 16; its LDS row is
    // 4 (k % 64) + k / 64 = 4 kt + 128 (i % 2) + i / 2, whose parity — the column flip — is that of i / 2: one source and two destination bases per thread
    const int kt = tid >> 4, n4 = tid & 15;
    const int src_off = kt * N + 4 * n4;                                       // + 32 i N + 64 c
    const int dst_off0 = 4 * kt * 64 + 4 * n4, dst_off1 = 4 * kt * 64 + ((4 * n4) ^ 16);   // + (128 (i % 2) + i / 2) * 64
    auto piece_src = [&](const float *w, int c, int i) -> const f32x4 * {
        return reinterpret_cast<const f32x4 *>(w + src_off + 32 * i * N + 64 * c);
    };
    auto piece_dst = [&](float *buf, int i) -> f32x4 * {
        return reinterpret_cast<f32x4 *>(buf + (((i >> 1) & 1) ? dst_off1 : dst_off0) + (128 * (i & 1) + (i >> 1)) * 64);
    };
    for (int t0 = 0; t0 < T; t0 += SB) {
        const int t1 = t0 + SB < T ? t0 + SB : T;
        int zoff = 0, jz = jl;
        DFX_OPAQUE(zoff);   // (keeps the block-invariant addresses out of registers held across the block loop)
        DFX_OPAQUE(jz);
        const float *w = reinterpret_cast<const float *>(A.wf[f]) + zoff;
        const float *bias = A.bias[f] + zoff;
        const int64_t clip = (int64_t)g * 16 + jz;
        const bool okc = clip < A.B;
        const float *xrow = A.x[f] + (okc ? clip : 0) * A.T * K + 64 * q;
        float *orow = A.gi[f] + (okc ? clip : 0) * A.T * N;
#pragma unroll
        for (int i = 0; i < PER_T; ++i) *piece_dst(wsf, i) = *piece_src(w, 0, i);
        if (tid == 0) {
            const unsigned int want = A.pbase + (unsigned int)t1;
            int spins = 0;
            while ((int)(__hip_atomic_load(A.yprog[f] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (++spins > A.spin_limit) {
                    dfx_raise(A.err + 2);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
        dfx_xcd_acquire(dfx_pf_xcd(A, f, g));
        float4 xv[2][16];
        int trow[2];
        bool okr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            trow[t] = t0 + 2 * wave + t;
            okr[t] = okc && trow[t] < t1;
            if (!okr[t]) trow[t] = t0;
            const float4 *p = reinterpret_cast<const float4 *>(xrow + trow[t] * K);
#pragma unroll
            for (int v = 0; v < 16; ++v) xv[t][v] = okr[t] ? p[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int c = 0; c < nchunks; ++c) {
            const float *wc = wsf + (size_t)(c & 1) * CHF + q * 64 + jl;   // element (row 4 ks + q, column 16 ct + jl) sits at + ks * 256 + 16 * (ct ^ (q & 1))
            float *dstb = wsf + (size_t)((c + 1) & 1) * CHF;
            f32x4 pre[HP];   // (the ext-vector type: a HIP float4 array with a conditional first write stays in scratch)
            if (c + 1 < nchunks) {
#pragma unroll
                for (int i = 0; i < HP; ++i) pre[i] = *piece_src(w, c + 1, i);
            }
            f32x4 acc[2][CT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            float fa[2][CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) fa[0][ct] = wc[16 * (ct ^ (q & 1))];
            auto ksteps = [&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks + 1 < 64) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) fa[(ks + 1) & 1][ct] = wc[(ks + 1) * 256 + 16 * (ct ^ (q & 1))];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 x4 = xv[t][ks >> 2];
                    const float xk = (ks & 3) == 0 ? x4.x : (ks & 3) == 1 ? x4.y : (ks & 3) == 2 ? x4.z : x4.w;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1][ct], xk, acc[t][ct], 0, 0, 0);
                }
                DFX_SCHED_BARRIER();
            };
            dfx_static_for<0, 32>(ksteps);
            if (c + 1 < nchunks) {   // the next chunk's first half lands in the other buffer, its second half is requested
#pragma unroll
                for (int i = 0; i < HP; ++i) *piece_dst(dstb, i) = pre[i];
#pragma unroll
                for (int i = 0; i < HP; ++i) pre[i] = *piece_src(w, c + 1, HP + i);
            }
            dfx_static_for<32, 64>(ksteps);
            DFX_MFMA_GUARD();
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int n = c * (16 * CT) + 16 * ct + 4 * q;
                const float4 bz = *reinterpret_cast<const float4 *>(bias + n);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (okr[t])
                        *reinterpret_cast<float4 *>(orow + trow[t] * N + n) =
                            make_float4(acc[t][ct][0] + bz.x, acc[t][ct][1] + bz.y, acc[t][ct][2] + bz.z, acc[t][ct][3] + bz.w);
            }
            if (c + 1 < nchunks) {
#pragma unroll
                for (int i = 0; i < HP; ++i) *piece_dst(dstb, HP + i) = pre[i];
            } else {
                DFX_VMEM_DRAIN();
            }
            __syncthreads();
        }
        if (tid == 0) dfx_xcd_release(dfx_pf_xcd(A, f, g), A.giprog[f] + g, A.pbase + (unsigned int)t1);
    }
}

// flag kernels of the persistent GRU phase: dfx_k_flag_set runs behind a producer on its stream (the kernel boundary in front of it
// has made the producer's writes visible), dfx_k_wait_ge holds its stream until all n flags have reached the target
__global__ void dfx_k_flag_set(unsigned int *flag, unsigned int value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void dfx_k_wait_ge(const unsigned int *flags, int n, unsigned int target, unsigned int *err, int spin_limit) {
    bool ok = false;
    int spins = 0;
    while (!ok) {
        ok = true;
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            ok = ok && (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
        ok = __all(ok);   // one wave
        if (!ok) {
            if (++spins > spin_limit) {
                if (threadIdx.x == 0) dfx_raise(err + 2);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Handshake of dfx_model_create (DFX_Q_HWQ_PROBE): one single-wave launch per stream of the persistent phase; each adds itself to a
// counter and waits, bounded, until all n have arrived — which only happens if the n streams really run at the same time (streams
// that share a hardware queue run one after the other: the first then waits in vain and raises *fail).
__global__ void dfx_k_probe_meet(unsigned int *cnt, unsigned int n, int spin_limit, unsigned int *fail) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
            if (++spins > spin_limit) {
                dfx_raise(fail);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
}

