// DeepFilterNet3 model handle: tensor manifest, BatchNorm folding + weight re-layout, workspace planning and the kernel
// sequence of DfNet.forward (deepfilternet3.py:389-456) and enhance() (enhance.py:206-250).
#include "dfx_manifest.h"
#include "dfx_nn_kernels.h"

#include <cmath>
#include <functional>
#include <atomic>
#include <map>
#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

// ------------------------------------------------------------------------------------------------ cfg validation
static int check_cfg(const dfx_model_cfg *c) {
    if (!c) DFX_FAIL(DFX_ERR_INVALID_ARG, "null model cfg");
    if (c->conv_ch != 16 && c->conv_ch != 32 && c->conv_ch != 64)
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "conv_ch=%d: the HIP kernels are instantiated for 16, 32, 64", c->conv_ch);
    if (c->emb_hidden_dim != 256 || c->df_hidden_dim != 256) DFX_FAIL(DFX_ERR_UNSUPPORTED, "GRU hidden size must be 256");
    if (c->nb_erb <= 0 || c->nb_erb % 8 || c->nb_erb > 64) DFX_FAIL(DFX_ERR_UNSUPPORTED, "nb_erb must be a multiple of 8, <= 64");
    if (c->nb_df <= 0 || c->nb_df % 2 || c->nb_df > c->fft_size / 2 + 1) DFX_FAIL(DFX_ERR_UNSUPPORTED, "nb_df must be even and <= F");
    if (c->df_order <= 0 || c->df_order > 16 || c->df_lookahead < 0 || c->df_lookahead >= c->df_order)
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "need 0 <= df_lookahead < df_order <= 16");
    if (c->conv_lookahead < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "conv_lookahead < 0");
    if (c->emb_num_layers < 2 || c->df_num_layers < 1) DFX_FAIL(DFX_ERR_UNSUPPORTED, "emb_num_layers >= 2 and df_num_layers >= 1 required");
    if (c->df_pathway_kernel_size_t < 1 || c->df_pathway_kernel_size_t > 8) DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_pathway_kernel_size_t must be 1..8");
    const int emb = c->conv_ch * c->nb_erb / 4;
    auto div_ok = [&](int I, int H, int G) { return G > 0 && I % G == 0 && H % G == 0 && (I / G) % 4 == 0 && (H / G) % 4 == 0; };
    if (!div_ok(c->conv_ch * c->nb_df / 2, emb, c->enc_lin_groups)) DFX_FAIL(DFX_ERR_UNSUPPORTED, "enc_linear_groups=%d does not tile df_fc_emb into multiples of 4", c->enc_lin_groups);
    if (!div_ok(emb, 256, c->lin_groups) || !div_ok(256, emb, c->lin_groups) || !div_ok(emb, 256, 8) ||
        !div_ok(256, c->nb_df * 2 * c->df_order, c->lin_groups))
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "linear_groups=%d does not tile the grouped linears into multiples of 4", c->lin_groups);
    const int G = dfx_gcd(c->conv_ch, 2 * c->df_order);
    if ((c->conv_ch / G) % 4 || (2 * c->df_order / G) > 16) DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_convp group shape unsupported");
    if (c->df_gru_skip == DFX_SKIP_IDENTITY && emb != 256) DFX_FAIL(DFX_ERR_INVALID_ARG, "df_gru_skip=identity needs emb_dim == 256");
    if (c->df_gru_skip < 0 || c->df_gru_skip > 2) DFX_FAIL(DFX_ERR_INVALID_ARG, "bad df_gru_skip");
    if (c->emb_gru_skip_enc < 0 || c->emb_gru_skip_enc > 2 || c->emb_gru_skip < 0 || c->emb_gru_skip > 2 || (c->enc_concat & ~1))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "bad emb_gru_skip_enc / emb_gru_skip / enc_concat");
    // deepfilternet3.py:138-146: with enc_concat the encoder GRU's input is twice as wide as its output; identity then fails the
    // reference's own assert and the grouped-linear skip (built for emb_out_dim inputs) cannot take it either
    if (c->enc_concat && c->emb_gru_skip_enc != DFX_SKIP_NONE) DFX_FAIL(DFX_ERR_INVALID_ARG, "enc_concat excludes emb_gru_skip_enc (dimensions do not match)");
    if (!div_ok(2 * emb, 256, c->lin_groups) && c->enc_concat) DFX_FAIL(DFX_ERR_UNSUPPORTED, "linear_groups does not tile the concatenated embedding");
    if ((c->emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR || c->emb_gru_skip == DFX_SKIP_GROUPEDLINEAR) && !div_ok(emb, emb, c->lin_groups))
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "linear_groups does not tile the embedding skip into multiples of 4");
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ manifest API
extern "C" int dfx_model_tensor_count(const dfx_model_cfg *cfg, int *count) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!count) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    *count = (int)dfx_build_manifest(*cfg).t.size();
    return DFX_OK;
}
extern "C" int dfx_model_tensor_info(const dfx_model_cfg *cfg, int index, char *name_out, int name_cap,
                                     int64_t shape_out[4], int *ndim_out, int64_t *offset_out) {
    if (int rc = check_cfg(cfg)) return rc;
    const DfxManifest m = dfx_build_manifest(*cfg);
    if (index < 0 || index >= (int)m.t.size()) DFX_FAIL(DFX_ERR_INVALID_ARG, "tensor index out of range");
    const DfxTensor &t = m.t[index];
    if (name_out && name_cap > 0) snprintf(name_out, (size_t)name_cap, "%s", t.name.c_str());
    if (shape_out)
        for (int i = 0; i < 4; ++i) shape_out[i] = i < t.ndim ? t.shape[i] : 1;
    if (ndim_out) *ndim_out = t.ndim;
    if (offset_out) *offset_out = t.offset;
    return DFX_OK;
}
extern "C" int dfx_model_blob_floats(const dfx_model_cfg *cfg, int64_t *n) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!n) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    *n = dfx_build_manifest(*cfg).total;
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ prepared weights
struct PwW {   // separable conv block: depthwise + pointwise(+BN) [+ pathway skip scalars]
    size_t dw = 0, wt = 0, bias = 0, sk_a = 0, sk_b = 0;
    bool has_skip = false;
    size_t wt_h3 = 0;       // pointwise weights as pre-scaled f16 hi/lo MFMA fragments (dfx_chain_stage_h3), 0 = none (C % 32 != 0)
    float unscale = 1.f;
};
struct GruW {
    size_t wih_t = 0, bias_i = 0, whh4 = 0, bhn = 0;
    size_t wih_h3 = 0;      // W_ih as pre-scaled f16 hi/lo MFMA fragments (dfx_k_proj256_h3)
    float wih_unscale = 1.f;
    size_t whh_h3 = 0;      // W_hh as pre-scaled f16 hi/lo MFMA fragments (dfx_k_gru_rec_h3)
    float whh_unscale = 1.f;
    size_t whh_pj = 0;      // W_hh in the projection kernel's fragment order (dfx_k_gru_step_h3: one time step of many streams), same scale
    size_t whh_x32 = 0;     // W_hh as fp32 fragments in dfx_k_gru_rec_h3's order (dfx_k_gru_rec_x32: the exact recurrence of the layer-pipelined phase)
};
struct GlinW {
    size_t w = 0;
    int G = 0, Kg = 0, Ng = 0;
};

#define DFX_MAX_LANES 4
#define DFX_LANE_EVENTS 12
#define DFX_THROTTLE_MIN_FRAMES 16384   /* passes of at least this many frames are enqueued one at a time (dfx_model::ev_pass) */
#define DFX_MAX_GRU_LAYERS 8   /* all GRU layers of the three stacks */
#define DFX_MAX_TCHUNKS 16     /* time chunks of the layer-pipelined GRU phase */
#define DFX_SEQ_GMAX 64        /* most 16-clip groups per layer the persistent GRU phase is used for (all workgroups must be co-resident) */
struct DfxLane;
struct DfxLane {
    hipStream_t main = nullptr;
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev[DFX_LANE_EVENTS] = {};
    hipStream_t gs[DFX_MAX_GRU_LAYERS] = {};                      // recurrence stream of GRU layer l
    hipStream_t ps[DFX_MAX_GRU_LAYERS] = {};                      // preparation stream of GRU layer l (linear_in, projection)
    hipStream_t ts[2] = {};                                       // tails of the ERB / DF decoder
    hipEvent_t gev[DFX_MAX_GRU_LAYERS][DFX_MAX_TCHUNKS] = {};     // layer l has produced time chunk k
    hipEvent_t pev[DFX_MAX_GRU_LAYERS][DFX_MAX_TCHUNKS] = {};     // gi of layer l, chunk k is ready
    hipEvent_t eev[DFX_MAX_TCHUNKS] = {};                         // emb chunk k is ready
};
// The internal streams and events of a process on one device (created on demand, kept for the life of the process) and what the handshake of
// the persistent GRU phase found about them (hwq_probe: -1 not run, 1 concurrent, 0 not).
struct DfxLaneSet {
    DfxLane lanes[DFX_MAX_LANES];
    int hwq_probe = -1;
};
// enhance() hands the synthesis to the model forward so that it runs behind the deep filter on the stream that carries the last coefficients
struct DfxFinish {
    const dfx_state *st;
    float *y;
    int64_t out_stride, out_skip, out_len;
    int64_t spec_stride;   // row stride (complex elements) of enhance()'s own spec / spec_e buffers: F rounded up to even, so that
                           // every row is 16-byte aligned (dfx_k_df_apply_rows); dfx_model_forward's caller-owned arrays are dense
    bool out_i16 = false;  // y points at int16_t PCM samples (dfx_enhance_pcm16)
};
// Streaming (dfx_stream_process): a forward pass over a window.  Every feature / activation array holds T = H + n frames per clip
// (H history frames, then the n new ones); only the new frames are computed (kernels take t_begin, per-frame kernels a DfxRowMap),
// the GRUs continue from h_state, the spec array has spec_T = T + lookahead frames and the enhanced spectra are stored compactly.
struct DfxStreamCtx {
    int64_t H;         // history frames in front of the new ones
    int64_t t_zero;    // local frames < t_zero precede the start of the stream (df_convp sees zero padding there)
    int64_t spec_T;    // frames per clip of the spec array
    int64_t spec_stride = 0;   // > 0: bins per row of spec and out (padded rows)
    int64_t feat_T = 0;    // > 0: frames per clip of feat_erb and feat_spec (windows inside the linear buffers; both share it)
    float *h_state;    // [GRU layers][B][256], in model order enc, erb_dec, df_dec
    float *h_next = nullptr;   // non-null (one new frame, ungated): the layers run as dfx_k_gru_step_h3 and leave their new states HERE
    void *c0ring = nullptr;    // non-null (one new frame, ungated): df_convp keeps the pending sums of its next kt - 1 outputs here (dfx_k_df_convp_step)
    int c0slot = 0;            //   slot of the new frame = its net position % (kt - 1)
    bool c0rebuild = false;    //   the sums are not current: recompute the older frames' taps from the feature window
    mutable bool c0ring_used = false;   //   out: dfx_k_df_convp_step ran in this pass (only then are the sums current afterwards)
    std::function<int(hipStream_t)> erb_pre;  // set: the ERB feature window's update, enqueued on the caller's stream BEHIND the event the DF branch
                                              //   starts on (that branch's chain to c1 is the longer one)
    std::function<int(hipStream_t)> df_pre;   // set: state updates that only the DF branch reads — enqueued on that branch's stream before its
                                              //   first kernel instead of in front of the encoder
    std::function<int(hipStream_t)> df_post;  // set: state updates that nothing before the final deep filter reads — enqueued behind df_convp on
                                              //   its stream (joined through EV_C0P before df_out), or with df_pre when that kernel does not run
    float pf_beta;     // < 0: the model's setting
    float *out;        // [B, out_T, F][2]: local frame t of clip b is stored at frame t - out_toff
    int64_t out_T, out_toff;
    const struct DfxGate *gate = nullptr;  // per-stream stage gating (one new frame per pass); null: every frame runs every stage
    int channels = 1;                      // > 1: consecutive streams are the channels of one multi-channel stream ...
    int reduce_mask = 0;                   // ... whose ERB masks are reduced over the channels: 0 none, 1 max, 2 mean (tract.rs:96-118,868-902)
};

// ---- per-stream stage gating of the streaming runtime (DfTract::process / apply_stages, tract.rs:509-616,658-672) ----------------
// The reference decides per frame, from the encoder's local-SNR estimate, whether the ERB decoder (stage 1) and the DF decoder
// (stage 2) run at all; a decoder that does not run keeps its state (tract's pulsed models only advance when they are run), and a
// stream that has been silent for more than 5 hops is not processed at all.  Lockstep streams take those decisions independently:
// every kernel still runs for every stream (a skipped stream costs the same as a busy one on a GPU), and the decisions are applied
// as data: flags[b] selects whose state is kept (dfx_k_gate_commit) and what the deep-filter kernel is fed (dfx_k_gate_edit).
enum { DFX_GATE_FROZEN = 1, DFX_GATE_GAINS = 2, DFX_GATE_ZEROS = 4, DFX_GATE_DF = 8 };
struct DfxGate {
    unsigned char *flags;  // [B]
    float thr[3];          // min_db_thresh, max_db_erb_thresh, max_db_df_thresh
    float *c0_win;         // [B, T, Fd, C]: slots T-kt .. T-2 = c0 of the last kt-1 frames the DF decoder ran on, T-1 = this frame
    void *pend2 = nullptr;           // fp16-split models: df_convp's pending sums instead, twice per stream (dfx_k_df_convp_step), with
    unsigned char *par = nullptr;    //   the half that is current and
    int *cnt = nullptr;              //   the frames each stream's DF decoder has consumed
    int channels;          // streams per multi-channel group: one skip counter and one stage decision (the first channel's lsnr) per group
};

// tract.rs:513-525: mean square of the hop (sequential f32 fold like the reference: the comparison with 1e-7 is then the same
// decision); below the threshold the counter goes up, else it is cleared; above 5 the stream is frozen for this hop.
// Multi-channel streams (ch consecutive rows): the fold runs over all channels of the hop, channel after channel, like the
// reference's `noisy.iter()` over its [ch, hop] array; every row of the group keeps an identical copy of the group's counter.
__global__ void dfx_k_gate_pre(const float *x, int64_t x_stride, int hop, int64_t B, int *skip_counter, unsigned char *flags, int ch) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t b0 = b - b % ch;
    float e = 0.f;
    for (int c = 0; c < ch; ++c) {
        const float *xp = x + (b0 + c) * x_stride;
        for (int i = 0; i < hop; ++i) e = __fadd_rn(e, __fmul_rn(xp[i], xp[i]));
    }
    const float ms = __fdiv_rn(e, (float)(hop * ch));
    int c = skip_counter[b];
    c = ms < 1e-7f ? c + 1 : 0;
    skip_counter[b] = c;
    flags[b] = c > 5 ? DFX_GATE_FROZEN : 0;
}

// The same for mono streams with the hop staged through LDS: 16 streams per 64-thread workgroup, rows loaded coalesced (a thread walking
// its own row touches 64 cache lines per load instruction: 50 us at 4096 streams, in front of everything else of a gated hop), then one
// lane per stream folds its row in the reference's order (rows padded by one float: the 16 lanes read different banks).
#define DFX_GATE_PRE_ROWS 16
__global__ void __launch_bounds__(64) dfx_k_gate_pre_lds(const float *x, int64_t x_stride, int hop, int64_t B, int *skip_counter, unsigned char *flags) {
    DFX_DYN_SMEM(float4, rows4);   // [16][hop / 4 + 1] float4: rows padded by one float4 (16 lanes x 16 bytes then read 64 different banks)
    const int64_t b0 = (int64_t)blockIdx.x * DFX_GATE_PRE_ROWS;
    const int q4 = hop >> 2, ld4 = q4 + 1;
    // all loads of eight rows are requested before the first LDS store (hop = 480: two float4 per thread and row)
    for (int r0 = 0; r0 < DFX_GATE_PRE_ROWS; r0 += 8) {
        for (int i0 = threadIdx.x; i0 < q4; i0 += 128) {
            float4 v[8][2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int64_t b = b0 + r0 + r < B ? b0 + r0 + r : B - 1;
                const float4 *xr = reinterpret_cast<const float4 *>(x + b * x_stride);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + 64 * u;
                    v[r][u] = xr[i < q4 ? i : i0];
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + 64 * u;
                    if (i < q4) rows4[(r0 + r) * ld4 + i] = v[r][u];
                }
        }
    }
    __syncthreads();
    const int64_t b = b0 + threadIdx.x;
    if (threadIdx.x >= DFX_GATE_PRE_ROWS || b >= B) return;
    const float4 *xp = rows4 + threadIdx.x * ld4;
    float e = 0.f;
    for (int i0 = 0; i0 < q4; i0 += 8) {   // eight LDS reads in flight, the fold itself in the reference's order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[i0 + u < q4 ? i0 + u : i0];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= q4) break;
            e = __fadd_rn(e, __fmul_rn(v[u].x, v[u].x));
            e = __fadd_rn(e, __fmul_rn(v[u].y, v[u].y));
            e = __fadd_rn(e, __fmul_rn(v[u].z, v[u].z));
            e = __fadd_rn(e, __fmul_rn(v[u].w, v[u].w));
        }
    }
    const float ms = __fdiv_rn(e, (float)hop);
    int c = skip_counter[b];
    c = ms < 1e-7f ? c + 1 : 0;
    skip_counter[b] = c;
    flags[b] = c > 5 ? DFX_GATE_FROZEN : 0;
}
static int launch_gate_pre(const float *x, int64_t x_stride, int hop, int64_t B, int *skip_counter, unsigned char *flags, int ch, hipStream_t s) {
    const size_t smem = (size_t)DFX_GATE_PRE_ROWS * (hop / 4 + 1) * 16;
    if (ch == 1 && smem <= 64 * 1024 && hop % 4 == 0 && x_stride % 4 == 0 && !((uintptr_t)x & 15)) {
        dfx_launch(dfx_k_gate_pre_lds, dim3((unsigned)dfx_ceil_div(B, DFX_GATE_PRE_ROWS)), dim3(64), smem, s, x, x_stride, hop, B, skip_counter, flags);
    } else {
        dfx_launch(dfx_k_gate_pre, dim3((unsigned)dfx_ceil_div(B, 64)), dim3(64), 0, s, x, x_stride, hop, B, skip_counter, flags, ch);
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// tract.rs:658-672 apply_stages on the newest frame's lsnr (lsnr[b*T + T-1])
// (multi-channel: the decision of a group is taken from its first channel's lsnr, tract.rs:468 `to_scalar`)
__global__ void dfx_k_gate_post(const float *lsnr, int64_t T, float thr_min, float thr_erb, float thr_df, unsigned char *flags,
                                int64_t B, int ch) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned char f = flags[b];
    if (f & DFX_GATE_FROZEN) return;
    const float v = lsnr[(b - b % ch) * T + T - 1];
    if (v < thr_min) f |= DFX_GATE_ZEROS;
    else if (v > thr_erb) f |= 0;
    else if (v > thr_df) f |= DFX_GATE_GAINS;
    else f |= DFX_GATE_GAINS | DFX_GATE_DF;
    flags[b] = f;
}

// The decisions as inputs of the (unchanged) deep-filter kernel, newest frame of every stream:
//   no stage 1: the band gains become 0 (zero mask, tract.rs:485-486) or 1 (gains absent: the spectrum passes, :565-567);
//   no stage 2: the low bins take the masked spectrum (:570-581) = a deep filter whose only non-zero tap is the current frame's,
//               with the band gain as a real coefficient (x*g - y*0 and 0-taps add exact zeros: the same bits as the mask path).
__global__ void dfx_k_gate_edit(const unsigned char *flags, float *mask, float *coefs, const unsigned char *bin2band, int64_t B,
                                int64_t T, int E, int Fd, int O, int tap0) {
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const unsigned char f = flags[b];
    float *mrow = mask + (b * T + T - 1) * E;
    if (!(f & DFX_GATE_GAINS)) {
        const float g = (f & DFX_GATE_ZEROS) ? 0.f : 1.f;
        for (int i = threadIdx.x; i < E; i += blockDim.x) mrow[i] = g;
    }
    __syncthreads();
    if (!(f & DFX_GATE_DF)) {
        for (int i = threadIdx.x; i < O * Fd; i += blockDim.x) {
            const int n = i / Fd, fq = i - n * Fd;
            float2 *cp = reinterpret_cast<float2 *>(coefs) + ((b * O + n) * T + T - 1) * Fd + fq;
            *cp = make_float2(n == tap0 ? mrow[bin2band[fq]] : 0.f, 0.f);
        }
    }
}

// Multi-channel streams: the ERB decoder's mask is reduced over the channels of a group and the reduced mask is applied to every
// channel (tract.rs:868-902: Reduce<Max> or Reduce<Sum> * (1/ch) wired behind the decoder; :547-556 the one mask for all channels).
// mask [B*T, E]; frames [t_begin, T) of every stream; mode 1 max, 2 mean.
__global__ void dfx_k_mask_reduce(float *mask, int64_t B, int64_t T, int64_t t_begin, int E, int ch, int mode) {
    const int64_t n = (B / ch) * (T - t_begin) * E;
    const float inv = 1.f / (float)ch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const int64_t r = i / E, g = r / (T - t_begin), t = t_begin + r % (T - t_begin);
        float *p = mask + ((g * ch) * T + t) * E + e;
        float acc = p[0];
        for (int c = 1; c < ch; ++c) {
            const float v = p[(int64_t)c * T * E];
            acc = mode == 1 ? fmaxf(acc, v) : acc + v;
        }
        if (mode == 2) acc *= inv;
        for (int c = 0; c < ch; ++c) p[(int64_t)c * T * E] = acc;
    }
}

// The stage decisions as dfx_stream_process_raw reports them: DFX_GATE_GAINS and DFX_GATE_ZEROS both mean "gains exist".
__global__ void dfx_k_gate_stages(const unsigned char *flags, unsigned char *stages, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned char f = flags[b];
    stages[b] = (unsigned char)((f & DFX_GATE_DF) | ((f & (DFX_GATE_GAINS | DFX_GATE_ZEROS)) ? DFX_GATE_GAINS : 0));
}

// State selection after a gated pass: entry e copies row b of src to dst when (flags[b] & mask) == want.  Frozen streams get all
// their state back (STFT memories, running means, history rings, hidden states), a stream whose stage 1 / stage 2 was skipped its
// decoder's hidden states.
#define DFX_GATE_MAX_ENTRIES 24
struct DfxGateTable {
    float *dst[DFX_GATE_MAX_ENTRIES];
    const float *src[DFX_GATE_MAX_ENTRIES];
    int64_t row[DFX_GATE_MAX_ENTRIES];
    unsigned char mask[DFX_GATE_MAX_ENTRIES], want[DFX_GATE_MAX_ENTRIES];
    int n;
};
__global__ void dfx_k_gate_commit(DfxGateTable G, const unsigned char *flags, int64_t B) {
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const unsigned char f = flags[b];
    for (int e = 0; e < G.n; ++e) {
        if ((f & G.mask[e]) != G.want[e]) continue;
        const float *sp = G.src[e] + b * G.row[e];
        float *dp = G.dst[e] + b * G.row[e];
        for (int64_t i = threadIdx.x; i < G.row[e]; i += blockDim.x) dp[i] = sp[i];
    }
}

// Streams whose DF decoder ran push the newest c0 frame into their window (slot j <- slot j+1, j = T-kt .. T-2); the others keep
// theirs: the delay line in front of df_convp only moves when the DF decoder runs.
__global__ void dfx_k_gate_c0_shift(const unsigned char *flags, float *c0_win, int64_t B, int64_t T, int kt, int64_t frame) {
    const int64_t b = blockIdx.x;
    if (b >= B || !(flags[b] & DFX_GATE_DF)) return;
    float *w = c0_win + (b * T + T - kt) * frame;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < frame; i += (int64_t)gridDim.y * blockDim.x)
        for (int j = 0; j + 1 < kt; ++j) w[j * frame + i] = w[(j + 1) * frame + i];
}

// Linear windows (dfx_stream_state::spec_lin, fe_lin, fs_lin) of gated handles: every stream's window slides by the hop, but a frozen stream's
// history must stay what it was.  After the hop's frame has been appended at frame pos + h, the frozen streams' h history frames
// [pos, pos + h) are moved up by one frame (the appended frame is overwritten): the next hop's history [pos + 1, pos + 1 + h) is then
// the old one.  One thread per element of a frame row, walking the frames from the newest down (an overlapping move within its column).
__global__ void dfx_k_gate_hold(const unsigned char *flags, float *win, int64_t cap, int64_t row, int64_t pos, int64_t h, int64_t B) {
    const int64_t b = blockIdx.x;
    if (b >= B || !(flags[b] & DFX_GATE_FROZEN)) return;
    float *w = win + (b * cap + pos) * row;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < row; i += (int64_t)gridDim.y * blockDim.x)
        for (int64_t j = h - 1; j >= 0; --j) w[(j + 1) * row + i] = w[j * row + i];
}

// The same for the pending-sum form of df_convp (dfx_k_df_convp_step): where the DF decoder ran the sums written by this pass become
// current and the stream's frame count goes up.
__global__ void dfx_k_gate_pend_commit(const unsigned char *flags, unsigned char *par, int *cnt, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || !(flags[b] & DFX_GATE_DF)) return;
    par[b] ^= 1;
    cnt[b] += 1;
}

// End of a gated hop: frozen streams answer zeros and lsnr = -15 (tract.rs:522-525); the others update the skip counter from the
// stage decision (:562-567: gains present -> 0, absent -> += 1).  warm: the hop produced no net position yet (no decision taken).
__global__ void dfx_k_gate_finish(const unsigned char *flags, int *skip_counter, float *y, int64_t y_stride, int hop, float *lsnr_out,
                                  int64_t lsnr_stride, int64_t B, int warm) {
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    const unsigned char f = flags[b];
    if (f & DFX_GATE_FROZEN) {
        for (int i = threadIdx.x; i < hop; i += blockDim.x) y[b * y_stride + i] = 0.f;
        if (threadIdx.x == 0 && lsnr_out) lsnr_out[b * lsnr_stride] = -15.f;
    } else if (threadIdx.x == 0 && !warm) {
        skip_counter[b] = (f & (DFX_GATE_GAINS | DFX_GATE_ZEROS)) ? 0 : skip_counter[b] + 1;
    }
}

enum { EV_START = 0, EV_C0, EV_C1, EV_C0P, EV_EMB, EV_COEFS, EV_FRONT, EV_DONE, EV_XA, EV_MASK, EV_FIN, EV_TICKET };

struct dfx_model {
    dfx_model_cfg cfg{};
    float *d_w = nullptr;  // all prepared weights, one device allocation
    size_t n_w = 0;
    // offsets (floats) into d_w
    size_t erb0_w = 0, erb0_b = 0;
    PwW erb1, erb2, erb3, dfc0, dfc1, ct3, ct2, ct1;
    size_t co_w = 0, co_ska = 0, co_skb = 0;
    float co_bias = 0.f;
    size_t tail_w0h3 = 0, tail_woh3 = 0;    // dfx_k_erb_tail's fp16-split fragments of erb_conv0 (e0 recomputed, bias in a constant-1 slot) and of conv0_out
    float tail_w0_unscale = 1.f, tail_wo_unscale = 1.f;
    GlinW fc_emb, enc_in, enc_out, dec_in, dec_out, dfg_in, df_skip, df_out, enc_skip, dec_skip;
    std::vector<GruW> enc_gru, dec_gru, df_gru;
    size_t lsnr_w = 0;
    float lsnr_b = 0.f;
    // dfx_k_emb_fan (one pass over the encoder GRU's output for emb and its consumers): fragments [chunks][8][64] float4, 0 chunks = the
    // grouped linears of this model do not nest the way the kernel needs (then the separate grouped GEMMs run); DFX_FUSE_EMB=0: off
    size_t fan_w = 0;
    int fan_chunks = 0;            // super-chunks of 32 hidden columns (0: not available)
    int fan_kind[3] = {0, 0, 0};   // per consumer (dec_in, dfg_in, df_skip): 0 absent, 1 narrow (32 -> 16 groups), 2 wide (64 -> 32 groups)
    bool fuse_emb = true;
    // dfx_k_enc_fan (df_fc_emb + the encoder GRU's linear_in in one pass over c1): fragment offsets, 0 groups = shapes do not nest
    size_t efan_w1 = 0, efan_w2 = 0;
    int efan_groups = 0;
    // dfx_k_df_enc_h3 (df_conv0 -> df_conv1 -> df_fc_emb -> linear_in in one kernel, c1 never stored): fc / linear_in fragments; 0 chunks = the
    // shapes do not fit (then dfx_k_df_conv01_h3 + dfx_k_enc_fan run); DFX_FUSE_DFENC=0: off
    size_t dfenc_fc = 0, dfenc_in = 0;
    int dfenc_chunks = 0;
    float dfenc_fc_unscale = 1.f, dfenc_in_unscale = 1.f;
    bool fuse_dfenc = true;
    // dfx_k_df_out_h3 (df_out + tanh + c0p as a row-streaming kernel of its own): fragments [G][ceil(Ng / 16)]; 0 = shapes do not fit
    // (dfx_k_ggemm then); DFX_DFOUT_LEAN=0: off
    size_t dfo_h3 = 0;
    int dfo_nu = 0;
    float dfo_unscale = 1.f;
    bool dfout_lean = true;
    bool fuse_encfan = true;       // DFX_FUSE_ENCFAN=0 (dev A/B): df_fc_emb and linear_in as two grouped GEMMs while the rest of DFX_FUSE_EMB stays on
    bool fuse_dfa = true;          // DFX_FUSE_DFA=0: deep filter and ISTFT of enhance() as two kernels with spec_e between them
    // The ERB decoder's convolutions as ONE launch (dfx_k_erb_tail: d3 / d2 / d1 stay in LDS, -12 KB per frame beside the GRU chain);
    // DFX_FUSE_TAIL=0: three launches (convt3, convt2, convt1 + conv0_out).  Measured at config 2 (profiles/r03_fusion_ab.log): the first
    // version (8 waves per CU) 2.19 ms alone against 1.5 ms for the three launches and +0.37 ms per step; the second (12 waves per CU, 8.5 KB
    // of strips per wave, fragments read from LDS per k-chunk) 1.32 ms alone and -0.25 ... -0.45 ms per step.
    bool fuse_tail = true;
    // e0 = erb_conv0's output (8 KB per frame) is not stored: the fused decoder tail recomputes it from the three feature rows it depends on
    // (DFX_E0_RECOMPUTE=0: written by dfx_k_erb_enc, read back by dfx_k_erb_tail)
    bool e0_recompute = true;
    size_t cp_w1 = 0, cp_w2 = 0, cp_b = 0;   // df_convp, tiled form (kt > 5)
    size_t cp_weff = 0, cp_b16 = 0;          // df_convp, folded sliding-window form (kt <= 5)
    size_t cin_weff = 0, cin_b = 0;          // enc.df_conv0 folded into a dense 3x3 conv 2 -> C
    // fp16-split MFMA fragments of the fused DF-encoder kernels (conv_ch % 32 == 0, kt <= 5): df_conv0, df_conv1 pointwise, df_convp
    size_t c0_h3 = 0, dfc1_h3 = 0, cp_h3 = 0;
    float c0_unscale = 1.f, dfc1_unscale = 1.f, cp_unscale = 1.f;
    int cp_G = 0, cp_NO = 0;
    // concurrency (created once; one forward / enhance at a time per handle):
    //   a lane = the streams of one batch chunk: `main` (only used when dfx_enhance pipelines chunks; otherwise the caller's
    //   stream plays that role) + two auxiliary streams for the independent branches of the forward pass + fork/join events
    //   Round 6: the lanes belong to the PROCESS (one set per device, DfxLaneSet), not to the handle: every handle of a process enqueues on the
    //   same ~14 internal streams.  Handles used to bring 14 streams each, the second handle's streams shared hardware queues with the first
    //   one's (GPU_MAX_HW_QUEUES = 24) and its handshake put it on the event-synchronised form for good; passes of different handles take turns
    //   anyway (DfxTurn), so nothing is lost.
    DfxLane *lanes = nullptr;
    struct DfxLaneSet *laneset = nullptr;
    hipEvent_t ev_fork = nullptr;
    // One big pass in flight per handle: a call that would enqueue a multi-stream pass while the previous one is still running first
    // waits (on the host) for that one to drain.  Packets queued ahead on the pass's ~13 hardware queues slow the running pass — measured
    // at config 2: 19.45 ms per step with free enqueue-ahead, 18.8 ms when the host holds the next step back (DF-apply inside the loop
    // 0.60 -> 0.50 ms, the rate it has alone).  DFX_ENQUEUE_AHEAD=1 restores the unthrottled enqueue.
    hipEvent_t ev_pass = nullptr;
    hipEvent_t ev_gate = nullptr;       // recorded behind every multi-stream pass of this handle (DfxTurn)
    mutable bool pass_pending = false;
    bool enqueue_ahead = false;
    bool concurrent = false;
    bool have_streams = false;
    int max_chunks = 1;       // batch chunks pipelined by dfx_enhance (DFX_CHUNKS; measured: no gain over time-chunk pipelining)
    int tchunks = 12;         // time chunks of the layer-pipelined GRU phase (DFX_TCHUNKS)
    int tchunk_min = 32;      // shortest chunk worth a launch (frames)
    bool run_df = true;       // DfNet(run_df=False): mask only (dfx_model_set_run_df)
    bool exact_fp32 = false;  // DFX_EXACT_FP32=1: keep the dense contractions on the exact fp32 MFMA path
    // df_conv0's output c0 is recomputed by its consumers instead of being stored when the pathway conv has the sliding-window kernel
    // (kt <= 5); DFX_FUSE_C0=0 restores the materialised c0 (dfx_k_conv_in_df -> dfx_k_pwconv / dfx_k_df_convp2).
    bool fuse_c0 = true;
    bool c0_batch_unfused = false;   // exact mode: batch passes materialise c0 (below)
    // frame-resident ERB encoder head / decoder tail (dfx_k_erb_enc, dfx_k_erb_dec10); DFX_FUSE_ERB=0: layer-by-layer kernels
    bool fuse_erb = true;
    // persistent GRU phase (dfx_k_gru_seq): flag words [ready: 8][emb: 1][pad][done: 8 * DFX_SEQ_GMAX], monotonic over the model's life
    unsigned int *d_sync = nullptr;
    mutable unsigned int seq_pbase = 0;       // step counter base of the follower hand-overs (yprog / giprog), monotonic like seq_base
    mutable int64_t passes_seq = 0, passes_ev = 0;   // passes that ran the persistent phase / that gave it up because another process held the device's ticket (DFX_Q_TICKET_*)
    mutable unsigned int seq_base = 0;  // flag value of "nothing of the current forward pass yet"
    unsigned long long *d_trace = nullptr;   // dev aid (DFX_SEQ_TRACE=1): chunk timestamps of the last persistent GRU launch
    mutable int trace_dims[3] = {0, 0, 0};
    bool gru_seq = true;                // DFX_GRU_SEQ=0: one launch per (layer, time chunk) synchronised with events (round-1 form)
    bool phase_late = true;             // DFX_PHASE_LATE=0: the GRU phase is enqueued right behind the front (no staged enqueue)
    int proj_rt = 0;                    // DFX_PROJ_RT=1|2|3: one form of the projection kernel for every launch size
    // switches of the persistent GRU phase and its side work, read when the handle is created like the rest; -1 = "the default"
    struct {
        int follow = 2;                 // DFX_SEQ_FOLLOW (see seq_follow_mode)
        int chunks = 0, ramp = 0;       // DFX_SEQ_CHUNKS (0: 12, or 16 without followers), DFX_SEQ_RAMP
        bool publish = true;            // DFX_SEQ_PUBLISH
        bool xcd_light = true;          // DFX_SEQ_XCD_LIGHT
        int convp_late = -1;            // DFX_CONVP_LATE (percent)
        int convp_after_p0 = -1;        // DFX_CONVP_AFTER_P0
        int p0_ahead = 3;               // DFX_SEQ_P0_AHEAD
        int tail_every = 1;             // DFX_SEQ_TAIL_EVERY
        int dftail_every = 0;           // DFX_SEQ_DFTAIL_EVERY (0: the rule in forward_impl)
        int64_t fan_few_rows = 4096;    // DFX_FAN_FEW_ROWS
        int64_t convp_elems = (int64_t)1 << 29;   // DFX_CONVP_ELEMS (test hook)
    } sw;
    // DFX_FRONT_GRAIN=k[,kp]: k (kp) times as many, shorter workgroups for df_conv0->1 (df_convp).  df_convp owns whole SIMDs (one wave of
    // 512 registers each) and is needed last (by df_out, deep in the GRU phase): as a persistent grid of long workgroups it held every SIMD
    // for 4.7 ms while the kernels on the front's critical path (ERB encoder convs -> embedding GEMMs) waited for slots (erb_conv2: 1.8 ms
    // instead of 0.3).  In 16 x shorter workgroups (40-frame segments, 10 % warm-up overhead) its slots come free every ~35 us and the
    // dispatcher lets the other queues in: the front ends 1.1 ms earlier, df_convp finishes under the first 2 ms of the GRU phase;
    // 17.7 -> 17.15 ms per step (8 ... 32: the same; df_conv0->1's own grain: no effect).
    int front_grain = 1, front_grain_p = 16;
    // Error words, written by kernels, read by the host (page-locked host memory the device can store to: dfx_env_err_words_alloc):
    // [0] unused, [1] fp16-split range, [2] a flag wait of the persistent GRU phase timed out.
    // The host looks at them wherever it waits for the device anyway (pass_begin, dfx_model_check) and at the start of every call
    // (model_poll: plain loads, no synchronisation), so a fault is reported by the NEXT call on the handle at the latest;
    // DFX_CHECK_EVERY_PASS=1 makes every call wait for its own pass and report its own faults.
    unsigned int *d_err = nullptr;      // device address
    unsigned int *h_err = nullptr;      // host address of the same words
    bool check_every_pass = false;
    int spin_limit = DFX_SYNC_SPIN_LIMIT;   // DFX_SYNC_SPIN_LIMIT=n (tests: force the timeouts)
    bool hwq_probe_pending = false;     // the handshake of the persistent phase's streams has not run yet (hwq_probe_run)
    int hwq_probe = -1;                 // -1 not run, 1: the phase's streams run concurrently, 0: they do not (event-based GRU phase instead)
    const float *p(size_t off) const { return d_w + off; }
};

namespace {
struct Prep {
    const float *blob;
    const DfxManifest &man;
    std::vector<float> out;
    std::string err;
    const float *get(const std::string &name, const DfxTensor **tt = nullptr) {
        const DfxTensor *t = man.find(name);
        if (!t) {
            err = "tensor not in manifest: " + name;
            return nullptr;
        }
        if (tt) *tt = t;
        return blob + t->offset;
    }
    size_t alloc(size_t n) {  // 16-byte aligned carve
        size_t off = (out.size() + 3) & ~(size_t)3;
        out.resize(off + n, 0.f);
        return off;
    }
    // BatchNorm2d eval: y = x*scale + shift
    bool bn(const std::string &p, int n, std::vector<float> &scale, std::vector<float> &shift) {
        const float *g = get(p + ".weight"), *b = get(p + ".bias"), *m = get(p + ".running_mean"), *v = get(p + ".running_var");
        if (!g || !b || !m || !v) return false;
        scale.resize(n);
        shift.resize(n);
        for (int i = 0; i < n; ++i) {
            const float s = g[i] / sqrtf(v[i] + 1e-5f);
            scale[i] = s;
            shift[i] = b[i] - m[i] * s;
        }
        return true;
    }
};

// depthwise [C,1,1,3] (+transposed [C,1,1,3]) + pointwise [C,C,1,1] + BN at Sequential indices .0/.1/.2
bool prep_sep(Prep &P, const std::string &name, int C, PwW &w) {
    const float *dw = P.get(name + ".0.weight"), *pw = P.get(name + ".1.weight");
    std::vector<float> sc, sh;
    if (!dw || !pw || !P.bn(name + ".2", C, sc, sh)) return false;
    w.dw = P.alloc(3 * C);
    w.wt = P.alloc((size_t)C * C);
    w.bias = P.alloc(C);
    for (int c = 0; c < C; ++c)
        for (int j = 0; j < 3; ++j) P.out[w.dw + j * C + c] = dw[c * 3 + j];
    for (int n = 0; n < C; ++n)
        for (int k = 0; k < C; ++k) P.out[w.wt + (size_t)k * C + n] = pw[(size_t)n * C + k] * sc[n];
    for (int n = 0; n < C; ++n) P.out[w.bias + n] = sh[n];
    return true;
}
// pathway conv{N}p: Conv2d(C,C,1,groups=C) [C,1,1,1] + BN at .0/.1  ->  y = relu(a*x + b)
bool prep_path(Prep &P, const std::string &name, int C, size_t &a_off, size_t &b_off) {
    const float *w = P.get(name + ".0.weight");
    std::vector<float> sc, sh;
    if (!w || !P.bn(name + ".1", C, sc, sh)) return false;
    a_off = P.alloc(C);
    b_off = P.alloc(C);
    for (int c = 0; c < C; ++c) {
        P.out[a_off + c] = w[c] * sc[c];
        P.out[b_off + c] = sh[c];
    }
    return true;
}
// fp16-split MFMA A fragments: nfrag fragments of [hi, lo][64 lanes][8 halves]; val(frag, lane, i) is the fp32 weight of that slot.
// All values are scaled by one power of two so that the largest is just below 2^14 (lo parts stay normal f16); *unscale undoes it.
template <typename F>
size_t pack_h3(Prep &P, int nfrag, F val, float *unscale) {
    float mx = 0.f;
    for (int fr = 0; fr < nfrag; ++fr)
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(val(fr, l, i)));
    int e = 0;
    if (mx > 0.f) {
        int ex;
        frexpf(mx, &ex);
        e = 14 - ex;
        if (e > 24) e = 24;
        if (e < -14) e = -14;
    }
    const float sc = ldexpf(1.f, e);
    *unscale = ldexpf(1.f, -e);
    const size_t off = P.alloc((size_t)nfrag * 2 * 64 * 8 / 2);
    uint16_t *dst = reinterpret_cast<uint16_t *>(&P.out[off]);
    for (int fr = 0; fr < nfrag; ++fr)
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) {
                const float w = val(fr, l, i) * sc;
                const uint16_t hb = dfx_f32_to_f16_bits(w);
                const uint16_t lb = dfx_f32_to_f16_bits(w - dfx_f16_bits_to_f32(hb));
                dst[(((size_t)fr * 2 + 0) * 64 + l) * 8 + i] = hb;
                dst[(((size_t)fr * 2 + 1) * 64 + l) * 8 + i] = lb;
            }
    return off;
}
// pointwise weights of a separable block as dfx_chain_stage_h3's A fragments: fragment (nt, kc), lane l, element i = the weight of output
// channel 16 nt + (l & 15) for input channel (C/4) (l >> 4) + 8 kc + i (lane (pos, q) owns the C/4 consecutive channels from (C/4) q)
void pack_pw_h3(Prep &P, int C, PwW &w) {
    if (C % 32 != 0) return;
    const int KC = C / 32;
    const std::vector<float> src(P.out.begin() + (long)w.wt, P.out.begin() + (long)w.wt + (size_t)C * C);   // pack_h3 may reallocate P.out
    w.wt_h3 = pack_h3(P, (C / 16) * KC, [&](int fr, int l, int i) {
        const int nt = fr / KC, kc = fr % KC;
        return src[(size_t)((C / 4) * (l >> 4) + 8 * kc + i) * C + 16 * nt + (l & 15)];
    }, &w.unscale);
}
bool prep_glin(Prep &P, const std::string &name, GlinW &g) {
    const DfxTensor *t = nullptr;
    const float *w = P.get(name, &t);
    if (!w) return false;
    g.G = (int)t->shape[0];
    g.Kg = (int)t->shape[1];
    g.Ng = (int)t->shape[2];
    g.w = P.alloc((size_t)t->numel());
    memcpy(&P.out[g.w], w, sizeof(float) * (size_t)t->numel());
    return true;
}
// Fragments of dfx_k_emb_fan.  Needs the nesting of DeepFilterNet3's released shape: linear_out of the encoder GRU in groups of
// 16 -> 32 (lin_groups = 16 over 256 -> 512), erb_dec's linear_in and df_skip in groups of 32 -> 16 ("narrow": lin_groups = 16 over
// 512 -> 256), df_gru's linear_in in groups of 64 -> 32 ("wide": SqueezedGRU_S's default of 8 groups, deepfilternet3.py:296-302), so that
// y[:, 32J:32J+32] -> emb[:, 64J:64J+64] -> out[:, 32J:32J+32] is closed for every super-chunk J.  Other shapes: fan_nj stays 0.
//   fragment (J, i), lane l = (m = l & 15, kq = l >> 4), component s:
//     i = 2 ch + t            stage 1, chunk ch of the super-chunk, output features 16 t + m of it: W_out[2J + ch][4 kq + s][16 t + m]
//     i = 4 + 8 c + 2 u + t   narrow consumer c, output column 16 u + m: W_c[2J + u][16 t + 4 kq + s][m]
//     i = 4 + 8 c + 4 u + tt  wide consumer c, output column 16 u + m:   W_c[J][16 tt + 4 kq + s][16 u + m]
void pack_fan(Prep &P, dfx_model *m) {
    const GlinW &o = m->enc_out;
    const GlinW *cons[DFX_FAN_NC] = {&m->dec_in, &m->dfg_in, m->cfg.df_gru_skip == DFX_SKIP_GROUPEDLINEAR ? &m->df_skip : nullptr};
    if (o.Kg != 16 || o.Ng != 32 || o.G % 2 != 0) return;
    const int nj = o.G / 2;
    for (int c = 0; c < DFX_FAN_NC; ++c) {
        const GlinW *g = cons[c];
        m->fan_kind[c] = 0;
        if (!g) continue;
        if (g->G == 2 * nj && g->Kg == 32 && g->Ng == 16) m->fan_kind[c] = 1;
        else if (g->G == nj && g->Kg == 64 && g->Ng == 32) m->fan_kind[c] = 2;
        else return;
    }
    // the instantiated combinations (launch_emb_fan)
    if (m->fan_kind[0] != 1 || m->fan_kind[1] != 2 || (m->fan_kind[2] != 1 && m->fan_kind[2] != 0)) return;
    const size_t off = P.alloc((size_t)nj * DFX_FAN_WPJ * 64 * 4);   // (may reallocate P.out: offsets only below)
    auto frag = [&](int J, int i, int l, int sidx) -> float & { return P.out[off + (((size_t)J * DFX_FAN_WPJ + i) * 64 + l) * 4 + sidx]; };
    for (int J = 0; J < nj; ++J)
        for (int l = 0; l < 64; ++l) {
            const int mm = l & 15, kq = l >> 4;
            for (int sidx = 0; sidx < 4; ++sidx) {
                for (int ch = 0; ch < 2; ++ch)
                    for (int t = 0; t < 2; ++t) frag(J, 2 * ch + t, l, sidx) = P.out[o.w + ((size_t)(2 * J + ch) * 16 + 4 * kq + sidx) * 32 + 16 * t + mm];
                for (int c = 0; c < DFX_FAN_NC; ++c) {
                    if (m->fan_kind[c] == 1) {
                        for (int u = 0; u < 2; ++u)
                            for (int t = 0; t < 2; ++t)
                                frag(J, 4 + 8 * c + 2 * u + t, l, sidx) = P.out[cons[c]->w + ((size_t)(2 * J + u) * 32 + 16 * t + 4 * kq + sidx) * 16 + mm];
                    } else if (m->fan_kind[c] == 2) {
                        for (int u = 0; u < 2; ++u)
                            for (int tt = 0; tt < 4; ++tt)
                                frag(J, 4 + 8 * c + 4 * u + tt, l, sidx) = P.out[cons[c]->w + ((size_t)J * 64 + 16 * tt + 4 * kq + sidx) * 32 + 16 * u + mm];
                    }
                }
            }
        }
    m->fan_w = off;
    m->fan_chunks = nj;
}
// Fragments of dfx_k_enc_fan: df_fc_emb in groups of 96 -> 16 (enc_lin_groups = 32 over 3072 -> 512) feeding linear_in of the encoder GRU in
// groups of 32 -> 16 (lin_groups = 16 over 512 -> 256), i.e. two groups of the first per group of the second.
//   w1 (g, i), lane (m, kq), component s: W_fc[g][16 i + 4 kq + s][m];   w2 (h, t), component r: W_in[h][16 t + 4 kq + r][m]
void pack_encfan(Prep &P, dfx_model *m) {
    const GlinW &a = m->fc_emb, &b = m->enc_in;
    if (a.Kg != 96 || a.Ng != 16 || a.G % 2 != 0 || b.G * 2 != a.G || b.Kg != 32 || b.Ng != 16) return;
    const size_t o1 = P.alloc((size_t)a.G * 6 * 64 * 4);
    const size_t o2 = P.alloc((size_t)b.G * 2 * 64 * 4);
    for (int l = 0; l < 64; ++l) {
        const int mm = l & 15, kq = l >> 4;
        for (int sidx = 0; sidx < 4; ++sidx) {
            for (int g = 0; g < a.G; ++g)
                for (int i = 0; i < 6; ++i) P.out[o1 + (((size_t)g * 6 + i) * 64 + l) * 4 + sidx] = P.out[a.w + ((size_t)g * 96 + 16 * i + 4 * kq + sidx) * 16 + mm];
            for (int h = 0; h < b.G; ++h)
                for (int t = 0; t < 2; ++t) P.out[o2 + (((size_t)h * 2 + t) * 64 + l) * 4 + sidx] = P.out[b.w + ((size_t)h * 32 + 16 * t + 4 * kq + sidx) * 16 + mm];
        }
    }
    m->efan_w1 = o1, m->efan_w2 = o2;
    m->efan_groups = a.G;
}
bool prep_gru(Prep &P, const std::string &name, int layers, std::vector<GruW> &out) {
    const int H = 256;
    for (int l = 0; l < layers; ++l) {
        const std::string s = std::to_string(l);
        const float *wih = P.get(name + ".weight_ih_l" + s), *whh = P.get(name + ".weight_hh_l" + s);
        const float *bih = P.get(name + ".bias_ih_l" + s), *bhh = P.get(name + ".bias_hh_l" + s);
        if (!wih || !whh || !bih || !bhh) return false;
        GruW g;
        g.wih_t = P.alloc((size_t)H * 3 * H);
        g.bias_i = P.alloc(3 * H);
        g.whh4 = P.alloc((size_t)H * 3 * H);
        g.bhn = P.alloc(H);
        for (int n = 0; n < 3 * H; ++n)
            for (int k = 0; k < H; ++k) P.out[g.wih_t + (size_t)k * 3 * H + n] = wih[(size_t)n * H + k];
        // r and z gates: both biases can be summed up front; the n gate keeps b_hn inside r*(...)
        for (int n = 0; n < 3 * H; ++n) P.out[g.bias_i + n] = bih[n] + (n < 2 * H ? bhh[n] : 0.f);
        for (int k4 = 0; k4 < H / 4; ++k4)
            for (int gate = 0; gate < 3; ++gate)
                for (int j = 0; j < H; ++j)
                    for (int e = 0; e < 4; ++e)
                        P.out[g.whh4 + ((((size_t)k4 * 3 + gate) * H + j) * 4 + e)] = whh[(size_t)(gate * H + j) * H + 4 * k4 + e];
        for (int j = 0; j < H; ++j) P.out[g.bhn + j] = bhh[2 * H + j];
        {   // fp16-split fragments of W[k][n] = wih[n][k]: [n/64][kc][ct][hi,lo][lane][8], scaled by 2^e so that lo is a normal f16
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)3 * H * H; ++i) mx = fmaxf(mx, fabsf(wih[i]));
            int e = 0;
            if (mx > 0.f) {
                int ex;
                frexpf(mx, &ex);          // mx = f * 2^ex, f in [0.5, 1)
                e = 14 - ex;              // scaled magnitude < 2^14
                if (e > 24) e = 24;
                if (e < -14) e = -14;
            }
            const float sc = ldexpf(1.f, e);
            g.wih_unscale = ldexpf(1.f, -e);
            const size_t nh = (size_t)3 * H * H * 2;  // halves
            g.wih_h3 = P.alloc(nh / 2);
            uint16_t *dst = reinterpret_cast<uint16_t *>(&P.out[g.wih_h3]);
            for (int ch = 0; ch < 3 * H / 64; ++ch)
                for (int kc = 0; kc < 8; ++kc)
                    for (int ct = 0; ct < 4; ++ct)
                        for (int l = 0; l < 64; ++l)
                            for (int i = 0; i < 8; ++i) {
                                const int n = ch * 64 + ct * 16 + (l & 15), k = 32 * kc + 8 * (l >> 4) + i;
                                const float w = wih[(size_t)n * H + k] * sc;
                                const uint16_t hb = dfx_f32_to_f16_bits(w);
                                const uint16_t lb = dfx_f32_to_f16_bits(w - dfx_f16_bits_to_f32(hb));
                                const size_t frag = (((size_t)ch * 8 + kc) * 4 + ct) * 2;
                                dst[((frag + 0) * 64 + l) * 8 + i] = hb;
                                dst[((frag + 1) * 64 + l) * 8 + i] = lb;
                            }
        }
        auto pack_whh_pj = [&](float sc) {   // W_hh in the same fragment order (dfx_k_gru_step_h3)
            g.whh_pj = P.alloc((size_t)3 * H * H);
            uint16_t *dst = reinterpret_cast<uint16_t *>(&P.out[g.whh_pj]);
            for (int ch = 0; ch < 3 * H / 64; ++ch)
                for (int kc = 0; kc < 8; ++kc)
                    for (int ct = 0; ct < 4; ++ct)
                        for (int l = 0; l < 64; ++l)
                            for (int i = 0; i < 8; ++i) {
                                const int n = ch * 64 + ct * 16 + (l & 15), k = 32 * kc + 8 * (l >> 4) + i;
                                const float w = whh[(size_t)n * H + k] * sc;
                                const uint16_t hb = dfx_f32_to_f16_bits(w);
                                const uint16_t lb = dfx_f32_to_f16_bits(w - dfx_f16_bits_to_f32(hb));
                                const size_t frag = (((size_t)ch * 8 + kc) * 4 + ct) * 2;
                                dst[((frag + 0) * 64 + l) * 8 + i] = hb;
                                dst[((frag + 1) * 64 + l) * 8 + i] = lb;
                            }
        };
        {   // W_hh fragments for dfx_k_gru_rec_h3: [16-unit tile][k-chunk][gate][hi,lo][lane][8]
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)3 * H * H; ++i) mx = fmaxf(mx, fabsf(whh[i]));
            int e = 0;
            if (mx > 0.f) {
                int ex;
                frexpf(mx, &ex);
                e = 14 - ex;
                if (e > 24) e = 24;
                if (e < -14) e = -14;
            }
            const float sc = ldexpf(1.f, e);
            g.whh_unscale = ldexpf(1.f, -e);
            pack_whh_pj(sc);
            g.whh_h3 = P.alloc((size_t)3 * H * H);  // 2 halves per weight
            uint16_t *dst = reinterpret_cast<uint16_t *>(&P.out[g.whh_h3]);
            for (int ut = 0; ut < 16; ++ut)          // 16-unit tile
                for (int kc = 0; kc < 8; ++kc)
                    for (int gate = 0; gate < 3; ++gate)
                        for (int l = 0; l < 64; ++l)
                            for (int i = 0; i < 8; ++i) {
                                const int unit = 16 * ut + (l & 15), k = 32 * kc + 8 * (l >> 4) + i;
                                const float v = whh[(size_t)(gate * H + unit) * H + k] * sc;
                                const uint16_t hb = dfx_f32_to_f16_bits(v);
                                const uint16_t lb = dfx_f32_to_f16_bits(v - dfx_f16_bits_to_f32(hb));
                                const size_t frag = (((size_t)ut * 8 + kc) * 3 + gate) * 2;
                                dst[((frag + 0) * 64 + l) * 8 + i] = hb;
                                dst[((frag + 1) * 64 + l) * 8 + i] = lb;
                            }
        }
        {   // the same fragment order in fp32: [16-unit tile][k-chunk][gate][half][lane][4], half h = weights 32 kc + 8 q + 4 h + 0..3 of the lane's unit
            g.whh_x32 = P.alloc((size_t)3 * H * H);
            float *dst = &P.out[g.whh_x32];
            for (int ut = 0; ut < 16; ++ut)
                for (int kc = 0; kc < 8; ++kc)
                    for (int gate = 0; gate < 3; ++gate)
                        for (int half = 0; half < 2; ++half)
                            for (int l = 0; l < 64; ++l)
                                for (int i = 0; i < 4; ++i) {
                                    const int unit = 16 * ut + (l & 15), k = 32 * kc + 8 * (l >> 4) + 4 * half + i;
                                    const size_t frag = (((size_t)ut * 8 + kc) * 3 + gate) * 2 + half;
                                    dst[(frag * 64 + l) * 4 + i] = whh[(size_t)(gate * H + unit) * H + k];
                                }
        }
        out.push_back(g);
    }
    return true;
}
}  // namespace

static std::mutex &dfx_laneset_mu() {
    static std::mutex mu;
    return mu;
}
static DfxLaneSet *dfx_laneset_of_device() {   // (never freed: streams of a process live as long as it does)
    static std::map<int, DfxLaneSet *> sets;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(dfx_laneset_mu());
    DfxLaneSet *&ls = sets[dev];
    if (!ls) ls = new DfxLaneSet();
    return ls;
}
// The persistent GRU phase across PROCESSES.  Its ~160 workgroups must all be resident (each owns a CU); two such passes of two processes on one
// GPU can each hold the CUs the other's missing workgroups wait for, and both end in flag-wait timeouts (a fault the host is told about, but two
// seconds late and with both passes lost).  Inside a process the passes take turns (DfxTurn); between processes there is a ticket per device: an
// advisory lock on /dev/shm/dfx_persistent_<PCI bus id>.lock, taken without waiting in front of a persistent phase and given back by a host callback
// behind the pass (on a side stream: the caller's stream does not wait for the callback).  A pass that does not get the ticket runs the
// event-synchronised form of the phase (no residency requirement; ~1.4 x slower), so nobody ever waits for another process.  DFX_DEVICE_TICKET=0:
// off (one process per GPU is guaranteed by the deployment); processes that do not share /dev/shm do not see each other.
struct DfxTicket {
    int fd = -2;                    // -2 not opened yet, -1 unavailable
    std::atomic<int> holds{0};      // passes of this process that hold it
};
static DfxTicket &dfx_ticket() {
    static DfxTicket t;
    return t;
}
static bool dfx_ticket_try() {   // (under the enqueue lock)
    DfxTicket &t = dfx_ticket();
    if (t.fd == -2) {
        const char *e = getenv("DFX_DEVICE_TICKET");
        t.fd = -1;
        if (!(e && e[0] == '0')) {
            int dev = 0;
            char bus[64] = "gpu";
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev);
            for (char *c = bus; *c; ++c)
                if (*c == ':' || *c == '.' || *c == '/') *c = '_';
            char path[160];
            snprintf(path, sizeof(path), "/dev/shm/dfx_persistent_%s.lock", bus);
            t.fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
            if (t.fd >= 0) (void)fchmod(t.fd, 0666);
        }
    }
    if (t.fd < 0) return true;                      // no ticket office: as before
    if (t.holds.load() > 0) {                       // this process already holds it (an earlier pass still in flight)
        t.holds.fetch_add(1);
        return true;
    }
    if (flock(t.fd, LOCK_EX | LOCK_NB) != 0) return false;
    t.holds.fetch_add(1);
    return true;
}
static void dfx_ticket_release_cb(void *) {
    DfxTicket &t = dfx_ticket();
    if (t.fd >= 0 && t.holds.fetch_sub(1) == 1) (void)flock(t.fd, LOCK_UN);
}
// One enqueue at a time per process: every entry point that puts work on the process's internal streams holds this lock while it does
// (forward, enhance, the streaming calls); see DfxTurn.
static std::mutex &dfx_enqueue_mu() {
    static std::mutex mu;
    return mu;
}
static bool dfx_create_lane(dfx_model *m, int l) {
    if (!m->laneset) {
        m->laneset = dfx_laneset_of_device();
        if (!m->laneset) return false;
        m->lanes = m->laneset->lanes;
    }
    // idempotent: whatever of the lane does not exist yet is created (a second handle finds the first one's streams; a deeper model adds its layers')
    std::lock_guard<std::mutex> lk(dfx_laneset_mu());
    DfxLane &ln = m->lanes[l];
    bool good = true;
    auto stream = [&](hipStream_t &st) {
        if (!st) good = good && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    };
    auto event = [&](hipEvent_t &e) {
        if (!e) good = good && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    };
    stream(ln.main);
    for (int i = 0; i < 2; ++i) stream(ln.aux[i]);
    for (int i = 0; i < DFX_LANE_EVENTS; ++i) event(ln.ev[i]);
    if (l == 0) {  // the layer-pipelined GRU phase runs on lane 0 only
        const int nl_model = 1 + (m->cfg.emb_num_layers - 1) + m->cfg.df_num_layers;
        const int nl = nl_model > 5 ? nl_model : 5;   // (at least the five layers of the shipped shapes: the handshake then covers what later handles use)
        for (int i = 0; i < nl && i < DFX_MAX_GRU_LAYERS; ++i) {
            if (i > 0) stream(ln.gs[i]);  // layer 0 recurs on the caller's stream
            stream(ln.ps[i]);             // (highest priority for these was measured: the spinning wait kernels then starve every other queue, seconds per step)
            for (int k = 0; k < DFX_MAX_TCHUNKS; ++k) event(ln.gev[i][k]), event(ln.pev[i][k]);
        }
        for (int i = 0; i < 2; ++i) stream(ln.ts[i]);
        for (int k = 0; k < DFX_MAX_TCHUNKS; ++k) event(ln.eev[k]);
    }
    return good;
}

// The persistent GRU phase synchronises through device flags: the stream of the persistent launch, the preparation stream of
// every layer and the two decoder-tail streams must make progress independently.  Streams that share a hardware queue
// (GPU_MAX_HW_QUEUES left at ROCm's default of 4, or set after HIP had initialised) would put a spinning wait in front of the
// launch it waits for — a timeout and an invalid pass.  Checked once per handle with a handshake between exactly those streams, the
// first time it matters (hwq_probe_pending).
static void hwq_probe_fallback(dfx_model *m) {   // the event-synchronised form (DFX_GRU_SEQ=0) needs no concurrency to be correct
    m->gru_seq = false;
    if (!(getenv("DFX_QUIET") && getenv("DFX_QUIET")[0] == '1'))
        fprintf(stderr, "dfx: the streams of the persistent GRU phase do not run concurrently (GPU_MAX_HW_QUEUES >= 16 must be in the environment "
                        "before HIP initialises): this model uses the slower event-synchronised GRU phase (dfx_model_query DFX_Q_HWQ_PROBE = 0)\n");
}
// (the caller holds the enqueue lock, DfxTurn: the handshake synchronises the process's internal streams, so whatever another handle has in flight
// on them is over before it starts; its result belongs to the lane set — one handshake per process and device)
static void hwq_probe_run(dfx_model *m) {
    if (!m->hwq_probe_pending) return;
    m->hwq_probe_pending = false;
    DfxLaneSet *ls = m->laneset;
    if (ls && ls->hwq_probe >= 0) {
        m->hwq_probe = ls->hwq_probe;
        if (m->hwq_probe == 0) hwq_probe_fallback(m);
        return;
    }
    const DfxLane &ln = m->lanes[0];
    std::vector<hipStream_t> ss;
    if (ln.gs[1]) ss.push_back(ln.gs[1]);
    for (int i = 0; i < DFX_MAX_GRU_LAYERS; ++i)
        if (ln.ps[i]) ss.push_back(ln.ps[i]);
    for (int i = 0; i < 2; ++i)
        if (ln.ts[i]) ss.push_back(ln.ts[i]);
    unsigned int *cnt = m->d_sync + 13;   // (a spare word of the flag block: ready 0-7 | emb 8 | probe 13 | done 16-)
    // warm-up: the kernel's code object is loaded and every stream's queue exists before the bounded handshake starts (a slow first
    // launch must not look like a shared queue); a failed handshake is tried once more with a longer bound before it counts
    for (hipStream_t st : ss) dfx_launch(dfx_k_probe_meet, dim3(1), dim3(64), 0, st, cnt, 0u, 1, m->d_err + 9);
    bool okp = hipGetLastError() == hipSuccess;
    for (hipStream_t st : ss) okp = hipStreamSynchronize(st) == hipSuccess && okp;
    m->hwq_probe = 0;
    for (int attempt = 0; attempt < 2 && okp && m->hwq_probe == 0; ++attempt) {
        (void)hipMemset(cnt, 0, sizeof(unsigned int));
        m->h_err[8] = 0u;
        for (hipStream_t st : ss) dfx_launch(dfx_k_probe_meet, dim3(1), dim3(64), 0, st, cnt, (unsigned int)ss.size(), 1 << (16 + 3 * attempt), m->d_err + 8);
        okp = hipGetLastError() == hipSuccess;
        for (hipStream_t st : ss) okp = hipStreamSynchronize(st) == hipSuccess && okp;
        m->hwq_probe = okp && ((volatile unsigned int *)m->h_err)[8] == 0u ? 1 : 0;
    }
    m->h_err[8] = 0u, m->h_err[9] = 0u;
    (void)hipMemset(cnt, 0, sizeof(unsigned int));
    if (ls) ls->hwq_probe = m->hwq_probe;
    if (m->hwq_probe == 0) hwq_probe_fallback(m);
}

extern "C" int dfx_model_create(const dfx_model_cfg *cfg, const float *blob, dfx_model **out) {
    if (int rc = check_cfg(cfg)) return rc;
    if (!blob || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_create: null");
    if (int rc = dfx_require_device()) return rc;
    const dfx_model_cfg &c = *cfg;
    const DfxManifest man = dfx_build_manifest(c);
    Prep P{blob, man, {}, {}};
    dfx_model *m = new dfx_model();
    m->cfg = c;
    const int C = c.conv_ch, O = c.df_order;
    bool ok = true;
    {   // enc.erb_conv0: pad(.0) conv(.1) bn(.2)
        const float *w = P.get("enc.erb_conv0.1.weight");
        std::vector<float> sc, sh;
        ok = ok && w && P.bn("enc.erb_conv0.2", C, sc, sh);
        if (ok) {
            m->erb0_w = P.alloc(9 * C);
            m->erb0_b = P.alloc(C);
            for (int ch = 0; ch < C; ++ch) {
                for (int k = 0; k < 9; ++k) P.out[m->erb0_w + k * C + ch] = w[ch * 9 + k] * sc[ch];
                P.out[m->erb0_b + ch] = sh[ch];
            }
        }
    }
    ok = ok && prep_sep(P, "enc.erb_conv1", C, m->erb1) && prep_sep(P, "enc.erb_conv2", C, m->erb2) &&
         prep_sep(P, "enc.erb_conv3", C, m->erb3) && prep_sep(P, "enc.df_conv1", C, m->dfc1);
    if (ok) {   // enc.df_conv0: pad(.0) conv groups=2 (.1) pointwise(.2) bn(.3), folded into one dense conv (K = 18 -> 20)
        const float *w = P.get("enc.df_conv0.1.weight"), *pw = P.get("enc.df_conv0.2.weight");
        std::vector<float> sc, sh;
        ok = w && pw && P.bn("enc.df_conv0.3", C, sc, sh);
        if (ok) {
            m->cin_weff = P.alloc((size_t)20 * C);
            m->cin_b = P.alloc(C);
            for (int n = 0; n < C; ++n) {
                for (int tap = 0; tap < 9; ++tap)
                    for (int ch = 0; ch < 2; ++ch) {
                        double acc = 0.0;  // groups=2: output channel c of the 3x3 conv sees input ch = c / (C/2)
                        for (int c = ch * (C / 2); c < (ch + 1) * (C / 2); ++c) acc += (double)pw[(size_t)n * C + c] * (double)w[c * 9 + tap];
                        P.out[m->cin_weff + (size_t)(tap * 2 + ch) * C + n] = (float)(acc * (double)sc[n]);
                    }
                P.out[m->cin_b + n] = sh[n];
            }
        }
    }
    ok = ok && prep_glin(P, "enc.df_fc_emb.0.weight", m->fc_emb) && prep_glin(P, "enc.emb_gru.linear_in.0.weight", m->enc_in) &&
         prep_gru(P, "enc.emb_gru.gru", 1, m->enc_gru) && prep_glin(P, "enc.emb_gru.linear_out.0.weight", m->enc_out);
    if (ok && c.emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR) ok = prep_glin(P, "enc.emb_gru.gru_skip.weight", m->enc_skip);
    if (ok && c.emb_gru_skip == DFX_SKIP_GROUPEDLINEAR) ok = prep_glin(P, "erb_dec.emb_gru.gru_skip.weight", m->dec_skip);
    if (ok) {
        const float *w = P.get("enc.lsnr_fc.0.weight"), *b = P.get("enc.lsnr_fc.0.bias");
        ok = w && b;
        if (ok) {
            const int emb = C * c.nb_erb / 4;
            m->lsnr_w = P.alloc(emb);
            memcpy(&P.out[m->lsnr_w], w, sizeof(float) * emb);
            m->lsnr_b = b[0];
        }
    }
    ok = ok && prep_glin(P, "erb_dec.emb_gru.linear_in.0.weight", m->dec_in) &&
         prep_gru(P, "erb_dec.emb_gru.gru", c.emb_num_layers - 1, m->dec_gru) &&
         prep_glin(P, "erb_dec.emb_gru.linear_out.0.weight", m->dec_out);
    ok = ok && prep_sep(P, "erb_dec.convt3", C, m->ct3) && prep_path(P, "erb_dec.conv3p", C, m->ct3.sk_a, m->ct3.sk_b) &&
         prep_sep(P, "erb_dec.convt2", C, m->ct2) && prep_path(P, "erb_dec.conv2p", C, m->ct2.sk_a, m->ct2.sk_b) &&
         prep_sep(P, "erb_dec.convt1", C, m->ct1) && prep_path(P, "erb_dec.conv1p", C, m->ct1.sk_a, m->ct1.sk_b) &&
         prep_path(P, "erb_dec.conv0p", C, m->co_ska, m->co_skb);
    m->ct3.has_skip = m->ct2.has_skip = m->ct1.has_skip = true;
    if (ok)
        for (PwW *w : {&m->erb1, &m->erb2, &m->erb3, &m->ct3, &m->ct2, &m->ct1}) pack_pw_h3(P, C, *w);
    if (ok) {   // erb_dec.conv0_out: conv [1,C,1,3] (.0) bn(1) (.1)
        const float *w = P.get("erb_dec.conv0_out.0.weight");
        std::vector<float> sc, sh;
        ok = w && P.bn("erb_dec.conv0_out.1", 1, sc, sh);
        if (ok) {
            m->co_w = P.alloc(3 * C);
            for (int ch = 0; ch < C; ++ch)
                for (int j = 0; j < 3; ++j) P.out[m->co_w + j * C + ch] = w[ch * 3 + j] * sc[0];
            m->co_bias = sh[0];
        }
    }
    if (ok && C % 32 == 0) {   // matrix-op forms of the two small contractions inside dfx_k_erb_tail (round 5)
        const int KC = C / 32;
        const std::vector<float> src(P.out.begin(), P.out.end());   // pack_h3 may reallocate P.out
        const size_t w0 = m->erb0_w, b0 = m->erb0_b, wo = m->co_w;
        // erb_conv0 as [C x 32] fragments per 16 channels: k-slot 8 (l >> 4) + i = tap 3 kt + kf for k < 9, the bias (against a constant 1) at k = 9
        m->tail_w0h3 = pack_h3(P, C / 16, [&](int nt, int l, int i) {
            const int k = 8 * (l >> 4) + i, ch = 16 * nt + (l & 15);
            return k < 9 ? src[w0 + (size_t)k * C + ch] : (k == 9 ? src[b0 + ch] : 0.f);
        }, &m->tail_w0_unscale);
        // conv0_out's three taps as the rows 0..2 of one 16-row tile; the contraction index is enumerated the way a D fragment leaves it (lane
        // (position, q): element 8 kc + i <-> channel 16 ((8 kc + i) >> 2) + 4 q + ((8 kc + i) & 3))
        m->tail_woh3 = pack_h3(P, KC, [&](int kc, int l, int i) {
            const int j = l & 15, e = 8 * kc + i, ch = 16 * (e >> 2) + 4 * (l >> 4) + (e & 3);
            return j < 3 ? src[wo + (size_t)j * C + ch] : 0.f;
        }, &m->tail_wo_unscale);
    }
    if (ok) {   // df_dec.df_convp
        const int kt = c.df_pathway_kernel_size_t, NO = 2 * O, G = dfx_gcd(C, NO), CG = C / G, OG = NO / G;
        const int idx = kt > 1 ? 1 : 0;
        const bool has_pw = kt > 1;  // separable only if groups > 1 and max(kernel) > 1; groups > 1 always holds (2 | C, 2O)
        const float *w = P.get("df_dec.df_convp." + std::to_string(idx) + ".weight");
        const float *pw = has_pw ? P.get("df_dec.df_convp." + std::to_string(idx + 1) + ".weight") : nullptr;
        std::vector<float> sc, sh;
        ok = w && (!has_pw || pw) && P.bn("df_dec.df_convp." + std::to_string(idx + (has_pw ? 2 : 1)), NO, sc, sh);
        if (ok) {
            m->cp_G = G;
            m->cp_NO = NO;
            m->cp_w1 = P.alloc((size_t)G * kt * CG * 16);
            m->cp_w2 = P.alloc((size_t)NO * NO);
            m->cp_b = P.alloc(NO);
            for (int g = 0; g < G; ++g)
                for (int k = 0; k < kt; ++k)
                    for (int ci = 0; ci < CG; ++ci)
                        for (int o = 0; o < OG; ++o) {
                            // weight [NO, CG, kt, 1]
                            float v = w[(((size_t)(g * OG + o) * CG + ci) * kt + k)];
                            if (!has_pw) v *= sc[g * OG + o];
                            P.out[m->cp_w1 + ((((size_t)g * kt + k) * CG + ci) * 16 + o)] = v;
                        }
            for (int n = 0; n < NO; ++n) {
                for (int o = 0; o < NO; ++o) P.out[m->cp_w2 + (size_t)n * NO + o] = has_pw ? pw[(size_t)n * NO + o] * sc[n] : (n == o ? 1.f : 0.f);
                P.out[m->cp_b + n] = sh[n];
            }
            // folded form: W_eff[k][c][n] = scale[n] * sum_o PW[n][o] * W1[o][c - group(o)*CG][k]   (PW = identity if absent).
            // One 16-wide MFMA tile of outputs: 2 * df_order <= 16; longer filters (BASELINE.json configs[4]: df_order = 10) take the
            // tiled kernel dfx_k_df_convp on a materialised c0.
            const bool folded = NO <= 16;
            m->cp_weff = P.alloc(folded ? (size_t)kt * C * 16 : 0);
            m->cp_b16 = P.alloc(16);
            for (int k = 0; folded && k < kt; ++k)
                for (int ch = 0; ch < C; ++ch) {
                    const int g = ch / CG, ci = ch - g * CG;
                    for (int n = 0; n < NO; ++n) {
                        double acc = 0.0;
                        for (int o = g * OG; o < (g + 1) * OG; ++o) {
                            const double p2 = has_pw ? (double)pw[(size_t)n * NO + o] : (n == o ? 1.0 : 0.0);
                            acc += p2 * (double)w[((size_t)o * CG + ci) * kt + k];
                        }
                        P.out[m->cp_weff + ((size_t)k * C + ch) * 16 + n] = (float)(acc * (double)sc[n]);
                    }
                }
            for (int n = 0; folded && n < NO; ++n) P.out[m->cp_b16 + n] = sh[n];
            if (C % 32 == 0 && folded) {  // fragments of the fused fp16-split DF-encoder kernels (dfx_k_df_conv01_h3, dfx_k_df_convp_h3)
                const int KC = C / 32;
                // channel a lane (q = l>>4) feeds as element i of k-chunk kc after dfx_c0_tile
                auto chan = [](int kc, int l, int i) { const int e = 8 * kc + i; return 16 * (e >> 2) + 4 * (l >> 4) + (e & 3); };
                const size_t cin = m->cin_weff, wt1 = m->dfc1.wt, cpw = m->cp_weff;
                // P.out may reallocate inside pack_h3 (alloc): read through offsets, never through cached pointers
                std::vector<float> src(P.out.begin(), P.out.end());
                m->c0_h3 = pack_h3(P, C / 16, [&](int nt, int l, int i) {
                    const int k = 8 * (l >> 4) + i;
                    return k < 18 ? src[cin + (size_t)k * C + 16 * nt + (l & 15)] : 0.f;
                }, &m->c0_unscale);
                m->dfc1_h3 = pack_h3(P, (C / 16) * KC, [&](int fr, int l, int i) {
                    const int nt = fr / KC, kc = fr % KC;
                    return src[wt1 + (size_t)chan(kc, l, i) * C + 16 * nt + (l & 15)];
                }, &m->dfc1_unscale);
                m->cp_h3 = pack_h3(P, kt * KC, [&](int fr, int l, int i) {
                    const int k = fr / KC, kc = fr % KC;
                    return src[cpw + ((size_t)k * C + chan(kc, l, i)) * 16 + (l & 15)];
                }, &m->cp_unscale);
            }
        }
    }
    ok = ok && prep_glin(P, "df_dec.df_gru.linear_in.0.weight", m->dfg_in) && prep_gru(P, "df_dec.df_gru.gru", c.df_num_layers, m->df_gru);
    if (ok && c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR) ok = prep_glin(P, "df_dec.df_skip.weight", m->df_skip);
    ok = ok && prep_glin(P, "df_dec.df_out.0.weight", m->df_out);
    if (ok) pack_fan(P, m);
    if (ok) pack_encfan(P, m);
    if (ok) {   // fragments of dfx_k_df_out_h3
        const GlinW a = m->df_out;
        const int NO = 2 * O, Fd = c.nb_df;
        if (a.Kg <= 32 && a.Kg % 8 == 0 && a.Ng % 2 == 0 && NO % 2 == 0 && Fd % 2 == 0 && (int64_t)a.G * a.Ng == (int64_t)NO * Fd &&
            DFX_DFO_SMEM(NO, Fd) <= (size_t)64 * 1024) {
            const std::vector<float> src(P.out.begin(), P.out.end());   // pack_h3 may reallocate P.out
            const int NU = (a.Ng + 15) / 16;
            m->dfo_h3 = pack_h3(P, a.G * NU, [&](int fr, int l, int i) {
                const int g = fr / NU, u = fr % NU, o = 16 * u + (l & 15), k = 8 * (l >> 4) + i;
                return o < a.Ng && k < a.Kg ? src[a.w + ((size_t)g * a.Kg + k) * a.Ng + o] : 0.f;
            }, &m->dfo_unscale);
            m->dfo_nu = NU;
        }
    }
    if (ok && C % 32 == 0 && m->c0_h3 && m->dfc1_h3) {   // fragments of the fused DF branch of the encoder (dfx_k_df_enc_h3)
        const GlinW a = m->fc_emb, b = m->enc_in;
        const int Fout = c.nb_df / 2, KC = C / 32;
        if (a.Kg % 32 == 0 && a.Ng == 16 && b.Kg == 32 && b.Ng == 16 && b.G * 2 == a.G && (int64_t)a.G * a.Kg == (int64_t)Fout * C) {
            const std::vector<float> src(P.out.begin(), P.out.end());   // pack_h3 may reallocate P.out
            // chunk ci = fo * KC + kc; lane (o = l & 15, q = l >> 4), element i <-> channel 16 ((8 kc + i) >> 2) + 4 q + (i & 3) of bin fo (the order
            // in which dfx_k_df_conv01_h3's D fragments hold df_conv1's output)
            m->dfenc_fc = pack_h3(P, Fout * KC, [&](int ci, int l, int i) {
                const int fo = ci / KC, kc = ci % KC, ch = 16 * ((8 * kc + i) >> 2) + 4 * (l >> 4) + (i & 3);
                const int idx = fo * C + ch, g = idx / a.Kg, kin = idx % a.Kg;
                return src[a.w + ((size_t)g * a.Kg + kin) * 16 + (l & 15)];
            }, &m->dfenc_fc_unscale);
            // linear_in group j: element i <-> feature 16 (i >> 2) + 4 q + (i & 3) of its 32 inputs (two finished fc groups)
            m->dfenc_in = pack_h3(P, b.G, [&](int j, int l, int i) {
                return src[b.w + ((size_t)j * 32 + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3)) * 16 + (l & 15)];
            }, &m->dfenc_in_unscale);
            m->dfenc_chunks = Fout * KC;
        }
    }
    if (!ok) {
        delete m;
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_create: %s", P.err.empty() ? "weight preparation failed" : P.err.c_str());
    }
    m->n_w = P.out.size();
    if (hipMalloc(reinterpret_cast<void **>(&m->d_w), m->n_w * sizeof(float)) != hipSuccess) {
        delete m;
        DFX_FAIL(DFX_ERR_ALLOC, "dfx_model_create: device allocation of %zu bytes failed", m->n_w * sizeof(float));
    }
    if (hipMemcpy(m->d_w, P.out.data(), m->n_w * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        dfx_model_free(m);
        DFX_FAIL(DFX_ERR_HIP, "dfx_model_create: upload failed");
    }
    {   // independent branches of the forward pass run on two auxiliary streams (DFX_STREAMS=0 keeps everything serial)
        const char *e = getenv("DFX_STREAMS");
        m->concurrent = !(e && e[0] == '0');
        const char *x = getenv("DFX_EXACT_FP32");
        m->exact_fp32 = x && x[0] == '1';
        const char *f0 = getenv("DFX_FUSE_C0"), *fe = getenv("DFX_FUSE_ERB");
        m->fuse_c0 = !(f0 && f0[0] == '0') && m->cfg.df_pathway_kernel_size_t <= 5 && 2 * m->cfg.df_order <= 16;
        // Exact mode, batch passes: c0 is written once and read by its two consumers instead of being recomputed by both — on fp32 matrix ops a c0
        // tile is 20 ops of 32 cycles (3 of 16 on the fp16-split path), and the two recomputing kernels side by side are the exact front's 10 ms:
        // 28.65 vs 29.45 ms per step (DFX_FUSE_C0=1 keeps the recomputing forms; the frame-by-frame runtime always uses them)
        m->c0_batch_unfused = m->exact_fp32 && m->fuse_c0 && !f0;
        m->fuse_erb = !(fe && fe[0] == '0');
        const char *gq = getenv("DFX_GRU_SEQ");
        m->gru_seq = !(gq && gq[0] == '0') && !dfx_env_is_emulator();
        const char *fem = getenv("DFX_FUSE_EMB");
        m->fuse_emb = !(fem && fem[0] == '0');
        const char *fef = getenv("DFX_FUSE_ENCFAN");
        m->fuse_encfan = !(fef && fef[0] == '0');
        const char *fdf = getenv("DFX_FUSE_DFA");
        m->fuse_dfa = !(fdf && fdf[0] == '0');
        const char *ftl = getenv("DFX_FUSE_TAIL");
        m->fuse_tail = !(ftl && ftl[0] == '0');
        const char *e0r = getenv("DFX_E0_RECOMPUTE");
        m->e0_recompute = !(e0r && e0r[0] == '0');
        const char *fde = getenv("DFX_FUSE_DFENC");
        m->fuse_dfenc = !(fde && fde[0] == '0');
        const char *dfl = getenv("DFX_DFOUT_LEAN");
        m->dfout_lean = !(dfl && dfl[0] == '0');
        const char *cep = getenv("DFX_CHECK_EVERY_PASS"), *spl = getenv("DFX_SYNC_SPIN_LIMIT");
        m->check_every_pass = cep && cep[0] == '1';
        if (spl && atoi(spl) > 0) m->spin_limit = atoi(spl);
        const char *pl = getenv("DFX_PHASE_LATE"), *prt = getenv("DFX_PROJ_RT");
        m->phase_late = !(pl && pl[0] == '0');
        m->proj_rt = prt ? atoi(prt) : 0;
        {
            auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
            auto env_pos = [](const char *name, int dflt) { const char *e = getenv(name); return e && atoi(e) > 0 ? atoi(e) : dflt; };
            auto env_on = [](const char *name) { const char *e = getenv(name); return !(e && e[0] == '0'); };
            m->sw.follow = env_int("DFX_SEQ_FOLLOW", 2);
            m->sw.chunks = env_pos("DFX_SEQ_CHUNKS", 0);
            m->sw.ramp = env_int("DFX_SEQ_RAMP", 0);
            m->sw.publish = env_on("DFX_SEQ_PUBLISH");
            m->sw.xcd_light = env_on("DFX_SEQ_XCD_LIGHT");
            m->sw.convp_late = env_int("DFX_CONVP_LATE", -1);
            if (m->sw.convp_late > 100) m->sw.convp_late = 100;
            m->sw.convp_after_p0 = env_int("DFX_CONVP_AFTER_P0", -1);
            m->sw.p0_ahead = env_pos("DFX_SEQ_P0_AHEAD", 3);
            m->sw.tail_every = env_pos("DFX_SEQ_TAIL_EVERY", 1);
            m->sw.dftail_every = env_pos("DFX_SEQ_DFTAIL_EVERY", 0);
            const char *ffr = getenv("DFX_FAN_FEW_ROWS"), *cel = getenv("DFX_CONVP_ELEMS");
            if (ffr && atoll(ffr) > 0) m->sw.fan_few_rows = atoll(ffr);
            if (cel && atoll(cel) > 0) m->sw.convp_elems = atoll(cel);
        }
        const char *fg = getenv("DFX_FRONT_GRAIN");
        if (fg) {
            m->front_grain = atoi(fg) > 1 ? atoi(fg) : 1;
            m->front_grain_p = m->front_grain;
            if (strchr(fg, ',')) m->front_grain_p = atoi(strchr(fg, ',') + 1) > 1 ? atoi(strchr(fg, ',') + 1) : 1;
        }
        {
            const char *tq = getenv("DFX_SEQ_TRACE");
            if (tq && tq[0] == '1') (void)hipMalloc(reinterpret_cast<void **>(&m->d_trace), (size_t)DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX * DFX_GS_MAX_CHUNKS * 3 * 8);
        }
        // ready 0-7 | emb 8 | probe 13 | done 16- | producers' completion counters (DfxPublish): 9 words (16) | yprog, giprog: steps per (layer, group)
        // | XCD registrations [3 kinds][layers][groups] (DfxXcd) | 64 words: [0] light hand-overs counted (dev aid), [8..48) the followers' claim counters
        const size_t sync_bytes = (size_t)(16 + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 16 + 2 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 3 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 64) * sizeof(unsigned int);
        if (hipMalloc(reinterpret_cast<void **>(&m->d_sync), sync_bytes) != hipSuccess || hipMemset(m->d_sync, 0, sync_bytes) != hipSuccess ||
            dfx_env_err_words_alloc(&m->h_err, &m->d_err, 256) != hipSuccess) {
            dfx_model_free(m);
            DFX_FAIL(DFX_ERR_ALLOC, "dfx_model_create: device allocation failed");
        }
        const char *tc = getenv("DFX_TCHUNKS");
        if (tc && atoi(tc) >= 1) m->tchunks = atoi(tc) < DFX_MAX_TCHUNKS ? atoi(tc) : DFX_MAX_TCHUNKS;
        const char *nc = getenv("DFX_CHUNKS");
        if (nc && atoi(nc) >= 1) m->max_chunks = atoi(nc) < DFX_MAX_LANES ? atoi(nc) : DFX_MAX_LANES;
        {
            // Streams are created sparingly: ROCm multiplexes them onto GPU_MAX_HW_QUEUES hardware queues and two live streams
            // that share a queue serialise each other.  Lane 0 gets exactly the streams its model needs; lanes 1.. (batch-chunk
            // pipelining, off by default) are created on demand by dfx_model_set_pipeline.
            bool good = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
            good = good && hipEventCreateWithFlags(&m->ev_pass, hipEventDisableTiming) == hipSuccess;
            good = good && hipEventCreateWithFlags(&m->ev_gate, hipEventDisableTiming) == hipSuccess;
            {
                const char *ea = getenv("DFX_ENQUEUE_AHEAD");
                m->enqueue_ahead = ea && ea[0] == '1';
            }
            good = good && dfx_create_lane(m, 0);
            for (int l = 1; l < m->max_chunks && good; ++l) good = dfx_create_lane(m, l);
            if (!good) {
                dfx_model_free(m);
                DFX_FAIL(DFX_ERR_HIP, "dfx_model_create: could not create the auxiliary streams/events");
            }
            m->have_streams = true;
        }
        // The persistent GRU phase synchronises through device flags and needs its streams to run concurrently: checked with a handshake
        // (hwq_probe_run) — not here but before the first pass that would use the persistent form, or when DFX_Q_HWQ_PROBE /
        // DFX_Q_GRU_PERSISTENT is asked for: a process that creates many handles it only streams through (df_create: one state per
        // stream, never a persistent phase) pays nothing for it.
        if (m->gru_seq && m->concurrent && !dfx_env_is_emulator()) {
            const char *pe = getenv("DFX_HWQ_PROBE");   // "0": skip the probe (trust the environment); "fail": dev / test hook
            if (pe && pe[0] == '0') {
            } else if (pe && pe[0] == 'f') {
                m->hwq_probe = 0;
                hwq_probe_fallback(m);
            } else {
                m->hwq_probe_pending = true;
            }
        }
    }
    *out = m;
    return DFX_OK;
}

static void pass_gate_forget(const dfx_model *m);
extern "C" void dfx_model_free(dfx_model *m) {
    if (!m) return;
    pass_gate_forget(m);   // (first: nobody may wait on ev_gate once it is destroyed; the lanes are the process's and stay)
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_pass) (void)hipEventDestroy(m->ev_pass);
    if (m->ev_gate) (void)hipEventDestroy(m->ev_gate);
    dfx_env_err_words_free(m->h_err);
    if (m->d_sync && m->d_trace) {   // dev aid (DFX_SEQ_TRACE=1): how many block hand-overs of the followers took the same-XCD form
        unsigned int n = 0;
        const size_t off = 16 + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 16 + 2 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 3 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX;
        if (hipMemcpy(&n, m->d_sync + off, sizeof(n), hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "[dfx] same-XCD (light) block hand-overs over the model's life: %u\n", n);
    }
    if (m->d_sync) (void)hipFree(m->d_sync);
    if (m->d_trace) (void)hipFree(m->d_trace);
    if (m->d_w) (void)hipFree(m->d_w);
    delete m;
}
extern "C" int dfx_model_set_streams(dfx_model *m, int enable) {
    if (!m) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    m->concurrent = enable != 0 && m->have_streams;
    return DFX_OK;
}
extern "C" int dfx_model_set_run_df(dfx_model *m, int enable) {
    if (!m) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    m->run_df = enable != 0;
    return DFX_OK;
}
extern "C" int dfx_model_set_pipeline(dfx_model *m, int time_chunks, int min_chunk_frames, int batch_chunks) {
    if (!m || time_chunks < 1 || min_chunk_frames < 1 || batch_chunks < 1) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_set_pipeline: bad arguments");
    m->tchunks = time_chunks < DFX_MAX_TCHUNKS ? time_chunks : DFX_MAX_TCHUNKS;
    m->tchunk_min = min_chunk_frames;
    m->max_chunks = batch_chunks < DFX_MAX_LANES ? batch_chunks : DFX_MAX_LANES;
    for (int l = 1; l < m->max_chunks && m->have_streams; ++l)
        if (!dfx_create_lane(m, l)) DFX_FAIL(DFX_ERR_HIP, "dfx_model_set_pipeline: could not create the streams of lane %d", l);
    return DFX_OK;
}
// Reads (and clears) the error words without waiting for anything: faults of work that has completed.  Each word is taken with one
// atomic exchange, so a word the device raises while the host is looking is either seen now or stays set for the next look — never
// lost.  (A fault of an unthrottled small pass can therefore still surface one call later than the pass that raised it: only
// dfx_model_check / DFX_CHECK_EVERY_PASS=1 wait for the device first.)
static int model_poll(const dfx_model *m) {
    unsigned int *h = m->h_err;
    if (!h) return DFX_OK;
    if (!(((volatile unsigned int *)h)[1] | ((volatile unsigned int *)h)[2])) return DFX_OK;
    const unsigned int e1 = __atomic_exchange_n(&h[1], 0u, __ATOMIC_ACQ_REL), e2 = __atomic_exchange_n(&h[2], 0u, __ATOMIC_ACQ_REL);
    if (!(e1 | e2)) return DFX_OK;
    if (e2)
        DFX_FAIL(DFX_ERR_HIP, "dfx: a flag wait of the persistent GRU phase timed out (bounded spin: the streams of the pass did not make progress "
                              "independently — hardware queues shared with other work, or the GPU shared with another process); the results of "
                              "the previous pass on this model are invalid.  DFX_GRU_SEQ=0 selects the event-synchronised form");
    DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx: an activation of magnitude >= 6e4 reached an fp16-split matrix kernel (GRU / DF-encoder / separable-conv "
                                  "path); the results of the previous pass on this model are invalid.  DFX_EXACT_FP32=1 selects the exact fp32 kernels");
}
extern "C" int dfx_model_poll(const dfx_model *m) {
    if (!m) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    return model_poll(m);
}
extern "C" int dfx_model_check(const dfx_model *m) {
    if (!m) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    DFX_HIP(hipDeviceSynchronize());   // every stream of the process: whatever this model has in flight is over
    return model_poll(m);
}
extern "C" int dfx_model_query(const dfx_model *m, int what, int64_t *value) {
    if (!m || !value) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_query: null");
    switch (what) {
        case DFX_Q_GRU_PERSISTENT:
        case DFX_Q_HWQ_PROBE: {
            std::lock_guard<std::mutex> lk(dfx_enqueue_mu());
            hwq_probe_run(const_cast<dfx_model *>(m));
            *value = what == DFX_Q_HWQ_PROBE ? m->hwq_probe : (m->gru_seq && m->concurrent ? 1 : 0);
            return DFX_OK;
        }
        case DFX_Q_EXACT_FP32: *value = m->exact_fp32 ? 1 : 0; return DFX_OK;
        case DFX_Q_PASSES_PERSISTENT: *value = m->passes_seq; return DFX_OK;
        case DFX_Q_PASSES_TICKET_BUSY: *value = m->passes_ev; return DFX_OK;
        case DFX_Q_SPIN_LIMIT: *value = m->spin_limit; return DFX_OK;
    }
    DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_query: unknown item %d", what);
}
// dev aid: chunk timestamps (100 MHz ticks) of the last persistent GRU launch, [layers][groups][chunks][3]; dims -> {layers, groups, chunks}
extern "C" int dfx_model_seq_trace(const dfx_model *m, unsigned long long *out_host, int64_t cap, int *dims) {
    if (!m || !m->d_trace || !out_host || !dims) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_seq_trace: tracing is off (DFX_SEQ_TRACE=1 at model creation)");
    const int64_t n = (int64_t)m->trace_dims[0] * m->trace_dims[1] * m->trace_dims[2] * 3;
    if (n > cap) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_seq_trace: buffer too small");
    DFX_HIP(hipMemcpy(out_host, m->d_trace, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; ++i) dims[i] = m->trace_dims[i];
    return DFX_OK;
}
extern "C" int dfx_model_cfg_get(const dfx_model *m, dfx_model_cfg *out) {
    if (!m || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "null");
    *out = m->cfg;
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ workspace plan
namespace {
struct Ws {
    // offsets in floats, each 64-float (256 B) aligned
    size_t e0, e1, e2, e3, c0, c1, emb_in, emb, xa, xb, gi, xa2, xb2, gi2, demb, d3, d2, d1, mask, c0p, xdf, coefs, lsnr, skp_e, skp_d, total;
    size_t pgi[DFX_MAX_GRU_LAYERS], py[DFX_MAX_GRU_LAYERS], ph[DFX_MAX_GRU_LAYERS];  // layer-pipelined GRU phase: gi, y, h state per layer
};
Ws plan_ws(const dfx_model_cfg &c, bool fuse_c0, int64_t R, int64_t B = 0) {
    Ws w{};
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) & ~(size_t)63;
        return o;
    };
    const size_t C = c.conv_ch, E = c.nb_erb, Fd = c.nb_df, emb = C * E / 4, NO = 2 * c.df_order;
    w.e0 = take(R * E * C);
    w.e1 = take(R * (E / 2) * C);
    w.e2 = take(R * (E / 4) * C);
    w.e3 = take(R * (E / 4) * C);
    w.c0 = fuse_c0 ? 0 : take(R * Fd * C);  // only materialised by the unfused DF-encoder path
    w.c1 = take(R * (Fd / 2) * C);
    w.emb_in = take(R * emb * (c.enc_concat ? 2 : 1));
    w.emb = take(R * emb);
    w.xa = take(R * 256);
    w.xb = take(R * 256);
    w.gi = take(R * 768);
    w.xa2 = take(R * 256);  // the DF decoder's GRU stack runs concurrently with the ERB decoder's
    w.xb2 = take(R * 256);
    w.gi2 = take(R * 768);
    w.demb = take(R * emb);
    w.d3 = take(R * (E / 4) * C);
    w.d2 = take(R * (E / 2) * C);
    w.d1 = take(R * E * C);
    w.mask = take(R * E);
    w.c0p = take(R * Fd * NO);
    w.xdf = take(R * 256);
    w.coefs = take(R * Fd * NO);
    w.lsnr = take(R);
    w.skp_e = take(c.emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR ? R * emb : 0);   // grouped-linear skips around the embedding GRUs
    w.skp_d = take(c.emb_gru_skip == DFX_SKIP_GROUPEDLINEAR ? R * emb : 0);
    const int nlayers = 1 + (c.emb_num_layers - 1) + c.df_num_layers;
    for (int l = 0; l < DFX_MAX_GRU_LAYERS; ++l) {
        const bool used = l < nlayers;
        w.pgi[l] = take(used ? R * 768 : 0);
        w.py[l] = take(used ? R * 256 : 0);
        w.ph[l] = take(used ? (B > 0 ? B : R) * 256 : 0);
    }
    w.total = off;
    return w;
}
}  // namespace

extern "C" int dfx_model_workspace_bytes(const dfx_model *m, int64_t B, int64_t T, int64_t *bytes) {
    if (!m || !bytes || B < 0 || T < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_workspace_bytes: bad arguments");
    *bytes = (int64_t)(plan_ws(m->cfg, m->fuse_c0 && !m->c0_batch_unfused, B * T, B).total * sizeof(float)) + 256;
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ launch helpers
static int nn_grid(int64_t tiles, int per_cu) {
    const int64_t cap = (int64_t)dfx_env_num_cus() * per_cu;
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

template <int C>
static int launch_pw(int mode, const dfx_model *m, const PwW &w, const float *x, const float *skip, float *out, int64_t R,
                     int Fin, int Fout, int stride, hipStream_t s, DfxRowMap rm = DfxRowMap{0, 0, 0}) {
    DfxPwArgs A;
    A.x = x;
    A.skip = skip;
    A.sk_a = skip ? m->p(w.sk_a) : nullptr;
    A.sk_b = skip ? m->p(w.sk_b) : nullptr;
    A.dw = m->p(w.dw);
    A.wt = m->p(w.wt);
    A.bias = m->p(w.bias);
    A.out = out;
    A.R = R;
    A.Fin = Fin;
    A.Fout = Fout;
    A.stride = stride;
    A.rm = rm;
    DfxKScope ks(DFX_K_PWCONV, s);
    // frame-staged form (coalesced loads / stores through wave-private LDS strips; same bits): whenever whole frames make whole tiles
    static const bool staged = [] { const char *e = getenv("DFX_PW_STAGED"); return !e || atoi(e) != 0; }();
    if (staged && dfx_pwf_ok(C, Fin, Fout)) {
        const size_t smem = dfx_pwf_smem(C, Fin, Fout);
        const int gridf = nn_grid(dfx_ceil_div(dfx_ceil_div(R, dfx_pwf_group(C, Fin, Fout)), 4), 2);
        const bool n4 = dfx_pwf_nvi(C, Fin, Fout) == 4;
        auto go = [&](auto kern) -> int {
            DFX_HIP(dfx_env_set_max_dyn_smem((const void *)kern, smem));
            dfx_launch(kern, dim3((unsigned)gridf), dim3(DFX_PW_THREADS), smem, s, A);
            return DFX_OK;
        };
        int rc;
        if constexpr (C % 32 == 0) {
            if (!m->exact_fp32 && w.wt_h3) {   // fp16-split pointwise contraction (default)
                A.wt_h3 = reinterpret_cast<const dfx_h8 *>(m->p(w.wt_h3));
                A.unscale = w.unscale;
                A.err = m->d_err;
                if (mode == DFX_PW_MODE_DW3) {
                    if (skip) rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, true, 4, true>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, true, DFX_PWF_MAXV, true>);
                    else rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, false, 4, true>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, false, DFX_PWF_MAXV, true>);
                } else {
                    if (skip) rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, true, 4, true>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, true, DFX_PWF_MAXV, true>);
                    else rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, false, 4, true>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, false, DFX_PWF_MAXV, true>);
                }
                if (rc) return rc;
                DFX_LAUNCH_CHECK();
                return DFX_OK;
            }
        }
        if (mode == DFX_PW_MODE_DW3) {
            if (skip) rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, true, 4>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, true, DFX_PWF_MAXV>);
            else rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, false, 4>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DW3, false, DFX_PWF_MAXV>);
        } else {
            if (skip) rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, true, 4>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, true, DFX_PWF_MAXV>);
            else rc = n4 ? go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, false, 4>) : go(dfx_k_pwconv_f<C, DFX_PW_MODE_DWT3, false, DFX_PWF_MAXV>);
        }
        if (rc) return rc;
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
    const int grid = nn_grid(dfx_ceil_div(R * Fout, 64), 8);
    if (mode == DFX_PW_MODE_DW3) {
        if (skip) dfx_launch(dfx_k_pwconv<C, DFX_PW_MODE_DW3, true>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
        else dfx_launch(dfx_k_pwconv<C, DFX_PW_MODE_DW3, false>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
    } else {
        if (skip) dfx_launch(dfx_k_pwconv<C, DFX_PW_MODE_DWT3, true>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
        else dfx_launch(dfx_k_pwconv<C, DFX_PW_MODE_DWT3, false>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

template <int C, int KT>
static int launch_convp2(const dfx_model *m, const float *c0, const float *feat_spec, float *out, int64_t B, int64_t T, int Fd,
                         int NO, hipStream_t s, int64_t t_begin = 0, int64_t t_zero = 0, int L = -1, int64_t t_end = -1) {
    if (t_end < 0) t_end = T;
    DfxCp2Args A;
    A.t_end = t_end;
    A.c0 = c0;
    A.feat = feat_spec;  // non-null: df_conv0 is recomputed on the fly, c0 is not read
    A.weff0 = m->p(m->cin_weff);
    A.bias0 = m->p(m->cin_b);
    A.L = L < 0 ? m->cfg.conv_lookahead : L;
    A.t_begin = t_begin;
    A.t_zero = t_zero;
    A.weff = m->p(m->cp_weff);
    A.bias = m->p(m->cp_b16);
    A.out = out;
    A.B = B;
    A.T = T;
    A.Fd = Fd;
    A.NO = NO;
    A.nfb = (Fd + 15) / 16;
    // enough independent wave-runs to fill the chip (each run re-reads KT-1 halo frames): target >= 8 waves per SIMD-slot
    const int64_t want = (int64_t)dfx_env_num_cus() * 4 * 8;
    int64_t nseg = dfx_ceil_div(want, B * A.nfb);
    const int64_t Tn = t_end - t_begin;  // frames produced
    const int64_t max_seg = dfx_ceil_div(Tn, (int64_t)8 * KT);
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    int64_t tseg = dfx_ceil_div(dfx_ceil_div(Tn, nseg), (int64_t)KT) * KT;
    A.tseg = (int)tseg;
    A.nseg = (int)dfx_ceil_div(Tn, tseg);
    const int64_t nruns = B * A.nfb * A.nseg;
    const int grid = nn_grid(dfx_ceil_div(nruns, 4), 8);
    DfxKScope ks(DFX_K_DF_CONVP, s);
    if (feat_spec) dfx_launch(dfx_k_df_convp2<C, KT, true>, dim3(grid), dim3(256), 0, s, A);
    else dfx_launch(dfx_k_df_convp2<C, KT, false>, dim3(grid), dim3(256), 0, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

template <int C, int KT>
static int launch_convp_h3(const dfx_model *m, const float *feat_spec, float *out, int64_t B, int64_t T, int Fd, int NO,
                           hipStream_t s, int64_t t_begin = 0, int64_t t_zero = 0, int L = -1, int64_t t_end = -1, int64_t feat_T = 0) {
    if constexpr (C % 32 != 0) {
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "fp16-split df_convp needs conv_ch %% 32 == 0");
    } else {
        if (t_end < 0) t_end = T;
        {   // 32-bit element offsets inside the kernel (B * feat_T * Fd < 2^29 per launch): a larger batch runs as several launches over whole clips
            const int64_t per_clip = (feat_T > 0 ? feat_T : T) * Fd;
            const int64_t lim = m->sw.convp_elems;   // (DFX_CONVP_ELEMS, test hook: the split at small sizes)
            if (per_clip >= lim) DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_convp: clip too long for 32-bit element offsets");
            const int64_t bmax = (lim - 1) / per_clip;
            if (B > bmax) {
                for (int64_t b0 = 0; b0 < B; b0 += bmax) {
                    const int64_t nb = B - b0 < bmax ? B - b0 : bmax;
                    if (int r = launch_convp_h3<C, KT>(m, feat_spec + b0 * per_clip * 2, out + b0 * (int64_t)(NO / 2) * T * Fd * 2, nb, T, Fd, NO, s, t_begin, t_zero, L,
                                                       t_end, feat_T))
                        return r;
                }
                return DFX_OK;
            }
        }
        DfxCphArgs A;
        A.t_end = t_end;
        A.feat = feat_spec;
        A.feat_T = feat_T;
        A.w0f = reinterpret_cast<const dfx_h8 *>(m->p(m->c0_h3));
        A.bias0 = m->p(m->cin_b);
        A.wf = reinterpret_cast<const dfx_h8 *>(m->p(m->cp_h3));
        A.bias = m->p(m->cp_b16);
        A.out = out;
        A.B = B;
        A.T = T;
        A.Fd = Fd;
        A.NO = NO;
        A.L = L < 0 ? m->cfg.conv_lookahead : L;
        A.t_begin = t_begin;
        A.t_zero = t_zero;
        A.unscale0 = m->c0_unscale;
        A.unscale = m->cp_unscale;
        A.err = m->d_err;
        A.nfb = (Fd + 15) / 16;
        const int64_t want = (int64_t)dfx_env_num_cus() * 4 * 4 * m->front_grain_p;  // two resident waves per SIMD, two rounds
        int64_t nseg = dfx_ceil_div(want, B * A.nfb);
        const int64_t Tn = t_end - t_begin;  // frames produced
        const int64_t max_seg = dfx_ceil_div(Tn, (int64_t)8 * KT);
        if (nseg > max_seg) nseg = max_seg;
        if (nseg < 1) nseg = 1;
        const int64_t tseg = dfx_ceil_div(dfx_ceil_div(Tn, nseg), (int64_t)KT) * KT;
        A.tseg = (int)tseg;
        A.nseg = (int)dfx_ceil_div(Tn, tseg);
        const int64_t nruns = B * A.nfb * A.nseg;
        // (capping the launch at 64 ... 192 resident workgroups, so that the rest of the chip is free for the front's critical path, measured
        // +0.1 ... +0.5 ms per step: profiles/r04_exact_and_convp_cap.log)
        DfxKScope ks(DFX_K_DF_CONVP, s);
        dfx_launch((dfx_k_df_convp_h3<C, KT>), dim3(nn_grid(dfx_ceil_div(nruns, 4), 2 * m->front_grain_p)), dim3(256), 0, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
}

// df_convp of the newest frame of every stream with the older frames' c0 tiles from the handle's ring (dfx_k_df_convp_step)
template <int C, int KT>
static int launch_convp_step(const dfx_model *m, const float *feat_spec, float *out, int64_t B, int64_t T, int Fd, int NO, hipStream_t s,
                             int64_t t_zero, int L, void *ring, int slot, bool rebuild, int64_t feat_T = 0, const unsigned char *par = nullptr,
                             const int *cnt = nullptr) {
    if constexpr (C % 32 != 0 || KT < 2) {
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_convp step kernel: conv_ch %% 32 == 0 and kt >= 2");
    } else {
        DfxCphArgs A;
        A.t_end = T;
        A.feat = feat_spec;
        A.feat_T = feat_T;
        A.w0f = reinterpret_cast<const dfx_h8 *>(m->p(m->c0_h3));
        A.bias0 = m->p(m->cin_b);
        A.wf = reinterpret_cast<const dfx_h8 *>(m->p(m->cp_h3));
        A.bias = m->p(m->cp_b16);
        A.out = out;
        A.B = B, A.T = T, A.Fd = Fd, A.NO = NO;
        A.L = L;
        A.t_begin = T - 1, A.t_zero = t_zero;
        A.unscale0 = m->c0_unscale, A.unscale = m->cp_unscale;
        A.err = m->d_err;
        A.nfb = (Fd + 15) / 16;
        A.nseg = 1, A.tseg = 1;
        const int grid = nn_grid(dfx_ceil_div(B * A.nfb, 4), 8);
        DfxKScope ks(DFX_K_DF_CONVP, s);
        if (rebuild) dfx_launch((dfx_k_df_convp_step<C, KT, true>), dim3(grid), dim3(256), 0, s, A, reinterpret_cast<f32x4 *>(ring), slot, par, cnt);
        else dfx_launch((dfx_k_df_convp_step<C, KT, false>), dim3(grid), dim3(256), 0, s, A, reinterpret_cast<f32x4 *>(ring), slot, par, cnt);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
}

template <int C>
static int launch_conv01_h3(const dfx_model *m, const PwW &w, const float *feat_spec, float *out, int64_t B, int64_t T, int Fin,
                            int Fout, int stride, hipStream_t s, int64_t t_begin = 0, int L = -1, int64_t t_end = -1, int64_t feat_T = 0) {
    if constexpr (C % 32 != 0) {
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "fp16-split df_conv1 needs conv_ch %% 32 == 0");
    } else {
        if (t_end < 0) t_end = T;
        if (B * T >= ((int64_t)1 << 31) || T * Fin >= ((int64_t)1 << 30)) DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_conv0->1: batch too large for one launch (32-bit frame index)");
        DfxC01hArgs A;
        A.t_end = t_end;
        A.feat = feat_spec;
        A.feat_T = feat_T;
        A.w0f = reinterpret_cast<const dfx_h8 *>(m->p(m->c0_h3));
        A.bias0 = m->p(m->cin_b);
        A.dw = m->p(w.dw);
        A.wpf = reinterpret_cast<const dfx_h8 *>(m->p(m->dfc1_h3));
        A.bias = m->p(w.bias);
        A.out = out;
        A.B = B;
        A.T = T;
        A.Fin = Fin;
        A.Fout = Fout;
        A.stride = stride;
        A.L = L < 0 ? m->cfg.conv_lookahead : L;
        A.t_begin = t_begin;
        A.unscale0 = m->c0_unscale;
        A.unscale = m->dfc1_unscale;
        A.err = m->d_err;
        const int grid = nn_grid(dfx_ceil_div(B * (t_end - t_begin) * Fout, 64), 3 * m->front_grain);   // three resident workgroups per CU
        DfxKScope ks(DFX_K_PWCONV, s);
        dfx_launch(dfx_k_df_conv01_h3<C>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
}

// erb_dec.convt1 -> conv0_out fused (dfx_k_erb_dec10); x = d2, writes the mask
template <int C>
static int launch_erb_dec10(const dfx_model *m, const float *d2, const float *e1, const float *e0, float *mask, int64_t R, int E,
                            hipStream_t s, DfxRowMap rm) {
    DfxDec10Args A;
    A.x = d2;
    A.skip1 = e1;
    A.sk1_a = m->p(m->ct1.sk_a);
    A.sk1_b = m->p(m->ct1.sk_b);
    A.dw = m->p(m->ct1.dw);
    A.wt = m->p(m->ct1.wt);
    A.bias = m->p(m->ct1.bias);
    A.skip0 = e0;
    A.sk0_a = m->p(m->co_ska);
    A.sk0_b = m->p(m->co_skb);
    A.wo = m->p(m->co_w);
    A.bias_o = m->co_bias;
    A.out = mask;
    A.R = R;
    A.E = E;
    A.rm = rm;
    DfxKScope ks(DFX_K_ERB_DEC, s);
    static const bool staged = [] { const char *e = getenv("DFX_PW_STAGED"); return !e || atoi(e) != 0; }();
    if (staged && dfx_dec10f_ok(C, E)) {   // whole frames streamed through LDS strips (dfx_k_erb_dec10_f)
        DfxDec10fArgs AA;
        AA.a = A;
        const size_t smemf = DFX_DEC10F_SMEM(C, E);
        const dim3 grid((unsigned)nn_grid(dfx_ceil_div(R, 4), 2));
        if constexpr (C % 32 == 0) {
            if (!m->exact_fp32 && m->ct1.wt_h3) {
                AA.wt_h3 = reinterpret_cast<const dfx_h8 *>(m->p(m->ct1.wt_h3));
                AA.unscale = m->ct1.unscale;
                AA.err = m->d_err;
                DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_erb_dec10_f<C, true>, smemf));
                dfx_launch(dfx_k_erb_dec10_f<C, true>, grid, dim3(256), smemf, s, AA);
                DFX_LAUNCH_CHECK();
                return DFX_OK;
            }
        }
        DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_erb_dec10_f<C, false>, smemf));
        dfx_launch(dfx_k_erb_dec10_f<C, false>, grid, dim3(256), smemf, s, AA);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
    const size_t smem = DFX_DEC10_SMEM(C, E);
    DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_erb_dec10<C>, smem));
    dfx_launch(dfx_k_erb_dec10<C>, dim3((unsigned)nn_grid(dfx_ceil_div(R, 4), 2)), dim3(256), smem, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// erb_dec.convt3 -> convt2 -> convt1 -> conv0_out in one kernel (dfx_k_erb_tail): d3 / d2 / d1 never reach HBM
template <int C>
static bool erb_tail_ok(const dfx_model *m, int E) {
    if constexpr (C % 32 != 0) return false;
    return m->fuse_tail && m->fuse_erb && !m->exact_fp32 && m->ct3.wt_h3 && m->ct2.wt_h3 && m->ct1.wt_h3 && m->tail_w0h3 && m->tail_woh3 && dfx_tail_ok(C, E);
}
template <int C>
// e0 == null: recomputed in the kernel from feat_erb (rows of T frames per clip, feat_T frames per clip in feat_erb, lookahead L)
static int launch_erb_tail(const dfx_model *m, const float *demb, const float *e3, const float *e2, const float *e1, const float *e0,
                           float *mask, int64_t R, int E, hipStream_t s, DfxRowMap rm, const float *feat_erb = nullptr, int64_t T = 0,
                           int64_t feat_T = 0, int L = 0) {
    if constexpr (C % 32 != 0) {
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "erb tail: conv_ch");
    } else {
        DfxTailArgs A;
        A.demb = demb, A.e3 = e3, A.e2 = e2, A.e1 = e1, A.e0 = e0;
        const PwW *L3[3] = {&m->ct3, &m->ct2, &m->ct1};
        for (int l = 0; l < 3; ++l) {
            A.dw[l] = m->p(L3[l]->dw);
            A.bias[l] = m->p(L3[l]->bias);
            A.wh3[l] = reinterpret_cast<const dfx_h8 *>(m->p(L3[l]->wt_h3));
            A.unscale[l] = L3[l]->unscale;
            A.ska[l] = m->p(L3[l]->sk_a);
            A.skb[l] = m->p(L3[l]->sk_b);
        }
        A.ska[3] = m->p(m->co_ska);
        A.skb[3] = m->p(m->co_skb);
        A.wo = m->p(m->co_w);
        A.woh3 = reinterpret_cast<const dfx_h8 *>(m->p(m->tail_woh3)), A.unscale_wo = m->tail_wo_unscale;
        A.w0h3 = reinterpret_cast<const dfx_h8 *>(m->p(m->tail_w0h3)), A.unscale_w0 = m->tail_w0_unscale;
        A.bias_o = m->co_bias;
        A.out = mask;
        A.R = R;
        A.E = E;
        A.rm = rm;
        A.err = m->d_err;
        if (!e0) {
            if (!feat_erb || T <= 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "erb tail: e0 or the features it is recomputed from");
            A.feat = feat_erb, A.w0 = m->p(m->erb0_w), A.b0 = m->p(m->erb0_b), A.T = T, A.feat_T = feat_T, A.L = L;
        }
        const size_t smem = DFX_TAIL_SMEM(C);
        DFX_HIP(dfx_env_set_max_dyn_smem(e0 ? (const void *)dfx_k_erb_tail<C, false> : (const void *)dfx_k_erb_tail<C, true>, smem));
        DfxKScope ks(DFX_K_ERB_TAIL, s);
        const dim3 grid((unsigned)nn_grid(dfx_ceil_div(R, DFX_TAIL_WAVES), 1));   // (capped at 64 ... 128 workgroups: +0.1 ... +0.5 ms per step)
        if (e0) dfx_launch((dfx_k_erb_tail<C, false>), grid, dim3(64 * DFX_TAIL_WAVES), smem, s, A);
        else dfx_launch((dfx_k_erb_tail<C, true>), grid, dim3(64 * DFX_TAIL_WAVES), smem, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
}

template <int C>
static int launch_erb_enc(const dfx_model *m, const float *feat_erb, float *e0, float *e1, int64_t B, int64_t T, hipStream_t s,
                          int64_t t_begin = 0, int L = -1, int64_t t_end = -1, int64_t feat_T = 0) {
    const dfx_model_cfg &c = m->cfg;
    if (t_end < 0) t_end = T;
    DfxEncArgs A;
    A.t_end = t_end;
    A.feat = feat_erb;
    A.feat_T = feat_T;
    A.w0 = m->p(m->erb0_w);
    A.b0 = m->p(m->erb0_b);
    A.dw = m->p(m->erb1.dw);
    A.wt = m->p(m->erb1.wt);
    A.bias = m->p(m->erb1.bias);
    A.e0 = e0;
    A.e1 = e1;
    A.B = B;
    A.T = T;
    A.E = c.nb_erb;
    A.L = L < 0 ? c.conv_lookahead : L;
    A.t_begin = t_begin;
    const size_t smem = DFX_ENC_SMEM(C, c.nb_erb);
    DfxKScope ks(DFX_K_ERB_ENC, s);
    const dim3 grid((unsigned)nn_grid(dfx_ceil_div(B * (t_end - t_begin), 4), 2));
    if constexpr (C % 32 == 0) {
        if (!m->exact_fp32 && m->erb1.wt_h3) {   // erb_conv1's pointwise contraction on the fp16-split path
            A.wt_h3 = reinterpret_cast<const dfx_h8 *>(m->p(m->erb1.wt_h3));
            A.unscale = m->erb1.unscale;
            A.err = m->d_err;
            DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_erb_enc<C, true>, smem));
            dfx_launch(dfx_k_erb_enc<C, true>, grid, dim3(256), smem, s, A);
            DFX_LAUNCH_CHECK();
            return DFX_OK;
        }
    }
    DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_erb_enc<C, false>, smem));
    dfx_launch(dfx_k_erb_enc<C, false>, grid, dim3(256), smem, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// enc.df_conv0 -> enc.df_conv1 without the c0 round trip (dfx_k_df_conv01)
template <int C>
static int launch_conv01(const dfx_model *m, const PwW &w, const float *feat_spec, float *out, int64_t B, int64_t T, int Fin,
                         int Fout, int stride, hipStream_t s, int64_t t_begin = 0, int L = -1, int64_t t_end = -1) {
    if (t_end < 0) t_end = T;
    DfxC01Args A;
    A.t_end = t_end;
    A.feat = feat_spec;
    A.weff0 = m->p(m->cin_weff);
    A.bias0 = m->p(m->cin_b);
    A.dw = m->p(w.dw);
    A.wt = m->p(w.wt);
    A.bias = m->p(w.bias);
    A.out = out;
    A.B = B;
    A.T = T;
    A.Fin = Fin;
    A.Fout = Fout;
    A.stride = stride;
    A.L = L < 0 ? m->cfg.conv_lookahead : L;
    A.t_begin = t_begin;
    const int grid = nn_grid(dfx_ceil_div(B * (t_end - t_begin) * Fout, 64), 8);
    DfxKScope ks(DFX_K_PWCONV, s);
    dfx_launch(dfx_k_df_conv01<C>, dim3(grid), dim3(DFX_PW_THREADS), 0, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

static int launch_ggemm(const float *a, int lda, const float *w, int G, int Kg, int Ng, const float *bias, int act,
                        const float *res, float *out, int ldo, int64_t M, hipStream_t s, int perm_inner = 0, int perm_F = 0,
                        int64_t perm_T = 1, DfxRowMap rm = DfxRowMap{0, 0, 0}, const float *a2 = nullptr) {
    if (M <= 0) return DFX_OK;
    if (Kg % 4 || Ng % 4 || lda % 4) DFX_FAIL(DFX_ERR_UNSUPPORTED, "grouped GEMM needs K, N, lda multiples of 4 (got %d, %d, %d)", Kg, Ng, lda);
    DfxGgArgs A;
    A.a = a;
    A.a2 = a2;
    A.w = w;
    A.bias = bias;
    A.res = res;
    A.out = out;
    A.M = M;
    A.lda = lda;
    A.ldo = ldo;
    A.G = G;
    A.Kg = Kg;
    A.Ng = Ng;
    A.act = act;
    A.perm_inner = perm_inner;
    A.perm_F = perm_F;
    A.perm_T = perm_T;
    A.rm = rm;
    const int BN = Ng <= 16 ? 16 : (Ng <= 32 ? 32 : 64);
    A.ntn = (Ng + BN - 1) / BN;
    const int64_t nblk = dfx_ceil_div(dfx_ceil_div(M, DFX_GG_BM), 8) * 8 * (int64_t)(G * A.ntn);
    if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "grouped GEMM grid too large");
    const dim3 grid((unsigned)nblk);
    DfxKScope ks(DFX_K_GGEMM, s);
    if (BN == 16) dfx_launch(dfx_k_ggemm<16>, grid, dim3(DFX_GG_THREADS), 0, s, A);
    else if (BN == 32) dfx_launch(dfx_k_ggemm<32>, grid, dim3(DFX_GG_THREADS), 0, s, A);
    else dfx_launch(dfx_k_ggemm<64>, grid, dim3(DFX_GG_THREADS), 0, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}
// GRU input projection [M,256] x [256,N] + bias on the weight-stationary kernel (N % 128 == 0), else the generic GEMM
static int launch_proj(const float *a, const float *w, const float *bias, float *out, int64_t M, int N, hipStream_t s,
                       DfxRowMap rm = DfxRowMap{0, 0, 0}) {
    if (M <= 0) return DFX_OK;
    if (N % DFX_PJ_BN) return launch_ggemm(a, 256, w, 1, 256, N, bias, DFX_ACT_NONE, nullptr, out, N, M, s, 0, 0, 1, rm);
    DfxPjArgs A;
    A.rm = rm;
    A.a = a;
    A.w = w;
    A.bias = bias;
    A.out = out;
    A.M = M;
    A.N = N;
    A.ncol = N / DFX_PJ_BN;
    const int64_t max_groups = dfx_ceil_div(dfx_ceil_div(M, 16), DFX_PJ_THREADS / 64);
    // one workgroup per CU (147 KB of LDS each) and, because block b runs on XCD b % 8, the same number of workgroups on
    // every XCD: 8 * floor(CUs_per_XCD / ncol) row groups (an XCD with one workgroup too many needs a second round)
    int64_t rg = (int64_t)16 * ((dfx_env_num_cus() / 8) / A.ncol);  // two balanced rounds (measured 8 % faster than one)
    if (rg < 8) rg = 8;
    if (rg > max_groups) rg = max_groups;
    A.rgroups = (int)rg;
    const int64_t nblk = dfx_ceil_div(rg, 8) * 8 * A.ncol;
    DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_proj256<0>, DFX_PJ_SMEM));
    DfxKScope ks(DFX_K_PROJ, s);
    dfx_launch(dfx_k_proj256<0>, dim3((unsigned)nblk), dim3(DFX_PJ_THREADS), DFX_PJ_SMEM, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// GRU input projection on the fp16-split matrix path (K = 256, N % 64 == 0)
static int launch_flag_set(unsigned int *flag, unsigned int value, hipStream_t s);
static int launch_proj_h3(const dfx_model *m, const GruW &g, const float *a, float *out, int64_t M, int N, hipStream_t s,
                          DfxRowMap rm = DfxRowMap{0, 0, 0}, const DfxPublish *pub = nullptr) {
    if (M <= 0) return pub ? launch_flag_set(pub->flag, pub->value, s) : DFX_OK;
    DfxPhArgs A;
    A.a = a;
    A.wf = reinterpret_cast<const dfx_h8 *>(m->p(g.wih_h3));
    A.bias = m->p(g.bias_i);
    A.out = out;
    A.M = M;
    A.N = N;
    A.unscale = g.wih_unscale;
    A.rm = rm;
    const int64_t nblk = dfx_ceil_div(M, DFX_PH_BM);
    if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "projection grid too large");
    DfxKScope ks(DFX_K_PROJ, s);
    // two row tiles per wave (256-row workgroups: half the fragment reads per row, 0.36 vs 0.40 ms for 256 k rows) unless the launch is a
    // single round of workgroups anyway — then the one-tile kernel's shorter workgroup latency wins (49 vs 79 us: the frame-by-frame
    // streaming runtime, 4096 rows per call).  DFX_PROJ_RT=1 / 2 / 3 forces one form.
    const int row_tiles = m->proj_rt;
    if (row_tiles == 3) {   // two workgroups of 4 waves per CU on 32-column chunks (measured 0.375 vs 0.363 ms: not the default)
        DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_proj256_h3x2<4, 2>, DFX_PH_SMEM / 2));
        if (pub) A.pub = *pub, A.pub.nblocks = (unsigned)dfx_ceil_div(M, 128);
        dfx_launch((dfx_k_proj256_h3x2<4, 2>), dim3((unsigned)dfx_ceil_div(M, 128)), dim3(256), DFX_PH_SMEM / 2, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
    if (row_tiles == 2 || (row_tiles == 0 && M > 8192)) {
        DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_proj256_h3x2<8, 4>, DFX_PH_SMEM));
        if (pub) A.pub = *pub, A.pub.nblocks = (unsigned)dfx_ceil_div(M, 256);
        dfx_launch((dfx_k_proj256_h3x2<8, 4>), dim3((unsigned)dfx_ceil_div(M, 256)), dim3(512), DFX_PH_SMEM, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
    // few row blocks (a streaming hop): the 64-column chunks of W are dealt to `parts` workgroups per row block — one workgroup per CU
    // (128 KB of LDS each) — so that the launch covers the chip instead of nblk CUs streaming all of W each
    {
        const int nch = N / DFX_PH_NC;
        int parts = 1;
        for (int d = 1; d <= nch; ++d)
            if (nch % d == 0 && nblk * d <= dfx_env_num_cus()) parts = d;
        A.parts = parts;
    }
    DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_proj256_h3, DFX_PH_SMEM));
    if (pub) A.pub = *pub, A.pub.nblocks = (unsigned)(nblk * A.parts);
    dfx_launch(dfx_k_proj256_h3, dim3((unsigned)(nblk * A.parts)), dim3(DFX_PH_THREADS), DFX_PH_SMEM, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

static int launch_glin(const dfx_model *m, const GlinW &g, const float *a, int act, const float *res, float *out, int64_t M,
                       hipStream_t s, DfxRowMap rm = DfxRowMap{0, 0, 0}, const float *a2 = nullptr) {
    return launch_ggemm(a, g.G * g.Kg, m->p(g.w), g.G, g.Kg, g.Ng, nullptr, act, res, out, g.G * g.Ng, M, s, 0, 0, 1, rm, a2);
}

// DFX_SEQ_FOLLOW (persistent GRU phase): 0 = every input projection a launch per time chunk; 1 = follower workgroups for the stacks' second layers;
// 2 (default since the same-XCD hand-over, M§R5.12) = followers for every decoder layer + the emb fan-out; 3 = the first layers + emb only.
static int seq_follow_mode(const dfx_model *m) { return m->sw.follow; }
// Row count up to which the fan-out kernels take their few-rows forms (one row tile per wave, a tile's chunks dealt to separate waves): made
// for a streaming hop (4096 rows).  Round 5: the time chunks of the persistent GRU phase (10-20 k rows at 16-24 chunks) take the large-launch
// forms — at the old bound of 16384 rows every chunking finer than 15 chunks fell onto the hop's forms (15.1 vs 14.1 ms per step).
static int64_t fan_few_rows(const dfx_model *m) { return m->sw.fan_few_rows; }
// df_fc_emb (+ e3) and the encoder GRU's linear_in in one pass over c1 (dfx_k_enc_fan)
static int launch_enc_fan(const dfx_model *m, const float *c1, const float *e3, float *emb_out, float *xa, int64_t M, hipStream_t s, DfxRowMap rm) {
    DfxEncFanArgs A;
    A.c1 = c1;
    A.w1 = reinterpret_cast<const float4 *>(m->p(m->efan_w1));
    A.w2 = reinterpret_cast<const float4 *>(m->p(m->efan_w2));
    A.e3 = e3;
    A.emb_out = emb_out;
    A.out = xa;
    A.R = M;
    A.ng = m->efan_groups;
    A.rm = rm;
    DfxKScope ks(DFX_K_GGEMM, s);
    if (M > fan_few_rows(m)) {
        constexpr int RT = 2;
        A.parts = 1;
        dfx_launch(dfx_k_enc_fan<RT>, dim3((unsigned)nn_grid(dfx_ceil_div(dfx_ceil_div(M, 16 * RT), 4), 8)), dim3(256), 0, s, A);
    } else {   // few rows (a streaming hop: 4096): one wave per (16 rows, pair of groups)
        A.parts = A.ng / 2;
        dfx_launch(dfx_k_enc_fan<1>, dim3((unsigned)nn_grid(dfx_ceil_div(dfx_ceil_div(M, 16) * A.parts, 4), 8)), dim3(256), 0, s, A);
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}
// The DF branch of the encoder in one kernel: feat_spec -> (c0 -> c1 -> df_fc_emb + e3 -> linear_in) -> xa (dfx_k_df_enc_h3); frames [t_begin, t_end)
template <int C>
static int launch_df_enc(const dfx_model *m, const float *feat_spec, const float *e3, float *emb_in, float *xa, int64_t B, int64_t T, int Fin,
                         hipStream_t s, int64_t t_begin, int L, int64_t t_end, int64_t feat_T) {
    if constexpr (C % 32 != 0) {
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "fused DF encoder needs conv_ch %% 32 == 0");
    } else {
        if (B * T * (int64_t)m->fc_emb.G * 16 >= ((int64_t)1 << 31) || B * (feat_T > 0 ? feat_T : T) * Fin >= ((int64_t)1 << 29))
            DFX_FAIL(DFX_ERR_UNSUPPORTED, "fused DF encoder: batch too large for one launch (32-bit element offsets)");
        DfxDfEncArgs A;
        A.feat = feat_spec;
        A.w0f = reinterpret_cast<const dfx_h8 *>(m->p(m->c0_h3));
        A.bias0 = m->p(m->cin_b);
        A.dw = m->p(m->dfc1.dw);
        A.wpf = reinterpret_cast<const dfx_h8 *>(m->p(m->dfc1_h3));
        A.bias = m->p(m->dfc1.bias);
        A.wfc = reinterpret_cast<const dfx_h8 *>(m->p(m->dfenc_fc));
        A.win = reinterpret_cast<const dfx_h8 *>(m->p(m->dfenc_in));
        A.e3 = e3, A.emb_in = emb_in, A.xa = xa;
        A.B = B, A.T = T;
        A.Fin = Fin, A.Fout = Fin / 2, A.stride = 2, A.L = L;
        A.cpg = m->fc_emb.Kg / 32;
        A.unscale0 = m->c0_unscale, A.unscale = m->dfc1_unscale, A.unscale_fc = m->dfenc_fc_unscale, A.unscale_in = m->dfenc_in_unscale;
        A.t_begin = t_begin, A.t_end = t_end;
        A.err = m->d_err;
        A.feat_T = feat_T;
        int64_t tiles = dfx_ceil_div(B * (t_end - t_begin), 16);
        // few frames (a streaming hop): deal the bins of a tile to several waves — parts of whole linear_in groups = (2 cpg / KC) bins each
        const int KC = C / 32, unit = 2 * A.cpg / KC > 0 && (2 * A.cpg) % KC == 0 ? 2 * A.cpg / KC : A.Fout;
        while (tiles * A.nsplit < (int64_t)dfx_env_num_cus() * 4 * 3 && A.Fout % (2 * A.nsplit) == 0 && (A.Fout / (2 * A.nsplit)) % unit == 0) A.nsplit *= 2;
        tiles *= A.nsplit;
        DfxKScope ks(DFX_K_PWCONV, s);
        dfx_launch(dfx_k_df_enc_h3<C>, dim3((unsigned)nn_grid(dfx_ceil_div(tiles, 4), 3)), dim3(DFX_PW_THREADS), 0, s, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    }
}
static DfxFanArgs emb_fan_args(const dfx_model *m, const float *y, const float *res, float *emb_out, float *dec_x, float *dfg_x, float *skp, float *lsnr) {
    const dfx_model_cfg &c = m->cfg;
    DfxFanArgs A;
    A.y = y;
    A.wfrag = reinterpret_cast<const float4 *>(m->p(m->fan_w));
    A.res = res;
    A.emb_out = emb_out;
    A.out[0] = dec_x, A.out[1] = dfg_x, A.out[2] = skp;
    A.act[0] = DFX_ACT_RELU, A.act[1] = DFX_ACT_RELU, A.act[2] = DFX_ACT_NONE;
    A.lsnr_w = lsnr ? m->p(m->lsnr_w) : nullptr;
    A.lsnr_b = m->lsnr_b, A.lsnr_scale = (float)(c.lsnr_max - c.lsnr_min), A.lsnr_off = (float)c.lsnr_min;
    A.lsnr = lsnr;
    A.R = 0;
    A.nj = m->fan_chunks;
    A.rm = DfxRowMap{0, 0, 0};
    A.parts = 1;
    return A;
}
// emb and everything that reads it, in one pass over the encoder GRU's output (dfx_k_emb_fan); outs[c] null = consumer not wanted
static int launch_emb_fan(const dfx_model *m, const float *y, const float *res, float *emb_out, float *dec_x, float *dfg_x, float *skp,
                          float *lsnr, int64_t M, hipStream_t s, DfxRowMap rm, float *embv_for_split = nullptr, const DfxPublish *pub = nullptr) {
    const dfx_model_cfg &c = m->cfg;
    DfxFanArgs A = emb_fan_args(m, y, res, emb_out, dec_x, dfg_x, skp, lsnr);
    A.R = M;
    A.rm = rm;
    // few rows (a streaming hop): one wave per (16 rows, super-chunk) instead of a wave walking all super-chunks — emb is then written out
    // (embv: 2 KB per row of a few thousand rows) and lsnr, the one consumer that needs all of a row's features, is a launch of its own
    const bool split = M <= fan_few_rows(m) && lsnr && embv_for_split;
    if (split) {
        A.parts = A.nj;
        A.emb_out = embv_for_split;
        A.lsnr = nullptr, A.lsnr_w = nullptr;
    }
    {
    DfxKScope ks(DFX_K_EMB_FAN, s);
    // (the kinds are what pack_fan accepted: dec_in narrow, dfg_in wide, df_skip narrow; a consumer that is not wanted drops out)
    if (M > fan_few_rows(m)) {
        constexpr int RT = 2;
        const dim3 grid((unsigned)nn_grid(dfx_ceil_div(dfx_ceil_div(M, 16 * RT), 4), 8));
        if (pub) A.pub = *pub, A.pub.nblocks = grid.x;
        if (dfg_x && skp) dfx_launch((dfx_k_emb_fan<RT, 1, 2, 1>), grid, dim3(256), 0, s, A);
        else if (dfg_x) dfx_launch((dfx_k_emb_fan<RT, 1, 2, 0>), grid, dim3(256), 0, s, A);
        else dfx_launch((dfx_k_emb_fan<RT, 1, 0, 0>), grid, dim3(256), 0, s, A);
    } else {   // few rows (a streaming hop): one row tile per wave — twice the waves, half the serial matrix work per wave
        constexpr int RT = 1;
        const dim3 grid((unsigned)nn_grid(dfx_ceil_div(dfx_ceil_div(M, 16 * RT) * A.parts, 4), 8));
        if (pub && !split) A.pub = *pub, A.pub.nblocks = grid.x;
        if (dfg_x && skp) dfx_launch((dfx_k_emb_fan<RT, 1, 2, 1>), grid, dim3(256), 0, s, A);
        else if (dfg_x) dfx_launch((dfx_k_emb_fan<RT, 1, 2, 0>), grid, dim3(256), 0, s, A);
        else dfx_launch((dfx_k_emb_fan<RT, 1, 0, 0>), grid, dim3(256), 0, s, A);
    }
    DFX_LAUNCH_CHECK();
    }
    if (split) {   // (needs identity rows: the streaming window's new frame is reached through rm — one wave per logical row)
        DfxKScope ks(DFX_K_LSNR, s);
        dfx_launch(dfx_k_lsnr_rows, dim3((unsigned)dfx_ceil_div(M * 64, 256)), dim3(256), 0, s, (const float *)embv_for_split, m->p(m->lsnr_w), m->lsnr_b,
                   (float)(c.lsnr_max - c.lsnr_min), (float)c.lsnr_min, lsnr, M, 64 * A.nj, rm);
        DFX_LAUNCH_CHECK();
        if (pub) return launch_flag_set(pub->flag, pub->value, s);   // (two launches: the flag follows the second)
    }
    return DFX_OK;
}


static int launch_gru_h3(const dfx_model *m, const GruW &g, const float *gi, float *y, const float *h_in, float *h_out,
                         int64_t B, int64_t T, int64_t t0, int64_t t1, hipStream_t s, int layer = -1) {
    DfxGhArgs A;
    A.gi = gi;
    A.whf = reinterpret_cast<const dfx_h8 *>(m->p(g.whh_h3));
    A.bhn = m->p(g.bhn);
    A.h_in = h_in;
    A.h_out = h_out;
    A.y = y;
    A.B = B;
    A.T = T;
    A.t0 = t0;
    A.t1 = t1;
    A.unscale = g.whh_unscale;
    const bool x32 = m->exact_fp32;   // exact fp32 matrix ops over fp32 fragments (dfx_k_gru_rec_x32)
    if (x32) A.whf = reinterpret_cast<const dfx_h8 *>(m->p(g.whh_x32)), A.unscale = 1.f;
    // A layer's workgroups are confined to 4 XCDs.  Measured at batch 256 inside the event-based pipeline: -0.7 ms per step against a plain grid
    // (the W_hh lines a layer streams every step are shared by more workgroups per L2); 1 and 2 XCDs +1.3 ms (L2 bandwidth).
    constexpr int xw = 4;
    const int64_t groups = dfx_ceil_div(B, DFX_GH_ROWS);
    A.xcd_mask = 0;
    if (layer >= 0 && groups <= 32 * xw) A.xcd_mask = (((1 << xw) - 1) << ((layer * xw) % 8)) & 0xff;
    const int64_t nblk = A.xcd_mask ? dfx_ceil_div(groups, xw) * 8 : groups;
    DFX_HIP(dfx_env_set_max_dyn_smem(x32 ? (const void *)dfx_k_gru_rec_x32 : (const void *)dfx_k_gru_rec_h3, DFX_GH_SMEM));
    DfxKScope ks(DFX_K_GRU_REC, s);
    if (x32) dfx_launch(dfx_k_gru_rec_x32, dim3((unsigned)nblk), dim3(DFX_GH_THREADS), DFX_GH_SMEM, s, A);
    else dfx_launch(dfx_k_gru_rec_h3, dim3((unsigned)nblk), dim3(DFX_GH_THREADS), DFX_GH_SMEM, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

static int launch_flag_set(unsigned int *flag, unsigned int value, hipStream_t s) {
    dfx_launch(dfx_k_flag_set, dim3(1), dim3(64), 0, s, flag, value);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}
static int launch_wait_ge(const dfx_model *m, const unsigned int *flags, int n, unsigned int target, hipStream_t s) {
    dfx_launch(dfx_k_wait_ge, dim3(1), dim3(64), 0, s, flags, n, target, m->d_err, m->spin_limit);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// SqueezedGRU_S without its linear_in/linear_out (modules.py:702-738): layers of (input projection GEMM, recurrence).
// x: [R,256] input; result pointer returned through *y (ping-pong between xa/xb).
// hstate != null (streaming): layer l continues from / leaves its state in hstate + l*B*256 and only the frames [t0, T) are run
static int run_gru_stack(const dfx_model *m, const std::vector<GruW> &layers, const float *x, float *bufa, float *bufb,
                         float *gi, int64_t B, int64_t T, const float **y, hipStream_t s, float *hstate = nullptr, int64_t t0 = 0,
                         DfxRowMap rm = DfxRowMap{0, 0, 0}, float *hnext = nullptr, bool twin = false) {
    const int64_t R = B * (T - t0);
    if (hstate && m->exact_fp32) DFX_FAIL(DFX_ERR_UNSUPPORTED, "streaming needs the fp16-split GRU kernels (unset DFX_EXACT_FP32)");
    const float *in = x;
    float *outb = (x == bufa) ? bufb : bufa;
    for (size_t l = 0; l < layers.size(); ++l) {
        const GruW &g = layers[l];
        if (hstate && hnext && T - t0 == 1 && !m->exact_fp32) {   // one time step of many streams: projection + recurrence + gates in one launch
            DfxGstArgs A;
            A.x = in, A.xrm = rm;
            A.h_in = hstate + l * B * 256, A.h_out = hnext + l * B * 256;
            A.y = outb, A.yrm = rm;
            A.wif = reinterpret_cast<const dfx_h8 *>(m->p(g.wih_h3));
            A.whf = reinterpret_cast<const dfx_h8 *>(m->p(g.whh_pj));
            A.bias_i = m->p(g.bias_i), A.bhn = m->p(g.bhn);
            A.unscale_i = g.wih_unscale, A.unscale_h = g.whh_unscale;
            A.B = B;
            // 32 hidden units per workgroup (twice the workgroups, half the chunk) unless 64-unit workgroups already fill the chip — by
            // themselves, or together with the other decoder's stack that runs at the same time (twin: at 4096 streams 0.456 vs 0.470 ms per hop)
            static const int ct_env = [] { const char *e = getenv("DFX_GRU_STEP_CT"); return e ? atoi(e) : 0; }();
            const bool wide = ct_env == 4 || (ct_env != 2 && dfx_ceil_div(B, DFX_PH_BM) * 4 * (twin ? 2 : 1) >= dfx_env_num_cus());
            DfxKScope ks(DFX_K_GRU_REC, s);
            const unsigned rb8 = (unsigned)(dfx_ceil_div(dfx_ceil_div(B, DFX_PH_BM), 8) * 8);   // row blocks, padded: the kernel deals them to the XCDs
            if (wide) {
                DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_gru_step_h3<4>, DFX_PH_SMEM));
                dfx_launch(dfx_k_gru_step_h3<4>, dim3(rb8 * 4), dim3(DFX_PH_THREADS), DFX_PH_SMEM, s, A);
            } else {
                DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_gru_step_h3<2>, DFX_PH_SMEM / 2));
                dfx_launch(dfx_k_gru_step_h3<2>, dim3(rb8 * 8), dim3(DFX_PH_THREADS), DFX_PH_SMEM / 2, s, A);
            }
            DFX_LAUNCH_CHECK();
            in = outb;
            outb = (outb == bufa) ? bufb : bufa;
            continue;
        }
        if (m->exact_fp32) {
            if (int rc = launch_proj(in, m->p(g.wih_t), m->p(g.bias_i), gi, R, 768, s)) return rc;
        } else {
            if (int rc = launch_proj_h3(m, g, in, gi, R, 768, s, rm)) return rc;
        }
        if (m->exact_fp32) {
            DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_gru_rec, DFX_GRU_SMEM));
            DfxKScope ks(DFX_K_GRU_REC, s);
            dfx_launch(dfx_k_gru_rec, dim3((unsigned)dfx_ceil_div(B, DFX_GRU_ROWS)), dim3(DFX_GRU_THREADS), DFX_GRU_SMEM, s,
                       (const float *)gi, reinterpret_cast<const float4 *>(m->p(g.whh4)), m->p(g.bhn), (const float *)nullptr,
                       (float *)nullptr, outb, B, T);
            DFX_LAUNCH_CHECK();
        } else {
            float *hl = hstate ? hstate + l * B * 256 : nullptr;
            if (int rc = launch_gru_h3(m, g, gi, outb, hl, hl, B, T, t0, T, s)) return rc;
        }
        in = outb;
        outb = (outb == bufa) ? bufb : bufa;
    }
    *y = in;
    return DFX_OK;
}

static int stream_copy_rows(const float *src, int64_t src_stride, int64_t src_len, int64_t src_off, float *dst, int64_t dst_stride,
                            int64_t dst_len, int64_t B, hipStream_t s);

// Work-skipping ablations exist only in dev builds (tools/dev/build_variant.sh <tag> -DDFX_DEV): the product library has no switch that leaves
// work out of a pass.  DFX_DEV_SKIP=bits: no ERB tail (1), DF tail (2), projections of layers > 0 (4), df_convp (8) — timing only, results
// invalid.  DFX_DEV_STAGE_LO / _HI: only the stages [lo, hi] of the serial forward (DFX_STREAMS=0) are enqueued (which kernel disturbs a neighbour).
#ifdef DFX_DEV
static int dfx_dev_skip() {
    static const int v = [] { const char *e = getenv("DFX_DEV_SKIP"); return e ? atoi(e) : 0; }();
    return v;
}
static bool dfx_dev_stage(int n) {
    static const int lo = [] { const char *e = getenv("DFX_DEV_STAGE_LO"); return e ? atoi(e) : 0; }();
    static const int hi = [] { const char *e = getenv("DFX_DEV_STAGE_HI"); return e ? atoi(e) : 99; }();
    return n >= lo && n <= hi;
}
#else
static constexpr int dfx_dev_skip() { return 0; }
static constexpr bool dfx_dev_stage(int) { return true; }
#endif

template <int C>
static int forward_impl(const dfx_model *m, const dfx_bands *bands, const float *spec, const float *feat_erb,
                        const float *feat_spec, int64_t B, int64_t T, float atten_lim, float *spec_e, float *mask_out,
                        float *lsnr_out, float *coefs_out, float *ws, hipStream_t s, const DfxLane *ln, bool signal_front,
                        const DfxFinish *fin, const DfxStreamCtx *sc = nullptr) {
    const dfx_model_cfg &c = m->cfg;
    const int64_t R = B * T;
    // (row maps of the time-chunked launches divide in 32 bits, dfx_row; the workspace of 2^31 frames would be ~170 TB)
    if (R >= ((int64_t)1 << 31)) DFX_FAIL(DFX_ERR_UNSUPPORTED, "forward: %lld x %lld frames in one call (32-bit row index)", (long long)B, (long long)T);
    // streaming window (sc): the arrays hold T = H + n frames per clip, only the n new ones are computed; per-frame kernels reach
    // their rows through rmw, the lookahead shift is already in the feature stream (kernel lookahead 0)
    const int64_t t_begin = sc ? sc->H : 0, Rn = B * (T - t_begin);
    const int64_t featT = sc ? sc->feat_T : 0;   // frames per clip of feat_erb / feat_spec when they are windows inside longer buffers (0: T)
    const DfxRowMap rmw = sc ? DfxRowMap{T, T - t_begin, t_begin} : DfxRowMap{0, 0, 0};
    const int Lk = sc ? 0 : c.conv_lookahead;
    const int64_t t_zero = sc ? sc->t_zero : 0;
    const Ws w = plan_ws(c, m->fuse_c0 && !m->c0_batch_unfused, R, B);
    const int64_t sstride = fin ? fin->spec_stride : 0;  // 0: dense rows of F bins
    hipStream_t fin_s = s;                               // stream of the finishing kernels (deep filter, synthesis)
    const int E = c.nb_erb, Fd = c.nb_df, O = c.df_order, NO = 2 * O, emb = C * E / 4, L = c.conv_lookahead;
    float *e0 = ws + w.e0, *e1 = ws + w.e1, *e2 = ws + w.e2, *e3 = ws + w.e3, *c0 = ws + w.c0, *c1 = ws + w.c1;
    float *emb_in = ws + w.emb_in, *embv = ws + w.emb, *xa = ws + w.xa, *xb = ws + w.xb, *gi = ws + w.gi;
    float *demb = ws + w.demb, *d3 = ws + w.d3, *d2 = ws + w.d2, *d1 = ws + w.d1;
    float *mask = mask_out ? mask_out : ws + w.mask;
    float *c0p = ws + w.c0p, *xdf = ws + w.xdf;
    float *coefs = coefs_out ? coefs_out : ws + w.coefs;
    float *lsnr = lsnr_out ? lsnr_out : ws + w.lsnr;
    float *xa2 = ws + w.xa2, *xb2 = ws + w.xb2, *gi2 = ws + w.gi2;
    float *skp_e = ws + w.skp_e, *skp_d = ws + w.skp_d;
    int rc;
    const bool run_df = m->run_df;
    // SqueezedGRU_S (modules.py:702-738): x = linear_out(gru(linear_in(in))) [+ gru_skip(in)]; the skip joins after linear_out's ReLU
    auto enc_out_skip = [&](const float *y, int64_t M, hipStream_t st, DfxRowMap rm) -> int {   // deepfilternet3.py:138-158
        const float *res = nullptr;
        if (c.emb_gru_skip_enc == DFX_SKIP_IDENTITY) res = emb_in;
        else if (c.emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR) {
            if (int r = launch_glin(m, m->enc_skip, emb_in, DFX_ACT_NONE, nullptr, skp_e, M, st, rm)) return r;
            res = skp_e;
        }
        return launch_glin(m, m->enc_out, y, DFX_ACT_RELU, res, embv, M, st, rm);
    };
    auto dec_out_skip = [&](const float *y, int64_t M, hipStream_t st, DfxRowMap rm) -> int {   // deepfilternet3.py:198-216
        const float *res = nullptr;
        if (c.emb_gru_skip == DFX_SKIP_IDENTITY) res = embv;
        else if (c.emb_gru_skip == DFX_SKIP_GROUPEDLINEAR) {
            if (int r = launch_glin(m, m->dec_skip, embv, DFX_ACT_NONE, nullptr, skp_d, M, st, rm)) return r;
            res = skp_d;
        }
        return launch_glin(m, m->dec_out, y, DFX_ACT_RELU, res, demb, M, st, rm);
    };
    // dfx_k_emb_fan: emb = enc_out_skip(y) and its consumers in one pass.  emb itself is only written when something outside the kernel
    // still reads it (the ERB decoder's skip connection, an identity skip around the DF GRU).  df_skip(emb) lands in xdf WITHOUT the
    // DF GRU's output (which does not exist yet): df_out then takes its operand as the sum y_df + xdf (DfxGgArgs::a2).
    const bool fan = m->fuse_emb && m->fan_chunks > 0 && !c.enc_concat && emb == 64 * m->fan_chunks;   // (exact fp32 matrix ops: also with DFX_EXACT_FP32=1)
    const bool fan_skp = fan && run_df && c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR && m->fan_kind[2] == 1;
    auto emb_fan = [&](const float *y, float *dec_x, int64_t M, hipStream_t st, DfxRowMap rm, const DfxPublish *pub = nullptr) -> int {
        const float *res = nullptr;
        if (c.emb_gru_skip_enc == DFX_SKIP_IDENTITY) res = emb_in;
        else if (c.emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR) {
            if (int r = launch_glin(m, m->enc_skip, emb_in, DFX_ACT_NONE, nullptr, skp_e, M, st, rm)) return r;
            res = skp_e;
        }
        const bool need_emb = c.emb_gru_skip != DFX_SKIP_NONE || (run_df && c.df_gru_skip == DFX_SKIP_IDENTITY);
        return launch_emb_fan(m, y, res, need_emb ? embv : nullptr, dec_x, run_df ? xa2 : nullptr, fan_skp ? xdf : nullptr, lsnr, M, st, rm, embv, pub);
    };
    // c = tanh(df_out(c)).view(b,t,F',2O) + c0p   (:329-330) of M rows; cfeat (+ cfeat2) is df_out's operand
    auto df_out_rows = [&](const float *cfeat, const float *cfeat2, int64_t M, hipStream_t st, DfxRowMap rm) -> int {
        if (m->dfout_lean && m->dfo_nu > 0 && !m->exact_fp32 && M > 0 && R * (int64_t)NO * Fd < ((int64_t)1 << 31)) {   // row-streaming form (dfx_k_df_out_h3)
            DfxDfOutArgs A;
            A.a = cfeat, A.a2 = cfeat2;
            A.wf = reinterpret_cast<const dfx_h8 *>(m->p(m->dfo_h3));
            A.c0p = c0p, A.out = coefs;
            A.R = M, A.T = T;
            A.G = m->df_out.G, A.Kg = m->df_out.Kg, A.Ng = m->df_out.Ng, A.NO = NO, A.Fd = Fd;
            A.unscale = m->dfo_unscale;
            A.rm = rm;
            A.err = m->d_err;
            const size_t smem = DFX_DFO_SMEM(NO, Fd);
            DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_df_out_h3, smem));
            DfxKScope ks(DFX_K_GGEMM, st);
            dfx_launch(dfx_k_df_out_h3, dim3((unsigned)nn_grid(dfx_ceil_div(M, 16), 2)), dim3(DFX_DFO_THREADS), smem, st, A);
            DFX_LAUNCH_CHECK();
            return DFX_OK;
        }
        return launch_ggemm(cfeat, m->df_out.G * m->df_out.Kg, m->p(m->df_out.w), m->df_out.G, m->df_out.Kg, m->df_out.Ng, nullptr, DFX_ACT_TANH,
                            c0p, coefs, m->df_out.G * m->df_out.Ng, M, st, NO, Fd, T, rm, cfeat2);
    };
    // Stream plan (s = caller's stream, x1/x2 = auxiliary; all joins are events, the host never blocks):
    //   s : e0..e3 ----------------(join c1)-- fc_emb, enc GRU, emb, lsnr --+-- ERB decoder: GRU stack, convt3..conv0_out --(join coefs)-- df_apply
    //   x1: c0 -+- c1 ------------------------------------------------------+-- DF decoder: GRU stack, skip, (join c0p) df_out -> coefs
    //   x2:     +- df_convp -> c0p
    const bool par = m->concurrent;
    hipStream_t x1 = par ? ln->aux[0] : s, x2 = par ? ln->aux[1] : s;
    auto signal = [&](int e, hipStream_t from) -> int {
        if (par) DFX_HIP(hipEventRecord(ln->ev[e], from));
        return DFX_OK;
    };
    auto wait = [&](int e, hipStream_t on) -> int {
        if (par) DFX_HIP(hipStreamWaitEvent(on, ln->ev[e], 0));
        return DFX_OK;
    };
    if ((rc = signal(EV_START, s)) || (rc = wait(EV_START, x1))) return rc;
    const bool post_behind_convp = sc && sc->df_post && m->run_df;
    if (sc && sc->erb_pre && (rc = sc->erb_pre(s))) return rc;
    if (sc && sc->df_pre && (rc = sc->df_pre(x1))) return rc;
    if (sc && sc->df_post && !post_behind_convp && (rc = sc->df_post(x1))) return rc;
    // ---- Encoder, DF branch on x1 (deepfilternet3.py:176-179).  By default c0 = df_conv0(feat_spec) never exists in HBM: its two
    // consumers (df_conv1 here, df_convp below) recompute the tiles they need from feat_spec on the matrix core.
    const bool fuse_c0 = m->fuse_c0 && !(m->c0_batch_unfused && !sc);
    if (sc && !fuse_c0) DFX_FAIL(DFX_ERR_UNSUPPORTED, "streaming needs the fused DF encoder (df_pathway_kernel_size_t <= 5, df_order <= 8, DFX_FUSE_C0 unset)");
    const float *cp_feat = fuse_c0 ? feat_spec : nullptr;
    const bool fuse_h3 = fuse_c0 && !m->exact_fp32 && C % 32 == 0 && m->cp_h3;  // fp16-split matrix ops (default)
    const bool fuse_dec = E % 2 == 0 && m->fuse_erb && 2 * DFX_DEC10_SMEM(C, E) <= (size_t)160 * 1024;
    const bool fuse_tail = fuse_dec && erb_tail_ok<C>(m, E);
    const bool fuse_enc = E % 2 == 0 && 3 * (E + 2) <= 192 && m->fuse_erb && 2 * DFX_ENC_SMEM(C, E) <= (size_t)160 * 1024;
    const bool no_e0 = fuse_tail && fuse_enc && m->e0_recompute && R < ((int64_t)1 << 31);   // e0 never exists in HBM
    const float *e0r = no_e0 ? nullptr : e0;   // what the decoder tail is handed
    if (sc && !fuse_enc) DFX_FAIL(DFX_ERR_UNSUPPORTED, "streaming needs the fused ERB encoder head (DFX_FUSE_ERB unset)");
    const DfxGate *gate = sc ? sc->gate : nullptr;
    if (gate && T - t_begin != 1) DFX_FAIL(DFX_ERR_INVALID_ARG, "gated streaming passes carry exactly one new frame");
    // ---- How the GRU phase will run — decided before the front, because its persistent form starts UNDER the front.
    // Layer-pipelined over time chunks when the fp16-split kernels are in use: every GRU layer has its own
    // stream; layer l may run chunk k as soon as layer l-1 has produced chunk k, so the three-layer chain
    // enc -> dec1 -> dec2 (and enc -> df1 -> df2) costs T*(1 + 2/K) steps instead of 3T.  Each layer-kernel occupies B/16
    // CUs; the per-chunk projections and grouped linears address their rows through a DfxRowMap.
    int K = m->tchunks;
    if (T / K < m->tchunk_min) K = (int)(T / m->tchunk_min);
    const int nenc = (int)m->enc_gru.size(), ndec = (int)m->dec_gru.size(), ndf = run_df ? (int)m->df_gru.size() : 0;
    // (DFX_EXACT_FP32=1: the same pipeline on dfx_k_gru_rec_x32 / dfx_k_proj256 — 16 CUs per layer instead of the VALU kernel's 128)
    const bool pipe = par && !sc && K > 1 && nenc == 1 && 1 + ndec + ndf <= DFX_MAX_GRU_LAYERS && ln == &m->lanes[0];
    const int nl = 1 + ndec + ndf;
    // persistent form (default on the GPU): ONE launch runs the recurrences of all layers for the whole sequence
    // (dfx_k_gru_seq); needs every (layer, group) workgroup resident at once (each owns a CU)
    const int groups = (int)dfx_ceil_div(B, DFX_GH_ROWS);
    if (pipe && m->gru_seq && m->hwq_probe_pending && groups <= DFX_SEQ_GMAX && nl * groups + 8 <= dfx_env_num_cus())
        hwq_probe_run(const_cast<dfx_model *>(m));   // first pass that would use the persistent form: do its streams run concurrently?
    const bool want_seq = pipe && m->gru_seq && groups <= DFX_SEQ_GMAX && nl * groups + 8 <= dfx_env_num_cus();
    // (another process in its persistent phase on this device: this pass takes the event-synchronised form, DfxTicket)
    const bool use_seq = want_seq && dfx_ticket_try();
    m->passes_seq += use_seq ? 1 : 0, m->passes_ev += (want_seq && !use_seq) ? 1 : 0;
    struct TicketGuard {   // an enqueue that fails half-way gives the ticket back at once
        bool armed;
        ~TicketGuard() {
            if (armed) dfx_ticket_release_cb(nullptr);
        }
    } ticket_guard{use_seq && dfx_ticket().fd >= 0};
    int sb[DFX_GS_MAX_CHUNKS + 1];   // chunk boundaries of the persistent form
    int Ks = 0;
    if (use_seq) {
        // short chunks at the start (the next layer can begin after the first chunk + its preparation: the
        // pipeline of 3 layers fills in ~3 short chunks instead of 3 long ones) and at the end (what is left to do after the last
        // recurrence step is one short chunk's decoder tail), uniform in between
        // measured at batch 256 x 1002 frames (ms per step): 8 body chunks + ramp from 32: 21.28; 12 + 16: 21.91; 12, no ramp: 21.47;
        // 6 + 32: 21.35; 4 + 32: 22.45; 16 + 16: 22.69 (the event-based form: 22.07); after the decoder convolutions went to the
        // staged fp16-split kernels (lighter background): 8 + 32: 20.1; 10 + 32: 19.85; 12 + 32: 19.99; 12 + 16: 20.27; 16 + 32: 21.0
        // round 4, after e0 / c1 / the grouped-GEMM df_out left the phase (lighter side work, shorter hand-overs), same-box A/B: 10 + ramp 32: 14.47;
        // 12 uniform chunks, no ramp: 14.12; 13: 14.17; 14: 14.14; 12 + ramp 48: 14.20; 15 + 48: 14.27; 16: 15.1 (chunks of < 16384 rows take the
        // small-launch forms of the fan-out kernels) -> 12 uniform chunks
        const int ramp0 = m->sw.ramp;
        // 16 chunks where the producers raise their flags themselves (DfxPublish: 17 launches per chunk), 12 where a one-thread launch does
        // (22 per chunk: the exact mode, DFX_SEQ_PUBLISH=0) — measured 13.24-13.28 (16) vs 13.37-13.49 (12) ms per step, measurements R5.10
        const int kenv = m->sw.chunks;
        const bool kpub = m->sw.publish;
        // (with followers only the encoder layer's projections and the decoder tails are still per chunk: 12 again, 12.98 vs 13.12 ms at 16)
        const int kbody = kenv > 0 ? kenv : (kpub && !m->exact_fp32 && seq_follow_mode(m) <= 0 ? 16 : 12);
        const int64_t body = std::max<int64_t>(dfx_ceil_div(T, (int64_t)kbody), m->tchunk_min);   // uniform chunk length: DFX_SEQ_CHUNKS=n gives n chunks (ceil: 1002 / 12 -> 84, not 83 and a 13th chunk)
        std::vector<int> sizes;
        int64_t left = T;
        for (int64_t r = ramp0; ramp0 > 0 && r < body && left > 4 * body; r *= 2) sizes.push_back((int)r), left -= r;   // up
        std::vector<int> down;
        for (int64_t r = ramp0; ramp0 > 0 && r < body && left > 3 * body; r *= 2) down.push_back((int)r), left -= r;     // down (round 5, the ramp at the end alone, 16 / 32 frames: 13.09-13.22 vs 13.11-13.18 ms, noise)
        const int nbody = (int)std::max<int64_t>(1, std::min<int64_t>(dfx_ceil_div(left, body), DFX_GS_MAX_CHUNKS - (int64_t)sizes.size() - (int64_t)down.size()));
        for (int i = 0; i < nbody; ++i) sizes.push_back((int)(left * (i + 1) / nbody - left * i / nbody));
        for (auto it = down.rbegin(); it != down.rend(); ++it) sizes.push_back(*it);
        Ks = (int)sizes.size();
        sb[0] = 0;
        for (int i = 0; i < Ks; ++i) sb[i + 1] = sb[i] + sizes[i];
    }
    const int kt = c.df_pathway_kernel_size_t;
    int64_t convp_split = T;   // frames [convp_split, T) of df_convp are enqueued under the GRU phase (DFX_CONVP_LATE)
    // df_conv0 -> df_conv1 of frames [t0, t1) (fuse_c0)
    auto df1_range = [&](int64_t t0, int64_t t1, hipStream_t st) -> int {
        if (fuse_h3) return launch_conv01_h3<C>(m, m->dfc1, feat_spec, c1, B, T, Fd, Fd / 2, 2, st, t0, Lk, t1, featT);
        return launch_conv01<C>(m, m->dfc1, feat_spec, c1, B, T, Fd, Fd / 2, 2, st, t0, Lk, t1);
    };
    // df_dec.df_convp of frames [t0, t1) (only needs c0 / feat_spec; :328)
    auto convp_range = [&](int64_t t0, int64_t t1, hipStream_t st) -> int {
        if (dfx_dev_skip() & 8) return DFX_OK;
        if (gate && kt > 1 && gate->pend2 && fuse_h3 && t1 - t0 == 1 && t1 == T) {   // gated, fp16-split: pending sums, two halves per stream
            switch (kt) {
                case 2: return launch_convp_step<C, 2>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, gate->pend2, 0, false, featT, gate->par, gate->cnt);
                case 3: return launch_convp_step<C, 3>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, gate->pend2, 0, false, featT, gate->par, gate->cnt);
                case 4: return launch_convp_step<C, 4>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, gate->pend2, 0, false, featT, gate->par, gate->cnt);
                default: return launch_convp_step<C, 5>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, gate->pend2, 0, false, featT, gate->par, gate->cnt);
            }
        }
        if (gate && kt > 1) {
            // gated streaming: the (kt-1)-frame delay line in front of df_convp belongs to the DF decoder and only moves on the frames
            // that decoder ran on, per stream.  c0 of the newest frame goes into the last slot of the per-stream window (exact fp32
            // matrix ops), the pathway conv reads the window; dfx_k_gate_c0_shift advances it where stage 2 ran.
            if (kt > 5) DFX_FAIL(DFX_ERR_UNSUPPORTED, "gated streaming needs df_pathway_kernel_size_t <= 5");
            DfxCinArgs A;
            A.feat = feat_spec;
            A.weff = m->p(m->cin_weff);
            A.bias = m->p(m->cin_b);
            A.out = gate->c0_win;
            A.B = B;
            A.T = T;
            A.Fin = Fd;
            A.L = Lk;
            A.t_begin = T - 1;
            A.out_T = T;
            A.out_toff = T - 1;
            {
                DfxKScope ks(DFX_K_CONV_IN_DF, st);
                dfx_launch(dfx_k_conv_in_df<C>, dim3((unsigned)nn_grid(dfx_ceil_div(B * Fd, 64), 8)), dim3(DFX_PW_THREADS), 0, st, A);
                DFX_LAUNCH_CHECK();
            }
            switch (kt) {
                case 2: return launch_convp2<C, 2>(m, gate->c0_win, nullptr, c0p, B, T, Fd, NO, st, T - 1, 0, Lk);
                case 3: return launch_convp2<C, 3>(m, gate->c0_win, nullptr, c0p, B, T, Fd, NO, st, T - 1, 0, Lk);
                case 4: return launch_convp2<C, 4>(m, gate->c0_win, nullptr, c0p, B, T, Fd, NO, st, T - 1, 0, Lk);
                default: return launch_convp2<C, 5>(m, gate->c0_win, nullptr, c0p, B, T, Fd, NO, st, T - 1, 0, Lk);
            }
        } else if (fuse_h3 && sc && sc->c0ring && !gate && t1 - t0 == 1 && t1 == T && kt >= 2) {
            sc->c0ring_used = true;
            switch (kt) {
                case 2: return launch_convp_step<C, 2>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, sc->c0ring, sc->c0slot, sc->c0rebuild, featT);
                case 3: return launch_convp_step<C, 3>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, sc->c0ring, sc->c0slot, sc->c0rebuild, featT);
                case 4: return launch_convp_step<C, 4>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, sc->c0ring, sc->c0slot, sc->c0rebuild, featT);
                default: return launch_convp_step<C, 5>(m, feat_spec, c0p, B, T, Fd, NO, st, t_zero, Lk, sc->c0ring, sc->c0slot, sc->c0rebuild, featT);
            }
        } else if (fuse_h3) {
            switch (kt) {
                case 1: return launch_convp_h3<C, 1>(m, feat_spec, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1, featT);
                case 2: return launch_convp_h3<C, 2>(m, feat_spec, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1, featT);
                case 3: return launch_convp_h3<C, 3>(m, feat_spec, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1, featT);
                case 4: return launch_convp_h3<C, 4>(m, feat_spec, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1, featT);
                default: return launch_convp_h3<C, 5>(m, feat_spec, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1, featT);
            }
        } else if (kt <= 5 && NO <= 16) {
            switch (kt) {
                case 1: return launch_convp2<C, 1>(m, c0, cp_feat, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1);
                case 2: return launch_convp2<C, 2>(m, c0, cp_feat, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1);
                case 3: return launch_convp2<C, 3>(m, c0, cp_feat, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1);
                case 4: return launch_convp2<C, 4>(m, c0, cp_feat, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1);
                default: return launch_convp2<C, 5>(m, c0, cp_feat, c0p, B, T, Fd, NO, st, t0, t_zero, Lk, t1);
            }
        }
        // tiled form (kt > 5 or more than 8 taps): whole sequences only
        DfxCpArgs A;
        A.c0 = c0;
        A.w1 = m->p(m->cp_w1);
        A.w2 = m->p(m->cp_w2);
        A.bias = m->p(m->cp_b);
        A.out = c0p;
        A.B = B;
        A.T = T;
        A.Fd = Fd;
        A.kt = kt;
        A.G = m->cp_G;
        A.NO = NO;
        A.tchunks = (int)dfx_ceil_div(T, DFX_CP_TT);
        A.fchunks = (Fd + DFX_CP_FB - 1) / DFX_CP_FB;
        const int CG = C / A.G;
        const size_t smem = ((size_t)(DFX_CP_TT + A.kt - 1) * DFX_CP_FB * (C + 2) + (size_t)A.G * A.kt * CG * 16 +
                             (size_t)DFX_CP_TT * DFX_CP_FB * NO) * sizeof(float);
        if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_df_convp<C>, smem));
        const int64_t nblk = B * A.tchunks * A.fchunks;
        if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "df_convp grid too large");
        DfxKScope ks(DFX_K_DF_CONVP, st);
        dfx_launch(dfx_k_df_convp<C>, dim3((unsigned)nblk), dim3(DFX_CP_THREADS), smem, st, A);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    };
    // Encoder, ERB branch (:168-171) for frames [t0, t1) = Rk rows reached through rm
    auto erb_range = [&](int64_t t0, int64_t t1, int64_t Rk, DfxRowMap rm, hipStream_t st) -> int {
        int r;
        if (fuse_enc) {
            if ((r = launch_erb_enc<C>(m, feat_erb, no_e0 ? nullptr : e0, e1, B, T, st, t0, Lk, t1, featT))) return r;
        } else {
            {
                const int64_t total = R * E * (C / 4);
                DfxKScope ks(DFX_K_CONV_IN_ERB, st);
                dfx_launch(dfx_k_conv_in_erb, dim3((unsigned)nn_grid(dfx_ceil_div(total, 256), 16)), dim3(256), 0, st, feat_erb,
                           m->p(m->erb0_w), m->p(m->erb0_b), e0, B, T, E, C, L);
                DFX_LAUNCH_CHECK();
            }
            if ((r = launch_pw<C>(DFX_PW_MODE_DW3, m, m->erb1, e0, nullptr, e1, R, E, E / 2, 2, st))) return r;
        }
        if ((r = launch_pw<C>(DFX_PW_MODE_DW3, m, m->erb2, e1, nullptr, e2, Rk, E / 2, E / 4, 2, st, rm))) return r;
        return launch_pw<C>(DFX_PW_MODE_DW3, m, m->erb3, e2, nullptr, e3, Rk, E / 4, E / 4, 1, st, rm);
    };
    // cemb = relu(df_fc_emb(c1.flatten)); emb_in = e3.flatten + cemb (:179-182), then enc.emb_gru's linear_in (SqueezedGRU_S :149-158).
    // (DFX_FUSE_EMB=0 also restores the two grouped GEMMs of the front)
    const bool enc_fan = m->fuse_emb && m->fuse_encfan && m->efan_groups > 0 && !c.enc_concat && emb == 16 * m->efan_groups;
    auto emb_range = [&](int64_t Rk, DfxRowMap rm, hipStream_t st) -> int {
        int r;
        if (enc_fan) return launch_enc_fan(m, c1, e3, c.emb_gru_skip_enc != DFX_SKIP_NONE ? emb_in : nullptr, xa, Rk, st, rm);
        if (c.enc_concat) {  // emb = cat(e3.flatten, cemb) (deepfilternet3.py:132-134,181): e3 rows into the left half, cemb written into the right half
            if ((r = stream_copy_rows(e3, emb, emb, 0, emb_in, 2 * emb, emb, R, st))) return r;   // (all rows: enc_concat excludes the ranged front)
            if ((r = launch_ggemm(c1, m->fc_emb.G * m->fc_emb.Kg, m->p(m->fc_emb.w), m->fc_emb.G, m->fc_emb.Kg, m->fc_emb.Ng, nullptr, DFX_ACT_RELU,
                                  nullptr, emb_in + emb, 2 * emb, Rk, st, 0, 0, 1, rm)))
                return r;
        } else if ((r = launch_glin(m, m->fc_emb, c1, DFX_ACT_RELU, e3, emb_in, Rk, st, rm))) return r;
        return launch_glin(m, m->enc_in, emb_in, DFX_ACT_RELU, nullptr, xa, Rk, st, rm);
    };
    // the DF branch of the encoder as one kernel behind the ERB convolutions (it adds e3), c1 never stored
    const bool dfenc = m->fuse_dfenc && fuse_h3 && enc_fan && m->dfenc_chunks > 0 && B * T * (int64_t)emb < ((int64_t)1 << 31) &&
                       B * (featT > 0 ? featT : T) * Fd < ((int64_t)1 << 29);   // (32-bit element offsets inside the kernel; beyond: the two kernels)
    {   // ---- the front: the frames [t_begin, T) that this pass computes
        if (fuse_c0) {
            if ((rc = signal(EV_C0, x1)) || (rc = wait(EV_C0, x2))) return rc;  // df_convp only needs feat_spec
            if (!dfenc && dfx_dev_stage(3) && (rc = df1_range(t_begin, T, x1))) return rc;
        } else {
            DfxCinArgs A;
            A.feat = feat_spec;
            A.weff = m->p(m->cin_weff);
            A.bias = m->p(m->cin_b);
            A.out = c0;
            A.B = B;
            A.T = T;
            A.Fin = Fd;
            A.L = L;
            A.t_begin = 0;
            A.out_T = T;
            A.out_toff = 0;
            DfxKScope ks(DFX_K_CONV_IN_DF, x1);
            dfx_launch(dfx_k_conv_in_df<C>, dim3((unsigned)nn_grid(dfx_ceil_div(R * Fd, 64), 8)), dim3(DFX_PW_THREADS), 0, x1, A);
            DFX_LAUNCH_CHECK();
            if ((rc = signal(EV_C0, x1)) || (rc = wait(EV_C0, x2))) return rc;
            if ((rc = launch_pw<C>(DFX_PW_MODE_DW3, m, m->dfc1, c0, nullptr, c1, R, Fd, Fd / 2, 2, x1))) return rc;
        }
        if ((rc = signal(EV_C1, x1))) return rc;
        // the pathway conv only has to finish before df_out: it starts right away on x2 and fills whatever the encoder kernels leave idle
        // (releasing it later — behind df_conv1, or behind the whole front — measured the same within noise, profiles/r01_gru_phase_ablation.log;
        // per time chunk inside the GRU phase: slower, profiles/r04_gru_floor_and_convp_phase.log)
        if (run_df) {
            // Round 5: with the persistent GRU phase the pathway conv is released only when the front has run, i.e. it runs UNDER the phase: the
            // front's critical path (ERB convolutions -> DF encoder) has the chip to itself, and since the decoder tail got 0.4 ms lighter the
            // phase has the room: 13.50 / 13.54 -> 13.18 / 13.22 ms per step (same box; 30 / 50 / 70 % of the frames deferred: 13.41 / 13.41 /
            // 13.48; in round 4, with the heavier tail, the same move measured as noise).  DFX_CONVP_LATE=p defers the last p percent (0: as before).
            // Exact mode with followers: in front of the phase, beside the (long) exact front — under the phase it starves the encoder layer's first
            // projections on the CUs the followers leave (25.9 vs 30.3 ms per step).
            const int late_env = m->sw.convp_late;
            const int late_pct = late_env >= 0 ? late_env : (m->exact_fp32 && seq_follow_mode(m) >= 2 ? 0 : 100);
            convp_split = use_seq && late_pct > 0 ? T - (T - t_begin) * late_pct / 100 : T;
            if (convp_split > t_begin && dfx_dev_stage(1) && (rc = convp_range(t_begin, convp_split, x2))) return rc;
            if (post_behind_convp && (rc = sc->df_post(x2))) return rc;
            if (convp_split >= T && (rc = signal(EV_C0P, x2))) return rc;
        }
        // (Round 5, timing only: the fused DF encoder on x1 BESIDE the ERB convolutions, its e3 dependency ignored — one VALU-bound, the others
        // HBM-bound — 13.68 / 13.71 vs 13.36 / 13.31 ms per step: slower; the encoder stays behind them.)
        if (dfx_dev_stage(2) && (rc = erb_range(t_begin, T, Rn, rmw, s))) return rc;
        if ((rc = wait(EV_C1, s))) return rc;
        if (dfenc) {
            if (dfx_dev_stage(3) && (rc = launch_df_enc<C>(m, feat_spec, e3, c.emb_gru_skip_enc != DFX_SKIP_NONE ? emb_in : nullptr, xa, B, T, Fd, s, t_begin, Lk, T, featT))) return rc;
        } else if (dfx_dev_stage(3) && (rc = emb_range(Rn, rmw, s))) return rc;
        // the chip-filling front of this chunk is enqueued: the next chunk of a pipelined dfx_enhance may start its own front
        // (it then overlaps this chunk's GRU chain, which occupies only a few CUs)
        if (signal_front && (rc = signal(EV_FRONT, s))) return rc;
    }
    // ---- GRU phase (planned above)
    float *hs_enc = sc ? sc->h_state : nullptr, *hs_dec = sc ? sc->h_state + (int64_t)nenc * B * 256 : nullptr;
    float *hs_df = sc ? sc->h_state + (int64_t)(nenc + ndec) * B * 256 : nullptr;
    float *hn_enc = sc && sc->h_next ? sc->h_next : nullptr, *hn_dec = hn_enc ? hn_enc + (int64_t)nenc * B * 256 : nullptr;
    float *hn_df = hn_enc ? hn_enc + (int64_t)(nenc + ndec) * B * 256 : nullptr;
    if (!pipe) {
        const float *y = xa;
        if (dfx_dev_stage(4) && (rc = run_gru_stack(m, m->enc_gru, xa, xa, xb, gi, B, T, &y, s, hs_enc, t_begin, rmw, hn_enc))) return rc;
        float *dec_x = y == xa ? xb : xa;   // input of the ERB decoder's GRU stack
        if (fan) {
            if (dfx_dev_stage(5) && (rc = emb_fan(y, dec_x, Rn, s, rmw))) return rc;
            if ((rc = signal(EV_EMB, s)) || (rc = wait(EV_EMB, x1))) return rc;
        } else {
            if ((rc = enc_out_skip(y, Rn, s, rmw))) return rc;
            if ((rc = signal(EV_EMB, s)) || (rc = wait(EV_EMB, x1))) return rc;
            {
                DfxKScope ks(DFX_K_LSNR, s);
                dfx_launch(dfx_k_lsnr, dim3((unsigned)dfx_ceil_div(R * 64, 256)), dim3(256), 0, s, (const float *)embv, m->p(m->lsnr_w),
                           m->lsnr_b, (float)(c.lsnr_max - c.lsnr_min), (float)c.lsnr_min, lsnr, R, emb);
            }
            DFX_LAUNCH_CHECK();
        }
        if (gate) {  // stage decisions of the newest frame (tract.rs:658-672)
            dfx_launch(dfx_k_gate_post, dim3((unsigned)dfx_ceil_div(B, 256)), dim3(256), 0, s, (const float *)lsnr, T, gate->thr[0],
                       gate->thr[1], gate->thr[2], gate->flags, B, gate->channels);
            DFX_LAUNCH_CHECK();
        }
        // ---- DfDecoder on x1 (:323-331)
        if (run_df && dfx_dev_stage(6)) {
            const float *y2 = nullptr;
            if (!fan && (rc = launch_glin(m, m->dfg_in, embv, DFX_ACT_RELU, nullptr, xa2, Rn, x1, rmw))) return rc;
            if ((rc = run_gru_stack(m, m->df_gru, xa2, xa2, xb2, gi2, B, T, &y2, x1, hs_df, t_begin, rmw, hn_df, par))) return rc;
            const float *cfeat = y2, *cfeat2 = nullptr;
            if (fan_skp) {
                cfeat2 = xdf;   // df_skip(emb), written by dfx_k_emb_fan
            } else if (c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR) {
                if ((rc = launch_glin(m, m->df_skip, embv, DFX_ACT_NONE, y2, xdf, Rn, x1, rmw))) return rc;
                cfeat = xdf;
            } else if (c.df_gru_skip == DFX_SKIP_IDENTITY) {
                DfxKScope ks(DFX_K_ADD, x1);
                dfx_launch(dfx_k_add, dim3((unsigned)nn_grid(dfx_ceil_div(R * 256, 256), 16)), dim3(256), 0, x1, y2, (const float *)embv,
                           xdf, R * 256);
                DFX_LAUNCH_CHECK();
                cfeat = xdf;
            }
            if ((rc = wait(EV_C0P, x1))) return rc;
            // c = tanh(df_out(c)).view(b,t,F',2O) + c0p   (:329-330); the reference's flat index f*2O + 2n + {re,im} is stored
            // tap-major, [B,O,T,F'][2] (DFX_COEF_BOTF == the reference's DfOutputReshapeMF layout), so the deep-filter kernel
            // reads coefficients coalesced over f
            if ((rc = df_out_rows(cfeat, cfeat2, Rn, x1, rmw))) return rc;
            if ((rc = signal(EV_COEFS, x1))) return rc;
        }
        // ---- ErbDecoder on s (:245-254)
        if (!fan && (rc = launch_glin(m, m->dec_in, embv, DFX_ACT_RELU, nullptr, dec_x, Rn, s, rmw))) return rc;
        if (dfx_dev_stage(7) && (rc = run_gru_stack(m, m->dec_gru, dec_x, xa, xb, gi, B, T, &y, s, hs_dec, t_begin, rmw, hn_dec, par && run_df))) return rc;
        if (dfx_dev_stage(8) && (rc = dec_out_skip(y, Rn, s, rmw))) return rc;
        if (fuse_tail) {
            if (dfx_dev_stage(9) && (rc = launch_erb_tail<C>(m, demb, e3, e2, e1, e0r, mask, Rn, E, s, rmw, feat_erb, T, featT, Lk))) return rc;
        } else if ((rc = launch_pw<C>(DFX_PW_MODE_DW3, m, m->ct3, demb, e3, d3, Rn, E / 4, E / 4, 1, s, rmw)) ||
                   (rc = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct2, d3, e2, d2, Rn, E / 4, E / 2, 2, s, rmw))) {
            return rc;
        } else if (fuse_dec) {
            if ((rc = launch_erb_dec10<C>(m, d2, e1, e0, mask, Rn, E, s, rmw))) return rc;
        } else {
            if ((rc = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct1, d2, e1, d1, Rn, E / 2, E, 2, s, rmw))) return rc;
            const int fpt = 64 / E > 0 ? 64 / E : 1;
            const size_t smem = ((size_t)fpt * E * (C + 1) + (size_t)fpt * E * 3 + 3 * C) * sizeof(float);
            DfxKScope ks(DFX_K_CONV_OUT, s);
            dfx_launch(dfx_k_conv_out<C>, dim3((unsigned)nn_grid(dfx_ceil_div(Rn, fpt), 8)), dim3(DFX_CO_THREADS), smem, s,
                       (const float *)d1, (const float *)e0, m->p(m->co_ska), m->p(m->co_skb), m->p(m->co_w), m->co_bias, mask,
                       Rn, E, fpt, rmw);
            DFX_LAUNCH_CHECK();
        }
    } else {
        auto tb = [&](int k) { return (int64_t)k * T / K; };
        auto rmk = [&](int k) { return DfxRowMap{T, tb(k + 1) - tb(k), tb(k)}; };
        auto Mk = [&](int k) { return B * (tb(k + 1) - tb(k)); };
        auto ewait = [&](hipEvent_t e, hipStream_t on) -> int {
            DFX_HIP(hipStreamWaitEvent(on, e, 0));
            return DFX_OK;
        };
        auto esig = [&](hipEvent_t e, hipStream_t from) -> int {
            DFX_HIP(hipEventRecord(e, from));
            return DFX_OK;
        };
        // Per layer l two streams: ps[l] prepares chunk k (linear_in of a stack's first layer + the input projection) as soon
        // as its input rows exist and signals pev[l][k]; gs[l] runs nothing but the recurrences, chunk after chunk, and
        // signals gev[l][k].  Two tail streams consume the last layers' chunks (linear_out / skip / df_out) and then run the
        // rest of their decoder.  The latency chain is therefore K+2 recurrence chunks and nothing else.
        auto proj_chunk = [&](const GruW &g, int l, int k, const float *xin, hipStream_t st) -> int {
            if ((dfx_dev_skip() & 4) && l > 0) return DFX_OK;
            if (m->exact_fp32) return launch_proj(xin, m->p(g.wih_t), m->p(g.bias_i), ws + w.pgi[l], Mk(k), 768, st, rmk(k));
            return launch_proj_h3(m, g, xin, ws + w.pgi[l], Mk(k), 768, st, rmk(k));
        };
        // ---- persistent form (default on the GPU): ONE launch runs the recurrences of all layers for the whole sequence
        // (dfx_k_gru_seq); the projections / grouped linears / decoder tails stay per time chunk on three streams and meet the
        // recurrences through flag words in device memory instead of events — no kernel boundary, no relaunch, no pending
        // cross-queue barrier packet inside the phase.  Chunk boundaries sb[0..Ks]: planned above.
        hipStream_t seq_tail = nullptr;   // the stream that carries the DF tail of the persistent form
        if (use_seq) {
            const int K = Ks;   // (shadows the uniform chunk count of the event-based form)
            auto tb = [&](int k) { return (int64_t)sb[k]; };
            auto rmk = [&](int k) { return DfxRowMap{T, tb(k + 1) - tb(k), tb(k)}; };
            auto Mk = [&](int k) { return B * (tb(k + 1) - tb(k)); };
            const unsigned int base = m->seq_base;
            m->seq_base += (unsigned int)K + 1u;
            unsigned int *ready = m->d_sync, *embf = m->d_sync + 8, *done = m->d_sync + 16;
            unsigned int *pcnt = m->d_sync + 16 + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX;   // one completion counter per producing stream (layer), [8] = emb
            // a layer's input projection of chunk k, and ready[l] = chunk k + 1 behind it: raised by the projection kernel's last workgroup
            // (DfxPublish; DFX_SEQ_PUBLISH=0 or the exact mode: by a one-thread launch behind it, as before round 5)
            const bool publish = m->sw.publish;
            // Follower workgroups (dfx_k_proj_follow) feed the decoder layers in blocks of 16 steps instead of time chunks (seq_follow_mode; default 2:
            // all of them — a follower of the encoder GRU, dfx_k_emb_follow, runs dfx_k_emb_fan's arithmetic per block of 8 steps and the stacks' first
            // layers' projection followers read what it wrote; 1: only the layers whose input is the output of the layer below; 0: launches per chunk).
            const int follow_env = seq_follow_mode(m);
            unsigned int *yprog = pcnt + 16, *giprog = yprog + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX;
            unsigned int *embprog = yprog + (size_t)(DFX_MAX_GRU_LAYERS - 1) * DFX_SEQ_GMAX;   // (the row of a layer that cannot exist: nl < 8 below)
            // same-XCD hand-overs (DfxXcd; DFX_SEQ_XCD_LIGHT=0: every block hand-over with the agent-scope release / acquire)
            const bool xcd_light = m->sw.xcd_light;
            unsigned int *xtab = xcd_light ? giprog + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX : nullptr;
            unsigned int *xstat = xtab ? xtab + 3 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX : nullptr;
            const unsigned int xtag = (m->seq_pbase & 0x0fffffffu) << 4;
            auto xword = [&](int kind, int layer) { return xtab + ((size_t)kind * DFX_MAX_GRU_LAYERS + layer) * DFX_SEQ_GMAX; };
            const unsigned int pbase = m->seq_pbase;
            bool followed[DFX_MAX_GRU_LAYERS] = {};
            int nfollow = 0;
            const int lfirst_df = 1 + ndec;
            const bool follow_emb = follow_env >= 2 && fan && c.emb_gru_skip_enc != DFX_SKIP_GROUPEDLINEAR && nl < DFX_MAX_GRU_LAYERS &&
                                    (nl + nl) * groups <= dfx_env_num_cus() * 3 / 4 && nl - 1 <= DFX_PF_MAX;
            if (follow_env >= 1) {
                for (int l = 1; l < nl; ++l) {
                    const bool first = l == 1 || l == lfirst_df;   // a stack's first layer reads a grouped linear of emb, the others the layer below
                    if (first ? follow_emb : follow_env != 3) followed[l] = true, ++nfollow;   // (3: the first layers only)
                }
                if (nfollow > DFX_PF_MAX || (nl + nfollow + 1) * groups > dfx_env_num_cus() * 3 / 4) {   // all of them or none (every workgroup must be resident; passes of other handles never overlap this one: PassTurn)
                    nfollow = 0;
                    for (int l = 0; l < nl; ++l) followed[l] = false;
                }
            }
            if (nfollow) m->seq_pbase += (unsigned int)T + 1u;
            auto proj_chunk = [&](const GruW &g, int l, int k, const float *xin, hipStream_t st) -> int {
                const unsigned int val = base + (unsigned int)k + 1u;
                if (m->exact_fp32 || !publish) {
                    const int r = m->exact_fp32 ? launch_proj(xin, m->p(g.wih_t), m->p(g.bias_i), ws + w.pgi[l], Mk(k), 768, st, rmk(k))
                                                : launch_proj_h3(m, g, xin, ws + w.pgi[l], Mk(k), 768, st, rmk(k));
                    return r ? r : launch_flag_set(ready + l, val, st);
                }
                DfxPublish pub;
                pub.cnt = pcnt + l, pub.flag = ready + l, pub.value = val;
                return launch_proj_h3(m, g, xin, ws + w.pgi[l], Mk(k), 768, st, rmk(k), &pub);
            };
            // (Finishing — deep filter + ISTFT — per time chunk behind the DF tail was built here and in the event-based form and measured:
            // 21.7 vs 20.2 ms per step; the chunks' traffic beside the chain costs more than the 1.2 ms it takes off the end.)
            auto donep = [&](int l) { return done + (size_t)l * DFX_SEQ_GMAX; };
            auto tgt = [&](int k) { return base + (unsigned int)k + 1u; };
            hipStream_t G = ln->gs[1], Eq = ln->ts[0], Dq = ln->ts[1], Pq = ln->ps[0];
            const int ev_go = EV_XA;   // the front is complete
            if ((rc = signal(EV_XA, s))) return rc;
            // Staged enqueue (big passes; off with DFX_ENQUEUE_AHEAD=1 or DFX_PHASE_LATE=0): the host enqueues the phase only once the front
            // has run, so that no barrier packets sit at the head of the phase's ~10 queues while the front's kernels run — measured
            // 18.83 -> 18.20 ms per step (the same effect as between passes, dfx_model::ev_pass).  The persistent launch goes out first
            // and the rest follows chunk-major, faster than the chain consumes it.
            if (m->phase_late && !m->enqueue_ahead && R >= DFX_THROTTLE_MIN_FRAMES) DFX_HIP(hipEventSynchronize(ln->ev[EV_XA]));
            if ((rc = wait(ev_go, G)) || (rc = wait(ev_go, Eq)) || (rc = wait(ev_go, Dq)) || (rc = wait(ev_go, Pq))) return rc;
            // (the followers' claim counters, dfx_xcd_claim: zeroed in front of the recurrences, whose registrations every follower waits for)
            if (nfollow && xtab) DFX_HIP(hipMemsetAsync(xstat + 8, 0, (size_t)(DFX_PF_MAX + 1) * 8 * sizeof(unsigned int), G));
            {   // the recurrences
                DfxGsArgs S;
                for (int l = 0; l < DFX_GS_MAX_LAYERS; ++l) S.gi[l] = nullptr, S.y[l] = nullptr, S.whf[l] = nullptr, S.bhn[l] = nullptr, S.unscale[l] = 1.f;
                for (int l = 0; l < nl; ++l) {
                    const GruW &g = l == 0 ? m->enc_gru[0] : (l <= ndec ? m->dec_gru[l - 1] : m->df_gru[l - 1 - ndec]);
                    S.gi[l] = ws + w.pgi[l];
                    S.y[l] = ws + w.py[l];
                    S.whf[l] = reinterpret_cast<const dfx_h8 *>(m->p(m->exact_fp32 ? g.whh_x32 : g.whh_h3));
                    S.bhn[l] = m->p(g.bhn);
                    S.unscale[l] = m->exact_fp32 ? 1.f : g.whh_unscale;
                }
                S.B = B, S.T = T, S.nlayers = nl, S.groups = groups, S.K = K;
                for (int i = 0; i <= K; ++i) S.tb[i] = sb[i];
                S.ready = ready, S.done = done, S.done_stride = DFX_SEQ_GMAX, S.base = base, S.err = m->d_err;
                S.trace = m->d_trace;
                S.spin_limit = m->spin_limit;
                S.pbase = pbase, S.sblk = 16;
                if (xtab) S.xtab = xtab, S.xstride = DFX_SEQ_GMAX, S.xtag = xtag, S.xstat = xstat;
                for (int l = 1; l < nl; ++l) {
                    if (!followed[l]) continue;
                    S.giprog[l] = giprog + (size_t)l * DFX_SEQ_GMAX;
                    const bool first = l == 1 || l == lfirst_df;
                    const int src = first ? 0 : l - 1;   // the recurrence whose output feeds the follower chain of layer l
                    S.yprog[src] = yprog + (size_t)src * DFX_SEQ_GMAX;
                    S.yblk[src] = first ? DFX_EF_STEPS : 16;
                    S.xcons_kind[src] = first ? 2 : 1, S.xcons_layer[src] = first ? 0 : l;
                }
                m->trace_dims[0] = nl, m->trace_dims[1] = groups, m->trace_dims[2] = K;
                DFX_HIP(dfx_env_set_max_dyn_smem(m->exact_fp32 ? (const void *)dfx_k_gru_seq_x32 : (const void *)dfx_k_gru_seq, DFX_GH_SMEM));
                DfxKScope ks(DFX_K_GRU_REC, G);
                if (m->exact_fp32) dfx_launch(dfx_k_gru_seq_x32, dim3((unsigned)(nl * groups)), dim3(DFX_GH_THREADS), DFX_GH_SMEM, G, S);
                else dfx_launch(dfx_k_gru_seq, dim3((unsigned)(nl * groups)), dim3(DFX_GH_THREADS), DFX_GH_SMEM, G, S);
                DFX_LAUNCH_CHECK();
            }
            if (nfollow) {   // the followers: right behind the recurrences, while the chip is still empty (each needs a CU's LDS)
                DfxPfArgs F;
                int f = 0;
                for (int l = 1; l < nl; ++l) {
                    if (!followed[l]) continue;
                    const GruW &g = l <= ndec ? m->dec_gru[l - 1] : m->df_gru[l - 1 - ndec];
                    const bool first = l == 1 || l == lfirst_df;
                    F.x[f] = first ? (l == 1 ? xb : xa2) : ws + w.py[l - 1], F.gi[f] = ws + w.pgi[l];
                    F.wf[f] = reinterpret_cast<const dfx_h8 *>(m->p(m->exact_fp32 ? g.wih_t : g.wih_h3)), F.bias[f] = m->p(g.bias_i), F.unscale[f] = g.wih_unscale;
                    F.yprog[f] = first ? embprog : yprog + (size_t)(l - 1) * DFX_SEQ_GMAX, F.giprog[f] = giprog + (size_t)l * DFX_SEQ_GMAX;
                    if (xtab) F.xme[f] = xword(1, l), F.xprod[f] = first ? xword(2, 0) : xword(0, l - 1), F.xcons[f] = xword(0, l);
                    ++f;
                }
                for (; f < DFX_PF_MAX; ++f) F.x[f] = nullptr, F.gi[f] = nullptr, F.wf[f] = nullptr, F.bias[f] = nullptr, F.unscale[f] = 1.f, F.yprog[f] = nullptr, F.giprog[f] = nullptr;
                F.xtag = xtag, F.xstat = xstat;
                if (xtab) {   // the followers choose their groups by XCD (dfx_xcd_claim): counters zeroed in front of the launches
                    F.xrec = xword(0, 0), F.xclaim = xstat + 8;
                }
                F.B = B, F.T = T, F.nf = nfollow, F.groups = groups, F.pbase = pbase, F.err = m->d_err, F.spin_limit = m->spin_limit;
                int lq = -1;
                for (int l = nl - 1; l >= 2 && lq < 0; --l)
                    if (followed[l]) lq = l;
                // the stream of a followed layer's projections has nothing else to carry (ps[1]: the emb follower); only layer 1 followed = no DF stack: its tail stream is free
                hipStream_t Fq = lq > 0 ? ln->ps[lq] : Dq;
                DFX_HIP(dfx_env_set_max_dyn_smem(m->exact_fp32 ? (const void *)dfx_k_proj_follow_x32 : (const void *)dfx_k_proj_follow, DFX_PH_SMEM));
                if ((rc = wait(ev_go, Fq))) return rc;
                DfxKScope ks(DFX_K_PROJ, Fq);
                if (m->exact_fp32) dfx_launch(dfx_k_proj_follow_x32, dim3((unsigned)(nfollow * groups)), dim3(512), DFX_PH_SMEM, Fq, F);
                else dfx_launch(dfx_k_proj_follow, dim3((unsigned)(nfollow * groups)), dim3(512), DFX_PH_SMEM, Fq, F);
                DFX_LAUNCH_CHECK();
            }
            if (followed[1]) {   // the follower of the encoder GRU: emb, lsnr and the inputs of both decoders' stacks per block of 8 steps
                const float *res = c.emb_gru_skip_enc == DFX_SKIP_IDENTITY ? emb_in : nullptr;
                const bool need_emb = c.emb_gru_skip != DFX_SKIP_NONE || (run_df && c.df_gru_skip == DFX_SKIP_IDENTITY);
                float *dfg_x = run_df ? xa2 : nullptr, *skp = fan_skp ? xdf : nullptr;
                DfxFanArgs EA = emb_fan_args(m, ws + w.py[0], res, need_emb ? embv : nullptr, xb, dfg_x, skp, lsnr);
                DfxFollowSync EY;
                if (xtab) {
                    EY.x.me = xword(2, 0), EY.x.prod = xword(0, 0), EY.x.cons = xword(1, 1), EY.x.cons2 = followed[lfirst_df] ? xword(1, lfirst_df) : nullptr;
                    EY.x.tag = xtag, EY.x.stat = xstat;
                    EY.xclaim = xstat + 8 + 8 * DFX_PF_MAX, EY.groups = groups;
                }
                EY.src = yprog, EY.dst = embprog, EY.pbase = pbase, EY.err = m->d_err, EY.spin_limit = m->spin_limit, EY.B = B, EY.T = T;
                hipStream_t Eq2 = ln->ps[1];
                if ((rc = wait(ev_go, Eq2))) return rc;
                {
                    DfxKScope ks(DFX_K_EMB_FAN, Eq2);
                    if (dfg_x && skp) dfx_launch((dfx_k_emb_follow<1, 2, 1>), dim3((unsigned)groups), dim3(512), 0, Eq2, EA, EY);
                    else if (dfg_x) dfx_launch((dfx_k_emb_follow<1, 2, 0>), dim3((unsigned)groups), dim3(512), 0, Eq2, EA, EY);
                    else dfx_launch((dfx_k_emb_follow<1, 0, 0>), dim3((unsigned)groups), dim3(512), 0, Eq2, EA, EY);
                    DFX_LAUNCH_CHECK();
                }
                if ((rc = signal(EV_EMB, Eq2))) return rc;   // the whole embedding exists (lsnr)
            }
            // the deferred part of the pathway conv: behind the front, beside the chain
            // (held back further, until the layer pipeline has filled — a flag wait on the last layer's first chunk in front of it — the fill is
            // 0.3 ms shorter and the layers then wait as long for the inputs of their next chunks: 13.20-13.23 vs 13.21 ms, not kept)
            // With followers the encoder layer's first projections go out in front of it: on the CUs the followers leave, a kernel that is enqueued
            // behind df_convp waits for it (exact mode: 6.4 ms for the first chunk's projection).
            const int convp_order = m->sw.convp_after_p0;
            const bool convp_after_p0 = convp_order >= 0 ? convp_order != 0 : nfollow > 0;
            auto convp_late = [&]() -> int {
                if (!(run_df && convp_split < T)) return DFX_OK;
                int r;
                if ((r = wait(ev_go, x2)) || (r = convp_range(convp_split, T, x2)) || (r = signal(EV_C0P, x2))) return r;
                return DFX_OK;
            };
            if (!convp_after_p0 && (rc = convp_late())) return rc;
            {
                // layer 0 (encoder GRU): its input xa is complete; one projection + flag per chunk
                // (stays two chunks ahead of the recurrence instead of flooding the chip with all K projections while the decoders'
                // first chunks are being prepared)
                for (int k = 0; k < K; ++k) {
                    const int p0_ahead = m->sw.p0_ahead;
                    if (k >= p0_ahead && (rc = launch_wait_ge(m, donep(0), groups, tgt(k - p0_ahead), Pq))) return rc;
                    if ((rc = proj_chunk(m->enc_gru[0], 0, k, xa, Pq))) return rc;
                    if (convp_after_p0 && k == (K < p0_ahead ? K : p0_ahead) - 1 && (rc = convp_late())) return rc;
                }
            }
            const int fpt = 64 / E > 0 ? 64 / E : 1;
            const size_t co_smem = ((size_t)fpt * E * (C + 1) + (size_t)fpt * E * 3 + 3 * C) * sizeof(float);
            // Every consumer has its own stream and walks the chunks in order: wait for its producer's flag, work, raise its own flag.
            //   ps[l]  (decoder layers): input of layer l, chunk k = linear_out / linear_in around the producer's y + the projection
            //   ts[0]  ERB tail (linear_out + the decoder's convolutions), ts[1] DF tail (skip + df_out), then the finishing kernels
            for (int l = 1; l < nl; ++l)
                if ((rc = wait(ev_go, ln->ps[l]))) return rc;
            // Host enqueue order: chunk-major (every stream still sees its own packets in chunk order).  (Consumers of equal pipeline depth
            // on one stream — 5 streams with 4 flag waits in flight instead of 8 with 7 — measured the same: 18.96 vs 18.80 ms.)
            const int lf = 1 + ndec;   // first DF layer
            seq_tail = Dq;
            // ---- ERB decoder layer j, chunk k
            auto prep_dec = [&](int j, int k) -> int {
                const int l = 1 + j;
                hipStream_t st = ln->ps[l];
                int r;
                if (followed[l]) return DFX_OK;
                if ((r = launch_wait_ge(m, donep(l - 1), groups, tgt(k), st))) return r;
                const float *xin = ws + w.py[l - 1];
                if (j == 0 && fan) {   // emb, lsnr and the inputs of both decoders' GRU stacks in one pass over the encoder GRU's chunk
                    if (publish && !m->exact_fp32) {
                        DfxPublish pub;
                        pub.cnt = pcnt + 8, pub.flag = embf, pub.value = tgt(k);
                        if ((r = emb_fan(ws + w.py[0], xb, Mk(k), st, rmk(k), &pub))) return r;
                    } else if ((r = emb_fan(ws + w.py[0], xb, Mk(k), st, rmk(k))) || (r = launch_flag_set(embf, tgt(k), st))) return r;
                    if (k == K - 1 && (r = signal(EV_EMB, st))) return r;
                    xin = xb;
                } else if (j == 0) {
                    if ((r = enc_out_skip(ws + w.py[0], Mk(k), st, rmk(k))) || (r = launch_flag_set(embf, tgt(k), st))) return r;
                    if (k == K - 1 && (r = signal(EV_EMB, st))) return r;   // the whole embedding exists (lsnr)
                    if ((r = launch_glin(m, m->dec_in, embv, DFX_ACT_RELU, nullptr, xb, Mk(k), st, rmk(k)))) return r;
                    xin = xb;
                }
                if ((r = proj_chunk(m->dec_gru[j], l, k, xin, st))) return r;
                return DFX_OK;
            };
            // ---- ERB tail, chunk k
            // (tails consume: they may take several hand-over chunks [k0, k1] in one launch — DFX_SEQ_TAIL_EVERY — when the chain is cut finer
            // than a decoder tail's launch is worth)
            auto erb_tail = [&](int k0, int k) -> int {
                const int64_t Rk = B * (tb(k + 1) - tb(k0));
                const DfxRowMap rm = DfxRowMap{T, tb(k + 1) - tb(k0), tb(k0)};
                int r;
                if ((r = launch_wait_ge(m, donep(ndec), groups, tgt(k), Eq))) return r;
                if (dfx_dev_skip() & 1) return DFX_OK;
                if ((r = dec_out_skip(ws + w.py[ndec], Rk, Eq, rm))) return r;
                if (fuse_tail) return launch_erb_tail<C>(m, demb, e3, e2, e1, e0r, mask, Rk, E, Eq, rm, feat_erb, T, featT, Lk);
                if ((r = launch_pw<C>(DFX_PW_MODE_DW3, m, m->ct3, demb, e3, d3, Rk, E / 4, E / 4, 1, Eq, rm))) return r;
                if ((r = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct2, d3, e2, d2, Rk, E / 4, E / 2, 2, Eq, rm))) return r;
                if (fuse_dec) return launch_erb_dec10<C>(m, d2, e1, e0, mask, Rk, E, Eq, rm);
                if ((r = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct1, d2, e1, d1, Rk, E / 2, E, 2, Eq, rm))) return r;
                DfxKScope ks(DFX_K_CONV_OUT, Eq);
                dfx_launch(dfx_k_conv_out<C>, dim3((unsigned)nn_grid(dfx_ceil_div(Rk, fpt), 8)), dim3(DFX_CO_THREADS), co_smem, Eq,
                           (const float *)d1, (const float *)e0, m->p(m->co_ska), m->p(m->co_skb), m->p(m->co_w), m->co_bias, mask, Rk, E,
                           fpt, rm);
                DFX_LAUNCH_CHECK();
                return DFX_OK;
            };
            // ---- DF decoder layer j, chunk k
            auto prep_df = [&](int j, int k) -> int {
                const int l = lf + j;
                hipStream_t st = ln->ps[l];
                int r;
                if (followed[l]) return DFX_OK;
                const float *xin = ws + w.py[l - 1];
                if (j == 0) {
                    if ((r = launch_wait_ge(m, embf, 1, tgt(k), st))) return r;
                    if (!fan && (r = launch_glin(m, m->dfg_in, embv, DFX_ACT_RELU, nullptr, xa2, Mk(k), st, rmk(k)))) return r;
                    xin = xa2;
                } else if ((r = launch_wait_ge(m, donep(l - 1), groups, tgt(k), st))) return r;
                if ((r = proj_chunk(m->df_gru[j], l, k, xin, st))) return r;
                return DFX_OK;
            };
            // ---- DF tail, chunk k
            auto df_tail = [&](int k0, int k) -> int {
                const int l = ndec + ndf;
                const int64_t Rk = B * (tb(k + 1) - tb(k0));
                const DfxRowMap rm = DfxRowMap{T, tb(k + 1) - tb(k0), tb(k0)};
                int r;
                if ((r = launch_wait_ge(m, donep(l), groups, tgt(k), Dq))) return r;
                if (dfx_dev_skip() & 2) return DFX_OK;
                if (c.df_gru_skip == DFX_SKIP_IDENTITY) {
                    if (k < K - 1) return DFX_OK;   // the identity-skip form is not chunked: one add + df_out over all frames at the end
                    {
                        DfxKScope ks(DFX_K_ADD, Dq);
                        dfx_launch(dfx_k_add, dim3((unsigned)nn_grid(dfx_ceil_div(R * 256, 256), 16)), dim3(256), 0, Dq,
                                   (const float *)(ws + w.py[l]), (const float *)embv, xdf, R * 256);
                    }
                    DFX_LAUNCH_CHECK();
                    return launch_ggemm(xdf, m->df_out.G * m->df_out.Kg, m->p(m->df_out.w), m->df_out.G, m->df_out.Kg, m->df_out.Ng,
                                        nullptr, DFX_ACT_TANH, c0p, coefs, m->df_out.G * m->df_out.Ng, R, Dq, NO, Fd, T);
                }
                const float *cfeat = ws + w.py[l], *cfeat2 = nullptr;
                if (fan_skp) {
                    cfeat2 = xdf;
                } else if (c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR) {
                    if ((r = launch_glin(m, m->df_skip, embv, DFX_ACT_NONE, ws + w.py[l], xdf, Rk, Dq, rm))) return r;
                    cfeat = xdf;
                }
                return df_out_rows(cfeat, cfeat2, Rk, Dq, rm);
            };
            if (run_df && (rc = wait(EV_C0P, Dq))) return rc;
            for (int k = 0; k < K; ++k) {
                for (int j = 0; j < (ndec > ndf ? ndec : ndf); ++j) {
                    if (j < ndec && (rc = prep_dec(j, k))) return rc;
                    if (j < ndf && (rc = prep_df(j, k))) return rc;
                }
                const int tail_every = m->sw.tail_every;
                if ((k + 1) % tail_every == 0 || k == K - 1) {
                    const int k0 = k - (k % tail_every);
                    if ((rc = erb_tail(k0, k))) return rc;
                }
                // the DF tail waits for ALL of df_convp, which — deferred under the phase, beside followers — ends with the phase: its launches then run
                // behind the chain anyway, and few large ones are through sooner than twelve small ones (DFX_SEQ_DFTAIL_EVERY=n chunks per launch)
                const int dft_env = m->sw.dftail_every;
                // (12.47-12.52 ms per step at 4 chunks per launch against 12.69-12.83 at 1, same box; 6: 12.49-12.57)
                const int dft_every = dft_env > 0 ? dft_env : (nfollow > 0 && convp_split < T && tail_every < 4 ? 4 : tail_every);
                if (run_df && ((k + 1) % dft_every == 0 || k == K - 1)) {
                    const int k0 = k - (k % dft_every);
                    if ((rc = df_tail(k0, k))) return rc;
                }
            }
            if ((rc = signal(EV_MASK, Eq))) return rc;
            // ---- lsnr on the caller's stream once the whole embedding exists (:163-165,184); dfx_k_emb_fan has written it per chunk
            if ((rc = wait(EV_EMB, s))) return rc;
            if (!fan) {
                DfxKScope ks(DFX_K_LSNR, s);
                dfx_launch(dfx_k_lsnr, dim3((unsigned)dfx_ceil_div(R * 64, 256)), dim3(256), 0, s, (const float *)embv, m->p(m->lsnr_w),
                           m->lsnr_b, (float)(c.lsnr_max - c.lsnr_min), (float)c.lsnr_min, lsnr, R, emb);
                DFX_LAUNCH_CHECK();
            }
            // the persistent launch and the layer-0 projections end before the decoders' last chunks do; join their streams all the same
            // (on the caller's stream, which has nothing else to do until the finishing kernels are through)
            DFX_HIP(hipEventRecord(ln->gev[0][0], G));
            DFX_HIP(hipStreamWaitEvent(s, ln->gev[0][0], 0));
            for (int l = 0; l < nl; ++l) {
                DFX_HIP(hipEventRecord(ln->pev[l][0], ln->ps[l]));
                DFX_HIP(hipStreamWaitEvent(s, ln->pev[l][0], 0));
            }
        } else {
        auto gru_chunk = [&](const GruW &g, int l, int k, hipStream_t st) -> int {
            float *hl = ws + w.ph[l];
            return launch_gru_h3(m, g, ws + w.pgi[l], ws + w.py[l], k == 0 ? nullptr : hl, hl, B, T, tb(k), tb(k + 1), st, l);
        };
        if ((rc = signal(EV_XA, s))) return rc;
        for (int l = 0; l < nl; ++l) {
            if (l > 0 && (rc = wait(EV_XA, ln->gs[l]))) return rc;
            if ((rc = wait(EV_XA, ln->ps[l]))) return rc;
        }
        if ((rc = wait(EV_XA, ln->ts[0])) || (rc = wait(EV_XA, ln->ts[1]))) return rc;
        // ---- layer 0 = encoder GRU: prep on ps[0] (x = xa is complete), recurrence on s
        for (int k = 0; k < K; ++k) {
            if ((rc = proj_chunk(m->enc_gru[0], 0, k, xa, ln->ps[0])) || (rc = esig(ln->pev[0][k], ln->ps[0]))) return rc;
        }
        for (int k = 0; k < K; ++k) {
            if ((rc = ewait(ln->pev[0][k], s)) || (rc = gru_chunk(m->enc_gru[0], 0, k, s)) || (rc = esig(ln->gev[0][k], s))) return rc;
        }
        // ---- ERB decoder stack (layers 1..ndec); its first prep stream also produces emb = relu(linear_out(y_enc)) per chunk
        for (int j = 0; j < ndec; ++j) {
            const int l = 1 + j;
            hipStream_t pst = ln->ps[l], gst = ln->gs[l];
            for (int k = 0; k < K; ++k) {
                if ((rc = ewait(ln->gev[l - 1][k], pst))) return rc;
                const float *xin = ws + w.py[l - 1];
                if (j == 0 && fan) {
                    if ((rc = emb_fan(ws + w.py[0], xb, Mk(k), pst, rmk(k)))) return rc;
                    if ((rc = esig(ln->eev[k], pst))) return rc;
                    xin = xb;
                } else if (j == 0) {
                    if ((rc = enc_out_skip(ws + w.py[0], Mk(k), pst, rmk(k)))) return rc;
                    if ((rc = esig(ln->eev[k], pst))) return rc;  // emb chunk k exists (the DF stack waits for it)
                    if ((rc = launch_glin(m, m->dec_in, embv, DFX_ACT_RELU, nullptr, xb, Mk(k), pst, rmk(k)))) return rc;
                    xin = xb;
                }
                if ((rc = proj_chunk(m->dec_gru[j], l, k, xin, pst)) || (rc = esig(ln->pev[l][k], pst))) return rc;
            }
            for (int k = 0; k < K; ++k) {
                if ((rc = ewait(ln->pev[l][k], gst)) || (rc = gru_chunk(m->dec_gru[j], l, k, gst)) || (rc = esig(ln->gev[l][k], gst))) return rc;
            }
        }
        {   // ERB tail: per time chunk linear_out and the convolutional half of the decoder (:250-253; all of it is per frame),
            // so it runs beside the GRU chain (which leaves most CUs idle) instead of after it
            hipStream_t st = ln->ts[0];
            const int fpt = 64 / E > 0 ? 64 / E : 1;
            const size_t smem = ((size_t)fpt * E * (C + 1) + (size_t)fpt * E * 3 + 3 * C) * sizeof(float);
            for (int k = 0; k < K; ++k) {
                const int64_t Rk = Mk(k);
                const DfxRowMap rm = rmk(k);
                if ((rc = ewait(ln->gev[ndec][k], st))) return rc;
                if (dfx_dev_skip() & 1) continue;
                if ((rc = dec_out_skip(ws + w.py[ndec], Rk, st, rm))) return rc;
                if (fuse_tail) {
                    if ((rc = launch_erb_tail<C>(m, demb, e3, e2, e1, e0r, mask, Rk, E, st, rm, feat_erb, T, featT, Lk))) return rc;
                } else if ((rc = launch_pw<C>(DFX_PW_MODE_DW3, m, m->ct3, demb, e3, d3, Rk, E / 4, E / 4, 1, st, rm)) ||
                           (rc = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct2, d3, e2, d2, Rk, E / 4, E / 2, 2, st, rm))) {
                    return rc;
                } else if (fuse_dec) {
                    if ((rc = launch_erb_dec10<C>(m, d2, e1, e0, mask, Rk, E, st, rm))) return rc;
                } else {
                    if ((rc = launch_pw<C>(DFX_PW_MODE_DWT3, m, m->ct1, d2, e1, d1, Rk, E / 2, E, 2, st, rm))) return rc;
                    DfxKScope ks(DFX_K_CONV_OUT, st);
                    dfx_launch(dfx_k_conv_out<C>, dim3((unsigned)nn_grid(dfx_ceil_div(Rk, fpt), 8)), dim3(DFX_CO_THREADS), smem, st,
                               (const float *)d1, (const float *)e0, m->p(m->co_ska), m->p(m->co_skb), m->p(m->co_w), m->co_bias, mask,
                               Rk, E, fpt, rm);
                }
                DFX_LAUNCH_CHECK();
            }
            if ((rc = signal(EV_MASK, st))) return rc;
        }
        // ---- DF decoder stack (layers 1+ndec ..)
        for (int j = 0; j < ndf; ++j) {
            const int l = 1 + ndec + j;
            hipStream_t pst = ln->ps[l], gst = ln->gs[l];
            for (int k = 0; k < K; ++k) {
                const float *xin = ws + w.py[l - 1];
                if (j == 0) {
                    if ((rc = ewait(ln->eev[k], pst))) return rc;
                    if (!fan && (rc = launch_glin(m, m->dfg_in, embv, DFX_ACT_RELU, nullptr, xa2, Mk(k), pst, rmk(k)))) return rc;
                    xin = xa2;
                } else if ((rc = ewait(ln->gev[l - 1][k], pst))) return rc;
                if ((rc = proj_chunk(m->df_gru[j], l, k, xin, pst)) || (rc = esig(ln->pev[l][k], pst))) return rc;
            }
            for (int k = 0; k < K; ++k) {
                if ((rc = ewait(ln->pev[l][k], gst)) || (rc = gru_chunk(m->df_gru[j], l, k, gst)) || (rc = esig(ln->gev[l][k], gst))) return rc;
            }
        }
        if (run_df) {   // DF tail: skip + df_out (+ c0p) per chunk (:324-330)
            hipStream_t st = ln->ts[1];
            const int l = ndec + ndf;
            if ((rc = wait(EV_C0P, st))) return rc;
            if (c.df_gru_skip != DFX_SKIP_IDENTITY) {
                for (int k = 0; k < K; ++k) {
                    if ((rc = ewait(ln->gev[l][k], st))) return rc;
                    if (dfx_dev_skip() & 2) continue;
                    const float *cfeat = ws + w.py[l], *cfeat2 = nullptr;
                    if (fan_skp) {
                        cfeat2 = xdf;
                    } else if (c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR) {
                        if ((rc = launch_glin(m, m->df_skip, embv, DFX_ACT_NONE, ws + w.py[l], xdf, Mk(k), st, rmk(k)))) return rc;
                        cfeat = xdf;
                    }
                    if ((rc = df_out_rows(cfeat, cfeat2, Mk(k), st, rmk(k)))) return rc;
                }
            } else {
                if ((rc = ewait(ln->gev[l][K - 1], st))) return rc;
                {
                    DfxKScope ks(DFX_K_ADD, st);
                    dfx_launch(dfx_k_add, dim3((unsigned)nn_grid(dfx_ceil_div(R * 256, 256), 16)), dim3(256), 0, st,
                               (const float *)(ws + w.py[l]), (const float *)embv, xdf, R * 256);
                }
                DFX_LAUNCH_CHECK();
                if ((rc = launch_ggemm(xdf, m->df_out.G * m->df_out.Kg, m->p(m->df_out.w), m->df_out.G, m->df_out.Kg, m->df_out.Ng,
                                       nullptr, DFX_ACT_TANH, c0p, coefs, m->df_out.G * m->df_out.Ng, R, st, NO, Fd, T)))
                    return rc;
            }
            if ((rc = signal(EV_COEFS, st))) return rc;
        }
        // ---- lsnr on the caller's stream once the whole embedding exists (:163-165,184); dfx_k_emb_fan has written it per chunk
        if ((rc = ewait(ln->eev[K - 1], s))) return rc;
        if (!fan) {
            DfxKScope ks(DFX_K_LSNR, s);
            dfx_launch(dfx_k_lsnr, dim3((unsigned)dfx_ceil_div(R * 64, 256)), dim3(256), 0, s, (const float *)embv, m->p(m->lsnr_w),
                       m->lsnr_b, (float)(c.lsnr_max - c.lsnr_min), (float)c.lsnr_min, lsnr, R, emb);
            DFX_LAUNCH_CHECK();
        }
        }   // !use_seq
        // The finishing kernels run on the DF tail's stream, directly behind its last df_out launch: a kernel that starts behind a
        // cross-queue join starts after ~45 us of idle chip and was measured 17 % slower for its whole duration (0.59 vs 0.50 ms
        // for the deep filter in the rocprofv3 trace, same data, nothing overlapping); behind a kernel of its own queue the gap is
        // 6 us.  The ERB tail's masks are normally complete by then (its event is already signalled).
        fin_s = seq_tail ? seq_tail : ln->ts[1];
        if ((rc = wait(EV_MASK, fin_s))) return rc;
    }
    if (run_df && fin_s == s && (rc = wait(EV_COEFS, s))) return rc;
    if (!run_df && coefs_out) DFX_HIP(hipMemsetAsync(coefs_out, 0, (size_t)R * Fd * NO * sizeof(float), fin_s));  // DfNet(run_df=False) has no coefficients
    // ---- Mask + MF.DF + combine + post filter + atten_lim (:426-454, enhance.py:238-240)
    if (sc) {  // spec has sc->spec_T frames per clip, coefficients / gains T; the n enhanced frames are stored compactly
        const float beta = sc->pf_beta >= 0.f ? sc->pf_beta : (c.mask_pf ? c.pf_beta : 0.f);
        if (sc->channels > 1 && sc->reduce_mask != 0) {
            dfx_launch(dfx_k_mask_reduce, dim3((unsigned)nn_grid(dfx_ceil_div(Rn * E / sc->channels, 256), 8)), dim3(256), 0, s, mask, B, T, t_begin,
                       E, sc->channels, sc->reduce_mask);
            DFX_LAUNCH_CHECK();
        }
        if (gate) {
            dfx_launch(dfx_k_gate_edit, dim3((unsigned)B), dim3(128), 0, s, (const unsigned char *)gate->flags, mask, coefs,
                       (const unsigned char *)bands->d_bin2band, B, T, E, Fd, O, O - 1 - c.df_lookahead);
            DFX_LAUNCH_CHECK();
        }
        // the real-time runtime filters with libDF's own post_filter (lib.rs:446-471 via tract.rs:603-610): Rust arithmetic and its
        // chunks_exact(4) walk over the stream's flattened [channels * F] frame
        return dfx_launch_df_apply(spec, coefs, DFX_COEF_BOTF, mask, bands, B, sc->spec_T, c.fft_size / 2 + 1, run_df ? Fd : 0, O, c.df_lookahead, beta,
                                   atten_lim, sc->out, s, t_begin, T, T, sc->out_T, sc->out_toff, sc->spec_stride, sc->spec_stride,
                                   sc->channels > 0 ? sc->channels : 1);
    }
    // enhance(): the deep filter + gains are applied on the way into the inverse transform (dfx_k_synthesis_rows): spec_e never exists.
    // DFX_FUSE_DFA=0: dfx_k_df_apply_rows -> spec_e -> dfx_k_synthesis (the stand-alone deep-filter kernel stays the API of
    // dfx_model_forward / dfx_df_apply and the roofline kernel of bench.py)
    if (fin && m->fuse_dfa && dfx_synthesis_rows_ok(fin->st, true, O, run_df ? Fd : 0, E) && bands == fin->st->bands) {
        if ((rc = dfx_launch_synthesis_rows(fin->st, spec, sstride, run_df ? coefs : nullptr, run_df ? Fd : 0, O, c.df_lookahead, mask,
                                            c.mask_pf ? c.pf_beta : 0.f, atten_lim, B, T, fin->y, fin->out_stride, fin->out_skip, fin->out_len, fin_s, fin->out_i16, m->d_err, m->d_sync ? m->d_sync + 14 : nullptr)))   // (d_sync[14]: a spare word of the flag block)
            return rc;
    } else {
        if (dfx_dev_stage(10) && (rc = dfx_launch_df_apply(spec, coefs, DFX_COEF_BOTF, mask, bands, B, T, c.fft_size / 2 + 1, run_df ? Fd : 0, O, c.df_lookahead,
                                      c.mask_pf ? c.pf_beta : 0.f, atten_lim, spec_e, fin_s, 0, -1, -1, -1, 0, sstride, sstride)))
            return rc;
        if (fin && (rc = dfx_launch_synthesis(fin->st, spec_e, B, T, nullptr, nullptr, fin->y, fin->out_stride, fin->out_skip, fin->out_len,
                                              fin_s, 0, -1, sstride, fin->out_i16)))
            return rc;
    }
    if (fin_s != s && ((rc = signal(EV_FIN, fin_s)) || (rc = wait(EV_FIN, s)))) return rc;
    if (use_seq && dfx_ticket().fd >= 0) {   // give the device's ticket back when this pass is through (side stream: s does not wait for the callback)
        hipStream_t ts = ln->main ? ln->main : s;
        if (ts != s) {
            DFX_HIP(hipEventRecord(ln->ev[EV_TICKET], s));
            DFX_HIP(hipStreamWaitEvent(ts, ln->ev[EV_TICKET], 0));
        }
        DFX_HIP(hipLaunchHostFunc(ts, dfx_ticket_release_cb, nullptr));
        ticket_guard.armed = false;
    }
    return DFX_OK;
}

static int model_forward_lane(const dfx_model *m, const dfx_bands *bands, const float *spec, const float *feat_erb,
                              const float *feat_spec, int64_t B, int64_t T, float atten_lim, float *spec_e, float *mask,
                              float *lsnr, float *df_coefs, void *workspace, int64_t workspace_bytes, void *stream,
                              const DfxLane *ln, bool signal_front, const DfxFinish *fin = nullptr) {
    if (!m || !bands || B < 0 || T < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: bad arguments");
    if (bands->nb != m->cfg.nb_erb || bands->F != m->cfg.fft_size / 2 + 1)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: band table does not match the model (nb_erb / fft_size)");
    if (atten_lim < 0.f || atten_lim >= 1.f) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: atten_lim must be in [0,1)");
    if (int rc = dfx_require_device()) return rc;
    if (B == 0 || T == 0) return DFX_OK;
    if (!spec || !feat_erb || !feat_spec || !spec_e || !workspace) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: null buffer");
    int64_t need = 0;
    dfx_model_workspace_bytes(m, B, T, &need);
    if (workspace_bytes < need) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)need);
    if (((uintptr_t)spec & 15) || ((uintptr_t)spec_e & 15) || ((uintptr_t)feat_erb & 15) || ((uintptr_t)feat_spec & 15))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: buffers must be 16-byte aligned");
    float *ws = reinterpret_cast<float *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t s = dfx_stream(stream);
    switch (m->cfg.conv_ch) {
        case 16: return forward_impl<16>(m, bands, spec, feat_erb, feat_spec, B, T, atten_lim, spec_e, mask, lsnr, df_coefs, ws, s, ln, signal_front, fin);
        case 32: return forward_impl<32>(m, bands, spec, feat_erb, feat_spec, B, T, atten_lim, spec_e, mask, lsnr, df_coefs, ws, s, ln, signal_front, fin);
        case 64: return forward_impl<64>(m, bands, spec, feat_erb, feat_spec, B, T, atten_lim, spec_e, mask, lsnr, df_coefs, ws, s, ln, signal_front, fin);
    }
    DFX_FAIL(DFX_ERR_UNSUPPORTED, "conv_ch");
}

// Whose turn it is.  The internal streams and events belong to the process (DfxLaneSet), so the enqueue of every entry point is serialised by
// one lock (a stream's event pairs must not interleave with another thread's), and the multi-stream passes of DIFFERENT handles also take
// turns on the device: a pass starts when the other handle's last pass is through (a stream wait on its event; passes of one handle are ordered
// by their caller's stream and ev_pass as before).  Why on the device too: a persistent GRU phase needs every one of its ~160 workgroups
// resident (each owns a CU) — two of them at once are 320 on 256 CUs, each can hold the CUs the other's missing workgroups wait for, and both
// end in flag-wait timeouts (seen with two handles on the persistent form, profiles/r06_two_handles.log).  Overlap buys nothing either: one
// pass fills the chip.  (Round 5 blamed the overlap for wrong samples; that was the packed-fp32 fault, measurements R6.1, and is gone with it.)
// The frame-by-frame streaming calls only take the lock: they start no persistent phase.
struct PassGate {
    const dfx_model *owner = nullptr;
    hipEvent_t done = nullptr;   // the owner's ev_gate, recorded behind its last pass
};
static PassGate &pass_gate() {
    static PassGate g;
    return g;
}
struct DfxTurn {
    std::unique_lock<std::mutex> lk;
    const dfx_model *m;
    hipStream_t s;
    bool big, recorded = false;
    DfxTurn(const dfx_model *m_, hipStream_t s_, bool big_pass) : m(m_), s(s_), big(big_pass && m_->concurrent && m_->ev_gate) {
        if (!m->have_streams) return;
        lk = std::unique_lock<std::mutex>(dfx_enqueue_mu());
        PassGate &g = pass_gate();
        if (big && g.owner && g.owner != m && g.done) (void)hipStreamWaitEvent(s, g.done, 0);
    }
    void passed() {   // the pass is enqueued and joined into s
        if (!big || !lk.owns_lock() || recorded) return;
        recorded = true;
        if (hipEventRecord(m->ev_gate, s) == hipSuccess) pass_gate().owner = m, pass_gate().done = m->ev_gate;
    }
    ~DfxTurn() { passed(); }   // also behind a pass that failed half-way: whatever it did enqueue is ordered in front of the next handle's pass
};
static void pass_gate_forget(const dfx_model *m) {
    std::lock_guard<std::mutex> lk(dfx_enqueue_mu());
    PassGate &g = pass_gate();
    if (g.owner == m) {
        if (g.done) (void)hipEventSynchronize(g.done);
        g.owner = nullptr, g.done = nullptr;
    }
}
// Enqueue throttle of the multi-stream pass (see dfx_model::ev_pass): big passes only — a small pass is over before the host has
// enqueued the next one, and holding the host back would serialise its launch overhead with the device's work.
static int pass_begin(const dfx_model *m, int64_t frames) {
    if (m->pass_pending && m->ev_pass && !m->enqueue_ahead && m->concurrent && frames >= DFX_THROTTLE_MIN_FRAMES) DFX_HIP(hipEventSynchronize(m->ev_pass));
    m->pass_pending = false;
    // the previous pass has drained (big passes) or may have (small ones): a fault it raised is reported now, before new work is enqueued
    return model_poll(m);
}
static int pass_end(const dfx_model *m, int64_t frames, hipStream_t s) {
    if (m->ev_pass && !m->enqueue_ahead && m->concurrent && frames >= DFX_THROTTLE_MIN_FRAMES) {
        DFX_HIP(hipEventRecord(m->ev_pass, s));
        m->pass_pending = true;
    }
    if (m->check_every_pass) {   // DFX_CHECK_EVERY_PASS=1: the call waits for its own pass and reports its own faults
        DFX_HIP(hipStreamSynchronize(s));
        m->pass_pending = false;
        return model_poll(m);
    }
    return DFX_OK;
}

extern "C" int dfx_model_forward(const dfx_model *m, const dfx_bands *bands, const float *spec, const float *feat_erb,
                                 const float *feat_spec, int64_t B, int64_t T, float atten_lim, float *spec_e,
                                 float *mask, float *lsnr, float *df_coefs, void *workspace, int64_t workspace_bytes,
                                 void *stream) {
    if (!m) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_forward: bad arguments");
    if (int rc = pass_begin(m, B * T)) return rc;
    DfxTurn turn(m, dfx_stream(stream), true);
    if (int rc = model_forward_lane(m, bands, spec, feat_erb, feat_spec, B, T, atten_lim, spec_e, mask, lsnr, df_coefs, workspace,
                                    workspace_bytes, stream, &m->lanes[0], false))
        return rc;
    turn.passed();
    return pass_end(m, B * T, dfx_stream(stream));
}

// ------------------------------------------------------------------------------------------------ enhance()
// row stride (complex elements) of enhance()'s spec / spec_e buffers: F rounded up to a multiple of 8 = rows that start on a
// 64-byte boundary (F = 481 -> 488): every access of the row-streaming deep-filter kernel is then a 16-byte access inside whole
// 64-byte sectors.  Measured (tools/dev/dfa_bench.hip, profiles/r02_dfa_bench.log): stride 481 (flat-stream kernel) 4.9 TB/s,
// 482 -> 6.0, 488 / 496 / 512 -> 6.2 TB/s.
static inline int64_t enh_spec_stride(const dfx_state *st) {
    const int64_t F = (int64_t)st->N / 2 + 1;
    return (F + 7) & ~(int64_t)7;
}
namespace {
struct EnhWs {
    size_t spec, spec_e, feat_erb, feat_spec, model, total;  // bytes
};
EnhWs plan_enh(const dfx_model *m, const dfx_state *st, int64_t B, int64_t T, int pad) {
    EnhWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const int64_t Tp = pad ? T + st->N : T, Tf = Tp / st->hop, F = enh_spec_stride(st);
    w.spec = take((size_t)B * Tf * F * 8);
    w.spec_e = take((size_t)B * Tf * F * 8);
    w.feat_erb = take((size_t)B * Tf * m->cfg.nb_erb * 4);
    w.feat_spec = take((size_t)B * Tf * m->cfg.nb_df * 8);
    int64_t mb = 0;
    dfx_model_workspace_bytes(m, B, Tf, &mb);
    w.model = take((size_t)mb);
    w.total = off + 256;
    return w;
}
}  // namespace

// Streaming history ring of one per-frame quantity (row floats per frame): work[b] = [hist_in[b] (h frames) ; new[b] (n frames, the
// first `skip` of them replaced by zeros)], and hist_out[b] = the last h frames of that window (hist_in != hist_out).
__global__ void dfx_k_ring_step(const float *hist_in, const float *nw, float *work, float *hist_out, int64_t B, int64_t h, int64_t n,
                                int64_t row, int64_t skip) {
    const int64_t wl = (h + n) * row, total = B * wl;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / wl, j = i - b * wl, fr = j / row;
        float v;
        if (fr < h) v = hist_in[b * h * row + j];
        else v = (fr - h < skip) ? 0.f : nw[b * n * row + (j - h * row)];
        work[i] = v;
        if (fr >= n) hist_out[b * h * row + (j - n * row)] = v;
    }
}

// copy rows with zero padding / offset: dst[b, i] = (i + src_off < src_len) ? src[b, i + src_off] : 0
__global__ void dfx_k_copy_rows(const float *src, int64_t src_stride, int64_t src_len, int64_t src_off, float *dst,
                                int64_t dst_stride, int64_t dst_len, int64_t B) {
    const int64_t n = B * dst_len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / dst_len, j = i - b * dst_len;
        const int64_t sj = j + src_off;
        dst[b * dst_stride + j] = sj < src_len ? src[b * src_stride + sj] : 0.f;
    }
}

__global__ void dfx_k_fill_rows(float *dst, int64_t dst_stride, int64_t len, int64_t B, float v) {
    const int64_t n = B * len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / len;
        dst[b * dst_stride + (i - b * len)] = v;
    }
}

// ------------------------------------------------------------------------------------------------ streaming (dfx_stream_*)
// Frame loop of DfTract::process (tract.rs:509-642) for many lockstep streams: every call runs the batch kernels on a window of
// H history + n new frames per stream (DfxStreamCtx), with all recurrent state carried in the handle.
struct dfx_stream_state {
    const dfx_model *m = nullptr;
    const dfx_state *st = nullptr;
    int64_t B = 0;
    int nmax = 0, H = 0, L = 0, layers = 0;
    int64_t frames = 0;       // hops consumed since the last reset
    float lim = 0.f;          // linear attenuation limit: 0 = off, 1 = bypass (tract.rs:387-398)
    float pf_beta = -1.f;     // < 0: the model's setting
    unsigned char *buf = nullptr;
    size_t bytes = 0;
    // byte offsets into buf
    size_t ana_mem[2], syn_mem[2], erb_state, unit_state, hist_fe[2], hist_fs[2], hist_spec[2], new_spec, new_fe, new_fs, work_fe, work_fs,
        work_spec, out_spec, h_state, h_state2, lsnr, model_ws;
    int hflip = 0;            // which of h_state / h_state2 holds the GRU states (the one-step kernel writes the other one: dfx_k_gru_step_h3)
    size_t c0ring = 0;        // pending sums of df_convp's next kt - 1 outputs (dfx_k_df_convp_step); c0ring_bytes == 0: not available
    size_t c0ring_bytes = 0;
    bool c0ring_ok = true;    // the sums are current (all zeros after a reset; stale after a pass that did not go through the step kernel)
    int64_t model_ws_bytes = 0;
    int flip = 0;             // which of the double-buffered STFT memories is current
    // The rolling spectra of an ungated handle live in a LINEAR buffer [B, lin_cap, F] through which the window [lin_pos, lin_pos + Hs + n)
    // slides: a call appends its n new frames and the deep filter reads the window in place (clip stride lin_cap frames); only when the
    // window reaches the end are its last Hs frames moved back to the front (once per lin_cap - Hs - n hops).  The ring form below
    // (hist_spec -> work_spec, dfx_k_ring_step) rewrites the whole window on every call — at 4096 streams 95 us of a 720 us hop — and stays
    // for gated handles (a frozen stream's spectra must not move) and graph replay (fixed addresses).  lin_owns: which form holds the state.
    int64_t Fp = 0;           // bins per spectrum row of the handle's buffers: F rounded up to a multiple of 8 (64-byte rows: the row-streaming deep filter takes them)
    size_t spec_lin = 0;
    int64_t lin_cap = 0, lin_pos = 0;
    bool lin_owns = false;
    // the encoder's feature windows in the same form (stream_body): [B, feat_cap, E] and [B, feat_cap, Fd, 2] at the same lin_pos
    size_t fe_lin = 0, fs_lin = 0;
    int64_t feat_cap = 0;
    bool feat_owns = false;
    // per-stream stage gating (dfx_stream_set_gating; DfTract::process, tract.rs:509-616,658-672): off by default
    bool gated = false;
    int channels = 1, reduce_mask = 2;    // multi-channel streams: ch consecutive rows per stream; ReduceMask::MEAN is the reference default
    float thr[3] = {-10.f, 30.f, 20.f};   // RuntimeParams::default_with_ch (tract.rs:177-189)
    unsigned char *gate_buf = nullptr;    // own allocation, made when gating is first switched on
    size_t g_flags = 0, g_counter = 0, g_sh_erb = 0, g_sh_unit = 0, g_sh_h = 0, g_c0_win = 0, g_mask = 0, g_coefs = 0, gate_bytes = 0;
    size_t g_pend2 = 0, g_par = 0, g_cnt = 0;   // pending-sum form of the gated df_convp (g_pend2_ok; then g_c0_win is not allocated)
    bool g_pend2_ok = false;
    // (Replaying a steady-state call from a hipGraph was built in round 1 and removed in round 4: on ROCm 7.2 the replay of the hop's kernel nodes
    // took 2.0-2.2 ms per call where plain launches take 0.4.)
};

static int stream_copy_rows(const float *src, int64_t src_stride, int64_t src_len, int64_t src_off, float *dst, int64_t dst_stride,
                            int64_t dst_len, int64_t B, hipStream_t s) {
    if (B <= 0 || dst_len <= 0) return DFX_OK;
    DfxKScope ks(DFX_K_COPY_ROWS, s);
    dfx_launch(dfx_k_copy_rows, dim3((unsigned)nn_grid(dfx_ceil_div(B * dst_len, 256), 16)), dim3(256), 0, s, src, src_stride, src_len,
               src_off, dst, dst_stride, dst_len, B);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_stream_create(const dfx_model *m, const dfx_state *st, int64_t streams, int max_frames, dfx_stream_state **out) {
    if (!m || !st || !out || streams <= 0 || max_frames <= 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_create: bad arguments");
    const dfx_model_cfg &c = m->cfg;
    if (st->N != c.fft_size || st->hop != c.hop_size || st->nb != c.nb_erb)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_create: the DF state does not match the model (fft/hop/nb_erb)");
    if (c.conv_lookahead != c.df_lookahead)
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_stream_create: conv_lookahead != df_lookahead is not supported by the streaming path");
    if (!m->fuse_c0 || !m->fuse_erb || m->exact_fp32 || !m->run_df)
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_stream_create: streaming needs the default (fused, fp16-split, DF stage on) engine configuration");
    if (int rc = dfx_require_device()) return rc;
    dfx_stream_state *s = new dfx_stream_state();
    s->m = m;
    s->st = st;
    s->B = streams;
    s->nmax = max_frames;
    s->L = c.df_lookahead;
    // history in front of the new frames: 2 frames for the 3-tap input convolutions + kt-1 frames of (recomputed) c0 for df_convp
    const int hist_conv = 2 + (c.df_pathway_kernel_size_t - 1), hist_df = c.df_order - 1 - c.df_lookahead;
    s->H = hist_conv > hist_df ? hist_conv : hist_df;
    s->layers = c.emb_num_layers + (c.emb_num_layers - 1) + c.df_num_layers;
    s->layers = (int)(m->enc_gru.size() + m->dec_gru.size() + m->df_gru.size());
    const int64_t B = streams, n = max_frames, H = s->H, Hs = s->H + s->L, F = (st->N / 2 + 1 + 7) & ~(int64_t)7 /* padded rows */, E = c.nb_erb, Fd = c.nb_df,
                  ML = st->N - st->hop;
    s->Fp = F;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    for (int i = 0; i < 2; ++i) s->ana_mem[i] = take((size_t)B * ML * 4), s->syn_mem[i] = take((size_t)B * ML * 4);
    s->erb_state = take((size_t)B * E * 4);
    s->unit_state = take((size_t)B * Fd * 4);
    for (int i = 0; i < 2; ++i) {
        s->hist_fe[i] = take((size_t)B * H * E * 4);
        s->hist_fs[i] = take((size_t)B * H * Fd * 8);
        s->hist_spec[i] = take((size_t)B * Hs * F * 8);
    }
    s->new_spec = take((size_t)B * n * F * 8);
    s->new_fe = take((size_t)B * n * E * 4);
    s->new_fs = take((size_t)B * n * Fd * 8);
    s->work_fe = take((size_t)B * (H + n) * E * 4);
    s->work_fs = take((size_t)B * (H + n) * Fd * 8);
    s->work_spec = take((size_t)B * (Hs + n) * F * 8);
    s->out_spec = take((size_t)B * n * F * 8);
    {   // linear rolling-spectra buffer: slack of at least one window (so that the move back to the front never overlaps), at most ~1 GB
        static const int lin_env = [] { const char *e = getenv("DFX_STREAM_LINEAR"); return e ? atoi(e) : 1; }();
        int64_t slack = lin_env > 1 ? lin_env : 32;   // (DFX_STREAM_LINEAR=0: ring form only; = n > 1: slack of n frames, tests)
        while (slack > Hs + n && (size_t)B * (Hs + n + slack) * F * 8 > ((size_t)1 << 30)) slack /= 2;
        if (slack < Hs + n) slack = Hs + n;
        if (lin_env && (size_t)B * (Hs + n + slack) * F * 8 <= ((size_t)3 << 29)) {
            s->lin_cap = Hs + n + slack;
            s->spec_lin = take((size_t)B * s->lin_cap * F * 8);
            static const bool feat_env = [] { const char *e = getenv("DFX_STREAM_LINEAR_FEAT"); return !(e && e[0] == '0'); }();
            if (feat_env) {
                s->feat_cap = H + n + slack;   // the same slack: the three windows reach the end in the same call
                s->fe_lin = take((size_t)B * s->feat_cap * E * 4);
                s->fs_lin = take((size_t)B * s->feat_cap * Fd * 8);
            }
        }
    }
    s->h_state = take((size_t)s->layers * B * 256 * 4);
    s->h_state2 = take((size_t)s->layers * B * 256 * 4);
    {   // pending sums of dfx_k_df_convp_step: [B][kt-1][nfb][64 lanes] x 16 bytes (4096 streams of the released model: 101 MB)
        const int kt = c.df_pathway_kernel_size_t;
        const size_t rb = kt >= 2 && c.conv_ch % 32 == 0 ? (size_t)B * (kt - 1) * ((Fd + 15) / 16) * 64 * 16 : 0;
        static const bool ring_env = [] { const char *e = getenv("DFX_STREAM_C0RING"); return !(e && e[0] == '0'); }();
        if (rb > 0 && rb <= ((size_t)1 << 30) && ring_env) {
            s->c0ring_bytes = rb;
            s->c0ring = take(rb);
        }
    }
    s->lsnr = take((size_t)B * (H + n) * 4);
    dfx_model_workspace_bytes(m, B, H + n, &s->model_ws_bytes);
    s->model_ws = take((size_t)s->model_ws_bytes);
    s->bytes = off;
    if (hipMalloc(reinterpret_cast<void **>(&s->buf), s->bytes) != hipSuccess) {
        delete s;
        DFX_FAIL(DFX_ERR_ALLOC, "dfx_stream_create: device allocation of %zu bytes failed", off);
    }
    if (int rc = dfx_stream_reset(s, nullptr)) {
        dfx_stream_free(s);
        return rc;
    }
    *out = s;
    return DFX_OK;
}

extern "C" void dfx_stream_free(dfx_stream_state *s) {
    if (!s) return;
    if (s->buf) (void)hipFree(s->buf);
    if (s->gate_buf) (void)hipFree(s->gate_buf);
    delete s;
}

extern "C" int dfx_stream_reset(dfx_stream_state *s, void *stream) {
    if (!s) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_reset: null handle");
    hipStream_t hs = dfx_stream(stream);
    DFX_HIP(hipMemsetAsync(s->buf, 0, s->model_ws, hs));  // every state and history buffer (all of buf but the model workspace)
    // running means start like a fresh erb_norm / unit_norm (lib.rs:12-13, transforms.rs:308-318,339-349): the same expressions as
    // dfx_k_norm_scan evaluates when it is given no state
    const dfx_model_cfg &c = s->m->cfg;
    const int E = c.nb_erb, Fd = c.nb_df;
    std::vector<float> es((size_t)s->B * E), us((size_t)s->B * Fd);
    for (int ch = 0; ch < E; ++ch) {
        volatile float step = E > 1 ? (-90.f - -60.f) / (float)(E - 1) : 0.f;
        volatile float prod = step * (float)ch;
        const float v = -60.f + prod;
        for (int64_t b = 0; b < s->B; ++b) es[(size_t)b * E + ch] = v;
    }
    for (int ch = 0; ch < Fd; ++ch) {
        volatile float step = Fd > 1 ? (0.0001f - 0.001f) / (float)(Fd - 1) : 0.f;
        volatile float prod = step * (float)ch;
        const float v = 0.001f + prod;
        for (int64_t b = 0; b < s->B; ++b) us[(size_t)b * Fd + ch] = v;
    }
    DFX_HIP(hipStreamSynchronize(hs));
    DFX_HIP(hipMemcpy(s->buf + s->erb_state, es.data(), es.size() * 4, hipMemcpyHostToDevice));
    DFX_HIP(hipMemcpy(s->buf + s->unit_state, us.data(), us.size() * 4, hipMemcpyHostToDevice));
    if (s->gate_buf) DFX_HIP(hipMemset(s->gate_buf, 0, s->gate_bytes));  // skip counters, c0 windows (zero = the causal padding)
    s->frames = 0;
    s->flip = 0;
    s->lin_pos = 0;
    s->lin_owns = false;   // (both forms are all zeros now)
    s->feat_owns = false;
    s->hflip = 0;
    s->c0ring_ok = true;   // (zeros = the causal padding in front of the stream)
    return DFX_OK;
}

// tract.rs:658-672 / RuntimeParams::with_thresholds (:160-170).  Gating needs the stream to be at a reset point only in the sense
// that the decoders' delay lines start empty when it is switched on.
extern "C" int dfx_stream_set_thresholds(dfx_stream_state *s, float min_db_thresh, float max_db_erb_thresh, float max_db_df_thresh) {
    if (!s) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_set_thresholds: null handle");
    s->thr[0] = min_db_thresh;
    s->thr[1] = max_db_erb_thresh;
    s->thr[2] = max_db_df_thresh;
    return DFX_OK;
}

// RuntimeParams::n_ch / with_mask_reduce (tract.rs:119-176): rows [k*ch, (k+1)*ch) are the channels of stream k
extern "C" int dfx_stream_set_channels(dfx_stream_state *s, int channels, int reduce_mask) {
    if (!s || channels < 1 || s->B % channels != 0 || reduce_mask < 0 || reduce_mask > 2)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_set_channels: channels must divide the number of rows; reduce_mask 0 none, 1 max, 2 mean");
    s->channels = channels;
    s->reduce_mask = reduce_mask;
    return DFX_OK;
}

extern "C" int dfx_stream_set_gating(dfx_stream_state *s, int enable) {
    if (!s) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_set_gating: null handle");
    if (!enable) {
        s->gated = false;
        return DFX_OK;
    }
    const dfx_model_cfg &c = s->m->cfg;
    if (c.df_lookahead > 5) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_stream_set_gating: lookahead > 5 hops is not supported");
    if (c.df_pathway_kernel_size_t > 5) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_stream_set_gating: df_pathway_kernel_size_t > 5 is not supported");
    if (!s->gate_buf) {
        const int64_t B = s->B, T = s->H + 1;
        size_t off = 0;
        auto take = [&](size_t bytes) {
            size_t o = off;
            off += (bytes + 255) & ~(size_t)255;
            return o;
        };
        s->g_flags = take((size_t)B);
        s->g_counter = take((size_t)B * 4);
        s->g_sh_erb = take((size_t)B * c.nb_erb * 4);
        s->g_sh_unit = take((size_t)B * c.nb_df * 4);
        s->g_sh_h = take((size_t)s->layers * B * 256 * 4);
        {   // df_convp's state of a gated handle: pending sums (fp16-split models; 2 x what the ungated handle keeps) or the window of c0 frames
            const int kt = c.df_pathway_kernel_size_t;
            s->g_pend2_ok = kt >= 2 && kt <= 5 && c.conv_ch % 32 == 0 && s->m->fuse_c0 && !s->m->exact_fp32 && s->m->cp_h3;
            if (s->g_pend2_ok) {
                s->g_pend2 = take((size_t)B * 2 * (kt - 1) * ((c.nb_df + 15) / 16) * 64 * 16);
                s->g_par = take((size_t)B);
                s->g_cnt = take((size_t)B * 4);
            }
            s->g_c0_win = take(kt > 1 && !s->g_pend2_ok ? (size_t)B * T * c.nb_df * c.conv_ch * 4 : 256);
        }
        s->g_mask = take((size_t)B * T * c.nb_erb * 4);                       // dfx_stream_process_raw: the pass's mask / coefficients
        s->g_coefs = take((size_t)B * c.df_order * T * c.nb_df * 8);
        s->gate_bytes = off;
        if (hipMalloc(reinterpret_cast<void **>(&s->gate_buf), off) != hipSuccess) {
            s->gate_buf = nullptr;
            DFX_FAIL(DFX_ERR_ALLOC, "dfx_stream_set_gating: device allocation of %zu bytes failed", off);
        }
        DFX_HIP(hipMemset(s->gate_buf, 0, off));
    }
    s->gated = true;
    return DFX_OK;
}

extern "C" int dfx_stream_frame_length(const dfx_stream_state *s) { return s ? s->st->hop : 0; }
extern "C" int dfx_stream_delay_frames(const dfx_stream_state *s) { return s ? s->L : 0; }

extern "C" int dfx_stream_set_atten_lim(dfx_stream_state *s, float lim_db) {
    if (!s) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_set_atten_lim: null handle");
    const float lim = fabsf(lim_db);  // tract.rs:387-398
    if (lim >= 100.f) s->lim = 0.f;
    else if (lim < 0.01f) s->lim = 1.f;
    else s->lim = powf(10.f, -lim / 20.f);
    return DFX_OK;
}

extern "C" int dfx_stream_set_post_filter_beta(dfx_stream_state *s, float beta) {
    if (!s || beta < 0.f) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_set_post_filter_beta: bad arguments");
    s->pf_beta = beta;
    return DFX_OK;
}

// one call's kernels, enqueued on s (and the model's auxiliary streams); does not advance the handle's counters
// x / y / lsnr_out rows may be strided (xs, ys, ls; -1: packed): a gated call of n hops is n one-hop passes over the caller's arrays
static int stream_body(dfx_stream_state *S, const float *x, int64_t n, float *y, float *lsnr_out, hipStream_t s, int64_t xs = -1,
                       int64_t ys = -1, int64_t ls = -1) {
    const dfx_model *m = S->m;
    const dfx_state *st = S->st;
    const dfx_model_cfg &c = m->cfg;
    const int64_t B = S->B, H = S->H, L = S->L, Hs = H + L, F = st->N / 2 + 1, E = c.nb_erb, Fd = c.nb_df, hop = st->hop, ML = st->N - hop;
    auto fp = [&](size_t o) { return reinterpret_cast<float *>(S->buf + o); };
    auto gp = [&](size_t o) { return reinterpret_cast<float *>(S->gate_buf + o); };
    if (xs < 0) xs = n * hop;
    if (ys < 0) ys = n * hop;
    if (ls < 0) ls = n;
    int rc;
    const bool gated = S->gated && S->gate_buf;
    if (gated && n != 1) DFX_FAIL(DFX_ERR_INVALID_ARG, "gated streaming passes carry one hop");
    // ---- rolling spectra: linear (sliding window, see dfx_stream_state::spec_lin) or ring.  spec_window() brings the form this call uses
    // up to date with the other one if that one holds the state, appends the call's new frames and returns the window [Hs + n frames]
    // and the clip stride (in frames) the deep filter has to use.
    const bool lin = S->lin_cap > 0;
    const int64_t Fp = S->Fp, F2 = Fp * 2;   // the handle's spectra have rows of Fp >= F bins
    // The feature windows of the encoder take the same form when the kernels that read them accept a clip stride (the fp16-split DF
    // encoder: DfxC01hArgs::feat_T): [B, feat_cap, E] and [B, feat_cap, Fd, 2] with the same slack as the spectra, so that all three
    // windows sit at lin_pos and go back to the front in the same call.  feat_owns: the linear form holds the feature history.
    const bool feat_lin_ok = lin && S->feat_cap > 0 && m->fuse_c0 && !m->exact_fp32 && c.conv_ch % 32 == 0 && m->cp_h3;
    struct RowCopy { const float *src; int64_t src_stride, src_len, src_off; float *dst; int64_t dst_stride, len; };
    struct CopyList {
        RowCopy c[4];
        int n = 0;
        void add(const float *src, int64_t src_stride, int64_t src_len, int64_t src_off, float *dst, int64_t dst_stride, int64_t len) {
            c[n++] = RowCopy{src, src_stride, src_len, src_off, dst, dst_stride, len};
        }
    } cp_spec, cp_fe, cp_fs;   // the copies of this call, by array: the caller decides which stream each list is enqueued on
    auto emit = [&](CopyList &l, hipStream_t on) -> int {
        for (int i = 0; i < l.n; ++i)
            if (int r = stream_copy_rows(l.c[i].src, l.c[i].src_stride, l.c[i].src_len, l.c[i].src_off, l.c[i].dst, l.c[i].dst_stride, l.c[i].len, B, on)) return r;
        l.n = 0;
        return DFX_OK;
    };
    const int64_t capf = S->feat_cap, E1 = E, D2 = Fd * 2;
    auto hold = [&](float *win, int64_t cap, int64_t row, int64_t pos, int64_t h, hipStream_t on) -> int {   // frozen streams keep their history (dfx_k_gate_hold)
        dfx_launch(dfx_k_gate_hold, dim3((unsigned)B, (unsigned)(row > 1024 ? 4 : 1)), dim3(256), 0, on, (const unsigned char *)(S->gate_buf + S->g_flags), win,
                   cap, row, pos, h, B);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    };
    auto feat_to_ring = [&]() {   // the feature windows' last H frames become the ring form's history
        cp_fe.add(fp(S->fe_lin), capf * E1, capf * E1, S->lin_pos * E1, fp(S->hist_fe[S->flip]), H * E1, H * E1);
        cp_fs.add(fp(S->fs_lin), capf * D2, capf * D2, S->lin_pos * D2, fp(S->hist_fs[S->flip]), H * D2, H * D2);
        S->feat_owns = false;
    };
    // spec_window(): host-side bookkeeping of the rolling spectra for this call (which form, where the window is) with the copies it takes
    // listed in cp_spec (and, when the windows go back to the front, in cp_fe / cp_fs); the ring form is stepped on `s` right away.
    // The caller advances lin_pos by n when it is done with the windows.
    const float *spec_ring_src = nullptr;
    auto spec_window = [&](const float *new_spec, const float **win, int64_t *win_T) -> int {
        if (lin) {
            float *L0 = fp(S->spec_lin);
            const int64_t cap = S->lin_cap;
            if (!S->lin_owns) {   // the ring form's history becomes the window's first Hs frames
                cp_spec.add(fp(S->hist_spec[S->flip]), Hs * F2, Hs * F2, 0, L0, cap * F2, Hs * F2);
                S->lin_pos = 0;
                S->lin_owns = true;
            } else if (S->lin_pos + Hs + n > cap) {   // the windows have reached the end: their last frames go back to the front (no overlap: lin_pos >= Hs)
                cp_spec.add(L0, cap * F2, cap * F2, S->lin_pos * F2, L0, cap * F2, Hs * F2);
                if (S->feat_owns) {
                    cp_fe.add(fp(S->fe_lin), capf * E1, capf * E1, S->lin_pos * E1, fp(S->fe_lin), capf * E1, H * E1);
                    cp_fs.add(fp(S->fs_lin), capf * D2, capf * D2, S->lin_pos * D2, fp(S->fs_lin), capf * D2, H * D2);
                }
                S->lin_pos = 0;
            }
            cp_spec.add(new_spec, n * F2, n * F2, 0, L0 + (S->lin_pos + Hs) * F2, cap * F2, n * F2);
            *win = L0 + S->lin_pos * F2;
            *win_T = cap;
            return DFX_OK;
        }
        if (S->lin_owns) {   // back to the ring form (gating was switched on): the windows' last frames are its history
            if (S->feat_owns) feat_to_ring();
            cp_spec.add(fp(S->spec_lin), S->lin_cap * F2, S->lin_cap * F2, S->lin_pos * F2, fp(S->hist_spec[S->flip]), Hs * F2, Hs * F2);
            S->lin_owns = false;
            int r;
            if ((r = emit(cp_spec, s)) || (r = emit(cp_fe, s)) || (r = emit(cp_fs, s))) return r;
        }
        spec_ring_src = new_spec;   // the ring step itself is enqueued by spec_ring(): like the copies, where the caller wants it
        *win = fp(S->work_spec);
        *win_T = Hs + n;
        return DFX_OK;
    };
    auto spec_ring = [&](hipStream_t on) -> int {
        if (!spec_ring_src) return DFX_OK;
        DfxKScope ks(DFX_K_COPY_ROWS, on);
        dfx_launch(dfx_k_ring_step, dim3((unsigned)nn_grid(dfx_ceil_div(B * (Hs + n) * F2, 256), 16)), dim3(256), 0, on,
                   (const float *)fp(S->hist_spec[S->flip]), spec_ring_src, fp(S->work_spec), fp(S->hist_spec[S->flip ^ 1]), B, Hs, n, F2, (int64_t)0);
        DFX_LAUNCH_CHECK();
        spec_ring_src = nullptr;
        return DFX_OK;
    };
    if (S->lim == 1.f) {
        // tract.rs:509-543 with atten_lim == 1: the silent-input counter, the STFT analysis and the rolling spectra still advance (so
        // that switching the limit back mid-stream continues from the right history); features, network and synthesis do not run, the
        // hop is passed through undelayed with lsnr = 35 — unless the stream has been silent for more than 5 hops (zeros, -15).
        unsigned char *gflags = gated ? S->gate_buf + S->g_flags : nullptr;
        if (gated) {
            if ((rc = launch_gate_pre(x, xs, (int)hop, B, reinterpret_cast<int *>(S->gate_buf + S->g_counter), gflags, S->channels, s))) return rc;
        }
        float *am_in = fp(S->ana_mem[S->flip]), *am_out = fp(S->ana_mem[S->flip ^ 1]);
        float *new_spec = fp(S->new_spec);
        if ((rc = dfx_launch_analysis(st, x, B, n * hop, xs, am_in, am_out, new_spec, nullptr, s, -1, Fp))) return rc;
        {
            const float *win = nullptr;
            int64_t win_T = 0;
            if (S->feat_owns) feat_to_ring();   // (the features do not advance here: their history waits in the ring form)
            if ((rc = spec_window(new_spec, &win, &win_T)) || (rc = emit(cp_fe, s)) || (rc = emit(cp_fs, s)) || (rc = emit(cp_spec, s)) || (rc = spec_ring(s))) return rc;
            if (gated && lin && (rc = hold(fp(S->spec_lin), S->lin_cap, F2, S->lin_pos, Hs, s))) return rc;
            if (lin) S->lin_pos += n;
        }
        // what this path does not touch keeps its contents across the parity flip
        DFX_HIP(hipMemcpyAsync(fp(S->syn_mem[S->flip ^ 1]), fp(S->syn_mem[S->flip]), (size_t)B * ML * 4, hipMemcpyDeviceToDevice, s));
        DFX_HIP(hipMemcpyAsync(fp(S->hist_fe[S->flip ^ 1]), fp(S->hist_fe[S->flip]), (size_t)B * H * E * 4, hipMemcpyDeviceToDevice, s));
        DFX_HIP(hipMemcpyAsync(fp(S->hist_fs[S->flip ^ 1]), fp(S->hist_fs[S->flip]), (size_t)B * H * Fd * 8, hipMemcpyDeviceToDevice, s));
        if ((rc = stream_copy_rows(x, xs, n * hop, 0, y, ys, n * hop, B, s))) return rc;
        if (lsnr_out) {
            dfx_launch(dfx_k_fill_rows, dim3((unsigned)nn_grid(dfx_ceil_div(B * n, 256), 16)), dim3(256), 0, s, lsnr_out, ls, n, B, 35.f);
            DFX_LAUNCH_CHECK();
        }
        if (gated) {  // frozen streams: zeros / -15, and their analysis memory and rolling spectra stay where they were
            DfxGateTable G;
            G.n = 0;
            const unsigned char FZ = DFX_GATE_FROZEN;
            G.dst[0] = am_out, G.src[0] = am_in, G.row[0] = ML, G.mask[0] = FZ, G.want[0] = FZ;
            G.dst[1] = fp(S->hist_spec[S->flip ^ 1]), G.src[1] = fp(S->hist_spec[S->flip]), G.row[1] = Hs * F2, G.mask[1] = FZ, G.want[1] = FZ;
            G.n = lin ? 1 : 2;   // (linear window: dfx_k_gate_hold above)
            dfx_launch(dfx_k_gate_commit, dim3((unsigned)B), dim3(128), 0, s, G, (const unsigned char *)gflags, B);
            DFX_LAUNCH_CHECK();
            dfx_launch(dfx_k_gate_finish, dim3((unsigned)B), dim3(128), 0, s, (const unsigned char *)gflags,
                       reinterpret_cast<int *>(S->gate_buf + S->g_counter), y, ys, (int)hop, lsnr_out, ls, B, 1 /* no stage decision was taken */);
            DFX_LAUNCH_CHECK();
        }
        return DFX_OK;
    }
    unsigned char *gflags = gated ? S->gate_buf + S->g_flags : nullptr;
    int *gcount = gated ? reinterpret_cast<int *>(S->gate_buf + S->g_counter) : nullptr;
    // one new hop, plain launches: every GRU layer is ONE launch (projection + recurrence + gates) that leaves the new states in the
    // other buffer (DFX_STREAM_STEP=0: the projection and the recurrence kernel of the batch path, in place)
    static const bool step_env = [] { const char *e = getenv("DFX_STREAM_STEP"); return !(e && e[0] == '0'); }();
    const int64_t skip_early = S->frames < L ? ((L - S->frames) < n ? (L - S->frames) : n) : 0;
    const bool step_all = step_env && n - skip_early == 1;
    if (gated) {
        // silent-input shortcut (tract.rs:513-525) + a copy of the in-place state, so that the streams that turn out not to advance
        // (frozen, or a decoder stage skipped) can be given their state back after the pass
        if ((rc = launch_gate_pre(x, xs, (int)hop, B, gcount, gflags, S->channels, s))) return rc;
        DFX_HIP(hipMemcpyAsync(gp(S->g_sh_erb), fp(S->erb_state), (size_t)B * E * 4, hipMemcpyDeviceToDevice, s));
        DFX_HIP(hipMemcpyAsync(gp(S->g_sh_unit), fp(S->unit_state), (size_t)B * Fd * 4, hipMemcpyDeviceToDevice, s));
        // (the GRU states: only when the layers run in place — the one-step kernel leaves the old states in the other buffer)
        if (!step_all) DFX_HIP(hipMemcpyAsync(gp(S->g_sh_h), fp(S->hflip ? S->h_state2 : S->h_state), (size_t)S->layers * B * 256 * 4, hipMemcpyDeviceToDevice, s));
    }
    // ---- STFT + features of the n new hops (state: analysis memory, running means)
    float *am_in = fp(S->ana_mem[S->flip]), *am_out = fp(S->ana_mem[S->flip ^ 1]);
    float *sm_in = fp(S->syn_mem[S->flip]), *sm_out = fp(S->syn_mem[S->flip ^ 1]);
    float *new_spec = fp(S->new_spec), *new_fe = fp(S->new_fe), *new_fs = fp(S->new_fs);
    // The linear form: what only the DF branch needs (the DF feature window) is enqueued on that branch's stream (DfxStreamCtx::df_pre), what
    // only the final deep filter or the NEXT call needs (the spectrum window, the analysis memory) behind df_convp on its stream
    // (DfxStreamCtx::df_post) — in front of the encoder these four small launches were 40 us of a 520 us hop at 4096 streams
    static const bool side_env = [] { const char *e = getenv("DFX_STREAM_SIDE"); return !(e && e[0] == '0'); }();
    const bool side = side_env;   // (either form of the windows: the ring steps are deferred like the copies)
    if ((rc = dfx_launch_analysis(st, x, B, n * hop, xs, am_in, side ? nullptr : am_out, new_spec, new_fe, s, -1, Fp))) return rc;
    // ---- windows: [history ; new].  Net position p uses the features of hop p + L, so the hops of this call are the positions
    // a0 - L .. a0 + n - 1 - L; positions < 0 do not exist: their features are zero for the taps of later positions (the causal
    // padding of pad_feat, deepfilternet3.py:357-361) and they are not computed.
    const int64_t a0 = S->frames, T = H + n;
    const int64_t skip = a0 < L ? ((L - a0) < n ? (L - a0) : n) : 0;
    float *work_fe = fp(S->work_fe), *work_fs = fp(S->work_fs), *work_spec = fp(S->work_spec);
    struct Ring { size_t *hist; float *nw, *work; int64_t h, row; bool zero_skipped; } rings[2] = {
        {S->hist_fe, new_fe, work_fe, H, E, true}, {S->hist_fs, new_fs, work_fs, H, Fd * 2, true}};
    const float *spec_win = work_spec;
    int64_t spec_win_T = Hs + n;
    auto ring_step = [&](const Ring &r, hipStream_t on) -> int {  // window = [history ; new], next call's history = its last h frames
        DfxKScope ks(DFX_K_COPY_ROWS, on);
        dfx_launch(dfx_k_ring_step, dim3((unsigned)nn_grid(dfx_ceil_div(B * (r.h + n) * r.row, 256), 16)), dim3(256), 0, on,
                   (const float *)fp(r.hist[S->flip]), (const float *)r.nw, r.work, fp(r.hist[S->flip ^ 1]), B, r.h, n, r.row,
                   r.zero_skipped ? skip : (int64_t)0);
        DFX_LAUNCH_CHECK();
        return DFX_OK;
    };
    // settle the three windows (host side), then enqueue their copies / ring steps: on s, or — side — on the streams that need them
    if ((rc = spec_window(new_spec, &spec_win, &spec_win_T))) return rc;
    const bool flin = feat_lin_ok && skip == 0;   // (warm-up hops zero their features: the ring step does that)
    const float *fe_win = work_fe, *fs_win = work_fs;
    int64_t feat_T = 0;
    float *norm_fe = new_fe, *norm_fs = new_fs;   // where the normalised features of the new hops go
    int64_t norm_fe_cs = 0, norm_fs_cs = 0;
    if (flin) {
        float *Lfe = fp(S->fe_lin), *Lfs = fp(S->fs_lin);
        if (!S->feat_owns) {   // the ring form's history becomes the windows' first H frames
            cp_fe.add(fp(S->hist_fe[S->flip]), H * E1, H * E1, 0, Lfe + S->lin_pos * E1, capf * E1, H * E1);
            cp_fs.add(fp(S->hist_fs[S->flip]), H * D2, H * D2, 0, Lfs + S->lin_pos * D2, capf * D2, H * D2);
            S->feat_owns = true;
        }
        fe_win = Lfe + S->lin_pos * E1, fs_win = Lfs + S->lin_pos * D2;
        feat_T = capf;
        if (n < 16) {   // the norms write the new frames straight into the windows (no append copies)
            norm_fe = Lfe + (S->lin_pos + H) * E1, norm_fs = Lfs + (S->lin_pos + H) * D2;
            norm_fe_cs = capf * E1, norm_fs_cs = capf * D2;
        } else {
            cp_fe.add(new_fe, n * E1, n * E1, 0, Lfe + (S->lin_pos + H) * E1, capf * E1, n * E1);
            cp_fs.add(new_fs, n * D2, n * D2, 0, Lfs + (S->lin_pos + H) * D2, capf * D2, n * D2);
        }
    } else if (S->feat_owns) {
        feat_to_ring();
    }
    // features of the new hops (state: the running means)
    if ((rc = dfx_launch_norm_scan(new_fe, norm_fe, (int)E, new_spec, Fp, norm_fs, (int)Fd, B, n, c.norm_alpha, fp(S->erb_state), fp(S->unit_state), s,
                                   norm_fe_cs, norm_fs_cs)))
        return rc;
    const int64_t lin_pos0 = S->lin_pos;
    if (lin) S->lin_pos += n;   // (advanced here: this form is never replayed from a graph nor walked hop by hop by the caller)
    bool side_done = false, erb_done = false;
    std::function<int(hipStream_t)> side_pre, side_post, erb_ring;
    erb_ring = [&](hipStream_t on) -> int {
        erb_done = true;
        if (int r = emit(cp_fe, on)) return r;
        if (!flin) return ring_step(rings[0], on);
        return gated ? hold(fp(S->fe_lin), capf, E1, lin_pos0, H, on) : DFX_OK;
    };
    side_pre = [&](hipStream_t on) -> int {
        if (int r = emit(cp_fs, on)) return r;
        if (!flin) return ring_step(rings[1], on);
        return gated ? hold(fp(S->fs_lin), capf, D2, lin_pos0, H, on) : DFX_OK;
    };
    side_post = [&](hipStream_t on) -> int {
        side_done = true;
        if (int r = emit(cp_spec, on)) return r;
        if (int r = spec_ring(on)) return r;
        if (gated && lin)
            if (int r = hold(fp(S->spec_lin), S->lin_cap, F2, lin_pos0, Hs, on)) return r;
        return side ? dfx_launch_analysis_mem(st, x, B, n * hop, xs, am_in, am_out, on) : DFX_OK;
    };
    if (!side && ((rc = side_post(s)) || (rc = erb_ring(s)) || (rc = side_pre(s)))) return rc;
    float *out_spec = fp(S->out_spec);
    if (skip > 0) DFX_HIP(hipMemsetAsync(out_spec, 0, (size_t)B * n * Fp * 8, s));  // warm-up hops: zero spectra (tract.rs rolling buffers)
    bool stepped = false;
    if (skip < n) {
        DfxStreamCtx sc;
        sc.H = H + skip;
        const int64_t pos0 = Hs - a0;  // local index of net position 0
        sc.t_zero = pos0 > 0 ? pos0 : 0;
        sc.spec_T = spec_win_T;
        sc.spec_stride = Fp;
        sc.feat_T = feat_T;
        sc.h_state = fp(S->hflip ? S->h_state2 : S->h_state);
        const bool step = step_all;
        sc.h_next = step ? fp(S->hflip ? S->h_state : S->h_state2) : nullptr;
        stepped = step;
        if (step && !gated && S->c0ring_bytes) {   // df_convp from its pending sums (dfx_k_df_convp_step; a gated handle keeps its per-stream delay line)
            const int ns = c.df_pathway_kernel_size_t - 1;
            sc.c0ring = S->buf + S->c0ring;
            sc.c0slot = (int)((((a0 + skip - L) % ns) + ns) % ns);
            sc.c0rebuild = !S->c0ring_ok;
        }
        if (side) sc.erb_pre = erb_ring, sc.df_pre = side_pre, sc.df_post = side_post;
        sc.pf_beta = S->pf_beta;
        sc.out = out_spec;  // local frame t of clip b lands at out_spec[(b*n + t - H) * Fp]
        sc.out_T = n;
        sc.out_toff = H;
        sc.channels = S->channels;
        sc.reduce_mask = S->reduce_mask;
        DfxGate gate;
        if (gated) {
            gate.channels = S->channels;
            gate.flags = gflags;
            gate.thr[0] = S->thr[0], gate.thr[1] = S->thr[1], gate.thr[2] = S->thr[2];
            gate.c0_win = gp(S->g_c0_win);
            if (S->g_pend2_ok) gate.pend2 = S->gate_buf + S->g_pend2, gate.par = S->gate_buf + S->g_par, gate.cnt = reinterpret_cast<int *>(S->gate_buf + S->g_cnt);
            sc.gate = &gate;
        }
        float *ws = reinterpret_cast<float *>(((uintptr_t)(S->buf + S->model_ws) + 255) & ~(uintptr_t)255);
        const DfxLane *ln = &m->lanes[0];
        switch (c.conv_ch) {
            case 16: rc = forward_impl<16>(m, st->bands, spec_win, fe_win, fs_win, B, T, S->lim, nullptr, nullptr, fp(S->lsnr), nullptr, ws, s, ln, false, nullptr, &sc); break;
            case 32: rc = forward_impl<32>(m, st->bands, spec_win, fe_win, fs_win, B, T, S->lim, nullptr, nullptr, fp(S->lsnr), nullptr, ws, s, ln, false, nullptr, &sc); break;
            case 64: rc = forward_impl<64>(m, st->bands, spec_win, fe_win, fs_win, B, T, S->lim, nullptr, nullptr, fp(S->lsnr), nullptr, ws, s, ln, false, nullptr, &sc); break;
            default: DFX_FAIL(DFX_ERR_UNSUPPORTED, "conv_ch");
        }
        if (rc) return rc;
        if (stepped) S->hflip ^= 1;   // (like lin_pos: this form is neither replayed from a graph nor walked hop by hop by the caller)
        S->c0ring_ok = sc.c0ring_used;   // any pass that did not go through the step kernel (several hops, gated, run_df off) leaves the sums behind
        if (gated && c.df_pathway_kernel_size_t > 1) {  // the DF decoder's delay line moves where that decoder ran
            if (S->g_pend2_ok) {
                dfx_launch(dfx_k_gate_pend_commit, dim3((unsigned)dfx_ceil_div(B, 256)), dim3(256), 0, s, (const unsigned char *)gflags,
                           S->gate_buf + S->g_par, reinterpret_cast<int *>(S->gate_buf + S->g_cnt), B);
            } else {
                const int64_t frame = (int64_t)Fd * c.conv_ch;
                dfx_launch(dfx_k_gate_c0_shift, dim3((unsigned)B, 4), dim3(256), 0, s, (const unsigned char *)gflags, gp(S->g_c0_win), B, T,
                           c.df_pathway_kernel_size_t, frame);
            }
            DFX_LAUNCH_CHECK();
        }
    }
    if (side && !erb_done && (rc = erb_ring(s))) return rc;                              // (no forward pass ran: warm-up hops)
    if (side && !side_done && ((rc = side_pre(s)) || (rc = side_post(s)))) return rc;
    // ---- ISTFT of the n enhanced hops (state: overlap-add memory)
    if ((rc = dfx_launch_synthesis(st, out_spec, B, n, sm_in, sm_out, y, ys, 0, n * hop, s, 0, -1, Fp))) return rc;
    if (lsnr_out) {  // the window's lsnr is [B, T]: take the n new frames (the entries of warm-up hops are not meaningful)
        if ((rc = stream_copy_rows(fp(S->lsnr), T, T, H, lsnr_out, ls, n, B, s))) return rc;
    }
    if (gated) {
        // ---- who keeps which state (dfx_k_gate_commit), then the frozen streams' answer and the skip counters
        DfxGateTable G;
        G.n = 0;
        auto entry = [&](float *dst, const float *src, int64_t row, unsigned char mask, unsigned char want) {
            G.dst[G.n] = dst, G.src[G.n] = src, G.row[G.n] = row, G.mask[G.n] = mask, G.want[G.n] = want;
            ++G.n;
        };
        const unsigned char FZ = DFX_GATE_FROZEN;
        entry(am_out, am_in, ML, FZ, FZ);
        entry(sm_out, sm_in, ML, FZ, FZ);
        if (!flin) {   // (linear windows: dfx_k_gate_hold)
            entry(fp(S->hist_fe[S->flip ^ 1]), fp(S->hist_fe[S->flip]), H * E, FZ, FZ);
            entry(fp(S->hist_fs[S->flip ^ 1]), fp(S->hist_fs[S->flip]), H * Fd * 2, FZ, FZ);
        }
        if (!lin) entry(fp(S->hist_spec[S->flip ^ 1]), fp(S->hist_spec[S->flip]), Hs * F2, FZ, FZ);
        entry(fp(S->erb_state), gp(S->g_sh_erb), E, FZ, FZ);
        entry(fp(S->unit_state), gp(S->g_sh_unit), Fd, FZ, FZ);
        const int nenc = (int)m->enc_gru.size(), ndec = (int)m->dec_gru.size();
        for (int l = 0; l < S->layers; ++l) {
            float *h = fp(S->hflip ? S->h_state2 : S->h_state) + (int64_t)l * B * 256;
            const float *hs = (stepped ? fp(S->hflip ? S->h_state : S->h_state2) : gp(S->g_sh_h)) + (int64_t)l * B * 256;   // the states before this pass
            if (l < nenc) entry(h, hs, 256, FZ, FZ);
            else if (l < nenc + ndec) entry(h, hs, 256, DFX_GATE_GAINS, 0);   // stage 1 did not run (frozen streams included)
            else entry(h, hs, 256, DFX_GATE_DF, 0);                           // stage 2 did not run
        }
        dfx_launch(dfx_k_gate_commit, dim3((unsigned)B), dim3(128), 0, s, G, (const unsigned char *)gflags, B);
        DFX_LAUNCH_CHECK();
        dfx_launch(dfx_k_gate_finish, dim3((unsigned)B), dim3(128), 0, s, (const unsigned char *)gflags, gcount, y, ys, (int)hop, lsnr_out,
                   ls, B, (int)(skip >= n));
        DFX_LAUNCH_CHECK();
    }
    return DFX_OK;
}

// Faults raised by kernels (dfx_model::h_err): a call reports what earlier passes on the model raised before it starts its own, and — with
// DFX_CHECK_EVERY_PASS=1 — waits for its own pass and reports that too.
static int stream_call_end(const dfx_model *m, hipStream_t s) {
    if (!m->check_every_pass) return DFX_OK;
    DFX_HIP(hipStreamSynchronize(s));
    return model_poll(m);
}
static int stream_process_impl(dfx_stream_state *S, const float *x, int64_t n, float *y, float *lsnr_out, hipStream_t s);
extern "C" int dfx_stream_process(dfx_stream_state *S, const float *x, int64_t n, float *y, float *lsnr_out, void *stream) {
    if (!S || n <= 0 || n > S->nmax || !x || !y) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_process: bad arguments (1 <= n_frames <= max_frames)");
    if (int rc = dfx_require_device()) return rc;
    if (int rc = model_poll(S->m)) return rc;
    hipStream_t s = dfx_stream(stream);
    {
        DfxTurn turn(S->m, s, false);   // (the enqueue lock only: a hop starts no persistent phase)
        if (int rc = stream_process_impl(S, x, n, y, lsnr_out, s)) return rc;
    }
    return stream_call_end(S->m, s);
}
static int stream_process_impl(dfx_stream_state *S, const float *x, int64_t n, float *y, float *lsnr_out, hipStream_t s) {
    const bool advances = S->lim != 1.f;  // the pass-through case (tract.rs:540-543) moves the STFT memory and the rolling spectra only
    const int64_t hop = S->st->hop;
    if (S->gated && S->gate_buf) {  // one hop per pass: the stage decisions of hop i shape the state hop i+1 starts from
        for (int64_t i = 0; i < n; ++i) {
            if (int rc = stream_body(S, x + i * hop, 1, y + i * hop, lsnr_out ? lsnr_out + i : nullptr, s, n * hop, n * hop, n)) return rc;
            if (advances) S->frames += 1;
            S->flip ^= 1;
        }
        return DFX_OK;
    }
    if (int rc = stream_body(S, x, n, y, lsnr_out, s)) return rc;
    if (advances) S->frames += n;
    S->flip ^= 1;
    return DFX_OK;
}

// DfTract::process_raw (tract.rs:441-507; exported as df_process_frame_raw, capi.rs:172-210): one *spectral* frame per stream in, the
// raw ERB gains and deep-filter coefficients of that pass out — features with the running means, encoder, stage decisions, the
// decoders that the decision selects (their state only moves when they run).  No STFT, no deep filtering, no synthesis, and (like the
// reference) neither the rolling spectra nor the silent-input counter are touched.  Needs gating (dfx_stream_set_gating); a handle
// should be driven either by dfx_stream_process or by this function, not by both.
//   spec [streams, F][2] -> gains [streams, nb_erb], coefs [streams, df_order, nb_df][2], stages [streams]: bit 1 (2) = gains present
//   (the network's mask, or zeros when lsnr < min_db_thresh), bit 3 (8) = coefficients present; a caller maps absent to NULL.
extern "C" int dfx_stream_process_raw(dfx_stream_state *S, const float *spec, float *gains, float *coefs, unsigned char *stages, float *lsnr_out,
                                      void *stream) {
    if (!S || !spec || !gains || !coefs || !stages) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_process_raw: null argument");
    if (!S->gated || !S->gate_buf) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_stream_process_raw: switch gating on first (dfx_stream_set_gating)");
    if (int rc = dfx_require_device()) return rc;
    if (int rc = model_poll(S->m)) return rc;
    hipStream_t s = dfx_stream(stream);
    DfxTurn turn(S->m, s, false);   // (the enqueue lock)
    const dfx_model *m = S->m;
    const dfx_state *st = S->st;
    const dfx_model_cfg &c = m->cfg;
    const int64_t B = S->B, H = S->H, L = S->L, Hs = H + L, F = st->N / 2 + 1, E = c.nb_erb, Fd = c.nb_df, hop = st->hop, ML = st->N - hop;
    const int64_t n = 1, T = H + n, a0 = S->frames;
    const int O = c.df_order;
    auto fp = [&](size_t o) { return reinterpret_cast<float *>(S->buf + o); };
    auto gp = [&](size_t o) { return reinterpret_cast<float *>(S->gate_buf + o); };
    unsigned char *gflags = S->gate_buf + S->g_flags;
    int rc;
    // this path keeps the windows in ring form: if an earlier call on the handle left them in the linear buffers, their last frames
    // become the ring form's history first (as stream_body does when it changes form)
    if (S->feat_owns) {
        const int64_t capf = S->feat_cap, D2 = Fd * 2;
        if ((rc = stream_copy_rows(fp(S->fe_lin), capf * E, capf * E, S->lin_pos * E, fp(S->hist_fe[S->flip]), H * E, H * E, B, s)) ||
            (rc = stream_copy_rows(fp(S->fs_lin), capf * D2, capf * D2, S->lin_pos * D2, fp(S->hist_fs[S->flip]), H * D2, H * D2, B, s)))
            return rc;
        S->feat_owns = false;
    }
    if (S->lin_owns) {
        const int64_t F2 = S->Fp * 2;
        if ((rc = stream_copy_rows(fp(S->spec_lin), S->lin_cap * F2, S->lin_cap * F2, S->lin_pos * F2, fp(S->hist_spec[S->flip]), Hs * F2, Hs * F2, B, s))) return rc;
        S->lin_owns = false;
    }
    DFX_HIP(hipMemsetAsync(gflags, 0, (size_t)B, s));  // no silent-input test on this path (tract.rs:441: process_raw starts at the features)
    DFX_HIP(hipMemcpyAsync(gp(S->g_sh_erb), fp(S->erb_state), (size_t)B * E * 4, hipMemcpyDeviceToDevice, s));
    DFX_HIP(hipMemcpyAsync(gp(S->g_sh_unit), fp(S->unit_state), (size_t)B * Fd * 4, hipMemcpyDeviceToDevice, s));
    DFX_HIP(hipMemcpyAsync(gp(S->g_sh_h), fp(S->hflip ? S->h_state2 : S->h_state), (size_t)S->layers * B * 256 * 4, hipMemcpyDeviceToDevice, s));
    // features of the given spectra (state: the running means): erb (dB) -> mean norm, low bins -> unit norm (lib.rs:206-217)
    float *new_fe = fp(S->new_fe), *new_fs = fp(S->new_fs);   // (the caller's dense [B, F] spectra are read in place)
    if ((rc = dfx_erb(st->bands, spec, B, 1, new_fe, s))) return rc;
    if ((rc = dfx_launch_norm_scan(new_fe, new_fe, (int)E, spec, F, new_fs, (int)Fd, B, n, c.norm_alpha, fp(S->erb_state),
                                   fp(S->unit_state), s)))
        return rc;
    const int64_t skip = a0 < L ? 1 : 0;
    float *work_fe = fp(S->work_fe), *work_fs = fp(S->work_fs);
    struct Ring { size_t *hist; float *nw, *work; int64_t h, row; } rings[2] = {{S->hist_fe, new_fe, work_fe, H, E}, {S->hist_fs, new_fs, work_fs, H, Fd * 2}};
    for (const Ring &r : rings) {
        DfxKScope ks(DFX_K_COPY_ROWS, s);
        dfx_launch(dfx_k_ring_step, dim3((unsigned)nn_grid(dfx_ceil_div(B * (r.h + n) * r.row, 256), 16)), dim3(256), 0, s,
                   (const float *)fp(r.hist[S->flip]), (const float *)r.nw, r.work, fp(r.hist[S->flip ^ 1]), B, r.h, n, r.row, skip);
        DFX_LAUNCH_CHECK();
    }
    // the buffers this path does not use keep their contents across the parity flip
    DFX_HIP(hipMemcpyAsync(fp(S->ana_mem[S->flip ^ 1]), fp(S->ana_mem[S->flip]), (size_t)B * ML * 4, hipMemcpyDeviceToDevice, s));
    DFX_HIP(hipMemcpyAsync(fp(S->syn_mem[S->flip ^ 1]), fp(S->syn_mem[S->flip]), (size_t)B * ML * 4, hipMemcpyDeviceToDevice, s));
    DFX_HIP(hipMemcpyAsync(fp(S->hist_spec[S->flip ^ 1]), fp(S->hist_spec[S->flip]), (size_t)B * Hs * S->Fp * 8, hipMemcpyDeviceToDevice, s));
    float *mask = gp(S->g_mask), *cbuf = gp(S->g_coefs);
    if (!skip) {
        DfxStreamCtx sc;
        sc.H = H;
        const int64_t pos0 = Hs - a0;
        sc.t_zero = pos0 > 0 ? pos0 : 0;
        sc.spec_T = Hs + n;
        sc.spec_stride = S->Fp;
        sc.h_state = fp(S->hflip ? S->h_state2 : S->h_state);
        sc.pf_beta = 0.f;
        sc.out = fp(S->out_spec);  // the deep-filter kernel still runs (on whatever the spectrum window holds); its output is not used
        sc.out_T = n;
        sc.out_toff = H;
        sc.channels = S->channels;
        sc.reduce_mask = S->reduce_mask;
        DfxGate gate;
        gate.channels = S->channels;
        gate.flags = gflags;
        gate.thr[0] = S->thr[0], gate.thr[1] = S->thr[1], gate.thr[2] = S->thr[2];
        gate.c0_win = gp(S->g_c0_win);
        if (S->g_pend2_ok) gate.pend2 = S->gate_buf + S->g_pend2, gate.par = S->gate_buf + S->g_par, gate.cnt = reinterpret_cast<int *>(S->gate_buf + S->g_cnt);
        sc.gate = &gate;
        float *ws = reinterpret_cast<float *>(((uintptr_t)(S->buf + S->model_ws) + 255) & ~(uintptr_t)255);
        const DfxLane *ln = &m->lanes[0];
        switch (c.conv_ch) {
            case 16: rc = forward_impl<16>(m, st->bands, fp(S->work_spec), work_fe, work_fs, B, T, 0.f, nullptr, mask, fp(S->lsnr), cbuf, ws, s, ln, false, nullptr, &sc); break;
            case 32: rc = forward_impl<32>(m, st->bands, fp(S->work_spec), work_fe, work_fs, B, T, 0.f, nullptr, mask, fp(S->lsnr), cbuf, ws, s, ln, false, nullptr, &sc); break;
            case 64: rc = forward_impl<64>(m, st->bands, fp(S->work_spec), work_fe, work_fs, B, T, 0.f, nullptr, mask, fp(S->lsnr), cbuf, ws, s, ln, false, nullptr, &sc); break;
            default: DFX_FAIL(DFX_ERR_UNSUPPORTED, "conv_ch");
        }
        if (rc) return rc;
        if (c.df_pathway_kernel_size_t > 1) {
            if (S->g_pend2_ok)
                dfx_launch(dfx_k_gate_pend_commit, dim3((unsigned)dfx_ceil_div(B, 256)), dim3(256), 0, s, (const unsigned char *)gflags,
                           S->gate_buf + S->g_par, reinterpret_cast<int *>(S->gate_buf + S->g_cnt), B);
            else
                dfx_launch(dfx_k_gate_c0_shift, dim3((unsigned)B, 4), dim3(256), 0, s, (const unsigned char *)gflags, gp(S->g_c0_win), B, T,
                           c.df_pathway_kernel_size_t, (int64_t)Fd * c.conv_ch);
            DFX_LAUNCH_CHECK();
        }
        // the newest frame's mask row and coefficient rows (coefficients are [B, O, T, F'][2]: one strided row per (stream, tap))
        if ((rc = stream_copy_rows(mask, T * E, T * E, (T - 1) * E, gains, E, E, B, s))) return rc;
        if ((rc = stream_copy_rows(cbuf, T * Fd * 2, T * Fd * 2, (T - 1) * Fd * 2, coefs, Fd * 2, Fd * 2, B * O, s))) return rc;
        if (lsnr_out && (rc = stream_copy_rows(fp(S->lsnr), T, T, H, lsnr_out, 1, 1, B, s))) return rc;
        // decoder states of the stages that did not run go back to what they were
        DfxGateTable G;
        G.n = 0;
        const int nenc = (int)m->enc_gru.size(), ndec = (int)m->dec_gru.size();
        for (int l = nenc; l < S->layers; ++l) {
            G.dst[G.n] = fp(S->hflip ? S->h_state2 : S->h_state) + (int64_t)l * B * 256, G.src[G.n] = gp(S->g_sh_h) + (int64_t)l * B * 256, G.row[G.n] = 256;
            G.mask[G.n] = l < nenc + ndec ? DFX_GATE_GAINS : DFX_GATE_DF, G.want[G.n] = 0;
            ++G.n;
        }
        dfx_launch(dfx_k_gate_commit, dim3((unsigned)B), dim3(128), 0, s, G, (const unsigned char *)gflags, B);
        DFX_LAUNCH_CHECK();
    } else if (lsnr_out) {
        dfx_launch(dfx_k_fill_rows, dim3((unsigned)nn_grid(dfx_ceil_div(B, 256), 16)), dim3(256), 0, s, lsnr_out, (int64_t)1, (int64_t)1, B, -15.f);
        DFX_LAUNCH_CHECK();
    }
    // stages: bit 2 = gains exist (the network's mask, or zeros below min_db_thresh: the reference returns Some(zeros) there,
    // tract.rs:485-486), bit 8 = coefficients exist
    dfx_launch(dfx_k_gate_stages, dim3((unsigned)dfx_ceil_div(B, 256)), dim3(256), 0, s, (const unsigned char *)gflags, stages, B);
    DFX_LAUNCH_CHECK();
    S->frames += 1;
    S->flip ^= 1;
    return stream_call_end(m, s);
}

// Batch-chunk pipelining: the GRU chain of a chunk is a long latency chain on a handful of CUs, so dfx_enhance splits the
// batch into up to DFX_MAX_LANES chunks (multiples of the 16 clips a GRU workgroup owns), each with its own streams; the
// chip-filling "front" (features, encoder convolutions) of chunk c+1 is released when chunk c has enqueued its front, and
// then overlaps chunk c's GRU chain; the tails overlap likewise.  Chunks are independent clips, so results do not change.
static int enh_chunks(const dfx_model *m, int64_t B, int64_t *sizes) {
    int nc = 1;
    if (m->concurrent && m->max_chunks > 1) {
        const int64_t groups = dfx_ceil_div(B, 16);
        nc = (int)(groups / 2 < m->max_chunks ? groups / 2 : m->max_chunks);  // at least 32 clips per chunk
        if (nc < 1) nc = 1;
    }
    const int64_t groups = dfx_ceil_div(B, 16);
    int64_t done = 0;
    for (int c = 0; c < nc; ++c) {
        int64_t g = groups / nc + (c < groups % nc ? 1 : 0);
        int64_t n = g * 16;
        if (done + n > B) n = B - done;
        sizes[c] = n;
        done += n;
    }
    return nc;
}

extern "C" int dfx_enhance_workspace_bytes(const dfx_model *m, const dfx_state *st, int64_t B, int64_t T, int pad, int64_t *bytes) {
    if (!m || !st || !bytes || B < 0 || T < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance_workspace_bytes: bad arguments");
    // sized for the finest chunking the handle may use, so toggling dfx_model_set_streams never needs a bigger workspace
    int64_t sizes[DFX_MAX_LANES];
    int64_t total = (int64_t)plan_enh(m, st, B, T, pad).total;
    if (m->max_chunks > 1 && m->have_streams) {
        const bool was = m->concurrent;
        const_cast<dfx_model *>(m)->concurrent = true;
        const int nc = enh_chunks(m, B, sizes);
        const_cast<dfx_model *>(m)->concurrent = was;
        int64_t sum = 0;
        for (int c = 0; c < nc; ++c) sum += (int64_t)plan_enh(m, st, sizes[c], T, pad).total;
        if (sum > total) total = sum;
    }
    *bytes = total;
    return DFX_OK;
}

// pcm16: x and y point at int16_t samples (same strides in samples); the conversions of df/io.py run in the STFT kernel's loads and the
// ISTFT kernel's stores
static int enhance_chunk(const dfx_model *m, const dfx_state *st, const float *x, int64_t B, int64_t T, int pad, float lim,
                         float *y, unsigned char *base, hipStream_t s, const DfxLane *ln, bool signal_front, bool pcm16) {
    const dfx_model_cfg &c = m->cfg;
    const EnhWs w = plan_enh(m, st, B, T, pad);
    const int64_t Tp = pad ? T + st->N : T, Tf = Tp / st->hop;
    float *spec = reinterpret_cast<float *>(base + w.spec), *spec_e = reinterpret_cast<float *>(base + w.spec_e);
    float *fe = reinterpret_cast<float *>(base + w.feat_erb), *fs = reinterpret_cast<float *>(base + w.feat_spec);
    // F.pad(audio, (0, n_fft)) (enhance.py:230-233) is implicit: the analysis reads zeros past the T samples of a row
    const int64_t sstride = enh_spec_stride(st);
    int rc = dfx_features_padded(st, x, B, Tp, T, T, c.nb_df, c.norm_alpha, spec, fe, fs, (void *)s, sstride, pcm16);
    if (rc) return rc;
    int64_t mb = 0;
    dfx_model_workspace_bytes(m, B, Tf, &mb);
    // the synthesis is enqueued by the model forward (per time chunk when the GRU phase is pipelined); with pad it stores exactly
    // the window audio[:, d : orig_len + d] of enhance.py:248-249
    DfxFinish fin;
    fin.st = st;
    fin.y = y;
    fin.out_stride = pad ? T : Tf * st->hop;
    fin.out_skip = pad ? st->N - st->hop : 0;
    fin.out_len = pad ? T : Tf * st->hop;
    fin.spec_stride = sstride;
    fin.out_i16 = pcm16;
    return model_forward_lane(m, st->bands, spec, fe, fs, B, Tf, lim, spec_e, nullptr, nullptr, nullptr, base + w.model, mb, (void *)s, ln,
                              signal_front, &fin);
}

static int enhance_any(const dfx_model *m, const dfx_state *st, const float *x, int64_t B, int64_t T, int pad,
                       float atten_lim_db, float *y, void *workspace, int64_t workspace_bytes, void *stream, bool pcm16) {
    if (!m || !st || B < 0 || T < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance: bad arguments");
    const dfx_model_cfg &c = m->cfg;
    if (st->N != c.fft_size || st->hop != c.hop_size || st->nb != c.nb_erb)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance: DF state does not match the model configuration");
    if (pad && st->N % st->hop) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance: pad requires fft_size %% hop_size == 0 (enhance.py:247)");
    if (int rc = dfx_require_device()) return rc;
    if (B == 0) return DFX_OK;
    if (!x || !y || !workspace) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance: null buffer");
    int64_t sizes[DFX_MAX_LANES];
    const int nc = enh_chunks(m, B, sizes);
    int64_t need = 0;
    for (int i = 0; i < nc; ++i) need += (int64_t)plan_enh(m, st, sizes[i], T, pad).total;
    if (workspace_bytes < need) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_enhance: workspace too small");
    unsigned char *base = reinterpret_cast<unsigned char *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t s = dfx_stream(stream);
    const int64_t Tp = pad ? T + st->N : T, Tf = Tp / st->hop;
    const int64_t out_len = pad ? T : Tf * st->hop;
    if (Tf == 0) {
        if (out_len > 0) DFX_HIP(hipMemsetAsync(y, 0, (size_t)B * out_len * (pcm16 ? 2 : 4), s));
        return DFX_OK;
    }
    float lim = 0.f;
    if (atten_lim_db != 0.f) {
        lim = powf(10.f, -fabsf(atten_lim_db) / 20.f);  // enhance.py:238-239
        if (lim >= 1.f) lim = 0.99999994f;              // |dB| tiny: the reference mixes with lim == 1.0f (the noisy signal passes)
    }
    if (int rc = pass_begin(m, B * Tf)) return rc;
    DfxTurn turn(m, s, true);
    if (nc == 1) {
        if (int rc = enhance_chunk(m, st, x, B, T, pad, lim, y, base, s, &m->lanes[0], false, pcm16)) return rc;
        turn.passed();
        return pass_end(m, B * Tf, s);
    }
    // ---- pipelined chunks: fork from the caller's stream, stagger the fronts, join back
    DFX_HIP(hipEventRecord(m->ev_fork, s));
    int64_t row = 0;
    for (int i = 0; i < nc; ++i) {
        const DfxLane *ln = &m->lanes[i];
        DFX_HIP(hipStreamWaitEvent(ln->main, m->ev_fork, 0));
        if (i > 0) DFX_HIP(hipStreamWaitEvent(ln->main, m->lanes[i - 1].ev[EV_FRONT], 0));
        // (16-bit samples: the float-typed pointers advance by half as many elements)
        const float *xi = pcm16 ? reinterpret_cast<const float *>(reinterpret_cast<const int16_t *>(x) + row * T) : x + row * T;
        float *yi = pcm16 ? reinterpret_cast<float *>(reinterpret_cast<int16_t *>(y) + row * out_len) : y + row * out_len;
        if (int rc = enhance_chunk(m, st, xi, sizes[i], T, pad, lim, yi, base, ln->main, ln, true, pcm16)) return rc;
        DFX_HIP(hipEventRecord(ln->ev[EV_DONE], ln->main));
        base += (plan_enh(m, st, sizes[i], T, pad).total + 255) & ~(size_t)255;
        row += sizes[i];
    }
    for (int i = 0; i < nc; ++i) DFX_HIP(hipStreamWaitEvent(s, m->lanes[i].ev[EV_DONE], 0));
    turn.passed();
    return pass_end(m, B * Tf, s);
}
extern "C" int dfx_enhance(const dfx_model *m, const dfx_state *st, const float *x, int64_t B, int64_t T, int pad,
                           float atten_lim_db, float *y, void *workspace, int64_t workspace_bytes, void *stream) {
    return enhance_any(m, st, x, B, T, pad, atten_lim_db, y, workspace, workspace_bytes, stream, false);
}
extern "C" int dfx_enhance_pcm16(const dfx_model *m, const dfx_state *st, const int16_t *x, int64_t B, int64_t T, int pad,
                                 float atten_lim_db, int16_t *y, void *workspace, int64_t workspace_bytes, void *stream) {
    return enhance_any(m, st, reinterpret_cast<const float *>(x), B, T, pad, atten_lim_db, reinterpret_cast<float *>(y), workspace, workspace_bytes,
                       stream, true);
}
