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
static_assert(DFX_MAX_GRU_LAYERS == DFX_GS_MAX_LAYERS, "the host lays the XCD / progress tables out with the stride the kernels index them with");
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
    unsigned int *d_psync = nullptr;    // pair form of the persistent GRU phase: [DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX / 2][48] words
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
        bool gru_pair_far = false;      // DFX_GRU_PAIR_FAR=1 (test hook): the pairs' hand-overs in their agent-scope form, as if the halves sat on different XCDs
        bool gru_pair = true;           // DFX_GRU_PAIR=0: every 16-clip group's recurrence on one CU (W_hh streamed from the L2) instead of 32 clips on a pair of CUs (dfx_gru_pair.h)
        int convp_late = -1;            // DFX_CONVP_LATE (percent)
        int convp_after_p0 = -1;        // DFX_CONVP_AFTER_P0
        int p0_ahead = 3;               // DFX_SEQ_P0_AHEAD
        int tail_every = 1;             // DFX_SEQ_TAIL_EVERY
        int dftail_every = 0;           // DFX_SEQ_DFTAIL_EVERY (0: the rule in forward_impl)
        bool tail_split = true;         // the ERB decoder's linear_out on a stream of its own beside the decoder tail (round 6; DFX_TAIL_SPLIT=0 in -DDFX_DEV builds)
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
        // (round 6: the DFX_FUSE_* / DFX_E0_RECOMPUTE / DFX_DFOUT_LEAN / DFX_PROJ_RT / DFX_SEQ_* ... switches of rounds 2-5 are gone — the forms they
        // selected were measured and lost; what remains below are properties of the model's shape, and tools/dev/patches/ keeps the experiments)
        m->fuse_c0 = m->cfg.df_pathway_kernel_size_t <= 5 && 2 * m->cfg.df_order <= 16;
        // Exact mode, batch passes: c0 is written once and read by its two consumers instead of being recomputed by both — on fp32 matrix ops a c0
        // tile is 20 ops of 32 cycles (3 of 16 on the fp16-split path), and the two recomputing kernels side by side are the exact front's 10 ms:
        // 28.65 vs 29.45 ms per step (the frame-by-frame runtime always uses the recomputing forms)
        m->c0_batch_unfused = m->exact_fp32 && m->fuse_c0;
        m->fuse_erb = true;
        const char *gq = getenv("DFX_GRU_SEQ");
        {
            const char *gp = getenv("DFX_GRU_PAIR");
            if (gp) m->sw.gru_pair = gp[0] != '0';
            const char *gf = getenv("DFX_GRU_PAIR_FAR");
            m->sw.gru_pair_far = gf && gf[0] == '1';
        }
        m->gru_seq = !(gq && gq[0] == '0') && !dfx_env_is_emulator();
        m->fuse_emb = m->fuse_encfan = m->fuse_dfa = m->fuse_tail = m->e0_recompute = m->fuse_dfenc = m->dfout_lean = true;
        const char *cep = getenv("DFX_CHECK_EVERY_PASS"), *spl = getenv("DFX_SYNC_SPIN_LIMIT");
        m->check_every_pass = cep && cep[0] == '1';
        if (spl && atoi(spl) > 0) m->spin_limit = atoi(spl);
        m->phase_late = true;
        m->proj_rt = 0;
        {
            const char *cel = getenv("DFX_CONVP_ELEMS");   // test hook: the 32-bit-offset split of df_convp at small sizes
            if (cel && atoll(cel) > 0) m->sw.convp_elems = atoll(cel);
        }
#ifdef DFX_DEV
        {   // dev A/Bs of the phase's side work (product builds have no such switches)
            const char *v;
            if ((v = getenv("DFX_TAIL_SPLIT"))) m->sw.tail_split = v[0] != '0';
            if ((v = getenv("DFX_SEQ_P0_AHEAD")) && atoi(v) > 0) m->sw.p0_ahead = atoi(v);
            if ((v = getenv("DFX_SEQ_DFTAIL_EVERY")) && atoi(v) > 0) m->sw.dftail_every = atoi(v);
            if ((v = getenv("DFX_SEQ_TAIL_EVERY")) && atoi(v) > 0) m->sw.tail_every = atoi(v);
            if ((v = getenv("DFX_SEQ_CHUNKS")) && atoi(v) > 0) m->sw.chunks = atoi(v);
            if ((v = getenv("DFX_CONVP_LATE")) && atoi(v) >= 0) m->sw.convp_late = atoi(v);
        }
#endif
        {
            const char *tq = getenv("DFX_SEQ_TRACE");
            if (tq && tq[0] == '1') (void)hipMalloc(reinterpret_cast<void **>(&m->d_trace), (size_t)DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX * DFX_GS_MAX_CHUNKS * 3 * 8);
        }
        // ready 0-7 | emb 8 | probe 13 | done 16- | producers' completion counters (DfxPublish): 9 words (16) | yprog, giprog: steps per (layer, group)
        // | XCD registrations [3 kinds][layers][groups] (DfxXcd) | 64 words: [0] light hand-overs counted (dev aid), [8..48) the followers' claim counters
        const size_t sync_bytes = (size_t)(16 + DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 16 + 2 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 3 * DFX_MAX_GRU_LAYERS * DFX_SEQ_GMAX + 64) * sizeof(unsigned int);
        const size_t psync_bytes = (size_t)DFX_MAX_GRU_LAYERS * (DFX_SEQ_GMAX / 2) * 48 * sizeof(unsigned int);
        if (hipMalloc(reinterpret_cast<void **>(&m->d_psync), psync_bytes) != hipSuccess || hipMemset(m->d_psync, 0, psync_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&m->d_sync), sync_bytes) != hipSuccess || hipMemset(m->d_sync, 0, sync_bytes) != hipSuccess ||
            dfx_env_err_words_alloc(&m->h_err, &m->d_err, 256) != hipSuccess) {
            dfx_model_free(m);
            DFX_FAIL(DFX_ERR_ALLOC, "dfx_model_create: device allocation failed");
        }
        const char *tc = getenv("DFX_TCHUNKS");
        if (tc && atoi(tc) >= 1) m->tchunks = atoi(tc) < DFX_MAX_TCHUNKS ? atoi(tc) : DFX_MAX_TCHUNKS;
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
    if (m->d_psync) (void)hipFree(m->d_psync);
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

// ---- the rest of this translation unit, by concern
#include "dfx_model_launch.h"
#include "dfx_model_forward.h"
#include "dfx_model_stream.h"
#include "dfx_model_enhance.h"
