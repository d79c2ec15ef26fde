// dfx: the frame-by-frame runtime (dfx_stream_*: DfTract::process for many lockstep streams).
// A part of dfx_model.hip (one translation unit: included from there, in this order — launch helpers, forward pass, streaming, enhance()).
#pragma once

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
        static const int lin_env = [] { const char *e = getenv("DFX_STREAM_LINEAR"); return e ? atoi(e) : 1; }();   // test hook: 0 = ring form, n > 1 = slack of n frames (the wrap of the linear buffers every few hops)
        int64_t slack = lin_env > 1 ? lin_env : 32;   // (DFX_STREAM_LINEAR=0: ring form only; = n > 1: slack of n frames, tests)
        while (slack > Hs + n && (size_t)B * (Hs + n + slack) * F * 8 > ((size_t)1 << 30)) slack /= 2;
        if (slack < Hs + n) slack = Hs + n;
        if (lin_env && (size_t)B * (Hs + n + slack) * F * 8 <= ((size_t)3 << 29)) {
            s->lin_cap = Hs + n + slack;
            s->spec_lin = take((size_t)B * s->lin_cap * F * 8);
            constexpr bool feat_env = true;
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
        constexpr bool ring_env = true;
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
    constexpr bool step_env = true;
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
    constexpr bool side_env = true;
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
