// dfx: enhance() as one C call (dfx_enhance / dfx_enhance_pcm16: workspace plan, STFT features, forward pass with the fused finishing kernel).
// A part of dfx_model.hip (one translation unit: included from there, in this order — launch helpers, forward pass, streaming, enhance()).
#pragma once

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
