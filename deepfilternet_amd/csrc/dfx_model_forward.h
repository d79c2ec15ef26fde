// dfx: DfNet.forward as a sequence of launches (forward_impl: front, GRU phase in its persistent / event-synchronised / serial form, finishing),
// whose turn it is (DfxTurn) and dfx_model_forward.
// A part of dfx_model.hip (one translation unit: included from there, in this order — launch helpers, forward pass, streaming, enhance()).
#pragma once

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
            DfxKScope ks(DFX_K_GGEMM, st);
            // (weight fragments resident in registers when a wave's share fits: <= 16 groups of <= 4 tiles — every shipped shape)
            const bool resident = m->dfo_nu == 4 && A.G <= 16 && (int64_t)(NO / 2) * 16 * (Fd / 2) <= (int64_t)DFX_DFO_NPT * DFX_DFO_THREADS;
            if (resident) {
                DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_df_out_h3r<4>, smem));
                dfx_launch(dfx_k_df_out_h3r<4>, dim3((unsigned)nn_grid(dfx_ceil_div(M, 16), 2)), dim3(DFX_DFO_THREADS), smem, st, A);
            } else {
                DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_df_out_h3, smem));
                dfx_launch(dfx_k_df_out_h3, dim3((unsigned)nn_grid(dfx_ceil_div(M, 16), 2)), dim3(DFX_DFO_THREADS), smem, st, A);
            }
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
            // the recurrences on pairs of CUs (dfx_gru_pair.h): 32 clips per pair, W_hh resident; fp16-split arithmetic only (the exact form's fragments are the same bytes, its matrix ops are not)
            const bool use_pair = m->sw.gru_pair && !m->exact_fp32 && groups >= 2 && m->d_psync;
            if (nfollow || use_pair) m->seq_pbase += (unsigned int)T + 1u;
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
                if (xtab) S.xtab = xtab, S.xstride = DFX_SEQ_GMAX, S.xstat = xstat;
                S.xtag = xtag, S.psync = m->d_psync, S.pair_far = m->sw.gru_pair_far ? 1 : 0;
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
                if (use_pair) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_gru_seq_p2, DFX_GP_SMEM));
                else DFX_HIP(dfx_env_set_max_dyn_smem(m->exact_fp32 ? (const void *)dfx_k_gru_seq_x32 : (const void *)dfx_k_gru_seq, DFX_GH_SMEM));
                DfxKScope ks(DFX_K_GRU_REC, G);
                if (use_pair) dfx_launch(dfx_k_gru_seq_p2, dim3(dfx_gp_grid(nl * (int)((groups + 1) / 2))), dim3(DFX_GP_THREADS), DFX_GP_SMEM, G, S);
                else if (m->exact_fp32) dfx_launch(dfx_k_gru_seq_x32, dim3((unsigned)(nl * groups)), dim3(DFX_GH_THREADS), DFX_GH_SMEM, G, S);
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
                // Round 6: linear_out of chunk k + 1 runs beside the decoder tail of chunk k.  On one stream (wait -> linear_out -> tail: 0.8 ms per
                // chunk under the phase's load against a chunk every 0.67 ms) the ERB tail fell two chunks behind the chain and ended 1.0 ms after it
                // (profiles/r06_timeline.txt).  With followers the projection stream of the decoder's second layer has nothing to carry: it takes the
                // flag wait and linear_out, an event per chunk hands demb's rows over (12.34 -> 12.22 ms per step, same box; the ERB tail now ends
                // 0.37 ms behind the chain, the three df_out launches that wait for all of df_convp 0.8 ms: profiles/r06_tail_split.log).
                hipStream_t Gq = (m->sw.tail_split && ndec >= 2 && followed[2]) ? ln->ps[2] : Eq;
                if ((r = launch_wait_ge(m, donep(ndec), groups, tgt(k), Gq))) return r;
                if (dfx_dev_skip() & 1) return DFX_OK;
                if ((r = dec_out_skip(ws + w.py[ndec], Rk, Gq, rm))) return r;
                if (Gq != Eq && ((r = esig(ln->pev[2][k], Gq)) || (r = ewait(ln->pev[2][k], Eq)))) return r;
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
    if (fin && m->fuse_dfa && dfx_synthesis_rows_ok(fin->st, true, O, run_df ? Fd : 0, E) && bands == fin->st->bands && sstride % 2 == 0 && sstride > 0) {
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
