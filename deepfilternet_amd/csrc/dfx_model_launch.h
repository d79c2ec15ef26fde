// dfx: the launchers of the network's kernels (one function per kernel family: arguments, grids, LDS sizes).
// A part of dfx_model.hip (one translation unit: included from there, in this order — launch helpers, forward pass, streaming, enhance()).
#pragma once

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
    constexpr bool staged = true;
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
    constexpr bool staged = true;
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
            const bool wide = dfx_ceil_div(B, DFX_PH_BM) * 4 * (twin ? 2 : 1) >= dfx_env_num_cus();
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
