// Tensor manifest of a DeepFilterNet3 checkpoint as the engine consumes it: reference state-dict names, shapes and the
// offset of each tensor inside the packed float32 blob handed to dfx_model_create().
//
// Names follow the reference's nn.Sequential indexing (DeepFilterNet/df/modules.py:18-126): a ConstantPad2d occupies
// index 0 only when the time kernel is > 1; the 1x1 pointwise conv exists only when groups > 1 and max(kernel) > 1.
// Structure: DeepFilterNet/df/deepfilternet3.py:100-185 (Encoder), :188-254 (ErbDecoder), :278-331 (DfDecoder).
#pragma once

#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

#include "dfx.h"

struct DfxTensor {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {1, 1, 1, 1};
    int64_t offset = 0;  // floats
    int64_t numel() const { return shape[0] * shape[1] * shape[2] * shape[3]; }
};

struct DfxManifest {
    std::vector<DfxTensor> t;
    int64_t total = 0;
    void add(const std::string &name, std::initializer_list<int64_t> shp) {
        DfxTensor x;
        x.name = name;
        x.ndim = (int)shp.size();
        int i = 0;
        for (int64_t v : shp) x.shape[i++] = v;
        x.offset = total;
        total += x.numel();
        t.push_back(x);
    }
    const DfxTensor *find(const std::string &name) const {
        for (auto &x : t)
            if (x.name == name) return &x;
        return nullptr;
    }
};

static inline int dfx_gcd(int a, int b) { return std::gcd(a, b); }

// keys of one Conv2dNormAct / ConvTranspose2dNormAct(bias=False) + BatchNorm2d
static inline void dfx_manifest_conv(DfxManifest &m, const std::string &p, int in_ch, int out_ch, int kt, int kf,
                                     bool transposed) {
    int idx = kt > 1 ? 1 : 0;
    const int groups = dfx_gcd(in_ch, out_ch);
    bool sep = groups > 1;
    if (!transposed && (kt > kf ? kt : kf) == 1) sep = false;
    if (transposed) m.add(p + "." + std::to_string(idx) + ".weight", {in_ch, out_ch / groups, kt, kf});
    else m.add(p + "." + std::to_string(idx) + ".weight", {out_ch, in_ch / groups, kt, kf});
    ++idx;
    if (sep) {
        m.add(p + "." + std::to_string(idx) + ".weight", {out_ch, out_ch, 1, 1});
        ++idx;
    }
    const std::string bn = p + "." + std::to_string(idx);
    m.add(bn + ".weight", {out_ch});
    m.add(bn + ".bias", {out_ch});
    m.add(bn + ".running_mean", {out_ch});
    m.add(bn + ".running_var", {out_ch});
}

static inline void dfx_manifest_gru(DfxManifest &m, const std::string &p, int H, int layers) {
    for (int l = 0; l < layers; ++l) {
        const std::string s = std::to_string(l);
        m.add(p + ".weight_ih_l" + s, {3 * H, H});
        m.add(p + ".weight_hh_l" + s, {3 * H, H});
        m.add(p + ".bias_ih_l" + s, {3 * H});
        m.add(p + ".bias_hh_l" + s, {3 * H});
    }
}

static inline DfxManifest dfx_build_manifest(const dfx_model_cfg &c) {
    DfxManifest m;
    const int C = c.conv_ch, E = c.nb_erb, Fd = c.nb_df, O = c.df_order, H = c.emb_hidden_dim;
    const int emb = C * E / 4;
    auto glin = [&](const std::string &name, int I, int Hh, int G) { m.add(name, {G, I / G, Hh / G}); };
    dfx_manifest_conv(m, "enc.erb_conv0", 1, C, 3, 3, false);
    dfx_manifest_conv(m, "enc.erb_conv1", C, C, 1, 3, false);
    dfx_manifest_conv(m, "enc.erb_conv2", C, C, 1, 3, false);
    dfx_manifest_conv(m, "enc.erb_conv3", C, C, 1, 3, false);
    dfx_manifest_conv(m, "enc.df_conv0", 2, C, 3, 3, false);
    dfx_manifest_conv(m, "enc.df_conv1", C, C, 1, 3, false);
    glin("enc.df_fc_emb.0.weight", C * Fd / 2, emb, c.enc_lin_groups);
    glin("enc.emb_gru.linear_in.0.weight", c.enc_concat ? 2 * emb : emb, H, c.lin_groups);
    dfx_manifest_gru(m, "enc.emb_gru.gru", H, 1);
    if (c.emb_gru_skip_enc == DFX_SKIP_GROUPEDLINEAR) glin("enc.emb_gru.gru_skip.weight", emb, emb, c.lin_groups);
    glin("enc.emb_gru.linear_out.0.weight", H, emb, c.lin_groups);
    m.add("enc.lsnr_fc.0.weight", {1, emb});
    m.add("enc.lsnr_fc.0.bias", {1});
    glin("erb_dec.emb_gru.linear_in.0.weight", emb, H, c.lin_groups);
    dfx_manifest_gru(m, "erb_dec.emb_gru.gru", H, c.emb_num_layers - 1);
    if (c.emb_gru_skip == DFX_SKIP_GROUPEDLINEAR) glin("erb_dec.emb_gru.gru_skip.weight", emb, emb, c.lin_groups);
    glin("erb_dec.emb_gru.linear_out.0.weight", H, emb, c.lin_groups);
    dfx_manifest_conv(m, "erb_dec.conv3p", C, C, 1, 1, false);
    dfx_manifest_conv(m, "erb_dec.convt3", C, C, 1, 3, false);
    dfx_manifest_conv(m, "erb_dec.conv2p", C, C, 1, 1, false);
    dfx_manifest_conv(m, "erb_dec.convt2", C, C, 1, 3, true);
    dfx_manifest_conv(m, "erb_dec.conv1p", C, C, 1, 1, false);
    dfx_manifest_conv(m, "erb_dec.convt1", C, C, 1, 3, true);
    dfx_manifest_conv(m, "erb_dec.conv0p", C, C, 1, 1, false);
    dfx_manifest_conv(m, "erb_dec.conv0_out", C, 1, 1, 3, false);
    dfx_manifest_conv(m, "df_dec.df_convp", C, 2 * O, c.df_pathway_kernel_size_t, 1, false);
    glin("df_dec.df_gru.linear_in.0.weight", emb, c.df_hidden_dim, 8);  // SqueezedGRU_S default linear_groups=8
    dfx_manifest_gru(m, "df_dec.df_gru.gru", c.df_hidden_dim, c.df_num_layers);
    if (c.df_gru_skip == DFX_SKIP_GROUPEDLINEAR) glin("df_dec.df_skip.weight", emb, c.df_hidden_dim, c.lin_groups);
    glin("df_dec.df_out.0.weight", c.df_hidden_dim, Fd * 2 * O, c.lin_groups);
    return m;
}
