// The reference's C API (libDF/src/capi.rs) on top of the dfx streaming runtime, and the .dfx model file it loads.
#include <cmath>
#include <deque>
#include <string>

#include "dfx_common.h"
#include "df_capi.h"

// ---------------------------------------------------------------------------------------------------- .dfx model files
// "DFXM" | u32 version | u32 sizeof(dfx_model_cfg) | dfx_model_cfg | i64 n_floats | float32[n_floats]   (little endian)
// The floats are the raw reference state-dict tensors packed per dfx_model_tensor_info — what dfx_model_create takes.
static const char DFX_FILE_MAGIC[4] = {'D', 'F', 'X', 'M'};
static const uint32_t DFX_FILE_VERSION = 2;  // 2: dfx_model_cfg grew emb_gru_skip_enc / emb_gru_skip / enc_concat

extern "C" int dfx_model_save_file(const dfx_model_cfg *cfg, const float *blob_host, const char *path) {
    if (!cfg || !blob_host || !path) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_save_file: null argument");
    int64_t n = 0;
    if (int rc = dfx_model_blob_floats(cfg, &n)) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_save_file: cannot open '%s' for writing", path);
    const uint32_t ver = DFX_FILE_VERSION, csz = (uint32_t)sizeof(dfx_model_cfg);
    bool ok = fwrite(DFX_FILE_MAGIC, 1, 4, f) == 4 && fwrite(&ver, 4, 1, f) == 1 && fwrite(&csz, 4, 1, f) == 1 &&
              fwrite(cfg, sizeof(*cfg), 1, f) == 1 && fwrite(&n, 8, 1, f) == 1 && fwrite(blob_host, 4, (size_t)n, f) == (size_t)n;
    ok = (fclose(f) == 0) && ok;
    if (!ok) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_save_file: short write to '%s'", path);
    return DFX_OK;
}

static int read_model_file(const char *path, dfx_model_cfg *cfg, std::vector<float> *blob) {
    FILE *f = fopen(path, "rb");
    if (!f) DFX_FAIL(DFX_ERR_INVALID_ARG, "cannot open model file '%s'", path);
    char magic[4];
    uint32_t ver = 0, csz = 0;
    int64_t n = 0, want = -1;
    // version 1 = version 2 without the last three configuration fields (emb_gru_skip_enc, emb_gru_skip, enc_concat: all "none" / off then)
    const size_t v1_cfg = sizeof(dfx_model_cfg) - 3 * sizeof(int32_t);
    memset(cfg, 0, sizeof(*cfg));
    bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, DFX_FILE_MAGIC, 4) == 0 && fread(&ver, 4, 1, f) == 1 && fread(&csz, 4, 1, f) == 1;
    if (ok && ver != 1 && ver != DFX_FILE_VERSION) {
        fclose(f);
        DFX_FAIL(DFX_ERR_INVALID_ARG, "'%s' is a .dfx model file of version %u; this library reads versions 1 and %u: re-export it with export_dfx", path, ver,
                 DFX_FILE_VERSION);
    }
    ok = ok && csz == (ver == 1 ? v1_cfg : sizeof(dfx_model_cfg)) && fread(cfg, csz, 1, f) == 1 && fread(&n, 8, 1, f) == 1;
    if (ok) ok = dfx_model_blob_floats(cfg, &want) == DFX_OK && want == n;
    if (ok) {
        blob->resize((size_t)n);
        ok = fread(blob->data(), 4, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    if (!ok) DFX_FAIL(DFX_ERR_INVALID_ARG, "'%s' is not a valid .dfx model file (version %u)", path, DFX_FILE_VERSION);
    return DFX_OK;
}

int dfx_read_onnx_targz(const char *path, dfx_model_cfg *cfg, std::vector<float> *blob, std::string *version);  // dfx_onnx.hip

// .dfx file or the reference's <model>_onnx.tar.gz (gzip magic 1f 8b); `version` receives the tar's version.txt
static int load_model_any(const char *path, dfx_model **out, std::string *version) {
    if (!path || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_model_load_file: null argument");
    unsigned char magic[2] = {0, 0};
    FILE *f = fopen(path, "rb");
    if (!f) DFX_FAIL(DFX_ERR_INVALID_ARG, "cannot open model file '%s'", path);
    const size_t got = fread(magic, 1, 2, f);
    fclose(f);
    dfx_model_cfg cfg;
    std::vector<float> blob;
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        if (int rc = dfx_read_onnx_targz(path, &cfg, &blob, version)) return rc;
    } else if (int rc = read_model_file(path, &cfg, &blob)) {
        return rc;
    }
    return dfx_model_create(&cfg, blob.data(), out);
}

extern "C" int dfx_model_load_file(const char *path, dfx_model **out) { return load_model_any(path, out, nullptr); }

// ---------------------------------------------------------------------------------------------------- df_* (capi.rs)
struct DFState {
    dfx_model *model = nullptr;
    dfx_state *st = nullptr;
    dfx_stream_state *rt = nullptr;
    float *d_io = nullptr;  // [hop] in, [hop] out, [1] lsnr
    float *d_raw = nullptr; // df_process_frame_raw: spectrum, gains, coefficients, lsnr, stage flags
    int hop = 0;
    bool logging = false;
    std::deque<std::string> log;
    void msg(const char *level, const std::string &text) {
        if (logging) log.push_back(std::string(level) + " | DF | " + text);  // capi.rs:47-57: "{level} | {target} | {message}"
    }
};

static void df_destroy(DFState *s) {
    if (!s) return;
    if (s->rt) dfx_stream_free(s->rt);
    if (s->st) dfx_state_free(s->st);
    if (s->model) dfx_model_free(s->model);
    if (s->d_io) (void)hipFree(s->d_io);
    if (s->d_raw) (void)hipFree(s->d_raw);
    delete s;
}

static void df_panic(const char *what) {  // the reference's `expect(...)`: message on stderr, then abort
    fprintf(stderr, "%s: %s\n", what, dfx_last_error());
    abort();
}

extern "C" DFState *df_create(const char *path, float atten_lim, const char *log_level) {
    if (!path) {
        dfx_set_error("df_create: null path");
        return nullptr;
    }
    DFState *s = new DFState();
    s->logging = log_level != nullptr;
    dfx_model_cfg c;
    std::string version;
    bool ok = load_model_any(path, &s->model, &version) == DFX_OK && dfx_model_cfg_get(s->model, &c) == DFX_OK &&
              dfx_state_create(c.sr, c.fft_size, c.hop_size, c.nb_erb, c.min_nb_freqs, &s->st) == DFX_OK &&
              dfx_stream_create(s->model, s->st, 1, 1, &s->rt) == DFX_OK &&
              dfx_stream_set_gating(s->rt, 1) == DFX_OK &&
              // capi.rs:27-34: with_thresholds(-15, 35, 35), with_post_filter(0), with_mask_reduce(ReduceMask::MAX)
              dfx_stream_set_thresholds(s->rt, -15.f, 35.f, 35.f) == DFX_OK && dfx_stream_set_channels(s->rt, 1, 1) == DFX_OK &&
              dfx_stream_set_post_filter_beta(s->rt, 0.f) == DFX_OK && dfx_stream_set_atten_lim(s->rt, atten_lim) == DFX_OK;
    if (ok) {
        s->hop = c.hop_size;
        ok = hipMalloc(reinterpret_cast<void **>(&s->d_io), ((size_t)2 * s->hop + 1) * sizeof(float)) == hipSuccess;
        if (!ok) dfx_set_error("df_create: device allocation failed");
    }
    if (!ok) {
        df_destroy(s);
        return nullptr;
    }
    if (!version.empty()) s->msg("INFO", "Loading model with id: " + version);  // tract.rs:56-59
    char buf[160];
    snprintf(buf, sizeof(buf), "Running with model type deepfilternet3 lookahead %d", dfx_stream_delay_frames(s->rt));  // tract.rs:318-322
    s->msg("INFO", buf);
    const float lim = fabsf(atten_lim);
    if (lim >= 100.f) {
    } else if (lim < 0.01f) {
        s->msg("WARN", "Attenuation limit too strong. No noise reduction will be performed");  // tract.rs:291-293
    } else {
        snprintf(buf, sizeof(buf), "Running with an attenuation limit of %.0f dB", lim);  // tract.rs:295
        s->msg("INFO", buf);
    }
    return s;
}

extern "C" size_t df_get_frame_length(DFState *st) {
    if (!st) df_panic("Invalid pointer");
    return (size_t)st->hop;
}

extern "C" char *df_next_log_msg(DFState *st) {
    if (!st) df_panic("Invalid pointer");
    if (st->log.empty()) return nullptr;
    const std::string m = st->log.front();
    st->log.pop_front();
    char *out = static_cast<char *>(malloc(m.size() + 1));
    if (out) memcpy(out, m.c_str(), m.size() + 1);
    return out;
}

extern "C" void df_free_log_msg(char *ptr) { free(ptr); }

extern "C" void df_set_atten_lim(DFState *st, float lim_db) {
    if (!st) df_panic("Invalid pointer");
    (void)dfx_stream_set_atten_lim(st->rt, lim_db);
}

extern "C" void df_set_post_filter_beta(DFState *st, float beta) {
    if (!st) df_panic("Invalid pointer");
    if (beta < 0.f) {  // tract.rs:380-384
        st->msg("WARN", "Post-filter beta cannot be smaller than 0.");
        beta = 0.f;
    }
    (void)dfx_stream_set_post_filter_beta(st->rt, beta);
}

extern "C" float df_process_frame(DFState *st, float *input, float *output) {
    if (!st || !input || !output) df_panic("Invalid pointer");
    const size_t hb = (size_t)st->hop * sizeof(float);
    float *dx = st->d_io, *dy = st->d_io + st->hop, *dl = st->d_io + 2 * st->hop;
    float lsnr = 0.f;
    if (hipMemcpy(dx, input, hb, hipMemcpyHostToDevice) != hipSuccess) df_panic("Failed to process DF frame (upload)");
    if (dfx_stream_process(st->rt, dx, 1, dy, dl, nullptr) != DFX_OK) df_panic("Failed to process DF frame");
    if (hipMemcpy(output, dy, hb, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&lsnr, dl, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        df_panic("Failed to process DF frame (download)");
    // the downloads above waited for the pass: a fault one of its kernels raised (invalid results) ends the process like the reference's panics
    if (dfx_model_poll(st->model) != DFX_OK) df_panic(dfx_last_error());
    return lsnr;
}

// capi.rs:172-210.  input [n_freqs, 2]; *out_gains_p -> [nb_erb], *out_coefs_p -> [df_order, nb_df, 2] (the shape the reference
// documents and views its output buffer with, capi.rs:183-184,205-206); a pointer is set to NULL when its stage did not run.
extern "C" float df_process_frame_raw(DFState *st, float *input, float **out_gains_p, float **out_coefs_p) {
    if (!st || !input || !out_gains_p || !out_coefs_p) df_panic("Invalid pointer");
    dfx_model_cfg c;
    if (dfx_model_cfg_get(st->model, &c) != DFX_OK) df_panic("Failed to process DF spectral frame");
    const size_t F = (size_t)c.fft_size / 2 + 1, ng = (size_t)c.nb_erb, nc = (size_t)c.df_order * c.nb_df * 2;
    if (!st->d_raw) {
        if (hipMalloc(reinterpret_cast<void **>(&st->d_raw), (F * 2 + ng + nc + 2) * sizeof(float)) != hipSuccess) df_panic("Failed to set input spectrum");
    }
    float *dspec = st->d_raw, *dg = dspec + F * 2, *dc = dg + ng, *dl = dc + nc;
    unsigned char *dflag = reinterpret_cast<unsigned char *>(dl + 1);
    if (hipMemcpy(dspec, input, F * 2 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) df_panic("Failed to set input spectrum");
    if (dfx_stream_process_raw(st->rt, dspec, dg, dc, dflag, dl, nullptr) != DFX_OK) df_panic("Failed to process DF spectral frame");
    float lsnr = 0.f;
    unsigned char flag = 0;
    if (hipMemcpy(&lsnr, dl, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&flag, dflag, 1, hipMemcpyDeviceToHost) != hipSuccess)
        df_panic("Failed to process DF spectral frame (download)");
    if (dfx_model_poll(st->model) != DFX_OK) df_panic(dfx_last_error());
    if ((flag & 2) && *out_gains_p) {
        if (hipMemcpy(*out_gains_p, dg, ng * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) df_panic("Failed to process DF spectral frame (download)");
    } else {
        *out_gains_p = nullptr;
    }
    if ((flag & 8) && *out_coefs_p) {
        if (hipMemcpy(*out_coefs_p, dc, nc * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) df_panic("Failed to process DF spectral frame (download)");
    } else {
        *out_coefs_p = nullptr;
    }
    return lsnr;
}

extern "C" void df_free(DFState *model) { df_destroy(model); }
