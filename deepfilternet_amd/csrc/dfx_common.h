// Internal declarations shared by the dfx translation units (not part of the public ABI).
#pragma once

#include "dfx_env.h"

#include <cstdarg>
#include <string>
#include <vector>

#include "dfx.h"

// ---- error plumbing ---------------------------------------------------------------------------------------------
void dfx_set_error(const char *fmt, ...);
#define DFX_FAIL(code, ...)         \
    do {                            \
        dfx_set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)
#define DFX_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) DFX_FAIL(DFX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));   \
    } while (0)
#define DFX_LAUNCH_CHECK()                                                                              \
    do {                                                                                                \
        hipError_t e__ = hipGetLastError();                                                             \
        if (e__ != hipSuccess) DFX_FAIL(DFX_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e__)); \
    } while (0)
int dfx_require_device();

// ---- FFT plan (passed by value into kernels) ----------------------------------------------------------------------
#define DFX_MAX_STAGES 12
struct DfxFftPlan {
    int N;       // real FFT size
    int M;       // complex FFT size = N/2
    int nstage;  // number of Stockham passes
    int radix[DFX_MAX_STAGES];
};

// ---- handles ------------------------------------------------------------------------------------------------------
struct dfx_bands {
    int nb = 0;
    int F = 0;
    std::vector<uint64_t> widths;
    int *d_start = nullptr;               // [nb+1] first bin of each band
    float *d_invw = nullptr;              // [nb]   1/width (f32 division, lib.rs:287)
    unsigned char *d_bin2band = nullptr;  // [F]
    int *d_segtab = nullptr;              // [3*64 + nb + 1] the bands cut into <= 64 near-equal segments (dfx_k_analysis), nseg > 0
    int nseg = 0;
    int segcap = 0, segparts = 0;         // longest segment in bins, most segments of one band
};

struct dfx_state {
    int sr = 0, N = 0, hop = 0, nb = 0, min_nb = 0;
    float wnorm = 0.f;
    DfxFftPlan plan{};
    std::vector<float> window_host;
    float *d_window = nullptr;  // [N]
    float2 *d_tw = nullptr;     // [N] exp(-2*pi*i*k/N)
    unsigned char *d_mfft = nullptr;   // N == 960 only: tables of the matrix-pipe 480-point transform, forward then inverse (DFX_MFFT_TABLE_BYTES each)
    dfx_bands *bands = nullptr;
};

static inline hipStream_t dfx_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t dfx_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- per-kernel timing (dfx_prof_* in dfx.h): hipEvents recorded on the launch stream around a kernel launch ------
enum DfxKernelId {
    DFX_K_ANALYSIS = 0, DFX_K_ANALYSIS_MEM, DFX_K_NORM_SCAN, DFX_K_SYNTHESIS, DFX_K_ERB, DFX_K_ERB_INV, DFX_K_DF_APPLY,
    DFX_K_CONV_IN_ERB, DFX_K_PWCONV, DFX_K_CONV_OUT, DFX_K_DF_CONVP, DFX_K_GGEMM, DFX_K_GRU_REC, DFX_K_LSNR, DFX_K_ADD,
    DFX_K_COPY_ROWS, DFX_K_CONV_IN_DF, DFX_K_PROJ, DFX_K_ERB_ENC, DFX_K_ERB_DEC, DFX_K_RESAMPLE, DFX_K_PCM, DFX_K_MF, DFX_K_EMB_FAN, DFX_K_ERB_TAIL, DFX_K_COUNT
};
bool dfx_prof_on(int kernel_id);
void dfx_prof_begin(int kernel_id, hipStream_t s);
void dfx_prof_end(int kernel_id, hipStream_t s);
// RAII: put one in the scope of a dfx_launch() call
struct DfxKScope {
    int id;
    hipStream_t s;
    bool on;
    DfxKScope(int id_, hipStream_t s_) : id(id_), s(s_), on(dfx_prof_on(id_)) {
        if (on) dfx_prof_begin(id, s);
    }
    ~DfxKScope() {
        if (on) dfx_prof_end(id, s);
    }
};

// ---- internal launchers shared between the DSP API and the model --------------------------------------------------
// x_len < T: the samples [x_len, T) of every row are implicit zeros (x_stride may then be as small as x_len); -1 = T
// spec_stride: row stride of spec in complex elements (0: F)
int dfx_launch_analysis(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride,
                        const float *mem_in, float *mem_out, float *spec, float *erb_db, hipStream_t s, int64_t x_len = -1,
                        int64_t spec_stride = 0, bool x_i16 = false);   // x_i16: x points at int16_t PCM samples (x / 32768 on the way in; no memories)
// only the analysis memory (the last N - hop samples in front of the next call's first hop) of a call over T samples per row
int dfx_launch_analysis_mem(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride, const float *mem_in, float *mem_out,
                            hipStream_t s);
int dfx_features_padded(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_len, int64_t x_stride, int nb_df,
                        float alpha, float *spec, float *erb_feat, float *spec_feat, void *stream, int64_t spec_stride = 0, bool x_i16 = false);
// dfx_synthesis storing only stream samples [out_skip, out_skip + out_len) of every row, at out[row * out_stride + n - out_skip]
int dfx_launch_synthesis(const dfx_state *st, const float *spec, int64_t B, int64_t Tf, const float *mem_in, float *mem_out,
                         float *out, int64_t out_stride, int64_t out_skip, int64_t out_len, hipStream_t s, int64_t f_begin = 0,
                         int64_t f_end = -1,   // only output frames [f_begin, f_end) (time-chunked finishing)
                         int64_t spec_stride = 0,   // row stride of spec in complex elements (0: F)
                         bool out_i16 = false);     // out points at int16_t PCM samples ((x * 2^15).to(int16) on the way out; no memories)
// The finishing pass of enhance() in one kernel (dfx_k_synthesis_rows): Mask + MF.DF [+ post filter + attenuation limit] applied on the way
// into the ISTFT, whole rows, no STFT memories.  coefs == null: spec IS the enhanced spectrum (no deep filter, the chunk carry alone).
// dfx_synthesis_rows_ok: the configuration the kernel is written for (N = 960, hop = 480, the in-place plan; order 5 taps in the tap-major
// layout, nb_df <= 128, <= 64 bands).
bool dfx_synthesis_rows_ok(const dfx_state *st, bool with_df, int order, int nb_df, int nbands);
int dfx_launch_synthesis_rows(const dfx_state *st, const float *spec, int64_t spec_stride, const float *coefs, int nb_df, int order, int lookahead,
                              const float *gains, float pf_beta, float atten_lim, int64_t B, int64_t Tf, float *out, int64_t out_stride,
                              int64_t out_skip, int64_t out_len, hipStream_t s, bool out_i16 = false,
                              const unsigned int *err = nullptr,    // the model's error words (host memory) and a device word for their verdict: a pass in
                              unsigned int *poison = nullptr);      // which a kernel raised a fault stores NaN (dfx_k_fault_mirror)
int dfx_launch_norm_scan(const float *erb_in, float *erb_out, int E, const float *spec_in, int64_t spec_frame_stride,
                         float *spec_out, int Fn, int64_t C, int64_t T, float alpha, float *erb_state,
                         float *unit_state, hipStream_t s, int64_t erb_out_cs = 0,   // > 0: floats between the clips of erb_out / spec_out (< 16 frames)
                         int64_t spec_out_cs = 0);
// Mask + MF.DF + post filter + atten_lim on frames [t_begin, t_end) of every clip (t_end < 0: T); coef_T: frames per clip of the
// coefficient / gain arrays (default T); out_T / out_toff: compacted output rows (default T / 0); spec_stride / out_stride: row
// strides in complex elements (0: F; even strides = 16-byte aligned rows take the row-streaming kernel)
int dfx_launch_df_apply(const float *spec, const float *coefs, int coef_layout, const float *gains, const dfx_bands *bands, int64_t B,
                        int64_t T, int F, int nb_df, int order, int lookahead, float pf_beta, float atten_lim, float *out, hipStream_t s,
                        int64_t t_begin = 0, int64_t t_end = -1, int64_t coef_T = -1, int64_t out_T = -1, int64_t out_toff = 0,
                        int64_t spec_stride = 0, int64_t out_stride = 0,
                        int pf_rs_channels = 0);  // > 0: the real-time runtime's post filter (lib.rs:446-471) over frames of that many rows
