// The steps either side of enhance() in the reference's file loop (df/enhance.py:73-89 main(): load_audio -> enhance -> resample back ->
// save_audio; df/io.py:25-116): PCM16 <-> float scaling and the windowed-sinc sample-rate conversion, on the device, so that a
// file -> file pipeline keeps its audio in HBM from decode to encode.
//
// The rate conversion is torchaudio.functional.resample (third party, torchaudio 2.x `_get_sinc_resample_kernel` /
// `_apply_sinc_resample_kernel`; absent from this image) with the parameter sets of df/io.py:92-111:
//   g = gcd(orig, new); orig /= g; new /= g; base = min(orig, new) * rolloff; width = ceil(lowpass_filter_width * orig / base)
//   W[j][k] = window(t) * sinc(pi t) * base / orig,   t = clamp((-j / new + (k - width) / orig) * base, +-lowpass_filter_width)
//             j in [0, new), k in [0, 2 width + orig); hann: window = cos(pi t / (2 lpw))^2; kaiser: I0(beta sqrt(1 - (t/lpw)^2)) / I0(beta)
//   y[n new + j] = sum_k W[j][k] xpad[n orig + k],  xpad = [0]*width ++ x ++ [0]*(width + orig),  length ceil(new T / orig)
// The bank is computed in float64 and rounded to float32 like torchaudio does for float32 input.
#include <cmath>

#include "dfx_common.h"

#define DFX_RS_JT 16  // output phases per register tile

struct dfx_resampler {
    int orig_sr = 0, new_sr = 0;
    int orig = 0, nw = 0, width = 0, K = 0, K4 = 0, nw_pad = 0;
    float *d_wt = nullptr;  // [K4][nw_pad]: W transposed, taps padded to a multiple of 4 and phases to a multiple of DFX_RS_JT (zeros)
    std::vector<float> w_host;  // [nw][K]
};

struct DfxRsArgs {
    int64_t B, T, x_stride, y_stride, out_len, frames;  // frames = ceil(out_len / nw) per clip
    int orig, nw, nw_pad, width, K4, tn, fw, jw;        // K4 = taps padded to a multiple of 4 (zero taps); tn = 64*fw frames per workgroup
};

// One thread per output frame n (nw consecutive output samples).  A workgroup is 4 waves = fw groups of 64 frames x jw groups of
// phase tiles (fw * jw == 4): the input segment of its 64*fw frames sits in LDS once (lane n reads xs[n*orig + k]: stride orig
// words, conflict-free when orig is odd) and is shared by the jw waves that split the DFX_RS_JT-phase register tiles between them;
// the filter taps are wave-uniform (s_load through the scalar cache); four taps per iteration: 4 LDS reads per 4*DFX_RS_JT FMAs.
// x [B, x_stride], y [B, y_stride], wt [K4][nw_pad] are separate __restrict__ parameters: only then may the compiler read the (never
// written) filter through s_load instead of 64 identical vector loads.
__global__ void __launch_bounds__(256) dfx_k_resample(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ wt,
                                                      DfxRsArgs A) {
    DFX_DYN_SMEM(float, xs);
    const int64_t tiles = (A.frames + A.tn - 1) / A.tn;
    const int64_t b = blockIdx.x / tiles;
    const int64_t n0 = (int64_t)(blockIdx.x - b * tiles) * A.tn;
    if (b >= A.B) return;
    const float *xb = x + b * A.x_stride;
    const int seg = (A.tn - 1) * A.orig + A.K4;
    const int64_t s0 = n0 * A.orig - A.width;  // stream index of xs[0]
    // eight loads in flight per lane before the first LDS store (a load -> store loop would wait out one memory latency per element)
    for (int i = threadIdx.x; i < seg; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t si = s0 + i + 256 * u;
            v[u] = (i + 256 * u < seg && si >= 0 && si < A.T) ? xb[si] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i + 256 * u < seg) xs[i + 256 * u] = v[u];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wf = dfx_wave_uniform(wave % A.fw), wj = dfx_wave_uniform(wave / A.fw);
    const int nl = wf * 64 + lane;
    const int64_t n = n0 + nl;
    if (n >= A.frames) return;
    const float *xp = xs + (int64_t)nl * A.orig;
    float *yb = y + b * A.y_stride + n * A.nw;
    const int64_t left = A.out_len - n * A.nw;  // output samples this frame may store
    for (int j0 = wj * DFX_RS_JT; j0 < A.nw; j0 += A.jw * DFX_RS_JT) {
        float acc[DFX_RS_JT];
#pragma unroll
        for (int i = 0; i < DFX_RS_JT; ++i) acc[i] = 0.f;
        const float *wp = wt + j0;
        for (int k = 0; k < A.K4; k += 4) {
            const float x0 = xp[k], x1 = xp[k + 1], x2 = xp[k + 2], x3 = xp[k + 3];
            const float *w0 = wp + (int64_t)k * A.nw_pad;
#pragma unroll
            for (int i = 0; i < DFX_RS_JT; ++i) {
                float a = acc[i];
                a = fmaf(w0[i], x0, a);
                a = fmaf(w0[A.nw_pad + i], x1, a);
                a = fmaf(w0[2 * A.nw_pad + i], x2, a);
                a = fmaf(w0[3 * A.nw_pad + i], x3, a);
                acc[i] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < DFX_RS_JT; ++i)
            if (j0 + i < A.nw && j0 + i < left) yb[j0 + i] = acc[i];
    }
}

// torchaudio.load's int16 normalisation (x / 32768) and save_audio's encoding (io.py:79-80: (audio * (1 << 15)).to(torch.int16):
// truncation toward zero; out-of-range values wrap like ATen's float -> int64 -> int16 conversion chain)
__global__ void dfx_k_pcm16_to_f32(const int16_t *in, int64_t n, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i] * (1.0f / 32768.0f);
}
__global__ void dfx_k_f32_to_pcm16(const float *in, int64_t n, int16_t *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = in[i] * 32768.0f;
        v = v != v ? 0.f : fminf(fmaxf(v, -9.0e18f), 9.0e18f);
        out[i] = (int16_t)(uint16_t)(uint64_t)(int64_t)v;
    }
}

static unsigned io_grid(int64_t n) {
    const int64_t cap = (int64_t)dfx_env_num_cus() * 8, want = dfx_ceil_div(n, 256);
    return (unsigned)(want < cap ? (want > 0 ? want : 1) : cap);
}

extern "C" int dfx_pcm16_to_f32(const int16_t *pcm, int64_t n, float *out, void *stream) {
    if (n < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_pcm16_to_f32: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (n == 0) return DFX_OK;
    if (!pcm || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_pcm16_to_f32: null buffer");
    DfxKScope ks(DFX_K_PCM, dfx_stream(stream));
    dfx_launch(dfx_k_pcm16_to_f32, dim3(io_grid(n)), dim3(256), 0, dfx_stream(stream), pcm, n, out);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_f32_to_pcm16(const float *x, int64_t n, int16_t *out, void *stream) {
    if (n < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_f32_to_pcm16: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (n == 0) return DFX_OK;
    if (!x || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_f32_to_pcm16: null buffer");
    DfxKScope ks(DFX_K_PCM, dfx_stream(stream));
    dfx_launch(dfx_k_f32_to_pcm16, dim3(io_grid(n)), dim3(256), 0, dfx_stream(stream), x, n, out);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

static double bessel_i0(double x) {  // torch.i0: power series (converges quickly for the beta range used here, <= ~15)
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

static int64_t gcd64(int64_t a, int64_t b) {
    while (b) {
        const int64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// the filter bank on the host: pure arithmetic, no device needed (dfx_resampler_kernel exposes it for the tests)
static int build_bank(dfx_resampler *r, int orig_sr, int new_sr, int lpw, double rolloff, int method, double beta) {
    if (orig_sr <= 0 || new_sr <= 0 || lpw <= 0 || rolloff <= 0.0 || rolloff > 1.0 || (method != 0 && method != 1))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resampler_create: bad arguments");
    const int64_t g = gcd64(orig_sr, new_sr);
    r->orig_sr = orig_sr, r->new_sr = new_sr;
    r->orig = (int)(orig_sr / g), r->nw = (int)(new_sr / g);
    const double base = (double)(r->orig < r->nw ? r->orig : r->nw) * rolloff;
    r->width = (int)std::ceil((double)lpw * r->orig / base);
    r->K = 2 * r->width + r->orig;
    r->K4 = (r->K + 3) / 4 * 4;
    r->nw_pad = (r->nw + DFX_RS_JT - 1) / DFX_RS_JT * DFX_RS_JT;
    if ((int64_t)r->K * r->nw_pad > ((int64_t)1 << 26)) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_resampler_create: filter bank too large (reduce the rates' ratio)");
    r->w_host.assign((size_t)r->nw * r->K, 0.f);
    const double pi = 3.14159265358979323846, scale = base / r->orig, i0b = method == 1 ? bessel_i0(beta) : 1.0;
    for (int j = 0; j < r->nw; ++j)
        for (int k = 0; k < r->K; ++k) {
            double t = ((double)(-j) / r->nw + (double)(k - r->width) / r->orig) * base;
            t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
            double win;
            if (method == 0) {
                const double c = std::cos(t * pi / lpw / 2.0);
                win = c * c;
            } else {
                const double u = t / lpw;
                win = bessel_i0(beta * std::sqrt(1.0 - u * u > 0.0 ? 1.0 - u * u : 0.0)) / i0b;
            }
            const double tp = t * pi;
            const double s = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
            r->w_host[(size_t)j * r->K + k] = (float)(s * win * scale);
        }
    return DFX_OK;
}

extern "C" int dfx_resampler_create(int orig_sr, int new_sr, int lowpass_filter_width, double rolloff, int method, double beta,
                                    dfx_resampler **out) {
    if (!out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resampler_create: null out");
    dfx_resampler *r = new dfx_resampler();
    if (int rc = build_bank(r, orig_sr, new_sr, lowpass_filter_width, rolloff, method, beta)) {
        delete r;
        return rc;
    }
    if (int rc = dfx_require_device()) {
        delete r;
        return rc;
    }
    std::vector<float> wt((size_t)r->K4 * r->nw_pad, 0.f);
    for (int j = 0; j < r->nw; ++j)
        for (int k = 0; k < r->K; ++k) wt[(size_t)k * r->nw_pad + j] = r->w_host[(size_t)j * r->K + k];
    if (hipMalloc(reinterpret_cast<void **>(&r->d_wt), wt.size() * 4) != hipSuccess) {
        delete r;
        DFX_FAIL(DFX_ERR_ALLOC, "dfx_resampler_create: device allocation failed");
    }
    if (hipMemcpy(r->d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(r->d_wt);
        delete r;
        DFX_FAIL(DFX_ERR_HIP, "dfx_resampler_create: upload failed");
    }
    *out = r;
    return DFX_OK;
}

extern "C" void dfx_resampler_free(dfx_resampler *r) {
    if (!r) return;
    if (r->d_wt) (void)hipFree(r->d_wt);
    delete r;
}

extern "C" int64_t dfx_resampler_out_len(const dfx_resampler *r, int64_t in_len) {
    if (!r || in_len < 0) return -1;
    return (in_len * r->nw + r->orig - 1) / r->orig;  // ceil(new * length / orig)
}

extern "C" int dfx_resampler_kernel(int orig_sr, int new_sr, int lowpass_filter_width, double rolloff, int method, double beta,
                                    int *phases, int *taps, int *width, float *w_host, int64_t cap) {
    dfx_resampler r;
    if (int rc = build_bank(&r, orig_sr, new_sr, lowpass_filter_width, rolloff, method, beta)) return rc;
    if (phases) *phases = r.nw;
    if (taps) *taps = r.K;
    if (width) *width = r.width;
    if (w_host) {
        if (cap < (int64_t)r.w_host.size()) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resampler_kernel: buffer too small");
        memcpy(w_host, r.w_host.data(), r.w_host.size() * 4);
    }
    return DFX_OK;
}

extern "C" int dfx_resample(const dfx_resampler *r, const float *x, int64_t B, int64_t T, int64_t x_stride, float *y, int64_t y_stride,
                            void *stream) {
    if (!r || B < 0 || T < 0 || x_stride < T) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resample: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    const int64_t out_len = dfx_resampler_out_len(r, T);
    if (y_stride < out_len) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resample: y_stride < output length %lld", (long long)out_len);
    if (B == 0 || out_len == 0) return DFX_OK;
    if (!x || !y) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_resample: null buffer");
    DfxRsArgs A;
    A.B = B, A.T = T, A.x_stride = x_stride, A.y_stride = y_stride, A.out_len = out_len;
    A.frames = dfx_ceil_div(out_len, r->nw);
    A.orig = r->orig, A.nw = r->nw, A.nw_pad = r->nw_pad, A.width = r->width, A.K4 = r->K4;
    // 4 waves per workgroup: jw of them split the phase tiles (as many as there are tiles, 1 / 2 / 4), fw = 4 / jw own 64 frames each
    const int ntiles = r->nw_pad / DFX_RS_JT;
    A.jw = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
    A.fw = 4 / A.jw;
    A.tn = 64 * A.fw;
    const size_t smem = ((size_t)(A.tn - 1) * r->orig + r->K4) * sizeof(float);
    if (smem > 160 * 1024) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_resample: rate ratio %d:%d needs more than 160 KB of LDS per workgroup", r->orig, r->nw);
    const int64_t nblk = B * dfx_ceil_div(A.frames, A.tn);
    if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_resample: grid too large");
    if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_resample, smem));
    DfxKScope ks(DFX_K_RESAMPLE, dfx_stream(stream));
    dfx_launch(dfx_k_resample, dim3((unsigned)nblk), dim3(256), smem, dfx_stream(stream), x, y, (const float *)r->d_wt, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}
