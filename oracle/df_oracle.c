/*
 * df_oracle.c — scalar C restatement of the libDF DSP core (TEST INFRASTRUCTURE ONLY, see df_oracle.h).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).  The arithmetic is
 * f32 exactly where the reference is f32 (window via f64 then cast, like lib.rs:128-133).
 *
 * FFT: the reference calls realfft 3.3.0 -> rustfft 6.2.0 (third party, not in /root/reference).  Restated here
 * as (a) a Stockham mixed-radix complex FFT of length N/2 and (b) the standard even/odd split for the real
 * transform, which is the algorithm realfft documents for even lengths.  Semantics matched: forward is the
 * unnormalised R2C; inverse is the unnormalised C2R (returns N*x) that ignores imag(DC) and imag(Nyquist)
 * (lib.rs:398-405 tolerates exactly that error case).  Bit-equality with rustfft is neither possible nor required.
 */
#define _GNU_SOURCE
#include "df_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DFO_MAX_FACTORS 32

typedef struct {
    float re, im;
} cf32;

typedef struct {
    int n;                         /* complex length */
    int nf;                        /* number of stages */
    int radix[DFO_MAX_FACTORS];    /* radix per stage */
    cf32 *tw[DFO_MAX_FACTORS];     /* per stage: tw[st][p*(r-1) + (j-1)] = exp(-2*pi*i*j*p/n_cur), p<m, j=1..r-1 */
    cf32 *dft[DFO_MAX_FACTORS];    /* per stage: r x r DFT matrix exp(-2*pi*i*j*l/r) (used for r not in {2,4}) */
    cf32 *scratch;                 /* n */
} dfo_cfft;

struct dfo_state {
    int sr, hop, fft, freq, nb_erb;
    float *window;        /* [fft] */
    float wnorm;
    uint64_t *erb;        /* [nb_erb] */
    float *analysis_mem;  /* [fft-hop] */
    float *synthesis_mem; /* [fft-hop] */
    dfo_cfft plan;        /* complex FFT of length fft/2 */
    cf32 *rtw;            /* [fft/2+1]: exp(-2*pi*i*k/fft) */
    cf32 *zbuf;           /* [fft/2] */
    float *tbuf;          /* [fft] */
};

/* ------------------------------------------------------------------------------------------------- FFT */

static int cfft_init(dfo_cfft *pl, int n) {
    memset(pl, 0, sizeof(*pl));
    pl->n = n;
    int rem = n;
    /* prefer radix 4, then 2, 3, 5, then any remaining prime */
    while (rem % 4 == 0) { pl->radix[pl->nf++] = 4; rem /= 4; }
    while (rem % 2 == 0) { pl->radix[pl->nf++] = 2; rem /= 2; }
    for (int r = 3; rem > 1; r += 2) {
        while (rem % r == 0) {
            if (pl->nf >= DFO_MAX_FACTORS) return -1;
            pl->radix[pl->nf++] = r;
            rem /= r;
        }
    }
    int ncur = n;
    for (int st = 0; st < pl->nf; ++st) {
        int r = pl->radix[st], m = ncur / r;
        pl->tw[st] = (cf32 *)malloc(sizeof(cf32) * (size_t)m * (size_t)(r - 1));
        for (int p = 0; p < m; ++p)
            for (int j = 1; j < r; ++j) {
                double a = -2.0 * M_PI * (double)j * (double)p / (double)ncur;
                pl->tw[st][p * (r - 1) + (j - 1)].re = (float)cos(a);
                pl->tw[st][p * (r - 1) + (j - 1)].im = (float)sin(a);
            }
        pl->dft[st] = (cf32 *)malloc(sizeof(cf32) * (size_t)r * (size_t)r);
        for (int j = 0; j < r; ++j)
            for (int l = 0; l < r; ++l) {
                double a = -2.0 * M_PI * (double)((j * l) % r) / (double)r;
                pl->dft[st][j * r + l].re = (float)cos(a);
                pl->dft[st][j * r + l].im = (float)sin(a);
            }
        ncur = m;
    }
    pl->scratch = (cf32 *)malloc(sizeof(cf32) * (size_t)(n > 0 ? n : 1));
    return 0;
}

static void cfft_free(dfo_cfft *pl) {
    for (int st = 0; st < pl->nf; ++st) {
        free(pl->tw[st]);
        free(pl->dft[st]);
    }
    free(pl->scratch);
    memset(pl, 0, sizeof(*pl));
}

static inline cf32 cmul(cf32 a, cf32 b) {
    cf32 c = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return c;
}

/* In-place (via scratch) Stockham autosort FFT.  inverse!=0 conjugates all twiddles (unnormalised inverse). */
static void cfft_exec(const dfo_cfft *pl, cf32 *data, int inverse) {
    cf32 *x = data, *y = pl->scratch;
    int ncur = pl->n, s = 1;
    const float sg = inverse ? -1.f : 1.f;
    for (int st = 0; st < pl->nf; ++st) {
        const int r = pl->radix[st], m = ncur / r;
        const cf32 *tw = pl->tw[st];
        for (int p = 0; p < m; ++p) {
            for (int q = 0; q < s; ++q) {
                cf32 a[64];
                cf32 b[64];
                if (r > 64) return; /* unsupported radix; never hit for DF sizes */
                for (int j = 0; j < r; ++j) a[j] = x[q + s * (p + m * j)];
                if (r == 2) {
                    b[0].re = a[0].re + a[1].re; b[0].im = a[0].im + a[1].im;
                    b[1].re = a[0].re - a[1].re; b[1].im = a[0].im - a[1].im;
                } else if (r == 4) {
                    cf32 t0 = {a[0].re + a[2].re, a[0].im + a[2].im};
                    cf32 t1 = {a[0].re - a[2].re, a[0].im - a[2].im};
                    cf32 t2 = {a[1].re + a[3].re, a[1].im + a[3].im};
                    cf32 t3 = {a[1].re - a[3].re, a[1].im - a[3].im};
                    /* forward: -i*t3 = (t3.im, -t3.re); inverse: +i*t3 = (-t3.im, t3.re) */
                    cf32 jt3 = {sg * t3.im, -sg * t3.re};
                    b[0].re = t0.re + t2.re; b[0].im = t0.im + t2.im;
                    b[1].re = t1.re + jt3.re; b[1].im = t1.im + jt3.im;
                    b[2].re = t0.re - t2.re; b[2].im = t0.im - t2.im;
                    b[3].re = t1.re - jt3.re; b[3].im = t1.im - jt3.im;
                } else {
                    const cf32 *D = pl->dft[st];
                    for (int j = 0; j < r; ++j) {
                        cf32 acc = {0.f, 0.f};
                        for (int l = 0; l < r; ++l) {
                            cf32 w = D[j * r + l];
                            w.im *= sg;
                            cf32 t = cmul(a[l], w);
                            acc.re += t.re;
                            acc.im += t.im;
                        }
                        b[j] = acc;
                    }
                }
                y[q + s * (r * p)] = b[0];
                for (int j = 1; j < r; ++j) {
                    cf32 w = tw[p * (r - 1) + (j - 1)];
                    w.im *= sg;
                    y[q + s * (r * p + j)] = cmul(b[j], w);
                }
            }
        }
        cf32 *t = x; x = y; y = t;
        ncur = m;
        s *= r;
    }
    if (x != data) memcpy(data, x, sizeof(cf32) * (size_t)pl->n);
}

/* Unnormalised real-to-complex FFT of even length N: in[N] -> out[N/2+1]. */
static void rfft_forward(dfo_state *st, const float *in, cf32 *out) {
    const int N = st->fft, M = N / 2;
    cf32 *z = st->zbuf;
    for (int k = 0; k < M; ++k) {
        z[k].re = in[2 * k];
        z[k].im = in[2 * k + 1];
    }
    cfft_exec(&st->plan, z, 0);
    for (int k = 0; k <= M; ++k) {
        cf32 zk = z[k % M];
        cf32 zc = z[(M - k) % M]; /* conj applied below */
        /* E = (Z[k] + conj(Z[M-k]))/2 ; O = (Z[k] - conj(Z[M-k]))/(2i) ; X = E + w^k O */
        float er = 0.5f * (zk.re + zc.re), ei = 0.5f * (zk.im - zc.im);
        float dr = 0.5f * (zk.re - zc.re), di = 0.5f * (zk.im + zc.im);
        /* O = (dr + i di)/i = di - i dr */
        cf32 o = {di, -dr};
        cf32 t = cmul(o, st->rtw[k]);
        out[k].re = er + t.re;
        out[k].im = ei + t.im;
    }
}

/* Unnormalised complex-to-real inverse FFT: in[N/2+1] -> out[N] (= N * x).  imag(in[0]), imag(in[N/2]) ignored. */
static void rfft_inverse(dfo_state *st, const cf32 *in, float *out) {
    const int N = st->fft, M = N / 2;
    cf32 *z = st->zbuf;
    for (int k = 0; k < M; ++k) {
        cf32 xk = in[k], xm = in[M - k];
        if (k == 0) { xk.im = 0.f; xm.im = 0.f; }
        /* E' = X[k] + conj(X[M-k]) ; O' = conj(w^k) (X[k] - conj(X[M-k])) ; Z = E' + i O' */
        float er = xk.re + xm.re, ei = xk.im - xm.im;
        cf32 d = {xk.re - xm.re, xk.im + xm.im};
        cf32 w = st->rtw[k];
        w.im = -w.im;
        cf32 o = cmul(d, w);
        z[k].re = er - o.im;
        z[k].im = ei + o.re;
    }
    cfft_exec(&st->plan, z, 1);
    for (int n = 0; n < M; ++n) {
        out[2 * n] = z[n].re;
        out[2 * n + 1] = z[n].im;
    }
}

/* --------------------------------------------------------------------------------------------- state */

/* libDF/src/lib.rs:42-47 */
static float freq2erb(float f) { return 9.265f * log1pf(f / (24.7f * 9.265f)); }
static float erb2freq(float e) { return 24.7f * 9.265f * (expf(e / 9.265f) - 1.f); }

/* libDF/src/lib.rs:68-100 */
int dfo_erb_fb(int sr, int fft_size, int nb_bands, int min_nb_freqs, uint64_t *erb) {
    if (sr <= 0 || fft_size <= 0 || nb_bands <= 0 || !erb) return -1;
    int nyq = sr / 2;
    float freq_width = (float)sr / (float)fft_size;
    float erb_low = freq2erb(0.f);
    float erb_high = freq2erb((float)nyq);
    float step = (erb_high - erb_low) / (float)nb_bands;
    int prev_freq = 0, freq_over = 0;
    for (int i = 1; i <= nb_bands; ++i) {
        float f = erb2freq(erb_low + (float)i * step);
        int fb = (int)roundf(f / freq_width); /* Rust f32::round: half away from zero, same as roundf */
        int nb_freqs = fb - prev_freq - freq_over;
        if (nb_freqs < min_nb_freqs) {
            freq_over = min_nb_freqs - nb_freqs;
            nb_freqs = min_nb_freqs;
        } else {
            freq_over = 0;
        }
        erb[i - 1] = (uint64_t)nb_freqs;
        prev_freq = fb;
    }
    erb[nb_bands - 1] += 1;
    int64_t sum = 0;
    for (int i = 0; i < nb_bands; ++i) sum += (int64_t)erb[i];
    int64_t too_large = sum - (fft_size / 2 + 1);
    if (too_large > 0) erb[nb_bands - 1] -= (uint64_t)too_large;
    return 0;
}

/* libDF/src/lib.rs:104-154 */
dfo_state *dfo_state_new(int sr, int fft_size, int hop_size, int nb_bands, int min_nb_freqs) {
    if (hop_size <= 0 || fft_size <= 0 || hop_size * 2 > fft_size || (fft_size & 1)) return NULL;
    dfo_state *st = (dfo_state *)calloc(1, sizeof(*st));
    st->sr = sr;
    st->hop = hop_size;
    st->fft = fft_size;
    st->freq = fft_size / 2 + 1;
    st->nb_erb = nb_bands;
    st->erb = (uint64_t *)calloc((size_t)nb_bands, sizeof(uint64_t));
    dfo_erb_fb(sr, fft_size, nb_bands, min_nb_freqs, st->erb);
    st->window = (float *)malloc(sizeof(float) * (size_t)fft_size);
    int window_size_h = fft_size / 2;
    for (int i = 0; i < fft_size; ++i) {
        double s = sin(0.5 * M_PI * ((double)i + 0.5) / (double)window_size_h);
        st->window[i] = (float)sin(0.5 * M_PI * s * s);
    }
    /* usize pow then as f32 (lib.rs:134) */
    st->wnorm = 1.f / ((float)((int64_t)fft_size * (int64_t)fft_size) / (float)(2 * hop_size));
    st->analysis_mem = (float *)calloc((size_t)(fft_size - hop_size), sizeof(float));
    st->synthesis_mem = (float *)calloc((size_t)(fft_size - hop_size), sizeof(float));
    cfft_init(&st->plan, fft_size / 2);
    st->rtw = (cf32 *)malloc(sizeof(cf32) * (size_t)st->freq);
    for (int k = 0; k < st->freq; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)fft_size;
        st->rtw[k].re = (float)cos(a);
        st->rtw[k].im = (float)sin(a);
    }
    st->zbuf = (cf32 *)malloc(sizeof(cf32) * (size_t)(fft_size / 2));
    st->tbuf = (float *)malloc(sizeof(float) * (size_t)fft_size);
    return st;
}

void dfo_state_free(dfo_state *st) {
    if (!st) return;
    free(st->erb);
    free(st->window);
    free(st->analysis_mem);
    free(st->synthesis_mem);
    cfft_free(&st->plan);
    free(st->rtw);
    free(st->zbuf);
    free(st->tbuf);
    free(st);
}

void dfo_state_reset(dfo_state *st) {
    memset(st->analysis_mem, 0, sizeof(float) * (size_t)(st->fft - st->hop));
    memset(st->synthesis_mem, 0, sizeof(float) * (size_t)(st->fft - st->hop));
}

int dfo_state_sr(const dfo_state *st) { return st->sr; }
int dfo_state_fft_size(const dfo_state *st) { return st->fft; }
int dfo_state_hop_size(const dfo_state *st) { return st->hop; }
int dfo_state_nb_erb(const dfo_state *st) { return st->nb_erb; }
float dfo_state_wnorm(const dfo_state *st) { return st->wnorm; }
void dfo_state_window(const dfo_state *st, float *out) { memcpy(out, st->window, sizeof(float) * (size_t)st->fft); }
void dfo_state_erb_widths(const dfo_state *st, uint64_t *out) {
    memcpy(out, st->erb, sizeof(uint64_t) * (size_t)st->nb_erb);
}

/* ------------------------------------------------------------------------------------ analysis/synthesis */

/* libDF/src/lib.rs:356-394 */
void dfo_frame_analysis(dfo_state *st, const float *in, float *out) {
    const int N = st->fft, H = st->hop, ML = N - H;
    float *buf = st->tbuf;
    for (int i = 0; i < ML; ++i) buf[i] = st->analysis_mem[i] * st->window[i];
    for (int i = 0; i < H; ++i) buf[ML + i] = in[i] * st->window[ML + i];
    const int split = ML - H;
    if (split > 0) memmove(st->analysis_mem, st->analysis_mem + H, sizeof(float) * (size_t)split); /* rotate_left; tail overwritten next */
    for (int i = 0; i < H; ++i) st->analysis_mem[split + i] = in[i];
    cf32 *o = (cf32 *)out;
    rfft_forward(st, buf, o);
    const float norm = st->wnorm;
    for (int k = 0; k < st->freq; ++k) {
        o[k].re *= norm;
        o[k].im *= norm;
    }
}

/* libDF/src/lib.rs:396-427 */
void dfo_frame_synthesis(dfo_state *st, const float *in, float *out) {
    const int N = st->fft, H = st->hop, ML = N - H;
    float *x = st->tbuf;
    rfft_inverse(st, (const cf32 *)in, x);
    for (int i = 0; i < N; ++i) x[i] *= st->window[i];
    for (int i = 0; i < H; ++i) out[i] = x[i] + st->synthesis_mem[i];
    const int split = ML - H;
    if (split > 0) {
        /* rotate_left(H): the first H entries wrap to the end and are overwritten below */
        memmove(st->synthesis_mem, st->synthesis_mem + H, sizeof(float) * (size_t)split);
    }
    const float *x_second = x + H;
    for (int i = 0; i < split; ++i) st->synthesis_mem[i] += x_second[i];
    for (int i = split; i < ML; ++i) st->synthesis_mem[i] = x_second[i];
}

/* pyDF/src/lib.rs:41-72 */
void dfo_analysis(dfo_state *st, const float *x, int64_t C, int64_t T, int reset, float *spec) {
    const int64_t Tf = T / st->hop;
    for (int64_t c = 0; c < C; ++c) {
        if (reset) dfo_state_reset(st);
        for (int64_t t = 0; t < Tf; ++t)
            dfo_frame_analysis(st, x + c * T + t * st->hop, spec + ((c * Tf + t) * st->freq) * 2);
    }
}

/* pyDF/src/lib.rs:74-107 */
void dfo_synthesis(dfo_state *st, const float *spec, int64_t C, int64_t Tf, int reset, float *out) {
    for (int64_t c = 0; c < C; ++c) {
        if (reset) dfo_state_reset(st);
        for (int64_t t = 0; t < Tf; ++t)
            dfo_frame_synthesis(st, spec + ((c * Tf + t) * st->freq) * 2, out + (c * Tf + t) * st->hop);
    }
}

/* --------------------------------------------------------------------------------------------- features */

/* libDF/src/lib.rs:280-295 compute_band_corr(x,x) + transforms.rs:249-251 dB */
void dfo_erb(const float *spec, int64_t rows, const uint64_t *widths, int nb, int db, float *out) {
    int64_t F = 0;
    for (int b = 0; b < nb; ++b) F += (int64_t)widths[b];
    for (int64_t r = 0; r < rows; ++r) {
        const cf32 *x = (const cf32 *)spec + r * F;
        float *o = out + r * nb;
        int64_t bcsum = 0;
        for (int b = 0; b < nb; ++b) {
            const int64_t w = (int64_t)widths[b];
            const float k = 1.f / (float)w;
            float acc = 0.f;
            for (int64_t j = 0; j < w; ++j) {
                const cf32 v = x[bcsum + j];
                acc += (v.re * v.re + v.im * v.im) * k;
            }
            o[b] = acc;
            bcsum += w;
        }
        if (db)
            for (int b = 0; b < nb; ++b) o[b] = log10f(o[b] + 1e-10f) * 10.f;
    }
}

/* libDF/src/lib.rs:339-348 interp_band_gain */
void dfo_erb_inv(const float *gains, int64_t rows, const uint64_t *widths, int nb, float *out) {
    int64_t F = 0;
    for (int b = 0; b < nb; ++b) F += (int64_t)widths[b];
    for (int64_t r = 0; r < rows; ++r) {
        int64_t bcsum = 0;
        for (int b = 0; b < nb; ++b) {
            for (int64_t j = 0; j < (int64_t)widths[b]; ++j) out[r * F + bcsum + j] = gains[r * nb + b];
            bcsum += (int64_t)widths[b];
        }
    }
}

static void linspace(float a, float b, int n, float *out) {
    /* ndarray::Array1::linspace: step = (b-a)/(n-1); out[i] = a + step*i  (lib.rs:183-191 uses the same form) */
    float step = n > 1 ? (b - a) / (float)(n - 1) : 0.f;
    for (int i = 0; i < n; ++i) out[i] = a + step * (float)i;
}

/* libDF/src/transforms.rs:301-330 + lib.rs:244-251 */
void dfo_erb_norm(float *x, int64_t C, int64_t T, int E, float alpha, float *state) {
    float *s = (float *)malloc(sizeof(float) * (size_t)E);
    for (int64_t c = 0; c < C; ++c) {
        if (state) memcpy(s, state + c * E, sizeof(float) * (size_t)E);
        else linspace(-60.f, -90.f, E, s);
        for (int64_t t = 0; t < T; ++t) {
            float *xs = x + (c * T + t) * E;
            for (int e = 0; e < E; ++e) {
                s[e] = xs[e] * (1.f - alpha) + s[e] * alpha;
                xs[e] -= s[e];
                xs[e] /= 40.f;
            }
        }
        if (state) memcpy(state + c * E, s, sizeof(float) * (size_t)E);
    }
    free(s);
}

/* libDF/src/transforms.rs:332-361 + lib.rs:253-259 */
void dfo_unit_norm(float *x, int64_t C, int64_t T, int F, float alpha, float *state) {
    float *s = (float *)malloc(sizeof(float) * (size_t)F);
    for (int64_t c = 0; c < C; ++c) {
        if (state) memcpy(s, state + c * F, sizeof(float) * (size_t)F);
        else linspace(0.001f, 0.0001f, F, s);
        for (int64_t t = 0; t < T; ++t) {
            cf32 *xs = (cf32 *)x + (c * T + t) * F;
            for (int f = 0; f < F; ++f) {
                /* num_complex norm() = hypot */
                float nrm = hypotf(xs[f].re, xs[f].im);
                s[f] = nrm * (1.f - alpha) + s[f] * alpha;
                float d = sqrtf(s[f]);
                xs[f].re /= d;
                xs[f].im /= d;
            }
        }
        if (state) memcpy(state + c * F, s, sizeof(float) * (size_t)F);
    }
    free(s);
}

/* libDF/src/lib.rs:314-326 */
void dfo_apply_band_gain(float *spec, int64_t rows, const float *gains, const uint64_t *widths, int nb) {
    int64_t F = 0;
    for (int b = 0; b < nb; ++b) F += (int64_t)widths[b];
    for (int64_t r = 0; r < rows; ++r) {
        cf32 *x = (cf32 *)spec + r * F;
        int64_t bcsum = 0;
        for (int b = 0; b < nb; ++b) {
            const float g = gains[r * nb + b];
            for (int64_t j = 0; j < (int64_t)widths[b]; ++j) {
                x[bcsum + j].re *= g;
                x[bcsum + j].im *= g;
            }
            bcsum += (int64_t)widths[b];
        }
    }
}

/* libDF/src/lib.rs:446-471 */
void dfo_post_filter(const float *noisy, float *enh, int64_t rows, int F, float beta) {
    const float beta_p1 = beta + 1.f, eps = 1e-12f, pi = 3.14159265358979323846f;
    const int F4 = F - (F % 4);
    for (int64_t r = 0; r < rows; ++r) {
        const cf32 *n = (const cf32 *)noisy + r * F;
        cf32 *e = (cf32 *)enh + r * F;
        for (int f = 0; f < F4; ++f) {
            float g = hypotf(e[f].re, e[f].im) / (hypotf(n[f].re, n[f].im) + eps);
            g = fmaxf(fminf(g, 1.f), eps);
            float g_sin = g * sinf(g * pi / 2.0f);
            float q = g / g_sin;
            float pf = (beta_p1 * g / (1.f + beta * (q * q))) / g;
            e[f].re *= pf;
            e[f].im *= pf;
        }
    }
}

void dfo_unit_norm_init(int n, float *out) { linspace(0.001f, 0.0001f, n, out); }
