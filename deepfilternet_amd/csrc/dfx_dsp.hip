// Host side of the DSP half of libdfx: handles, argument validation, launch geometry and the C ABI declared in
// include/dfx.h.  Device pointers in, device pointers out; nothing here computes on the CPU except the ERB width table
// (pure index arithmetic done once at state creation, libDF/src/lib.rs:68-100).
#include "dfx_dsp_kernels.h"

#include <cmath>
#include <cstdarg>

// ------------------------------------------------------------------------------------------------ errors / misc
static thread_local char g_dfx_err[512] = "";

void dfx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_dfx_err, sizeof(g_dfx_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dfx_last_error(void) { return g_dfx_err; }
extern "C" int dfx_version(void) { return DFX_VERSION; }
extern "C" int dfx_is_emulator(void) { return dfx_env_is_emulator() ? 1 : 0; }
extern "C" const char *dfx_status_string(int s) {
    switch (s) {
        case DFX_OK: return "ok";
        case DFX_ERR_INVALID_ARG: return "invalid argument";
        case DFX_ERR_UNSUPPORTED: return "unsupported configuration";
        case DFX_ERR_HIP: return "HIP runtime error";
        case DFX_ERR_NO_DEVICE: return "no HIP device";
        case DFX_ERR_ALLOC: return "allocation failed";
        default: return "unknown";
    }
}
extern "C" int dfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int dfx_require_device() {
    if (dfx_device_count() <= 0)
        DFX_FAIL(DFX_ERR_NO_DEVICE, "no HIP device visible: libdfx has no CPU fallback (MI355X / gfx950 required)");
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ kernel timing
namespace {
struct ProfPair {
    int id;
    hipEvent_t a, b;
};
struct Prof {
    uint32_t mask = 0;
    std::vector<ProfPair> pending;
    std::vector<hipEvent_t> pool;
    hipEvent_t cur = nullptr;
    double total_ms[DFX_K_COUNT] = {0};
    int64_t launches[DFX_K_COUNT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void drain() {
        for (auto &p : pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(p.b);
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                total_ms[p.id] += ms;
                launches[p.id] += 1;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
};
Prof g_prof;
const char *const g_kernel_names[DFX_K_COUNT] = {
    "dfx_k_analysis", "dfx_k_analysis_mem_out", "dfx_k_norm_scan", "dfx_k_synthesis", "dfx_k_erb", "dfx_k_erb_inv",
    "dfx_k_df_apply", "dfx_k_conv_in_erb", "dfx_k_pwconv", "dfx_k_conv_out", "dfx_k_df_convp", "dfx_k_ggemm",
    "dfx_k_gru_rec", "dfx_k_lsnr", "dfx_k_add", "dfx_k_copy_rows", "dfx_k_conv_in_df", "dfx_k_proj256", "dfx_k_erb_enc",
    "dfx_k_erb_dec", "dfx_k_resample", "dfx_k_pcm", "dfx_k_mf_filter", "dfx_k_emb_fan", "dfx_k_erb_tail"};
}  // namespace

bool dfx_prof_on(int id) { return g_prof.mask != 0 && id >= 0 && id < DFX_K_COUNT && ((g_prof.mask >> id) & 1u); }
// An empty kernel in front of the start event: a marker that directly follows hipStreamWaitEvent packets was seen to carry a
// timestamp from BEFORE the awaited events (the in-loop df_apply interval then included the wait for the last decoder tail:
// 0.69 ms by events against 0.60 ms in the rocprofv3 kernel trace of the same run).  Behind a dispatch the marker is ordered.
__global__ void dfx_k_prof_fence() {}
void dfx_prof_begin(int, hipStream_t s) {
    g_prof.cur = g_prof.get();
    dfx_launch(dfx_k_prof_fence, dim3(1), dim3(64), 0, s);
    (void)hipEventRecord(g_prof.cur, s);
}
void dfx_prof_end(int id, hipStream_t s) {
    hipEvent_t b = g_prof.get();
    (void)hipEventRecord(b, s);
    g_prof.pending.push_back(ProfPair{id, g_prof.cur, b});
    g_prof.cur = nullptr;
}
extern "C" int dfx_prof_kernel_count(void) { return DFX_K_COUNT; }
extern "C" const char *dfx_prof_kernel_name(int id) { return (id >= 0 && id < DFX_K_COUNT) ? g_kernel_names[id] : ""; }
extern "C" int dfx_prof_enable(uint32_t mask) {
    g_prof.drain();
    g_prof.mask = mask;
    return DFX_OK;
}
extern "C" int dfx_prof_reset(void) {
    g_prof.drain();
    for (int i = 0; i < DFX_K_COUNT; ++i) {
        g_prof.total_ms[i] = 0;
        g_prof.launches[i] = 0;
    }
    return DFX_OK;
}
extern "C" int dfx_prof_read(int id, double *total_ms, int64_t *launches) {
    if (id < 0 || id >= DFX_K_COUNT) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_prof_read: bad kernel id");
    g_prof.drain();
    if (total_ms) *total_ms = g_prof.total_ms[id];
    if (launches) *launches = g_prof.launches[id];
    return DFX_OK;
}

template <typename T>
static int upload(T **dst, const T *src, size_t n) {
    DFX_HIP(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    DFX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ band table
extern "C" int dfx_bands_create(const uint64_t *widths, int nb, dfx_bands **out) {
    if (!widths || nb <= 0 || nb > 255 || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_bands_create: bad arguments (1..255 bands)");
    if (int rc = dfx_require_device()) return rc;
    dfx_bands *b = new dfx_bands();
    b->nb = nb;
    b->widths.assign(widths, widths + nb);
    std::vector<int> start(nb + 1, 0);
    std::vector<float> invw(nb);
    for (int i = 0; i < nb; ++i) {
        if (widths[i] == 0 || widths[i] > (1u << 20)) {
            delete b;
            DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_bands_create: band %d has width %llu", i, (unsigned long long)widths[i]);
        }
        start[i + 1] = start[i] + (int)widths[i];
        invw[i] = 1.f / (float)widths[i];
    }
    b->F = start[nb];
    std::vector<unsigned char> b2b(b->F);
    for (int i = 0; i < nb; ++i)
        for (int f = start[i]; f < start[i + 1]; ++f) b2b[f] = (unsigned char)i;
    // the bands cut into at most 64 segments of at most `cap` bins (smallest cap that fits): one lane of the analysis kernel per segment
    std::vector<int> segtab(3 * 64 + nb + 1, 0);
    if (nb <= 64) {
        uint64_t cap = 1;
        for (;; ++cap) {
            uint64_t cnt = 0;
            for (int i = 0; i < nb; ++i) cnt += (widths[i] + cap - 1) / cap;
            if (cnt <= 64) break;
        }
        int ns = 0;
        for (int i = 0; i < nb; ++i) {
            segtab[3 * 64 + i] = ns;
            const int w = (int)widths[i], parts = (int)((widths[i] + cap - 1) / cap);
            for (int k = 0, at = 0; k < parts; ++k) {      // near-equal parts, the longer ones first
                const int len = w / parts + (k < w % parts ? 1 : 0);
                segtab[ns] = start[i] + at;
                segtab[64 + ns] = len;
                memcpy(&segtab[128 + ns], &invw[i], 4);
                at += len;
                ++ns;
            }
        }
        segtab[3 * 64 + nb] = ns;
        b->nseg = ns;
        b->segcap = (int)cap;
        for (int i = 0; i < nb; ++i) b->segparts = std::max(b->segparts, (int)((widths[i] + cap - 1) / cap));
    }
    int rc = upload(&b->d_start, start.data(), start.size());
    if (!rc) rc = upload(&b->d_invw, invw.data(), invw.size());
    if (!rc) rc = upload(&b->d_segtab, segtab.data(), segtab.size());
    if (!rc) rc = upload(&b->d_bin2band, b2b.data(), b2b.size());
    if (rc) {
        dfx_bands_free(b);
        return rc;
    }
    *out = b;
    return DFX_OK;
}
extern "C" void dfx_bands_free(dfx_bands *b) {
    if (!b) return;
    if (b->d_start) (void)hipFree(b->d_start);
    if (b->d_invw) (void)hipFree(b->d_invw);
    if (b->d_bin2band) (void)hipFree(b->d_bin2band);
    if (b->d_segtab) (void)hipFree(b->d_segtab);
    delete b;
}
extern "C" int dfx_bands_nb(const dfx_bands *b) { return b ? b->nb : 0; }
extern "C" int dfx_bands_nfreqs(const dfx_bands *b) { return b ? b->F : 0; }

// ------------------------------------------------------------------------------------------------ state
static float freq2erb(float f) { return 9.265f * log1pf(f / (24.7f * 9.265f)); }           // lib.rs:42-44
static float erb2freq(float e) { return 24.7f * 9.265f * (expf(e / 9.265f) - 1.f); }        // lib.rs:45-47

// libDF/src/lib.rs:68-100 (f32 arithmetic, round-half-away like f32::round)
extern "C" int dfx_erb_fb(int sr, int fft_size, int nb_bands, int min_nb_freqs, uint64_t *erb) {
    if (sr <= 0 || fft_size <= 0 || nb_bands <= 0 || !erb) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb_fb: bad arguments");
    const int nyq = sr / 2;
    const float freq_width = (float)sr / (float)fft_size;
    const float erb_low = freq2erb(0.f), erb_high = freq2erb((float)nyq);
    const float step = (erb_high - erb_low) / (float)nb_bands;
    int prev_freq = 0, freq_over = 0;
    for (int i = 1; i <= nb_bands; ++i) {
        const float f = erb2freq(erb_low + (float)i * step);
        const int fb = (int)roundf(f / freq_width);
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
    const int64_t too_large = sum - (fft_size / 2 + 1);
    if (too_large > 0) erb[nb_bands - 1] -= (uint64_t)too_large;
    return DFX_OK;
}

static int make_plan(int N, DfxFftPlan *pl) {
    pl->N = N;
    pl->M = N / 2;
    pl->nstage = 0;
    int rem = pl->M;
    const int radices[4] = {4, 2, 3, 5};
    for (int ri = 0; ri < 4; ++ri)
        while (rem % radices[ri] == 0 && rem > 1) {
            if (pl->nstage >= DFX_MAX_STAGES) return -1;
            pl->radix[pl->nstage++] = radices[ri];
            rem /= radices[ri];
        }
    return rem == 1 ? 0 : -1;
}

// Tables of dfx_fft480_mfma for one direction (sg = -1 forward, +1 inverse): see dfx_dsp_kernels.h.  Built in double precision; matrix
// entries scaled by 2^13 and split into f16 hi / lo, the twiddle factors carry 2^-17.
static void build_mfft_table(int sg, unsigned char *dst) {
    uint16_t *fr = reinterpret_cast<uint16_t *>(dst);
    const double s13 = 8192.0;
    auto put = [&](int frag, int lane, int i, double v) {   // fragment pair (frag, frag + 1) = (hi, lo)
        const float w = (float)(v * s13);
        const uint16_t hb = dfx_f32_to_f16_bits(w);
        const uint16_t lb = dfx_f32_to_f16_bits(w - dfx_f16_bits_to_f32(hb));
        fr[((size_t)frag * 64 + lane) * 8 + i] = hb;
        fr[((size_t)(frag + 1) * 64 + lane) * 8 + i] = lb;
    };
    for (int l = 0; l < 64; ++l) {
        const int jl = l & 15, q = l >> 4;
        for (int i = 0; i < 8; ++i) {
            // first product, B operand: column k1 = jl, k-slot 8 q + i = (re | im) of n1
            const int k = 8 * q + i, n1 = k & 15;
            const double th = 2.0 * M_PI * (double)((n1 * jl) % 16) / 16.0, wr = cos(th), wi = sg * sin(th);
            put(0, l, i, k < 16 ? wr : -wi);   // -> Yr
            put(2, l, i, k < 16 ? wi : wr);    // -> Yi
            // second product, A operand: row k2 = 16 t2 + jl, k-slot i <-> n2 = 16 (i >> 2) + 4 q + (i & 3)
            const int n2 = 16 * (i >> 2) + 4 * q + (i & 3);
            for (int t2 = 0; t2 < 2; ++t2) {
                const int k2 = 16 * t2 + jl;
                const bool ok = n2 < 30 && k2 < 30;
                const double ph = 2.0 * M_PI * (double)((n2 * k2) % 30) / 30.0;
                put(DFX_MFFT_FRAG1 + (0 * 2 + t2) * 2, l, i, ok ? cos(ph) : 0.0);
                put(DFX_MFFT_FRAG1 + (1 * 2 + t2) * 2, l, i, ok ? sg * sin(ph) : 0.0);
            }
        }
        // twiddle factors of the lane's eight values: n2 = 16 mt + 4 q + r, k1 = jl
        float2 *tw = reinterpret_cast<float2 *>(dst + (size_t)(DFX_MFFT_FRAG1 + DFX_MFFT_FRAG3) * 64 * 16) + l * 8;
        for (int mt = 0; mt < 2; ++mt)
            for (int r = 0; r < 4; ++r) {
                const int n2 = 16 * mt + 4 * q + r;
                const double a = 2.0 * M_PI * (double)((n2 * jl) % 480) / 480.0, sc = n2 < 30 ? 1.0 / 131072.0 : 0.0;
                tw[4 * mt + r] = make_float2((float)(cos(a) * sc), (float)(sg * sin(a) * sc));
            }
    }
}

extern "C" int dfx_state_create(int sr, int fft_size, int hop_size, int nb_bands, int min_nb_erb_freqs, dfx_state **out) {
    if (!out || sr <= 0 || fft_size <= 0 || hop_size <= 0 || nb_bands <= 0)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_state_create: bad arguments");
    if (hop_size * 2 > fft_size) DFX_FAIL(DFX_ERR_INVALID_ARG, "assertion failed: hop_size * 2 <= fft_size");  // lib.rs:111
    if (fft_size & 1) DFX_FAIL(DFX_ERR_UNSUPPORTED, "fft_size must be even");
    if (fft_size > 4096) DFX_FAIL(DFX_ERR_UNSUPPORTED, "fft_size > 4096 is not supported by the LDS-resident FFT");
    if ((fft_size + hop_size - 1) / hop_size > DFX_DSP_TEAMS)
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "fft_size/hop_size > %d overlapping frames is not supported", DFX_DSP_TEAMS);
    if (int rc = dfx_require_device()) return rc;
    dfx_state *st = new dfx_state();
    st->sr = sr;
    st->N = fft_size;
    st->hop = hop_size;
    st->nb = nb_bands;
    st->min_nb = min_nb_erb_freqs;
    if (make_plan(fft_size, &st->plan) != 0) {
        delete st;
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "fft_size/2 = %d has a prime factor other than 2, 3, 5", fft_size / 2);
    }
    // vorbis window in f64, stored as f32 (lib.rs:126-133)
    st->window_host.resize(fft_size);
    const int window_size_h = fft_size / 2;
    for (int i = 0; i < fft_size; ++i) {
        const double s = sin(0.5 * M_PI * ((double)i + 0.5) / (double)window_size_h);
        st->window_host[i] = (float)sin(0.5 * M_PI * s * s);
    }
    st->wnorm = 1.f / ((float)((int64_t)fft_size * (int64_t)fft_size) / (float)(2 * hop_size));  // lib.rs:134
    std::vector<float2> tw(fft_size);
    for (int k = 0; k < fft_size; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)fft_size;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    std::vector<uint64_t> widths(nb_bands);
    dfx_erb_fb(sr, fft_size, nb_bands, min_nb_erb_freqs, widths.data());
    int rc = upload(&st->d_window, st->window_host.data(), st->window_host.size());
    if (!rc) rc = upload(&st->d_tw, tw.data(), tw.size());
    if (!rc && dfx_plan_is_480(st->plan)) {
        std::vector<unsigned char> mt(2 * DFX_MFFT_TABLE_BYTES);
        build_mfft_table(-1, mt.data());
        build_mfft_table(+1, mt.data() + DFX_MFFT_TABLE_BYTES);
        rc = upload(&st->d_mfft, mt.data(), mt.size());
    }
    if (!rc) rc = dfx_bands_create(widths.data(), nb_bands, &st->bands);
    if (!rc && st->bands->F != fft_size / 2 + 1) {
        dfx_set_error("ERB widths sum to %d, expected %d", st->bands->F, fft_size / 2 + 1);
        rc = DFX_ERR_INVALID_ARG;
    }
    if (rc) {
        dfx_state_free(st);
        return rc;
    }
    *out = st;
    return DFX_OK;
}
extern "C" void dfx_state_free(dfx_state *st) {
    if (!st) return;
    if (st->d_window) (void)hipFree(st->d_window);
    if (st->d_tw) (void)hipFree(st->d_tw);
    if (st->d_mfft) (void)hipFree(st->d_mfft);
    dfx_bands_free(st->bands);
    delete st;
}
extern "C" int dfx_state_sr(const dfx_state *st) { return st->sr; }
extern "C" int dfx_state_fft_size(const dfx_state *st) { return st->N; }
extern "C" int dfx_state_hop_size(const dfx_state *st) { return st->hop; }
extern "C" int dfx_state_nb_erb(const dfx_state *st) { return st->nb; }
extern "C" float dfx_state_wnorm(const dfx_state *st) { return st->wnorm; }
extern "C" const dfx_bands *dfx_state_bands(const dfx_state *st) { return st->bands; }
extern "C" int dfx_state_erb_widths(const dfx_state *st, uint64_t *out) {
    if (!st || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_state_erb_widths: null");
    for (int i = 0; i < st->nb; ++i) out[i] = st->bands->widths[i];
    return DFX_OK;
}
extern "C" int dfx_state_fft_window(const dfx_state *st, float *out) {
    if (!st || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_state_fft_window: null");
    memcpy(out, st->window_host.data(), sizeof(float) * st->window_host.size());
    return DFX_OK;
}
extern "C" int dfx_unit_norm_init(int n, float *out) {
    if (n <= 0 || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_unit_norm_init: bad arguments");
    const float step = n > 1 ? (0.0001f - 0.001f) / (float)(n - 1) : 0.f;
    for (int i = 0; i < n; ++i) out[i] = 0.001f + step * (float)i;
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ launch helpers
static size_t dsp_smem_bytes(const dfx_state *st) {
    return (size_t)st->N * 12 + (size_t)DFX_DSP_TEAMS * 2 * (size_t)(st->plan.M + 2) * 8;
}
// analysis: the 480-point plan transforms in place — one buffer per frame (dfx_plan_is_480, dfx_k_analysis)
static bool ana_in_place(const dfx_state *st) {
    const DfxFftPlan &pl = st->plan;   // (the 48 kHz / 20 ms plan transforms in place; any other plan takes the two-buffer passes)
    return pl.M == 480 && pl.nstage == 5 && pl.radix[0] == 4 && pl.radix[1] == 4 && pl.radix[2] == 2 && pl.radix[3] == 3 && pl.radix[4] == 5;
}
// the 480-point transform on the matrix pipe (dfx_fft480_mfma) instead of the radix passes: DFX_FFT_MFMA=1.  Not the default: built, validated
// and measured in round 5 — analysis 0.535 vs 0.478 ms, the finishing kernel 0.74 vs 0.63 (profiles/r05_dft_mfma.log)
// (round 6: no switch any more and no instance of the MF forms in the library; dfx_fft480_mfma stays in dfx_dsp_kernels.h as the record of the
// experiment, with its tables)
static constexpr bool fft_mfma(const dfx_state *) { return false; }
static size_t ana_smem_bytes(const dfx_state *st, bool mf = false) {
    return (mf ? (size_t)DFX_MFFT_FRAG3 * 64 * 16 : 0) + (size_t)st->N * 12 + (size_t)DFX_DSP_TEAMS * (ana_in_place(st) ? (size_t)DFX_FFT480_BUF : 2 * (size_t)(st->plan.M + 2)) * 8 +   // (in place: one buffer per frame, with room for the transform's padded layout)
           (((size_t)(2 * st->nb + 1 + 3 * 64 + st->nb + 1 + DFX_DSP_TEAMS * 64) * 4 + 15) & ~(size_t)15) + 64;   // + the ERB band tables and segment sums (analysis), and slack behind the last frame's sums (fixed-trip reads)
}
static int grid_for(int64_t work_groups, int per_cu = 8) {
    const int64_t cap = (int64_t)dfx_env_num_cus() * per_cu;  // memory-bound: ~8 workgroups per CU, grid-stride the rest
    return (int)(work_groups < cap ? (work_groups > 0 ? work_groups : 1) : cap);
}

int dfx_launch_analysis(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride,
                        const float *mem_in, float *mem_out, float *spec, float *erb_db, hipStream_t s, int64_t x_len,
                        int64_t spec_stride, bool x_i16) {
    if (x_i16 && (mem_in || mem_out)) DFX_FAIL(DFX_ERR_INVALID_ARG, "analysis of 16-bit PCM input carries no memories (whole rows only)");
    const int64_t Tf = T / st->hop;
    if (B > 0 && Tf > 0) {
        DfxAnaArgs A;
        A.x = x;
        A.mem_in = mem_in;
        A.spec = reinterpret_cast<float2 *>(spec);
        A.erb_db = erb_db;
        A.window = st->d_window;
        A.tw = st->d_tw;
        A.band_start = st->bands->d_start;
        A.band_invw = st->bands->d_invw;
        A.seg_tab = st->bands->d_segtab;
        A.nseg = st->bands->nseg;   // (0: more than 64 bands, one lane per band)
        // fixed-trip band sums: a lane reads up to segcap (rounded up to 6) power values from its segment's start, `part` up to segparts (rounded up
        // to 4) from its band's first segment — inside the frame buffer / the slack behind `part` for the shipped band layouts
        A.segcap = st->bands->segcap <= 24 && st->bands->segparts <= 12 ? st->bands->segcap : 0;
        A.segparts = st->bands->segparts;
        A.B = B;
        A.Tf = Tf;
        A.x_stride = x_stride;
        A.x_len = x_len < 0 ? T : x_len;
        A.spec_stride = spec_stride > 0 ? spec_stride : st->N / 2 + 1;
        A.hop = st->hop;
        A.nb = st->nb;
        A.wnorm = st->wnorm;
        A.plan = st->plan;
        const bool ip = ana_in_place(st), mf = ip && fft_mfma(st);
        const size_t smem = ana_smem_bytes(st, mf);
        A.mfft = mf ? st->d_mfft : nullptr;
        const int grid = grid_for(dfx_ceil_div(B * Tf, DFX_DSP_TEAMS), ip ? (mf ? 8 : 9) : 8);   // (three resident workgroups per CU in place, two on the matrix pipe: whole rounds)
        DfxKScope ks(DFX_K_ANALYSIS, s);
        auto go = [&](auto kern) -> int {
            if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)kern, smem));
            dfx_launch(kern, dim3(grid), dim3(DFX_DSP_THREADS), smem, s, A);
            return DFX_OK;
        };
        if (int rc = ip ? (x_i16 ? go(dfx_k_analysis<true, true>) : go(dfx_k_analysis<true, false>))
                        : (x_i16 ? go(dfx_k_analysis<false, true>) : go(dfx_k_analysis<false, false>)))
            return rc;
        DFX_LAUNCH_CHECK();
    }
    if (mem_out && B > 0) return dfx_launch_analysis_mem(st, x, B, T, x_stride, mem_in, mem_out, s);
    return DFX_OK;
}
// the analysis memory after the T samples of a call (what dfx_launch_analysis does last when it is given mem_out; on its own for callers
// that want it off the stream the spectra are waited for on: only the NEXT call reads it)
int dfx_launch_analysis_mem(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride, const float *mem_in, float *mem_out,
                            hipStream_t s) {
    const int64_t Tf = T / st->hop;
    const int ML = st->N - st->hop;
    if (!mem_out || B <= 0) return DFX_OK;
    const int64_t n = B * ML;
    DfxKScope ks(DFX_K_ANALYSIS_MEM, s);
    dfx_launch(dfx_k_analysis_mem_out, dim3((unsigned)dfx_ceil_div(n, 256)), dim3(256), 0, s, x, mem_in, mem_out, B, Tf, x_stride, st->hop, ML);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

int dfx_launch_norm_scan(const float *erb_in, float *erb_out, int E, const float *spec_in, int64_t spec_frame_stride,
                         float *spec_out, int Fn, int64_t C, int64_t T, float alpha, float *erb_state,
                         float *unit_state, hipStream_t s, int64_t erb_out_cs, int64_t spec_out_cs) {
    const int nch = (erb_in ? E : 0) + (spec_in ? Fn : 0);
    const int64_t n = C * nch;
    if (n <= 0) return DFX_OK;
    if ((erb_out_cs > 0 || spec_out_cs > 0) && T >= 16) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_launch_norm_scan: strided outputs are for calls of < 16 frames");
    DfxKScope ks(DFX_K_NORM_SCAN, s);
    // four lanes per (row, channel) when there are frames to share (the frame-by-frame streaming runtime keeps one lane per channel)
    if (T >= 16) {
        dfx_launch(dfx_k_norm_scan4, dim3((unsigned)dfx_ceil_div(4 * n, 256)), dim3(256), 0, s, erb_in, erb_out, E,
                   reinterpret_cast<const float2 *>(spec_in), spec_frame_stride, reinterpret_cast<float2 *>(spec_out), Fn, C,
                   T, alpha, erb_state, unit_state);
    } else {
        dfx_launch(dfx_k_norm_scan, dim3((unsigned)dfx_ceil_div(n, 64)), dim3(64), 0, s, erb_in, erb_out, E,
                   reinterpret_cast<const float2 *>(spec_in), spec_frame_stride, reinterpret_cast<float2 *>(spec_out), Fn, C,
                   T, alpha, erb_state, unit_state, erb_out_cs, spec_out_cs / 2);
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI: transforms
extern "C" int dfx_analysis(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride,
                            const float *mem_in, float *mem_out, float *spec, void *stream) {
    if (!st || B < 0 || T < 0 || x_stride < T) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_analysis: bad arguments");
    if (B > 0 && T / st->hop > 0 && (!x || !spec)) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_analysis: null buffer");
    if (int rc = dfx_require_device()) return rc;
    return dfx_launch_analysis(st, x, B, T, x_stride, mem_in, mem_out, spec, nullptr, dfx_stream(stream));
}

extern "C" int dfx_synthesis(const dfx_state *st, const float *spec, int64_t B, int64_t Tf, const float *mem_in,
                             float *mem_out, float *out, int64_t out_stride, void *stream) {
    if (!st || B < 0 || Tf < 0 || out_stride < Tf * st->hop) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_synthesis: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (B == 0 || (Tf == 0 && !mem_out)) return DFX_OK;
    if ((Tf > 0 && (!spec || !out))) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_synthesis: null buffer");
    return dfx_launch_synthesis(st, spec, B, Tf, mem_in, mem_out, out, out_stride, 0, Tf * st->hop, dfx_stream(stream));
}

int dfx_launch_synthesis(const dfx_state *st, const float *spec, int64_t B, int64_t Tf, const float *mem_in, float *mem_out,
                         float *out, int64_t out_stride, int64_t out_skip, int64_t out_len, hipStream_t stream, int64_t f_begin,
                         int64_t f_end, int64_t spec_stride, bool out_i16) {
    if (out_i16 && (mem_in || mem_out)) DFX_FAIL(DFX_ERR_INVALID_ARG, "synthesis to 16-bit PCM carries no memories (whole rows only)");
    DfxSynArgs A;
    A.spec_stride = spec_stride > 0 ? spec_stride : st->N / 2 + 1;
    const int Rr = (st->N + st->hop - 1) / st->hop;
    A.f_begin = f_begin;
    A.f_end = f_end < 0 ? Tf + (mem_out ? Rr - 1 : 0) : f_end;
    A.out_skip = out_skip;
    A.out_len = out_len;
    A.spec = reinterpret_cast<const float2 *>(spec);
    A.mem_in = mem_in;
    A.mem_out = mem_out;
    A.out = out;
    A.window = st->d_window;
    A.tw = st->d_tw;
    A.B = B;
    A.Tf = Tf;
    A.out_stride = out_stride;
    A.hop = st->hop;
    A.R = (st->N + st->hop - 1) / st->hop;
    A.outf = DFX_DSP_TEAMS - (A.R - 1);
    A.chunks = (int)dfx_ceil_div(A.f_end - A.f_begin, A.outf);
    A.plan = st->plan;
    if (A.chunks <= 0) return DFX_OK;
    // in place like the analysis (one buffer per frame, six waves per SIMD; without the register prefetch of the next frame, which would spill
    // there): 1.07-1.13 -> 0.93-1.03 ms alone
    const bool ip = ana_in_place(st), mf = ip && fft_mfma(st);
    const size_t smem = ip ? ana_smem_bytes(st, mf) : dsp_smem_bytes(st);
    A.mfft = mf ? st->d_mfft + DFX_MFFT_TABLE_BYTES : nullptr;   // (the inverse tables)
    int64_t nblk = B * A.chunks;
    // persistent workgroups (the twiddle / window tables are staged once per workgroup): a few per CU, grid-stride over the work items
    const int64_t cap = (int64_t)dfx_env_num_cus() * (ip ? (mf ? 8 : 9) : 8);   // (three resident workgroups per CU in place, two on the matrix pipe: whole rounds)
    if (nblk > cap) nblk = cap;
    DfxKScope ks(DFX_K_SYNTHESIS, stream);
    auto go = [&](auto kern) -> int {
        if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)kern, smem));
        dfx_launch(kern, dim3((unsigned)nblk), dim3(DFX_DSP_THREADS), smem, stream, A);
        return DFX_OK;
    };
    if (int rc = ip ? (out_i16 ? go(dfx_k_synthesis<true, true>) : go(dfx_k_synthesis<true, false>))
                    : (out_i16 ? go(dfx_k_synthesis<false, true>) : go(dfx_k_synthesis<false, false>)))
        return rc;
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

bool dfx_synthesis_rows_ok(const dfx_state *st, bool with_df, int order, int nb_df, int nbands) {
    if (!st || st->N != 960 || st->hop != 480 || !ana_in_place(st)) return false;
    if (!with_df) return true;
    return (order == 5 || nb_df == 0) && nb_df >= 0 && nb_df <= 128 && nb_df % 2 == 0 && nbands <= 64 && st->bands && st->bands->F == 481;   // (bins in pairs: 16-byte accesses)
}
int dfx_launch_synthesis_rows(const dfx_state *st, const float *spec, int64_t spec_stride, const float *coefs, int nb_df, int order, int lookahead,
                              const float *gains, float pf_beta, float atten_lim, int64_t B, int64_t Tf, float *out, int64_t out_stride,
                              int64_t out_skip, int64_t out_len, hipStream_t s, bool out_i16, const unsigned int *err, unsigned int *poison) {
    if (B <= 0 || Tf <= 0) return DFX_OK;
    const bool with_df = coefs != nullptr || gains != nullptr;
    if (!dfx_synthesis_rows_ok(st, with_df, order, coefs ? nb_df : 0, gains ? st->bands->nb : 0))
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "synthesis_rows: configuration (needs fft 960 / hop 480, deep-filter order 5, nb_df <= 128)");
    DfxSynRowsArgs A;
    A.spec = reinterpret_cast<const float2 *>(spec);
    A.coefs = reinterpret_cast<const float2 *>(coefs);
    A.gains = gains;
    A.bin2band = st->bands ? st->bands->d_bin2band : nullptr;
    A.out = out;
    A.window = st->d_window;
    A.tw = st->d_tw;
    A.B = B, A.Tf = Tf;
    A.spec_stride = spec_stride > 0 ? spec_stride : st->N / 2 + 1;
    if (with_df && ((A.spec_stride & 1) || ((uintptr_t)spec & 15) || ((uintptr_t)coefs & 15)))
        DFX_FAIL(DFX_ERR_UNSUPPORTED, "synthesis_rows: the deep-filter form reads bins in pairs (rows of an even stride, 16-byte aligned arrays)");
    A.out_stride = out_stride, A.out_skip = out_skip, A.out_len = out_len;
    const int64_t nd = coefs ? nb_df : 0;
    A.cs_b = (int64_t)order * Tf * nd, A.cs_n = Tf * nd, A.cs_t = nd;   // [B, O, Tf, nd] (DFX_COEF_BOTF)
    A.nbdf = (int)nd, A.lookahead = lookahead, A.nb = gains ? st->bands->nb : 0;
    A.pf_beta = pf_beta, A.atten_lim = atten_lim;
    // segments: enough (row, segment) items to fill the workgroups a CU holds — three of the plain ISTFT; two with the deep filter, whose next
    // frame stays in flight in ~60 registers (four waves per SIMD) —, but at least 4 chunks each (a
    // segment that does not start a row costs one extra single-wave item)
    const bool pf = pf_beta > 0.f || atten_lim > 0.f;
    const int wgs = !with_df ? DFX_SYNR_WGS : 2;
    const int64_t chunks = dfx_ceil_div(Tf, DFX_SYNR_TEAMS);
    const int64_t want = dfx_ceil_div((int64_t)dfx_env_num_cus() * wgs, B);
    int64_t segs = want < 1 ? 1 : want;
    if (segs > chunks / 4) segs = chunks / 4 > 0 ? chunks / 4 : 1;
    A.poison = nullptr;
    if (err && poison) {   // the pass's faults so far (every one that can reach this kernel's inputs), as a device word
        dfx_launch(dfx_k_fault_mirror, dim3(1), dim3(1), 0, s, err, poison);
        DFX_LAUNCH_CHECK();
        A.poison = poison;
    }
    A.seg_chunks = (int)dfx_ceil_div(chunks, segs);
    A.segs = (int)dfx_ceil_div(chunks, A.seg_chunks);
    int64_t nblk = B * A.segs;
    const bool mf = fft_mfma(st);
    A.mfft = mf ? st->d_mfft + DFX_MFFT_TABLE_BYTES : nullptr;   // (the inverse tables)
    const int64_t cap = (int64_t)dfx_env_num_cus() * (mf ? (wgs < 2 ? wgs : 2) : wgs);
    if (nblk > cap) nblk = cap;
    const size_t smem = mf ? DFX_SYNR_SMEM_MF : DFX_SYNR_SMEM;
    DfxKScope ks(DFX_K_SYNTHESIS, s);
    auto go = [&](auto kern) { dfx_launch(kern, dim3((unsigned)nblk), dim3(DFX_SYNR_THREADS), smem, s, A); };
    if (!with_df) {
        (out_i16 ? go(dfx_k_synthesis_rows<0, false, true>) : go(dfx_k_synthesis_rows<0, false, false>));
    } else if (pf) {
        (out_i16 ? go(dfx_k_synthesis_rows<5, true, true>) : go(dfx_k_synthesis_rows<5, true, false>));
    } else {
        (out_i16 ? go(dfx_k_synthesis_rows<5, false, true>) : go(dfx_k_synthesis_rows<5, false, false>));
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_erb(const dfx_bands *bands, const float *spec, int64_t rows, int db, float *out, void *stream) {
    if (!bands || rows < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (rows == 0) return DFX_OK;
    if (!spec || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb: null buffer");
    const int64_t n = rows * bands->nb;
    DfxKScope ks(DFX_K_ERB, dfx_stream(stream));
    dfx_launch(dfx_k_erb, dim3((unsigned)dfx_ceil_div(n, 256)), dim3(256), 0, dfx_stream(stream),
               reinterpret_cast<const float2 *>(spec), rows, bands->F, bands->nb, bands->d_start, bands->d_invw, db, out);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_erb_inv(const dfx_bands *bands, const float *gains, int64_t rows, float *out, void *stream) {
    if (!bands || rows < 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb_inv: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (rows == 0) return DFX_OK;
    if (!gains || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb_inv: null buffer");
    const int64_t n = rows * bands->F;
    DfxKScope ks(DFX_K_ERB_INV, dfx_stream(stream));
    dfx_launch(dfx_k_erb_inv, dim3((unsigned)dfx_ceil_div(n, 256)), dim3(256), 0, dfx_stream(stream), gains, rows,
               bands->F, bands->nb, bands->d_bin2band, out);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_erb_norm(float *x, int64_t C, int64_t T, int E, float alpha, float *state, void *stream) {
    if (C < 0 || T < 0 || E <= 0) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb_norm: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (C == 0 || T == 0) return DFX_OK;
    if (!x) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_erb_norm: null buffer");
    return dfx_launch_norm_scan(x, x, E, nullptr, 0, nullptr, 0, C, T, alpha, state, nullptr, dfx_stream(stream));
}

extern "C" int dfx_unit_norm(const float *x, int64_t x_frame_stride, float *out, int64_t C, int64_t T, int F,
                             float alpha, float *state, void *stream) {
    if (C < 0 || T < 0 || F <= 0 || x_frame_stride < F) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_unit_norm: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    if (C == 0 || T == 0) return DFX_OK;
    if (!x || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_unit_norm: null buffer");
    return dfx_launch_norm_scan(nullptr, nullptr, 0, x, x_frame_stride, out, F, C, T, alpha, nullptr, state,
                                dfx_stream(stream));
}

extern "C" int dfx_features(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride, int nb_df,
                            float alpha, float *spec, float *erb_feat, float *spec_feat, void *stream) {
    if (x_stride < T) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_features: bad arguments");
    return dfx_features_padded(st, x, B, T, T, x_stride, nb_df, alpha, spec, erb_feat, spec_feat, stream);
}

// dfx_features over rows of T samples of which only the first x_len exist in memory (the rest are zeros): enhance()'s end padding
int dfx_features_padded(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_len, int64_t x_stride, int nb_df,
                        float alpha, float *spec, float *erb_feat, float *spec_feat, void *stream, int64_t spec_stride, bool x_i16) {
    if (spec_stride <= 0) spec_stride = st ? st->N / 2 + 1 : 0;
    if (!st || B < 0 || T < 0 || x_len < 0 || x_len > T || x_stride < x_len || nb_df <= 0 || nb_df > st->N / 2 + 1)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_features: bad arguments");
    if (int rc = dfx_require_device()) return rc;
    const int64_t Tf = T / st->hop;
    if (B == 0 || Tf == 0) return DFX_OK;
    if (!x || !spec || !erb_feat || !spec_feat) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_features: null buffer");
    // enhance.py:190-197: spec = analysis(x); erb_norm(erb(spec)); unit_norm(spec[..., :nb_df])
    if (int rc = dfx_launch_analysis(st, x, B, T, x_stride, nullptr, nullptr, spec, erb_feat, dfx_stream(stream), x_len, spec_stride, x_i16)) return rc;
    return dfx_launch_norm_scan(erb_feat, erb_feat, st->nb, spec, spec_stride, spec_feat, nb_df, B, Tf, alpha, nullptr,
                                nullptr, dfx_stream(stream));
}

// frames [t_begin, t_end) of every clip (t_end < 0: T); coef_T: frames per clip of the coefficient / gain arrays (default T);
// out_T / out_toff: compacted output rows (default T / 0); spec_stride / out_stride: row strides in complex elements (0: F).
// Rows that are 16-byte aligned (even strides: the engine's own padded spec buffers) take the row-streaming kernel
// dfx_k_df_apply_rows; dense rows of an odd F (the public dfx_df_apply on [B,T,F] arrays) the flat-stream kernel dfx_k_df_apply.
template <int NPC, bool PF>
static int launch_dfa_rows(const DfxDfrArgs &A, int order, unsigned grid, hipStream_t s) {
    switch (order) {
#define DFX_DFR_CASE(O_) case O_: dfx_launch((dfx_k_df_apply_rows<O_, NPC, PF>), dim3(grid), dim3(256), 0, s, A); break;
        DFX_DFR_CASE(1) DFX_DFR_CASE(2) DFX_DFR_CASE(3) DFX_DFR_CASE(4) DFX_DFR_CASE(5) DFX_DFR_CASE(6) DFX_DFR_CASE(7) DFX_DFR_CASE(8)
        DFX_DFR_CASE(9) DFX_DFR_CASE(10) DFX_DFR_CASE(11) DFX_DFR_CASE(12) DFX_DFR_CASE(13) DFX_DFR_CASE(14) DFX_DFR_CASE(15) DFX_DFR_CASE(16)
#undef DFX_DFR_CASE
        default: DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: order %d > 16", order);
    }
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

int dfx_launch_df_apply(const float *spec, const float *coefs, int coef_layout, const float *gains,
                        const dfx_bands *bands, int64_t B, int64_t T, int F, int nb_df, int order, int lookahead,
                        float pf_beta, float atten_lim, float *out, hipStream_t s, int64_t t_begin, int64_t t_end, int64_t coef_T,
                        int64_t out_T, int64_t out_toff, int64_t spec_stride, int64_t out_stride, int pf_rs_channels) {
    if (spec_stride <= 0) spec_stride = F;
    if (out_stride <= 0) out_stride = F;
    if (t_end < 0) t_end = T;
    if (t_end <= t_begin) return DFX_OK;
    const int nbands = (gains && bands) ? bands->nb : 0;
    {
        const int64_t nd = nb_df, O = order, Tc = coef_T < 0 ? T : coef_T;
        const bool rows_ok = spec_stride % 2 == 0 && out_stride % 2 == 0 && coef_layout != DFX_COEF_BTFO && nd % 2 == 0 && nd / 2 <= 64 &&
                             nbands <= 64 && O <= 16 && !((uintptr_t)spec & 15) && !((uintptr_t)out & 15) && !((uintptr_t)coefs & 15);
        if (rows_ok) {
            DfxDfrArgs R;
            R.spec = spec;
            R.coefs = coefs;
            R.gains = nbands ? gains : nullptr;
            R.bin2band = bands ? bands->d_bin2band : nullptr;
            R.out = out;
            R.B = B;
            R.T = T;
            if (coef_layout == DFX_COEF_BOTF) R.cs_b = O * Tc * nd, R.cs_n = Tc * nd, R.cs_t = nd;   // [B,O,Tc,nd]
            else R.cs_b = Tc * O * nd, R.cs_t = O * nd, R.cs_n = nd;                                  // [B,Tc,O,nd]
            R.gT = Tc;
            R.out_T = out_T < 0 ? T : out_T;
            R.out_toff = out_toff;
            R.Fs = (int)spec_stride;
            R.Fso = (int)out_stride;
            R.F = F;
            R.nbdf = nb_df;
            R.lookahead = lookahead;
            R.nb = nbands;
            R.pf_beta = pf_beta;
            R.atten_lim = atten_lim;
            R.pf_ch = pf_rs_channels > 0 ? pf_rs_channels : 0;
            R.t_begin = (int)t_begin;
            R.t_end = (int)t_end;
            R.rpw = 1;   // one frame per wave measured fastest (6.2 TB/s vs 5.9 at 4 and 5.4 at 16: the O-1 extra rows a wave reads are L2 hits)
            R.chunks = (int)dfx_ceil_div(t_end - t_begin, R.rpw);
            R.zcols = (F + 1) / 2;
            if ((out_stride * 8) % 64 == 0) {   // 64-byte aligned output rows: complete the last sector of every row with zeros
                const int64_t z = dfx_ceil_div((int64_t)F * 8, 64) * 4;
                R.zcols = (int)(z < out_stride / 2 ? z : out_stride / 2);
            }
            R.items = dfx_ceil_div(B, 8) * 8 * dfx_ceil_div(R.chunks, 4);
            // one workgroup per work item (a persistent grid of n workgroups per CU walking the items measured slower: 0.54 vs 0.47 ms
            // stand-alone, profiles/r02_dfa_bench.log)
            const int64_t nblk = R.items;
            if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: batch too large for one launch");
            const int np = ((F + 1) / 2 + 63) / 64;
            DfxKScope ks(DFX_K_DF_APPLY, s);
            const bool pf = pf_beta > 0.f || atten_lim > 0.f;
            if (np == 4) return pf ? launch_dfa_rows<4, true>(R, order, (unsigned)nblk, s) : launch_dfa_rows<4, false>(R, order, (unsigned)nblk, s);
            return pf ? launch_dfa_rows<0, true>(R, order, (unsigned)nblk, s) : launch_dfa_rows<0, false>(R, order, (unsigned)nblk, s);
        }
        if (spec_stride != F || out_stride != F)
            DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: padded rows need even strides, a tap-/frame-major coefficient layout, an even nb_df <= 128, "
                                          "<= 64 bands, order <= 16 and 16-byte aligned buffers");
    }
    DfxDfaArgs A;
    A.spec = reinterpret_cast<const float2 *>(spec);
    A.coefs = reinterpret_cast<const float2 *>(coefs);
    A.gains = gains;
    A.bin2band = bands ? bands->d_bin2band : nullptr;
    A.out = reinterpret_cast<float2 *>(out);
    A.B = B;
    A.T = T;
    A.F = F;
    A.nbdf = nb_df;
    A.order = order;
    A.lookahead = lookahead;
    A.nb = (gains && bands) ? bands->nb : 0;
    if (!(gains && bands)) A.gains = nullptr;
    const int64_t nd = nb_df, O = order;
    const int64_t Tc = coef_T < 0 ? T : coef_T;
    if (coef_layout == DFX_COEF_BOTF) {         // [B,O,Tc,nd]
        A.cs_b = O * Tc * nd, A.cs_n = Tc * nd, A.cs_t = nd, A.cs_f = 1;
    } else if (coef_layout == DFX_COEF_BTFO) {  // [B,Tc,nd,O]
        A.cs_b = Tc * nd * O, A.cs_t = nd * O, A.cs_f = O, A.cs_n = 1;
    } else {                                    // DFX_COEF_BTOF [B,Tc,O,nd]
        A.cs_b = Tc * O * nd, A.cs_t = O * nd, A.cs_n = nd, A.cs_f = 1;
    }
    A.gT = Tc;
    A.out_T = out_T < 0 ? T : out_T;
    A.out_toff = out_toff;
    A.pf_beta = pf_beta;
    A.atten_lim = atten_lim;
    A.pf_ch = pf_rs_channels > 0 ? pf_rs_channels : 0;
    const int ROWS = DFX_DFA_ROWS;
    A.t_begin = (int)t_begin;
    A.t_end = (int)t_end;
    A.chunks = (int)dfx_ceil_div(t_end - t_begin, ROWS);
    if (T * (int64_t)F > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: T*F exceeds 2^31 elements per clip");
    const int64_t nblk = dfx_ceil_div(B, 8) * 8 * A.chunks;
    if (nblk > 0x7fffffff) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: batch too large for one launch");
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t smem = al((size_t)(ROWS + order - 1) * nb_df * 8) + al((size_t)ROWS * (A.nb > 0 ? A.nb : 1) * 4) + al((size_t)F);
    if (smem > 160 * 1024) DFX_FAIL(DFX_ERR_UNSUPPORTED, "dfx_df_apply: nb_df*order too large for LDS staging (%zu B)", smem);
    DfxKScope ks(DFX_K_DF_APPLY, s);
    if (smem > 64 * 1024) DFX_HIP(dfx_env_set_max_dyn_smem((const void *)dfx_k_df_apply<DFX_DFA_ROWS>, smem));
    dfx_launch(dfx_k_df_apply<DFX_DFA_ROWS>, dim3((unsigned)nblk), dim3(DFX_DFA_THREADS), smem, s, A);
    DFX_LAUNCH_CHECK();
    return DFX_OK;
}

extern "C" int dfx_df_apply(const float *spec, const float *coefs, int coef_layout, const float *gains,
                            const dfx_bands *bands, int64_t B, int64_t T, int F, int nb_df, int order, int lookahead,
                            float pf_beta, float atten_lim, float *out, void *stream) {
    if (B < 0 || T < 0 || F <= 0 || nb_df <= 0 || nb_df > F || order <= 0 || lookahead < 0 || lookahead >= order)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: bad sizes (need 0 <= lookahead < order, 0 < nb_df <= F)");
    if (coef_layout != DFX_COEF_BOTF && coef_layout != DFX_COEF_BTFO && coef_layout != DFX_COEF_BTOF)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: bad coef_layout");
    if (gains && (!bands || bands->F != F)) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: gains need a band table covering F bins");
    if (atten_lim < 0.f || atten_lim >= 1.f) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: atten_lim must be in [0,1)");
    if (int rc = dfx_require_device()) return rc;
    if (B == 0 || T == 0) return DFX_OK;
    if (!spec || !coefs || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: null buffer");
    if (spec == out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: out must not alias spec");
    if (((uintptr_t)spec & 15) || ((uintptr_t)out & 15) || ((uintptr_t)coefs & 7))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply: spec and out must be 16-byte aligned, coefs 8-byte aligned");
    return dfx_launch_df_apply(spec, coefs, coef_layout, gains, bands, B, T, F, nb_df, order, lookahead, pf_beta,
                               atten_lim, out, dfx_stream(stream));
}

// dfx_df_apply on rows with a stride: spec [B,T,spec_stride][2], out [B,T,out_stride][2] (strides in complex elements, >= F).  Even
// strides make every row 16-byte aligned (F = fft/2 + 1 is odd for every shipped model; the engine pads its own buffers to F + 1)
// and select the row-streaming kernel.  The pad bins of `out` are written as zeros, those of `spec` are ignored.
extern "C" int dfx_df_apply_strided(const float *spec, int64_t spec_stride, const float *coefs, int coef_layout, const float *gains,
                                    const dfx_bands *bands, int64_t B, int64_t T, int F, int nb_df, int order, int lookahead,
                                    float pf_beta, float atten_lim, float *out, int64_t out_stride, void *stream) {
    if (B < 0 || T < 0 || F <= 0 || nb_df <= 0 || nb_df > F || order <= 0 || lookahead < 0 || lookahead >= order || spec_stride < F ||
        out_stride < F)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: bad sizes (need 0 <= lookahead < order, 0 < nb_df <= F <= strides)");
    if (coef_layout != DFX_COEF_BOTF && coef_layout != DFX_COEF_BTFO && coef_layout != DFX_COEF_BTOF)
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: bad coef_layout");
    if (gains && (!bands || bands->F != F)) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: gains need a band table covering F bins");
    if (atten_lim < 0.f || atten_lim >= 1.f) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: atten_lim must be in [0,1)");
    if (int rc = dfx_require_device()) return rc;
    if (B == 0 || T == 0) return DFX_OK;
    if (!spec || !coefs || !out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: null buffer");
    if (spec == out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: out must not alias spec");
    if (((uintptr_t)spec & 15) || ((uintptr_t)out & 15) || ((uintptr_t)coefs & 7))
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_df_apply_strided: spec and out must be 16-byte aligned, coefs 8-byte aligned");
    return dfx_launch_df_apply(spec, coefs, coef_layout, gains, bands, B, T, F, nb_df, order, lookahead, pf_beta, atten_lim, out,
                               dfx_stream(stream), 0, -1, -1, -1, 0, spec_stride, out_stride);
}
